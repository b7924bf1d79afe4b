"""
User-supplied right-hand sides (hiprtc): the compile step needs no GPU, so the CPU suite checks that both forms of
user source (explicit f / fjac, and scalar-generic rhs + AutoJac duals) build for gfx950 and that compiler errors
surface through rk_last_error(); the GPU suite (tests/test_gpu_user_rhs.py) checks the numbers.
"""
import numpy as np
import pytest

SEIR_SRC = r"""
// SIR-type epidemic model with three compartments, scalar-generic: Jacobian by forward-mode duals
struct Sir3 {
    static constexpr int D = 3;
    static constexpr int NTHETA = 2;
    static constexpr int NDEP = 1;
    template <class T, int P>
    __device__ __forceinline__ static void rhs(const T (&X)[D][P], double t, const double (&th)[NTHETA], T (&out)[D]) {
        const T S = X[0][0], I = X[1][0], R = X[2][0];
        const double beta = th[0], gamma = th[1];
        out[0] = -beta * S * I;
        out[1] = beta * S * I - gamma * I;
        out[2] = gamma * I + 0.0 * R;
    }
};
"""

FN_SRC = r"""
struct MyFitz {
    static constexpr int D = 2;
    static constexpr int NTHETA = 3;
    static constexpr int NDEP = 1;
    static constexpr bool HAS_TILE_FORM = false;
    template <int P>
    __device__ __forceinline__ static void f(const double (&X)[D][P], double, const double (&th)[NTHETA], double (&out)[D]) {
        const double a = th[0], b = th[1], c = th[2], V = X[0][0], R = X[1][0];
        out[0] = c * (V - V * V * V / 3 + R);
        out[1] = -1 / c * (V - a + b * R);
    }
    template <int P>
    __device__ __forceinline__ static void fjac(const double (&X)[D][P], double t, const double (&th)[NTHETA],
                                                double (&out)[D], double (&J)[D][P]) {
        f<P>(X, t, th, out);
        for (int b = 0; b < D; ++b) for (int j = 0; j < P; ++j) J[b][j] = 0.0;
        J[0][0] = th[2] * (1.0 - X[0][0] * X[0][0]);
        J[1][0] = -th[1] / th[2];
    }
};
"""


def sir_host(X, t, theta):
    theta = np.asarray(theta, dtype=np.float64)
    beta, gamma = theta[..., 0], theta[..., 1]
    S, I = X[..., 0, 0], X[..., 1, 0]
    return np.stack([-beta * S * I, beta * S * I - gamma * I, gamma * I], axis=-1)[..., None]


def test_user_sources_compile_for_gfx950_without_gpu():
    import rodeo_amd as ra
    from rodeo_amd import _lib
    sir = ra.ode.from_source("AutoJac<Sir3>", SEIR_SRC, 3, (("theta", 2),), sir_host, name="sir3")
    assert sir.rhs_id >= _lib.RHS_USER_BASE and sir.n_block == 3 and sir.n_theta == 2
    ra.ode.compile_check(sir, 3, _lib.INTERROGATE_KRAMER)
    ra.ode.compile_check(sir, 4, _lib.INTERROGATE_CHKREBTII)
    fn = ra.ode.from_source("MyFitz", FN_SRC, 2, (("theta", 3),), name="myfitz")
    ra.ode.compile_check(fn, 3, _lib.INTERROGATE_RODEO)
    bad = ra.ode.from_source("Broken", "struct Broken { static constexpr int D = 1; this is not C++ };", 1)
    with pytest.raises(_lib.RodeoKalmanError) as e:
        ra.ode.compile_check(bad, 3)
    assert "Broken" in str(e.value) or "error" in str(e.value)
    with pytest.raises(TypeError):
        fn(np.zeros((2, 3)), 0.0, theta=np.ones(3))          # registered without a host twin
