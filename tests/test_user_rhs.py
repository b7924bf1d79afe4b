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


def test_trace_python_rhs_to_source_and_compile():
    """ode.from_python: the generated struct, NDEP detection, powers / elementary functions, host twin, hiprtc build."""
    import rodeo_amd as ra
    from rodeo_amd import trace

    def fitz_fun(X, t, **params):
        a, b, c = params["theta"]
        V, R = X[:, 0]
        return np.array([[c * (V - V * V * V / 3 + R)], [-1 / c * (V - a + b * R)]])
    src, ndep, _ = trace.trace_source(fitz_fun, 2, 2, (("theta", 3),), "Probe")
    assert ndep == 1 and "static constexpr int D = 2;" in src and "NTHETA = 3" in src
    assert "out[0] = (th[2] * ((X[0][0] - (((X[0][0] * X[0][0]) * X[0][0]) / 3.0)) + X[1][0]));" in src

    def second(X, t, k):
        return np.array([[np.sin(2 * t) - k[0] * X[0, 0] - 0.1 * X[0, 1] ** 3 + np.exp(-X[0, 0] ** 2) + np.sqrt(1.0 + X[0, 0] ** 2)]])
    src2, ndep2, _ = trace.trace_source(second, 1, 2, (("k", 1),), "Probe2")
    assert ndep2 == 2 and "sin((2.0 * t))" in src2 and "exp(" in src2 and "sqrt(" in src2
    ode = ra.ode.from_python(fitz_fun, 2, theta=3)
    assert ra.ode.from_python(fitz_fun, 2, theta=3) is ode                    # cached
    ra.ode.compile_check(ode, 3)
    ra.ode.compile_check(ra.ode.from_python(second, 1, k=1), 4)
    X = np.array([[-1., 0, 0], [1., 0, 0]])
    np.testing.assert_allclose(ode(X, 0.0, theta=np.array([.2, .2, 3.])).ravel(), [1.0, 1.0 / 3.0], rtol=1e-15)
    assert ode(np.zeros((4, 2, 3)), 0.0, theta=np.ones((4, 3))).shape == (4, 2, 1)
    with pytest.raises(TypeError):
        trace.trace_source(lambda X, t: np.array([[abs(X[0, 0]) if X[0, 0] > 0 else 0.0]]), 1, 2, (), "Bad")


def test_non_block_form_beyond_the_lane_kernels_builds_the_dense_interrogation_kernel():
    """An ode_fun in the reference's non-block form (X (1, p) -> (1, n_vars), prior.indep_init) with more states than the
    lane-per-trajectory kernels take: traced, and hiprtc builds the dense path's interrogation kernel around it
    (csrc/solve_dense_itg_kernels.hpp) -- no GPU needed for the build.  Several blocks keep the limit of four measurements."""
    import rodeo_amd as ra
    from rodeo_amd import _lib, trace
    n_vars, n_deriv = 8, 3

    def ring(X, t, kc):
        x = X[0, ::n_deriv]
        return np.array([[-kc[0] * x[i] + kc[1] * np.sin(x[(i + 1) % n_vars]) - 0.1 * x[i] ** 3 + 0.05 * np.cos(t)
                          for i in range(n_vars)]])
    dev = trace.from_python(ring, 1, n_vars * n_deriv, kc=2)
    assert (dev.n_block, dev.n_bmeas) == (1, n_vars) and "static constexpr int M = 8;" in dev.source
    for itg in (_lib.INTERROGATE_KRAMER, _lib.INTERROGATE_SCHOBER, _lib.INTERROGATE_RODEO):
        ra.ode.compile_check(dev, n_vars * n_deriv, itg)
    out = dev(np.zeros((1, n_vars * n_deriv)), 0.0, kc=np.array([1.0, 1.0]))
    assert out.shape == (1, n_vars) and np.allclose(out, 0.05)
    with pytest.raises(ValueError):
        trace.trace_source(lambda X, t: np.array([[X[0, 0]] * 5, [X[1, 0]] * 5]), 2, 2, (), "FiveMeasurementsTwoBlocks")


def test_run_time_builds_survive_another_rocm_in_the_process():
    """A process that imported torch first carries torch's own libhiprtc / comgr under the same soname, and the library's
    hiprtc calls land there; an -mllvm option that compiler does not know makes LLVM call exit() (found when the GPU suite
    died without a message in its first traced test).  The option is therefore only passed to the system's hiprtc
    (rhs_jit.hip, hiprtc_takes_backend_options); here: the build goes through in a child process with torch loaded."""
    import os, subprocess, sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = ("def f(X, t, theta):\n"
            "    return np.array([[theta[0] * X[0, 0] - X[1, 0] ** 2], [np.sin(X[0, 0]) + theta[1]]])\n"
            "ra.ode.compile_check(ra.ode.from_python(f, 2, theta=2), 3)\n"
            "print('built')\n")
    # torch before the library (its libhiprtc serves the calls), and the library before torch (the system's libhiprtc, but
    # torch's libamd_comgr is in the process by the time of the first build): the second order is what killed the CPU
    # suite once the gloo tests imported torch lazily
    orders = ("import torch, sys, numpy as np; sys.path.insert(0, %r); import rodeo_amd as ra\n" % root,
              "import sys, numpy as np; sys.path.insert(0, %r); import rodeo_amd as ra; ra._lib.load(); import torch\n" % root)
    for head in orders:
        out = subprocess.run([sys.executable, "-c", head + body], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "built" in out.stdout, (head, out.stderr[-2000:])


def test_first_order_pad_with_plain_python_function_is_batch_safe():
    """utils.first_order_pad on the reference's own kind of ode_fun (written for one trajectory): a batch of initial
    values and parameters is evaluated per element, and equals the built-in functor's initialisation."""
    import rodeo_amd as ra

    def fitz_fun(X, t, **params):
        a, b, c = params["theta"]
        V, R = X[:, 0]
        return np.array([[c * (V - V * V * V / 3 + R)], [-1 / c * (V - a + b * R)]])
    W, init = ra.utils.first_order_pad(fitz_fun, 2, 3)
    th = np.array([[.2, .2, 3.], [.3, .2, 2.]])              # B = 2 = n_vars: indexing over the batch axis would "work"
    x0 = np.array([[-1., 1.], [0.5, 0.2]])
    ref = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)[1](x0, 0.0, theta=th)
    np.testing.assert_allclose(init(x0, 0.0, theta=th), ref, rtol=1e-15)
    assert init(x0[0], 0.0, theta=th[0]).shape == (2, 3) and W.shape == (2, 1, 3)
