"""Kernel times of the headline configuration (C2) with the right-hand side given as USER source (hiprtc):
once with a hand-written Jacobian, once scalar-generic with the Jacobian by forward-mode duals (AutoJac<>)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
import bench

HAND = r"""
struct MyFitz {
    static constexpr int D = 2;
    static constexpr int NTHETA = 3;
    static constexpr int NDEP = 1;
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
AUTO = r"""
struct FitzG {
    static constexpr int D = 2;
    static constexpr int NTHETA = 3;
    static constexpr int NDEP = 1;
    template <class T, int P>
    __device__ __forceinline__ static void rhs(const T (&X)[D][P], double t, const double (&th)[NTHETA], T (&out)[D]) {
        const double a = th[0], b = th[1], c = th[2];
        const T V = X[0][0], R = X[1][0];
        out[0] = c * (V - V * V * V / 3.0 + R);
        out[1] = (-1.0 / c) * (V - a + b * R);
    }
};
"""
W, x0, theta, prior = bench.make_problem(ra, 0)
for label, typ, src in (("built-in (tile form)", None, None), ("user source, hand-written Jacobian", "MyFitz", HAND),
                        ("user source, AutoJac<> duals", "AutoJac<FitzG>", AUTO)):
    ode = ra.ode.fitzhugh_nagumo if typ is None else ra.ode.from_source(typ, src, 2, (("theta", 3),),
                                                                         ra.ode.fitzhugh_nagumo._host_fun, name="t_" + typ[:6])
    plan = ra.SolvePlan(ode, W, x0, 0.0, 40.0, 4000, ra.interrogate.interrogate_kramer, prior, theta=theta)
    dev = plan.dev
    plan.mv(None); plan.mv(None)
    dev.profile_enable(True)
    acc = {}
    for _ in range(10):
        plan.mv(None)
        for k, ms in dev.profile_last():
            acc.setdefault(k, []).append(ms)
    dev.profile_enable(False)
    print(f"{label:40s}", {k: round(float(np.mean(v)), 4) for k, v in acc.items()}, flush=True)
