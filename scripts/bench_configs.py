#!/usr/bin/env python3
"""
Timings of the other BASELINE.json configurations (not the bench.py headline): C3 Lorenz63, C4 pseudo-marginal
log-posterior (per GPU: 1024 draws), C5 dense 160-dim block.  Prints one JSON line per configuration.
    python scripts/bench_configs.py [c3] [c4] [c5] [--c5-batch B] [--c5-steps N]
"""
import argparse, functools, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
from rodeo_amd.inference import gauss_obs_logpost, obs_index


def timeit(fn, dev, reps):
    fn(); dev.sync()
    dev.timer_start()
    for _ in range(reps):
        fn()
    return dev.timer_stop() / reps


def c3(args):
    B, N, p = args.c3_batch, 20000, 4
    rng = np.random.default_rng(20241)
    theta = np.array([28., 10., 8. / 3.])
    W, init = ra.utils.first_order_pad(ra.ode.lorenz63, 3, p)
    x0 = init(np.array([-12., -5., 38.]) + 1e-3 * rng.standard_normal((B, 3)), 0.0, theta=theta)
    prior = ra.ibm_init(20.0 / N, p, np.array([5e7] * 3))
    plan = ra.SolvePlan(ra.ode.lorenz63, W, x0, 0.0, 20.0, N, ra.interrogate.interrogate_kramer, prior, theta=theta)
    ms = timeit(lambda: plan.mv(None), plan.dev, 5)
    a = 3 * 3 * p * (p + 1) * 8
    plan.dev.profile_enable(True)
    plan.mv(None); plan.dev.sync()
    kern = dict(plan.dev.profile_last())
    plan.dev.profile_enable(False)
    return {"config": "C3 Lorenz63 d=3 p=4 N=20000 B=%d solve_mv+kramer" % B, "ms": ms, "traj_steps_per_s": B * N / ms * 1e3,
            "hbm_frac_solve": a * B * N / (ms * 1e-3) / 8e12, "layout": plan.layout, "kernels_ms": kern}


def c4(args):
    B, N = 1024, 800
    rng = np.random.default_rng(20242)
    u0 = np.concatenate([np.log([.2, .2, 3.]), [-1., 1.], [.1, .1]])
    upars = u0 + np.array([0.01, 0.1, 0.01, 0.01, 0.01, 0.01, 0.01]) * rng.standard_normal((B, 7))
    theta, x0v, sigma = np.exp(upars[:, :3]), upars[:, 3:5], upars[:, 5:]
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    X0 = init(x0v, 0., theta=theta)
    prior = ra.ibm_init(40.0 / N, 3, sigma)
    g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard")
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, X0, 0., 40., N, g, prior, theta=theta)
    obs_t = np.linspace(0, 40, 41)
    ind = obs_index(0., 40., N, obs_t)
    Y = rng.standard_normal((41, 2))
    from rodeo_amd.inference import sim_logpost
    if args.c4_unfused:                                  # round 3's form: two device calls with the parameters uploaded in between
        def run():
            plan.sim(20242)
            return gauss_obs_logpost(plan, Y, ind, np.sqrt(0.005), upars=upars, n_prior=5, reuse_out=True)
    else:                                                # one call (rk_solve_sim_logpost): the sampler reduces the log-posterior itself;
        from rodeo_amd.inference import stage_upars      # inputs resident before the timed region (bench contract): the parameters are
        d_up = stage_upars(plan, upars, 5)               # uploaded ONCE here, like x0 / theta / prior in SolvePlan, not once per evaluation
        def run():
            return sim_logpost(plan, 20242, Y, ind, np.sqrt(0.005), upars=d_up)
    ms = timeit(run, plan.dev, 5)
    a = (2 * 2 * 3 * 4 + 2 * 3) * 8
    plan.dev.profile_enable(True)
    run(); plan.dev.sync()
    kern = dict(plan.dev.profile_last())
    plan.dev.profile_enable(False)
    lp = run().to_host()
    return {"config": "C4 FN pseudo-marginal: 1024 draws/GPU, N=800, solve_sim+chkrebtii+logpost", "ms": ms,
            "traj_steps_per_s": B * N / ms * 1e3, "hbm_frac_solve": a * B * N / (ms * 1e-3) / 8e12,
            "kernels_ms": kern, "finite": bool(np.all(np.isfinite(lp)))}


def c5(args):
    from scipy.linalg import block_diag
    n_vars, n_deriv, N, B = 32, 5, args.c5_steps, args.c5_batch
    rng = np.random.default_rng(20243)
    lam = np.logspace(0, 3, n_vars)
    A = -np.diag(lam) + 0.1 * rng.standard_normal((n_vars, n_vars)) / np.sqrt(n_vars)
    Wb, _ = ra.utils.first_order_pad(lambda x, t: x, n_vars, n_deriv)
    W = block_diag(*[w for w in Wb])[None]
    prior = ra.indep_init(ra.ibm_init(1.0 / 2000, n_deriv, np.ones(n_vars)))
    if args.c5_kalman == "square-root":
        prior = (prior[0], np.linalg.cholesky(prior[1]))
    x0v = 1.0 + 0.01 * rng.standard_normal((B, n_vars))
    X0 = np.zeros((B, n_vars, n_deriv)); X0[..., 0] = x0v; X0[..., 1] = x0v @ A.T
    X0 = X0.reshape(B, 1, -1)
    plan = ra.SolvePlan(ra.ode.linear_dense(n_vars, n_deriv), W, X0, 0.0, N / 2000.0, N,
                        getattr(ra.interrogate, "interrogate_" + args.c5_itg), prior, kalman_type=args.c5_kalman, A=A)
    dev = plan.dev
    dev.profile_enable(True)
    t0 = time.perf_counter(); plan.mv(None); dev.sync(); wall = time.perf_counter() - t0
    prof = dict(dev.profile_last())
    p, m = 160, 32
    F = (12 + 2 / 3) * p ** 3 + 4 * m * p * p + 4 * m * m * p + (2 / 3) * m ** 3
    ms = sum(prof.values())
    if os.environ.get("RK_DENSE_STAMPS") and args.c5_kalman == "square-root":
        # library built with `make -C rodeo_amd/csrc stamps`, loaded through RK_LIB_PATH (solve_dense_sqrt.hpp)
        nb = ["S gemms", "L^-1 Q", "X Sigma_f", "L^-T .", "J^T", "stack gemms", "mean", "QR 3p x p", "store"]
        nq = ["QR panel", "QR T + W pass", "QR T^T W", "QR rank-16 update"]
        nf = ["(QL)^T, R^T", "QR 2p x p", "stores + mu-", "interrogation", "W~L, stack", "QR (p+kv) x m", "tri solves m", "K gemms",
              "D, stack", "QR (p+kv) x p"]
        sq_stride = (6 * p * p + 4 * m * p + (p + m) * m + m * m) + 2 * p + 2 * m + 16
        bw = plan._ws.to_host()[:sq_stride][::-1][1:16]              # stamps_[-2 - k] of trajectory 0 after the backward pass
        print("sqrt bwd phase cycles per step (wg 0):", {k: int(v / (N - 1)) for k, v in zip(nb, bw[:9])},
              {k: int(v / (N - 1)) for k, v in zip(nq, bw[10:14])}, file=sys.stderr)
        plan.filter(None); dev.sync()
        fw = plan._ws.to_host()[:sq_stride][::-1][1:16]
        print("sqrt fwd phase cycles per step (wg 0):", {k: int(v / N) for k, v in zip(nf, fw[:10])},
              {k: int(v / N) for k, v in zip(nq, fw[10:14])}, file=sys.stderr)
    elif os.environ.get("RK_DENSE_STAMPS") == "fwd":
        plan.filter(None); dev.sync()
        ws = plan._ws.to_host().reshape(B, -1)
        names = ["Q Sigma", "(Q Sigma) Q^T + R", "mu-", "interrogation + W~ Sigma-", "S", "Sigma- W~^T", "LU + mean + downdate"]
        print("fwd phase cycles per step (wg 0):", {k: int(v) for k, v in zip(names, ws[0, -2:-9:-1] / N)}, file=sys.stderr)
    elif os.environ.get("RK_DENSE_STAMPS"):
        ws = plan._ws.to_host().reshape(B, -1)
        names = ["predict: rest", "LU panel", "LU swaps / gather", "LU trsm", "LU gemm", "LU: wave 1 strips", "predict: fused pass", "mean", "G D", "GDG^T",
                 "backsub: store", "backsub: stage", "backsub: trsm", "backsub: update"]
        cyc = ws[0, -2:-16:-1] / (N - 1)            # library built with -DRK_DENSE_STAMPS (solve_dense.hip)
        print("bwd phase cycles per step (wg 0):", {k: int(v) for k, v in zip(names, cyc)}, "total", int(cyc.sum()), file=sys.stderr)
    err = None
    if args.c5_check:                                   # distance of trajectory 0's solution estimate from expm(A t) x0 at the last step
        from scipy.linalg import expm
        err = float(np.max(np.abs(plan.mean_state.slice0_host(0)[N, 0, ::n_deriv] - expm(A * N / 2000.0) @ x0v[0])))
    return {"config": f"C5 dense p=160 m=32 N={N} B={B} solve_mv+{args.c5_itg} kalman_type={args.c5_kalman}", "ms": ms,
            "kernels_ms": prof, "flop_count": "covariance form (SURVEY 8d)", "abs_err_vs_expm_last_step_traj0": err,
            "traj_steps_per_s": B * N / ms * 1e3, "tflops": F * B * N / (ms * 1e-3) / 1e12,
            "frac_fp64_peak_78.6TF": F * B * N / (ms * 1e-3) / 78.6e12, "wall_s": wall}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["c3", "c4", "c5"])
    ap.add_argument("--c3-batch", type=int, default=512)
    ap.add_argument("--c5-batch", type=int, default=256)
    ap.add_argument("--c5-steps", type=int, default=2000)      # BASELINE config 5: N = 2000
    ap.add_argument("--c5-itg", default="kramer", choices=["kramer", "rodeo", "schober"])
    ap.add_argument("--c5-kalman", default="standard", choices=["standard", "square-root"])
    ap.add_argument("--c5-check", action="store_true")
    ap.add_argument("--c4-unfused", action="store_true")
    args = ap.parse_args()
    for w in args.which:
        print(json.dumps({"c3": c3, "c4": c4, "c5": c5}[w](args)), flush=True)
