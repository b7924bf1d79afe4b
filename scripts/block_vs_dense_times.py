#!/usr/bin/env python3
"""
One ODE, the reference's two forms: a nonlinear ring of 32 variables with n_deriv = 5 solved
  * in BLOCK form  -- ibm_init's 32 blocks of 5 x 5, O(d p^3) per step (src/rodeo/solve.py:47-68; the blocked MFMA tiles with an
    eight-wave workgroup per trajectory), and
  * in NON-BLOCK form -- prior.indep_init's one dense 160 x 160 block, O((d p)^3) per step (BASELINE config 5's path),
same trajectories, solve_mv + interrogate_kramer; also at n_deriv = 3.  Prints one JSON line per (form, n_deriv).
    python scripts/block_vs_dense_times.py [--batch 256] [--steps 200]
"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
from scipy.linalg import block_diag

D = 32


def ring(X, t, **params):
    k, c = params["kc"]
    x = X[:, 0]
    return np.array([[k * (x[(i + 1) % D] - 2 * x[i] + x[(i - 1) % D]) - c * x[i] ** 3 + np.sin(t + i)] for i in range(D)])


def ring_nb(X, t, **params):                      # the same right-hand side on the non-block state (1, D * p), variable-major
    k, c = params["kc"]
    p = X.shape[1] // D
    x = X[0, ::p]
    return np.array([[k * (x[(i + 1) % D] - 2 * x[i] + x[(i - 1) % D]) - c * x[i] ** 3 + np.sin(t + i) for i in range(D)]])


def timeit(fn, dev, reps):
    fn(); dev.sync()
    dev.timer_start()
    for _ in range(reps):
        fn()
    return dev.timer_stop() / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    B, N = args.batch, args.steps
    t_max = N / 2000.0
    rng = np.random.default_rng(5)
    kc = np.array([0.7, 0.2]) * np.exp(0.05 * rng.standard_normal((B, 2)))
    xv = rng.standard_normal((B, D))
    for p in (3, 5):
        dev_b = ra.ode.from_python(ring, D, kc=2)
        W, init = ra.utils.first_order_pad(dev_b, D, p)
        x0 = init(xv, 0.0, kc=kc)
        prior = ra.ibm_init(t_max / N, p, np.ones(D))
        plan = ra.SolvePlan(dev_b, W, x0, 0.0, t_max, N, ra.interrogate.interrogate_kramer, prior, kc=kc)
        ms_b = timeit(lambda: plan.mv(None), plan.dev, 3)
        mb = plan.state_host()[0][0, -1, :, 0]
        print(json.dumps({"form": "block (ibm_init, 32 blocks)", "n_deriv": p, "B": B, "N": N, "ms": ms_b,
                          "ms_per_step": ms_b / N}), flush=True)
        # non-block form of the same problem
        fn_nb = ra.ode.from_python(ring_nb, 1, n_deriv_used=D * p, kc=2) if False else None
        Wn = block_diag(*[w for w in W])[None]
        X0 = x0.reshape(B, 1, D * p)
        pn = ra.indep_init(prior)
        plan_n = ra.SolvePlan(ring_nb, Wn, X0, 0.0, t_max, N, ra.interrogate.interrogate_kramer, pn, kc=kc)
        ms_n = timeit(lambda: plan_n.mv(None), plan_n.dev, 1)
        mn = plan_n.state_host()[0][0, -1, 0, ::p]
        print(json.dumps({"form": "non-block (indep_init, one 160-dim block)" if p == 5 else "non-block (indep_init, one 96-dim block)",
                          "n_deriv": p, "B": B, "N": N, "ms": ms_n, "ms_per_step": ms_n / N, "block_speedup": ms_n / ms_b,
                          "max_abs_diff_of_the_two_forms_x_at_t_max_traj0": float(np.max(np.abs(mb - mn)))}), flush=True)


if __name__ == "__main__":
    main()
