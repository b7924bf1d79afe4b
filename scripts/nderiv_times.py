"""Per-kernel times of the headline shape (FitzHugh-Nagumo, 1024 trajectories x 4000 steps, solve_mv + kramer) for
n_deriv = 3 .. 8, and of solve_sim + chkrebtii at the C4 shape (1024 draws x 800 steps) for n_deriv = 3 .. 6: which kernel
family a configuration lands on, and what the step from one family to the next costs."""
import functools, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra

rng = np.random.default_rng(20240)
B = 1024
eps = rng.standard_normal((B, 5))
theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.1 * eps[:, :3])
x0v = np.array([-1.0, 1.0]) + 0.1 * eps[:, 3:]


def timed(plan, call, reps=5):
    dev = plan.dev
    call(); call()
    dev.profile_enable(True)
    acc = {}
    for _ in range(reps):
        call()
        for k, ms in dev.profile_last():
            acc.setdefault(k, []).append(ms)
    dev.profile_enable(False)
    return {k: round(float(np.mean(v)), 4) for k, v in acc.items()}


for p in [int(v) for v in sys.argv[1:]] or [3, 4, 5, 6, 7, 8]:
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0 = init(x0v, 0.0, theta=theta)
    N = 4000
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, 40.0, N, ra.interrogate.interrogate_kramer,
                        ra.ibm_init(40.0 / N, p, np.array([0.1, 0.1])), theta=theta)
    k = timed(plan, lambda: plan.mv(None))
    tot = sum(k.values())
    a_mv = 3 * 2 * p * (p + 1) * 8
    print(json.dumps({"config": f"FN n_deriv={p}, B=1024, N=4000, solve_mv+kramer", "layout": plan.layout, "kernels_ms": k,
                      "total_ms": round(tot, 4), "traj_steps_per_s": B * N / (tot * 1e-3),
                      "hbm_frac_algorithmic": a_mv * B * N / (tot * 1e-3) / 8e12}), flush=True)
    del plan
    if p <= 6:
        N = 800
        g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard")
        plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, 40.0, N, g, ra.ibm_init(40.0 / N, p, np.array([0.1, 0.1])),
                            theta=theta)
        k = timed(plan, lambda: plan.sim(7))
        print(json.dumps({"config": f"FN n_deriv={p}, B=1024, N=800, solve_sim+chkrebtii", "layout": plan.layout,
                          "kernels_ms": k, "total_ms": round(sum(k.values()), 4)}), flush=True)
        del plan


# kalman_type = "square-root" on the same shape (lane-per-trajectory kernels, solve_sqrt.hip): what the factor form costs
# next to the covariance form's MFMA tiles
for p in [int(v) for v in sys.argv[1:] if int(v) <= 6] or [3, 4, 5]:
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0 = init(x0v, 0.0, theta=theta)
    N = 4000
    Q, R = ra.ibm_init(40.0 / N, p, np.array([0.1, 0.1]))
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, 40.0, N, ra.interrogate.interrogate_kramer,
                        (Q, np.linalg.cholesky(R)), kalman_type="square-root", theta=theta)
    k = timed(plan, lambda: plan.mv(None), reps=3)
    tot = sum(k.values())
    print(json.dumps({"config": f"FN n_deriv={p}, B=1024, N=4000, solve_mv+kramer, kalman_type=square-root", "layout": plan.layout,
                      "kernels_ms": k, "total_ms": round(tot, 4), "traj_steps_per_s": B * N / (tot * 1e-3),
                      "hbm_frac_algorithmic": 3 * 2 * p * (p + 1) * 8 * B * N / (tot * 1e-3) / 8e12}), flush=True)
    del plan
