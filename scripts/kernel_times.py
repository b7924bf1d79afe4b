"""Per-kernel times of the headline configuration (C2) from the library's HIP-event profile, solve_mv and filter-only streams; optional batch sizes as arguments."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
import bench
W, x0, theta, prior = bench.make_problem(ra, 0)
for B in [int(v) for v in sys.argv[1:]] or [1024]:
    reps = -(-B // 1024)
    x0b, thb = np.concatenate([x0] * reps)[:B], np.concatenate([theta] * reps)[:B]
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0b, 0.0, 40.0, 4000, ra.interrogate.interrogate_kramer, prior, theta=thb)
    dev = plan.dev
    plan.mv(None); plan.mv(None)
    dev.profile_enable(True)
    acc = {}
    for _ in range(10):
        plan.mv(None)
        for k, ms in dev.profile_last():
            acc.setdefault(k, []).append(ms)
    dev.profile_enable(False)
    print("B =", B, {k: round(float(np.mean(v)), 4) for k, v in acc.items()})
    dev.profile_enable(True)
    acc = {}
    for _ in range(10):
        plan.filter(None)
        for k, ms in dev.profile_last():
            acc.setdefault(k, []).append(ms)
    dev.profile_enable(False)
    print("        filter only:", {k: round(float(np.mean(v)), 4) for k, v in acc.items()})
    del plan
