"""Workload for the PMC passes on the B = 2048 forward slowdown: three filter-only launches, then three solve_mv launches
(forward + backward), in this order -- the dispatch order in the counter CSV identifies the neighbourhood of every forward
launch.  See scripts/fwd_slowdown_probe.py and DESIGN.md section 4."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
import bench
W, x0, theta, prior = bench.make_problem(ra, 0)
B = 2048
x0b, thb = np.concatenate([x0] * 2)[:B], np.concatenate([theta] * 2)[:B]
plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0b, 0.0, 40.0, 4000, ra.interrogate.interrogate_kramer, prior, theta=thb)
for _ in range(3):
    plan.filter(None)
plan.sync()
for _ in range(3):
    plan.mv(None)
plan.sync()
