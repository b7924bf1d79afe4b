import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import rodeo_amd as ra
from oracle import scan, odes, interrogations as oi
from test_gpu_dense import dense_problem
np.set_printoptions(linewidth=220, precision=3)
N, t_max = 24, 0.3
for B in (2, 3):
    s = dense_problem(ra, 4, 3, N, t_max, B=B)
    plan = ra.SolvePlan(ra.ode.linear_dense(4, 3), s["W"], s["x0"], 0.0, t_max, N, ra.interrogate.interrogate_kramer, s["prior"], A=s["A"])
    plan.filter(None)
    mf, vf = plan.state_host()
    fo = scan.solve_filter(None, odes.make_linear_dense(s["A"], 3), s["W"], s["x0"], 0.0, t_max, N, oi.interrogate_kramer, *s["prior"])
    print("B", B, "filter err mean per traj/step\n", np.abs(mf - fo["state_filt"][0]).max(axis=(2, 3)))
    print("cond S-ish: var scale", np.abs(fo["state_filt"][1]).max(), "mean scale", np.abs(fo["state_filt"][0]).max(axis=(0,1,2)))
