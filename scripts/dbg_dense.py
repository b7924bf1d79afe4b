import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import rodeo_amd as ra
from oracle import scan, odes, interrogations as oi
from test_gpu_dense import dense_problem
np.set_printoptions(linewidth=200, precision=4)
for itg in ("schober", "kramer"):
    for N in (1, 2, 5):
        s = dense_problem(ra, 4, 3, N, 0.0125 * N, B=2)
        g, o = getattr(ra.interrogate, "interrogate_" + itg), getattr(oi, "interrogate_" + itg)
        plan = ra.SolvePlan(ra.ode.linear_dense(4, 3), s["W"], s["x0"], 0.0, 0.0125 * N, N, g, s["prior"], A=s["A"])
        plan.filter(None)
        mf, vf = plan.state_host()
        fo = scan.solve_filter(None, odes.make_linear_dense(s["A"], 3), s["W"], s["x0"], 0.0, 0.0125 * N, N, o, *s["prior"])
        em = np.abs(mf - fo["state_filt"][0]).max(axis=(0, 2, 3)); ev = np.abs(vf - fo["state_filt"][1]).max(axis=(0, 2, 3, 4))
        print(itg, "N", N, "filter err mean per step", em, "var", ev, "scale var", np.abs(fo["state_filt"][1]).max())
        plan.mv(None)
        m, v = plan.state_host()
        mo, vo = scan.solve_mv(None, odes.make_linear_dense(s["A"], 3), s["W"], s["x0"], 0.0, 0.0125 * N, N, o, s["prior"])
        print("     smooth err mean per step", np.abs(m - mo).max(axis=(0, 2, 3)), "var", np.abs(v - vo).max(axis=(0, 2, 3, 4)))
