import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import rodeo_amd as ra
rng = np.random.default_rng(20240); B = 1024
eps = rng.standard_normal((B, 5)); theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.1 * eps[:, :3]); x0v = np.array([-1.0, 1.0]) + 0.1 * eps[:, 3:]
for p in (5, 8):
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p); x0 = init(x0v, 0.0, theta=theta); N = 4000
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, 40.0, N, ra.interrogate.interrogate_kramer, ra.ibm_init(40.0 / N, p, np.array([0.1, 0.1])), theta=theta)
    plan.mv(None); plan.dev.sync()
    d = plan._ws.to_host()[:10]
    print(p, "per step: chain work/wait", d[0] / N, d[1] / N, "producers work/wait", [(d[2 * w] / N, d[2 * w + 1] / N) for w in range(1, 5)])
