import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import rodeo_amd as ra
B, N = 256, 200
t_max = N / 2000.0
rng = np.random.default_rng(5)
kc = np.array([0.7, 0.2]) * np.exp(0.05 * rng.standard_normal((B, 2)))
for D in (8, 16, 32):
  for with_sin in (True, False):
    def ring(X, t, **params):
        k, c = params["kc"]
        x = X[:, 0]
        return np.array([[k * (x[(i + 1) % D] - 2 * x[i] + x[(i - 1) % D]) - c * x[i] ** 3 + (np.sin(t + i) if with_sin else 0.1 * i)] for i in range(D)])
    xv = rng.standard_normal((B, D))
    for p in (3, 5):
        dev_b = ra.ode.from_python(ring, D, kc=2)
        W, init = ra.utils.first_order_pad(dev_b, D, p)
        x0 = init(xv, 0.0, kc=kc)
        prior = ra.ibm_init(t_max / N, p, np.ones(D))
        plan = ra.SolvePlan(dev_b, W, x0, 0.0, t_max, N, ra.interrogate.interrogate_kramer, prior, kc=kc)
        plan.filter(None); plan.dev.sync()
        plan.dev.profile_enable(True)
        plan.filter(None); plan.dev.sync()
        print(D, "sin" if with_sin else "poly", p, {k: round(v, 3) for k, v in plan.dev.profile_last()})
        plan.dev.profile_enable(False)
