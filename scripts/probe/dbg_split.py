import os, sys, functools
import numpy as np
sys.path.insert(0, os.getcwd())
import rodeo_amd as ra
theta = np.array([0.2, 0.2, 3.0])
W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
B = 12
rng = np.random.default_rng(1)
x0 = init(np.array([-1., 1.]) + 0.1 * rng.standard_normal((B, 2)), 0.0, theta=theta)
N, t_max = 100, 5.0
prior = ra.ibm_init(t_max / N, 3, np.array([.1, .1]))
g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard")
out = {}
for v in ("1", "0"):
    os.environ["RK_TILE3_SPLIT"] = v
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, t_max, N, g, prior, theta=theta)
    plan.filter(7)
    out[v] = plan.state_host()
dm = np.abs(out["1"][0] - out["0"][0]).max(axis=(0, 2, 3))
dv = np.abs(out["1"][1] - out["0"][1]).max(axis=(0, 2, 3, 4))
print("mean diff per step:", np.array2string(dm[:40], precision=2))
print("var  diff per step:", np.array2string(dv[:40], precision=2))
np.set_printoptions(linewidth=200)
d = np.abs(out["1"][0] - out["0"][0])          # (B, N+1, 2, 3)
print("per step max (first 24):", np.array2string(d.max(axis=(0, 2, 3))[:24], precision=3, floatmode="maxprec"))
print("traj 0, steps 3..8, block 0/1:\n", out["1"][0][0, 3:9], "\n", out["0"][0][0, 3:9])
print("which trajectories differ at step 5:", d[:, 5].max(axis=(1, 2)))
print("signed diff traj 0 steps 0..4 (split - one-wave):\n", (out["1"][0] - out["0"][0])[0, :5])
print("signed diff traj 1 steps 0..4:\n", (out["1"][0] - out["0"][0])[1, :5])
