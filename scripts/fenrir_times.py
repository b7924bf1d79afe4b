"""Kernel times of inference.fenrir on the headline shape (FitzHugh-Nagumo, p = 3, 4000 steps, 1024 parameter sets,
41 observations per variable)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
import bench
from rodeo_amd.inference import fenrir
W, x0, theta, prior = bench.make_problem(ra, 0)
n_obs = 41
obs_t = np.linspace(0, 40, n_obs)
rng = np.random.default_rng(0)
Y = rng.standard_normal((n_obs, 2, 1))
Dw = np.zeros((n_obs, 2, 1, 3)); Dw[..., 0] = 1.0
Om = np.full((n_obs, 2, 1, 1), 0.005)
dev = ra.device.default_device()
for rep in range(3):
    dev.sync(); t0 = time.perf_counter()
    dev.profile_enable(True)
    ll = fenrir(None, ra.ode.fitzhugh_nagumo, W, x0, 0.0, 40.0, 4000, ra.interrogate.interrogate_kramer, prior, Y, obs_t, Dw, Om, theta=theta)
    dev.sync(); t1 = time.perf_counter()
    print("fenrir: wall ms %.2f" % ((t1 - t0) * 1e3), {k: round(v, 4) for k, v in dev.profile_last()}, ll[:2], flush=True)
