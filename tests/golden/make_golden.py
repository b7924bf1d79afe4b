"""Writes tests/golden/oracle_fn_small.npz from oracle/ (seeded, deterministic).  Run from the repository root."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import scan, odes, priors, interrogations as oi

theta = np.array([0.2, 0.2, 3.0])
W = np.zeros((2, 1, 3)); W[:, :, 1] = 1.0                               # first_order_pad (utils.py:94-96)
x0v = np.array([-1.0, 1.0])
f0 = odes.fitzhugh_nagumo(np.concatenate([x0v[:, None], np.zeros((2, 2))], axis=1), 0.0, theta=theta)
x0 = np.stack([x0v, f0[:, 0], np.zeros(2)], axis=1)
N, t_max = 40, 2.0
prior = priors.ibm_init(t_max / N, 3, np.array([0.1, 0.1]))
out = {"theta": theta, "W": W, "x0": x0, "N": N, "t_max": t_max, "Q": prior[0], "R": prior[1]}
for name, g in (("kramer", oi.interrogate_kramer), ("rodeo", oi.interrogate_rodeo), ("schober", oi.interrogate_schober)):
    m, v = scan.solve_mv(None, odes.fitzhugh_nagumo, W, x0, 0.0, t_max, N, g, prior, theta=theta)
    filt = scan.solve_filter(None, odes.fitzhugh_nagumo, W, x0, 0.0, t_max, N, g, *prior, theta=theta)
    out[f"mv_mean_{name}"], out[f"mv_var_{name}"] = m, v
    out[f"filt_mean_{name}"], out[f"filt_var_{name}"] = filt["state_filt"]
out["sim_rodeo_seed5"] = scan.solve_sim(5, odes.fitzhugh_nagumo, W, x0, 0.0, t_max, N, oi.interrogate_rodeo, prior, theta=theta)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_fn_small.npz"), **out)
print("written", {k: np.shape(v) for k, v in out.items()})
