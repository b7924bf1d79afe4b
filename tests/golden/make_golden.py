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


# ---- round 3: the paths added since (square-root form, n_deriv = 5, a dense block in both forms, fenrir in both forms) ----------
from scipy.linalg import block_diag
from oracle import fenrir as ofen

r3 = {}
cholR = np.linalg.cholesky(prior[1])
for name, g in (("kramer", oi.interrogate_kramer), ("rodeo", oi.interrogate_rodeo)):
    m, L = scan.solve_mv(None, odes.fitzhugh_nagumo, W, x0, 0.0, t_max, N, g, (prior[0], cholR), kalman_type="square-root", theta=theta)
    r3[f"sqrt_mv_mean_{name}"], r3[f"sqrt_mv_var_{name}"] = m, L @ np.swapaxes(L, -1, -2)       # (factors: unique up to signs)
# n_deriv = 5 on the same problem
W5 = np.zeros((2, 1, 5)); W5[:, :, 1] = 1.0
X5 = np.zeros((2, 5)); X5[:, 0] = x0v; X5[:, 1] = f0[:, 0]
# (second derivative of the initial value: the Jacobian of f applied to f, as first_order_pad's init does it -- taken from the oracle)
W5o, init5 = priors.first_order_pad(odes.fitzhugh_nagumo, 2, 5)
x05 = init5(x0v, 0.0, theta=theta)
prior5 = priors.ibm_init(t_max / N, 5, np.array([0.1, 0.1]))
m5, v5 = scan.solve_mv(None, odes.fitzhugh_nagumo, W5o, x05, 0.0, t_max, N, oi.interrogate_kramer, prior5, theta=theta)
r3.update(W5=W5o, x05=x05, Q5=prior5[0], R5=prior5[1], mv5_mean=m5, mv5_var=v5)
# a dense block: 4 variables x 3 derivatives, linear right-hand side x' = A x, both filter forms (exact measurement: same posterior)
rng = np.random.default_rng(3)
nv, nd, Nd, td = 4, 3, 12, 0.5
A = -np.diag(np.linspace(0.5, 2.0, nv)) + 0.1 * rng.standard_normal((nv, nv)) / np.sqrt(nv)
Wb, _ = priors.first_order_pad(lambda x, t: x, nv, nd)
Wd = block_diag(*[w for w in Wb])[None]
Qd, Rd = priors.indep_init(priors.ibm_init(td / Nd, nd, np.ones(nv)))
xv = np.ones(nv)
X0d = np.zeros((nv, nd)); X0d[:, 0] = xv; X0d[:, 1] = A @ xv
X0d = X0d.reshape(1, -1)
ode_d = odes.make_linear_dense(A, nd)
md, vd = scan.solve_mv(None, ode_d, Wd, X0d, 0.0, td, Nd, oi.interrogate_kramer, (Qd, Rd))
mds, Lds = scan.solve_mv(None, ode_d, Wd, X0d, 0.0, td, Nd, oi.interrogate_kramer, (Qd, np.linalg.cholesky(Rd)), kalman_type="square-root")
r3.update(dense_A=A, dense_W=Wd, dense_x0=X0d, dense_Q=Qd, dense_R=Rd, dense_N=Nd, dense_t=td, dense_mv_mean=md, dense_mv_var=vd,
          dense_sqrt_mv_mean=mds, dense_sqrt_mv_var=Lds @ np.swapaxes(Lds, -1, -2))
# fenrir: five observations of the first component of both blocks
obs_t = np.linspace(0.0, t_max, 5)
yobs = rng.standard_normal((5, 2, 1))
Dw = np.zeros((5, 2, 1, 3)); Dw[..., 0] = 1.0
Om = np.full((5, 2, 1, 1), 0.05)
r3.update(fen_obs_t=obs_t, fen_y=yobs, fen_D=Dw, fen_Om=Om)
r3["fen_ll_standard"] = ofen.fenrir(None, odes.fitzhugh_nagumo, W, x0, 0.0, t_max, N, oi.interrogate_kramer, prior, yobs, obs_t, Dw, Om, theta=theta)
r3["fen_ll_sqrt"] = ofen.fenrir(None, odes.fitzhugh_nagumo, W, x0, 0.0, t_max, N, oi.interrogate_kramer, (prior[0], cholR), yobs, obs_t, Dw,
                                np.sqrt(Om), kalman_type="square-root", theta=theta)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_round3.npz"), **r3)
print("written", {k: np.shape(v) for k, v in r3.items()})
