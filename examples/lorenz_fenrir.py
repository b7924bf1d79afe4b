"""
docs/examples/lorenz.md of the reference on this build: the chaotic Lorenz63 system solved without data
(``solve_mv``) and with noisy observations (``rodeo.inference.fenrir.solve_mv``), with the document's own Python
``lorenz`` function (traced into device code), its settings (n_deriv = 3, sigma = 5e7, 20 observations, 200 solver steps
between observations).  (The document's third solver, dalton, is not part of this build.)

    python examples/lorenz_fenrir.py        (needs an MI355X)
"""
import os
import sys
import numpy as np
from scipy.integrate import odeint
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as rodeo
from rodeo_amd.utils import first_order_pad
from rodeo_amd.prior import ibm_init
from rodeo_amd.interrogate import interrogate_kramer
from rodeo_amd.inference.fenrir import solve_mv as fsolve


def lorenz0(X_t, t, theta):
    rho, sigma, beta = theta
    x, y, z = X_t
    return np.array([-sigma * x + sigma * y, rho * x - y - x * z, -beta * z + x * y])


def lorenz(X_t, t, theta):
    rho, sigma, beta = theta
    x, y, z = X_t[:, 0]
    dx = -sigma * x + sigma * y
    dy = rho * x - y - x * z
    dz = -beta * z + x * y
    return np.array([[dx], [dy], [dz]])


def main():
    tmin, tmax = 0., 20.
    theta = np.array([28, 10, 8 / 3])
    ode0 = np.array([-12., -5., 38.])
    n_obs = 20
    obs_times = np.linspace(tmin, tmax, n_obs + 1)
    exact_obs = odeint(lorenz0, ode0, obs_times, args=(theta,), rtol=1e-12, atol=1e-12)
    gamma = np.sqrt(.005)
    obs = exact_obs + gamma * np.random.default_rng(0).normal(loc=0.0, scale=1, size=exact_obs.shape)

    n_deriv, n_vars = 3, 3
    sigma = np.array([5e7] * n_vars)
    W, lorenz_init_pad = first_order_pad(lorenz, n_vars, n_deriv)
    x0 = lorenz_init_pad(ode0, 0, theta=theta)
    n_res = 200
    n_steps = n_obs * n_res
    dt = (tmax - tmin) / n_steps
    prior_pars = ibm_init(dt, n_deriv, sigma)
    key = 0

    n_meas = 1
    obs_data = np.expand_dims(obs, -1)
    obs_weight = np.zeros((len(obs_data), n_vars, n_meas, n_deriv)); obs_weight[:, :, :, 0] = 1
    obs_var = np.zeros((len(obs_data), n_vars, n_meas, n_meas)); obs_var[:, :, :, 0] = gamma ** 2

    rsol, _ = rodeo.solve_mv(key, lorenz, W, x0, tmin, tmax, n_steps, interrogate_kramer, prior_pars, theta=theta)
    fsol, _ = fsolve(key, lorenz, W, x0, tmin, tmax, n_steps, interrogate_kramer, prior_pars,
                     obs_data, obs_times, obs_weight, obs_var, theta=theta)

    # The document's finding (lorenz.md, last paragraph): only dalton recovers the true solution beyond t > 7.5 -- the
    # data-free solver leaves the chaotic trajectory, Fenrir is pulled through the observations (it conditions on them)
    # but, linearised around the forward filter's path, does not follow the truth in between.
    idx = np.arange(n_obs + 1) * n_res
    at_obs_r = float(np.max(np.abs(rsol[idx, :, 0] - obs)[n_obs // 2:]))
    at_obs_f = float(np.max(np.abs(fsol[idx, :, 0] - obs)[1:]))
    tseq_sim = np.linspace(tmin, tmax, n_steps + 1)
    exact = odeint(lorenz0, ode0, tseq_sim, args=(theta,), rtol=1e-12, atol=1e-12)
    early = tseq_sim <= 5.0
    err_early = float(np.max(np.abs(rsol[early, :, 0] - exact[early])))
    print(f"solve_mv against odeint on t <= 5: {err_early:.3f};  distance to the observations on t >= 10: "
          f"solve_mv {at_obs_r:.2f}, fenrir.solve_mv {at_obs_f:.3f}")
    return err_early, at_obs_r, at_obs_f


if __name__ == "__main__":
    main()
