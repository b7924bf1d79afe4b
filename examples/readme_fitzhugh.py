"""
The quick-start of the reference's README (README.md:88-152) on this build, call for call: NumPy in place of
jax.numpy, an integer seed in place of a PRNG key -- and the same plain Python ``fitz_fun``.  The scan runs on the GPU,
so the function is traced once into device code (rodeo_amd/trace.py, the counterpart of JAX tracing it inside
``lax.scan``) and compiled with hiprtc; the Jacobian that ``interrogate_kramer`` needs comes from forward-mode duals
(``jax.jacfwd`` in src/rodeo/interrogate.py:76).

    python examples/readme_fitzhugh.py            (needs an MI355X; prints the error against scipy's odeint)
"""
import os
import sys
import numpy as np
from scipy.integrate import odeint
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as rodeo


def fitz_fun(X, t, **params):
    # V' = c (V - V^3/3 + R),  R' = -(V - a + b R) / c;  X[:, 0] holds (V, R), the return value one row per variable
    a, b, c = params["theta"]
    V, R = X[:, 0]
    return np.array(
        [[c * (V - V * V * V / 3 + R)],
         [-1 / c * (V - a + b * R)]]
    )


def main():
    n_vars, n_deriv = 2, 3
    x0 = np.array([-1., 1.])
    theta = np.array([.2, .2, 3])
    W, fitz_init_pad = rodeo.utils.first_order_pad(fitz_fun, n_vars, n_deriv)
    X0 = fitz_init_pad(x0, 0., theta=theta)          # (n_vars, n_deriv): [x0, f(x0), 0]
    t_min, t_max = 0., 40.
    sigma = np.array([.1] * n_vars)
    n_steps = 800
    dt = (t_max - t_min) / n_steps
    prior_pars = rodeo.prior.ibm_init(dt=dt, n_deriv=n_deriv, sigma=sigma)
    key = 0
    Xt, _ = rodeo.solve_mv(
        key=key,
        ode_fun=fitz_fun,                  # the plain Python function above
        ode_weight=W,
        ode_init=X0,
        t_min=t_min,
        t_max=t_max,
        theta=theta,
        n_steps=n_steps,
        interrogate=rodeo.interrogate.interrogate_kramer,
        prior_pars=prior_pars
    )
    # the built-in device functor of the same ODE gives the same numbers
    Xb, _ = rodeo.solve_mv(key, rodeo.ode.fitzhugh_nagumo, W, X0, t_min, t_max, n_steps,
                           rodeo.interrogate.interrogate_kramer, prior_pars, theta=theta)
    tseq = np.linspace(t_min, t_max, n_steps + 1)
    exact = odeint(lambda X, t: [theta[2] * (X[0] - X[0] ** 3 / 3 + X[1]), -(X[0] - theta[0] + theta[1] * X[1]) / theta[2]],
                   x0, tseq, rtol=1e-10, atol=1e-10)
    err = float(np.max(np.abs(Xt[:, :, 0] - exact)))
    print(f"solve_mv: output {Xt.shape}, max |rodeo - odeint| = {err:.3e}, "
          f"traced Python function vs built-in: {float(np.max(np.abs(Xt - Xb))):.1e}")
    return err


if __name__ == "__main__":
    main()
