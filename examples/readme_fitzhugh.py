"""
The quick-start of the reference's README (README.md:88-152) on this build: same calls, same keywords, NumPy arrays
in place of jax.numpy, an integer seed in place of a PRNG key.  The one real difference is the ODE: the scan runs on
the GPU, so ``ode_fun`` is a device functor -- a built-in (``rodeo_amd.ode.fitzhugh_nagumo``) or HIP source compiled
at first use (``rodeo_amd.ode.from_source``; the Jacobian that ``interrogate_kramer`` needs comes from forward-mode
duals when the right-hand side is written on a generic scalar type, like ``jax.jacfwd`` in interrogate.py:76).

    python examples/readme_fitzhugh.py            (needs an MI355X; prints the error against scipy's odeint)
"""
import os
import sys
import numpy as np
from scipy.integrate import odeint
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as rodeo


# --- the ODE, written once on a generic scalar type (compare fitz_fun of the README) ------------------------------------
FITZ_SRC = r"""
struct Fitz {
    static constexpr int D = 2;            // number of variables (blocks)
    static constexpr int NTHETA = 3;       // length of the `theta` keyword argument
    static constexpr int NDEP = 1;         // f depends on X[:, 0] only
    template <class T, int P>
    __device__ static void rhs(const T (&X)[D][P], double t, const double (&th)[NTHETA], T (&out)[D]) {
        const double a = th[0], b = th[1], c = th[2];
        const T V = X[0][0], R = X[1][0];
        out[0] = c * (V - V * V * V / 3.0 + R);
        out[1] = -1.0 / c * (V - a + b * R);
    }
};
"""


def fitz_host(X, t, theta):
    """NumPy twin used on the host by first_order_pad (initial derivatives)."""
    a, b, c = np.moveaxis(np.asarray(theta, dtype=np.float64), -1, 0)
    V, R = X[..., 0, 0], X[..., 1, 0]
    return np.stack([c * (V - V * V * V / 3 + R), -1 / c * (V - a + b * R)], axis=-1)[..., None]


def main():
    fitz_fun = rodeo.ode.from_source("AutoJac<Fitz>", FITZ_SRC, 2, (("theta", 3),), fitz_host, name="readme_fitz")
    n_vars, n_deriv = 2, 3
    x0 = np.array([-1., 1.])
    theta = np.array([.2, .2, 3])
    W, fitz_init_pad = rodeo.utils.first_order_pad(fitz_fun, n_vars, n_deriv)
    X0 = fitz_init_pad(x0, 0., theta=theta)
    t_min, t_max = 0., 40.
    sigma = np.array([.1] * n_vars)
    n_steps = 800
    dt = (t_max - t_min) / n_steps
    prior_pars = rodeo.prior.ibm_init(dt=dt, n_deriv=n_deriv, sigma=sigma)
    key = 0
    Xt, _ = rodeo.solve_mv(key=key, ode_fun=fitz_fun, ode_weight=W, ode_init=X0, t_min=t_min, t_max=t_max, theta=theta,
                           n_steps=n_steps, interrogate=rodeo.interrogate.interrogate_kramer, prior_pars=prior_pars)
    # the built-in functor gives the same numbers (it is the same ODE)
    Xb, _ = rodeo.solve_mv(key, rodeo.ode.fitzhugh_nagumo, W, X0, t_min, t_max, n_steps,
                           rodeo.interrogate.interrogate_kramer, prior_pars, theta=theta)
    tseq = np.linspace(t_min, t_max, n_steps + 1)
    exact = odeint(lambda X, t: [theta[2] * (X[0] - X[0] ** 3 / 3 + X[1]), -(X[0] - theta[0] + theta[1] * X[1]) / theta[2]],
                   x0, tseq, rtol=1e-10, atol=1e-10)
    err = float(np.max(np.abs(Xt[:, :, 0] - exact)))
    print(f"solve_mv: output {Xt.shape}, max |rodeo - odeint| = {err:.3e}, "
          f"user source vs built-in: {float(np.max(np.abs(Xt - Xb))):.1e}")
    return err


if __name__ == "__main__":
    main()
