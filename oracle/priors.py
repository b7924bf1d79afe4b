"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  Restatement of the prior / padding helpers:
src/rodeo/prior/ibm.py:21-88, src/rodeo/prior/indep_init.py:8-23, src/rodeo/utils.py:80-102.
"""
import numpy as np
from scipy.special import gammaln
from scipy.linalg import block_diag


def _factorial(x):
    """ibm.py:21-34: the reference's factorial is exp(gammaln(x+1)) (so not exactly integral)."""
    with np.errstate(over="ignore"):
        return np.exp(gammaln(np.asarray(x, dtype=np.float64) + 1.0))


def ibm_state(dt, q, sigma):
    """ibm.py:54-62.  Q_ij = 1[i<=j] dt^(j-i)/(j-i)!, R_ij = s^2 dt^(2q+1-i-j)/((2q+1-i-j)(q-i)!(q-j)!)."""
    I, J = np.meshgrid(np.arange(q + 1), np.arange(q + 1), indexing="ij", sparse=True)
    mesh = (J - I).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        Q = np.nan_to_num(float(dt) ** mesh / _factorial(mesh), nan=0.0)
    mesh = (2.0 * q + 1.0) - I - J
    num = float(dt) ** mesh
    den = mesh * _factorial(q - I) * _factorial(q - J)
    R = sigma ** 2 * num / den
    return Q, R


def ibm_init(dt, n_deriv, sigma):
    """ibm.py:81-88.  sigma (n_block,) -> Q (n_block,p,p), R (n_block,p,p)."""
    sigma = np.asarray(sigma, dtype=np.float64)
    n_block = len(sigma)
    Q1, R1 = ibm_state(dt, n_deriv - 1, 1)
    wgt_state = np.repeat(Q1[None], n_block, axis=0)
    var_state = np.stack([sigma[b] ** 2 * R1 for b in range(n_block)])
    return wgt_state, var_state


def indep_init(prior_pars):
    """indep_init.py:20-23: block_diag all blocks into a single dense block."""
    prior_weight, prior_var = prior_pars
    return block_diag(*prior_weight)[None, :], block_diag(*prior_var)[None, :]


def first_order_pad(ode_fun, n_vars, n_deriv):
    """utils.py:96-102.  W (n_vars,1,n_deriv) with W[:,:,1]=1; ode_init(x0,t) = [x0, f(x0,t), 0...]."""
    def ode_init(x0, t, **params):
        x0 = np.asarray(x0, dtype=np.float64)[:, None]
        return np.hstack([x0, ode_fun(x0, t, **params), np.zeros((n_vars, n_deriv - 2))])

    W = np.zeros((n_vars, 1, n_deriv))
    W[:, :, 1] = 1.0
    return W, ode_init
