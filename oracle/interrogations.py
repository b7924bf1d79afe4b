"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  Restatement of src/rodeo/interrogate.py.

Signature is the reference's keyword protocol (src/rodeo/solve.py:70-78):
    interrogate(key=, ode_fun=, ode_weight=, t=, mean_state_pred=, var_state_pred=, **params)
        -> (wgt_meas (.., d, m, p), mean_meas (.., d, m), var_meas (.., d, m, m))
with optional leading batch dims on mean_state_pred / var_state_pred / params.

``ode_fun`` must be an ``oracle.odes.ODE`` for ``interrogate_kramer`` (it supplies the block-diagonal of
``jax.jacfwd``, interrogate.py:76-79); plain callables work for the other three.

``key`` for ``interrogate_chkrebtii`` is a ``StepKey`` (seed, global trajectory indices, step) addressing the
Philox stream of oracle/counter_rng.py, or ``z=`` may inject the standard normals directly.
"""
from collections import namedtuple
import numpy as np
from . import counter_rng

StepKey = namedtuple("StepKey", ["seed", "traj", "step"])


def _T(A):
    return np.swapaxes(A, -1, -2)


def psd_factor(A):
    """
    Lower-triangular F with F F^T = A for symmetric positive *semi*-definite A (reads the lower triangle),
    column-by-column Cholesky in which a non-positive pivot zeroes its column instead of producing NaN.
    This is the factor the HIP kernels use; for SPD input it is the Cholesky factor that
    jax.random.multivariate_normal(method='cholesky') uses (interrogate.py:30-34).
    """
    A = np.asarray(A, dtype=np.float64)
    p = A.shape[-1]
    L = np.zeros_like(A)
    for j in range(p):
        d = A[..., j, j] - np.sum(L[..., j, :j] ** 2, axis=-1)
        ok = d > 0.0
        dj = np.sqrt(np.where(ok, d, 1.0))
        L[..., j, j] = np.where(ok, dj, 0.0)
        for i in range(j + 1, p):
            s = A[..., i, j] - np.sum(L[..., i, :j] * L[..., j, :j], axis=-1)
            L[..., i, j] = np.where(ok, s / dj, 0.0)
    return L


def _wvw(ode_weight, var_state_pred):
    return np.matmul(np.matmul(ode_weight, var_state_pred), _T(ode_weight))


def interrogate_rodeo(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, **params):
    """interrogate.py:109-115."""
    var_meas = _wvw(ode_weight, var_state_pred)
    mean_meas = -ode_fun(mean_state_pred, t, **params)
    return np.zeros(np.shape(ode_weight)), mean_meas, var_meas


def interrogate_schober(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, **params):
    """interrogate.py:59-62."""
    n_block, n_bmeas, _ = np.shape(ode_weight)[-3:]
    mean_meas = -ode_fun(mean_state_pred, t, **params)
    var_meas = np.zeros(mean_meas.shape[:-2] + (n_block, n_bmeas, n_bmeas))
    return np.zeros(np.shape(ode_weight)), mean_meas, var_meas


def interrogate_kramer(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, **params):
    """interrogate.py:74-84.  wgt_meas = -J, mean_meas = -f + J mu, var_meas = 0."""
    n_block, n_bmeas, _ = np.shape(ode_weight)[-3:]
    fun_meas = -ode_fun(mean_state_pred, t, **params)
    jac = ode_fun.jac(mean_state_pred, t, **params)                         # (.., d, m, p) block diagonal
    wgt_meas = -jac
    mean_meas = fun_meas + np.matmul(jac, np.asarray(mean_state_pred)[..., None])[..., 0]
    var_meas = np.zeros(mean_meas.shape[:-2] + (n_block, n_bmeas, n_bmeas))
    return wgt_meas, mean_meas, var_meas


def interrogate_chkrebtii(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred,
                          kalman_type="standard", z=None, **params):
    """interrogate.py:22-47 (standard branch) and :35-42 (square-root branch)."""
    mean_state_pred = np.asarray(mean_state_pred, dtype=np.float64)
    n_block, n_bstate = mean_state_pred.shape[-2:]
    if z is None:
        z = counter_rng.normals(key.seed, key.traj, key.step, n_block, n_bstate, counter_rng.PURPOSE_INTERROGATE)
        z = z.reshape(mean_state_pred.shape)
    if kalman_type == "standard":
        var_meas = _wvw(ode_weight, var_state_pred)
        x_state = mean_state_pred + np.matmul(psd_factor(var_state_pred), z[..., None])[..., 0]
    elif kalman_type == "square-root":
        # reference quirk (interrogate.py:36-42): var_meas = W L has shape (m, p), and the draw is mu + (W L) z
        var_meas = np.matmul(ode_weight, var_state_pred)
        # factors are unique only up to column signs (QR): draws use the sign-normalised factor (diag >= 0) so that the
        # device and this oracle sample the same path whatever signs their Householder steps produce
        sgn = np.where(np.einsum("...ii->...i", var_state_pred) < 0, -1.0, 1.0)
        x_state = mean_state_pred + np.matmul(var_meas, (sgn * z)[..., None])[..., 0]
    else:
        raise NotImplementedError
    mean_meas = -ode_fun(x_state, t, **params)
    return np.zeros(np.shape(ode_weight)), mean_meas, var_meas
