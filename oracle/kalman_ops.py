"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  NumPy restatement of the covariance-form Kalman ops,
following src/rodeo/kalmantv/standard.py (line numbers in each function) and the LU solve of
src/rodeo/utils.py:105-119.

Every function accepts arbitrary leading batch dimensions on every array argument (NumPy broadcasting);
the trailing dims are the reference's: vectors (n_state,) / (n_meas,), matrices (n_state, n_state) etc.
Like the reference, unknown extra positional / keyword arguments are swallowed (standard.py:36).
"""
import numpy as np


def _mv(A, x):
    """Batched matrix-vector product A @ x with x a (..., n) vector."""
    return np.matmul(A, x[..., None])[..., 0]


def _T(A):
    return np.swapaxes(A, -1, -2)


def solve_var(V, B):
    """X = V^{-1} B by LU with partial pivoting (utils.py:119 -> LAPACK gesv)."""
    V = np.asarray(V, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    lead = np.broadcast_shapes(V.shape[:-2], B.shape[:-2])
    V = np.broadcast_to(V, lead + V.shape[-2:])
    B = np.broadcast_to(B, lead + B.shape[-2:])
    return np.linalg.solve(V, B)


def predict(mean_state_past, var_state_past, mean_state, wgt_state, var_state, *args, **kwargs):
    """standard.py:57-59."""
    mean_state_pred = _mv(wgt_state, mean_state_past) + mean_state
    var_state_pred = np.matmul(np.matmul(wgt_state, var_state_past), _T(wgt_state)) + var_state
    return mean_state_pred, var_state_pred


def update(mean_state_pred, var_state_pred, x_meas, mean_meas, wgt_meas, var_meas, *args, **kwargs):
    """standard.py:93-102."""
    mean_meas_pred = _mv(wgt_meas, mean_state_pred) + mean_meas
    var_meas_state_pred = np.matmul(wgt_meas, var_state_pred)
    var_meas_meas_pred = np.matmul(np.matmul(wgt_meas, var_state_pred), _T(wgt_meas)) + var_meas
    var_state_meas_pred = np.matmul(var_state_pred, _T(wgt_meas))
    var_state_temp = _T(solve_var(var_meas_meas_pred, _T(var_state_meas_pred)))
    mean_state_filt = mean_state_pred + _mv(var_state_temp, x_meas - mean_meas_pred)
    var_state_filt = var_state_pred - np.matmul(var_state_temp, var_meas_state_pred)
    return mean_state_filt, var_state_filt


def filter(mean_state_past, var_state_past, mean_state, wgt_state, var_state,
           x_meas, mean_meas, wgt_meas, var_meas, *args, **kwargs):
    """standard.py:142-157."""
    mean_state_pred, var_state_pred = predict(mean_state_past, var_state_past, mean_state, wgt_state, var_state)
    mean_state_filt, var_state_filt = update(mean_state_pred, var_state_pred, x_meas, mean_meas, wgt_meas, var_meas)
    return mean_state_pred, var_state_pred, mean_state_filt, var_state_filt


def _smooth(var_state_filt, var_state_pred, wgt_state):
    """standard.py:175-176."""
    var_state_temp = np.matmul(var_state_filt, _T(wgt_state))
    var_state_temp_tilde = _T(solve_var(var_state_pred, _T(var_state_temp)))
    return var_state_temp, var_state_temp_tilde


def smooth_mv(mean_state_next, var_state_next, mean_state_filt, var_state_filt,
              mean_state_pred, var_state_pred, wgt_state, *args, **kwargs):
    """standard.py:210-216."""
    _, G = _smooth(var_state_filt, var_state_pred, wgt_state)
    mean_state_smooth = mean_state_filt + _mv(G, mean_state_next - mean_state_pred)
    var_state_smooth = var_state_filt + np.matmul(np.matmul(G, var_state_next - var_state_pred), _T(G))
    return mean_state_smooth, var_state_smooth


def smooth_sim(x_state_next, mean_state_filt, var_state_filt, mean_state_pred, var_state_pred,
               wgt_state, *args, **kwargs):
    """standard.py:248-254."""
    T, G = _smooth(var_state_filt, var_state_pred, wgt_state)
    mean_state_sim = mean_state_filt + _mv(G, x_state_next - mean_state_pred)
    var_state_sim = var_state_filt - np.matmul(G, _T(T))
    return mean_state_sim, var_state_sim


def smooth(x_state_next, mean_state_next, var_state_next, mean_state_filt, var_state_filt,
           mean_state_pred, var_state_pred, wgt_state, *args, **kwargs):
    """standard.py:290-305."""
    T, G = _smooth(var_state_filt, var_state_pred, wgt_state)
    mean_state_sim = mean_state_filt + _mv(G, x_state_next - mean_state_pred)
    var_state_sim = var_state_filt - np.matmul(G, _T(T))
    mean_state_smooth = mean_state_filt + _mv(G, mean_state_next - mean_state_pred)
    var_state_smooth = var_state_filt + np.matmul(np.matmul(G, var_state_next - var_state_pred), _T(G))
    return mean_state_sim, var_state_sim, mean_state_smooth, var_state_smooth


def forecast(mean_state_pred, var_state_pred, mean_meas, wgt_meas, var_meas, *args, **kwargs):
    """standard.py:333-335."""
    mean_fore = _mv(wgt_meas, mean_state_pred) + mean_meas
    var_fore = np.matmul(np.matmul(wgt_meas, var_state_pred), _T(wgt_meas)) + var_meas
    return mean_fore, var_fore


def smooth_cond(mean_state_filt, var_state_filt, mean_state_pred, var_state_pred, wgt_state, *args, **kwargs):
    """standard.py:366-370."""
    T, wgt_state_cond = _smooth(var_state_filt, var_state_pred, wgt_state)
    mean_state_cond = mean_state_filt - _mv(wgt_state_cond, mean_state_pred)
    var_state_cond = var_state_filt - np.matmul(wgt_state_cond, _T(T))
    return wgt_state_cond, mean_state_cond, var_state_cond
