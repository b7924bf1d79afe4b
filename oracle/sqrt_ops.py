"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  NumPy restatement of the square-root Kalman ops of
src/rodeo/kalmantv/square_root.py; every ``var_*`` argument/return is a (lower) square-root factor, except
``forecast`` which returns the full variance (square_root.py:343-344).  ``add_sqrt`` follows
src/rodeo/utils.py:22-24 (reduced QR of the stacked transposes, returns R^T; no sign normalisation).
Arbitrary leading batch dims are accepted.
"""
import numpy as np
from scipy.linalg import solve_triangular as _st


def _mv(A, x):
    return np.matmul(A, x[..., None])[..., 0]


def _T(A):
    return np.swapaxes(A, -1, -2)


def add_sqrt(sqrt_A, sqrt_B):
    """utils.py:22-24.  sqrt_A (.., n, ka), sqrt_B (.., n, kb) -> (.., n, n) factor of A + B."""
    lead = np.broadcast_shapes(sqrt_A.shape[:-2], sqrt_B.shape[:-2])
    sqrt_A = np.broadcast_to(sqrt_A, lead + sqrt_A.shape[-2:])
    sqrt_B = np.broadcast_to(sqrt_B, lead + sqrt_B.shape[-2:])
    stacked = np.concatenate([_T(sqrt_A), _T(sqrt_B)], axis=-2)
    R = np.linalg.qr(stacked, mode="r")
    return _T(R)


def _solve_tri(L, B, lower):
    """Batched triangular solve L X = B."""
    lead = np.broadcast_shapes(L.shape[:-2], B.shape[:-2])
    L = np.broadcast_to(L, lead + L.shape[-2:]).reshape((-1,) + L.shape[-2:])
    B2 = np.broadcast_to(B, lead + B.shape[-2:]).reshape((-1,) + B.shape[-2:])
    X = np.stack([_st(L[i], B2[i], lower=lower) for i in range(L.shape[0])])
    return X.reshape(lead + B.shape[-2:])


def predict(mean_state_past, var_state_past, mean_state, wgt_state, var_state, *args, **kwargs):
    """square_root.py:56-57."""
    mean_state_pred = _mv(wgt_state, mean_state_past) + mean_state
    var_state_pred = add_sqrt(np.matmul(wgt_state, var_state_past), var_state)
    return mean_state_pred, var_state_pred


def update(mean_state_pred, var_state_pred, x_meas, mean_meas, wgt_meas, var_meas, *args, **kwargs):
    """square_root.py:88-99."""
    mean_meas_pred = _mv(wgt_meas, mean_state_pred) + mean_meas
    var_meas_meas_pred = add_sqrt(np.matmul(wgt_meas, var_state_pred), var_meas)
    inter = _solve_tri(var_meas_meas_pred, wgt_meas, lower=True)
    inter = np.matmul(np.matmul(inter, var_state_pred), _T(var_state_pred))
    var_state_temp = _T(_solve_tri(_T(var_meas_meas_pred), inter, lower=False))
    mean_state_filt = mean_state_pred + _mv(var_state_temp, x_meas - mean_meas_pred)
    var_state_filt = add_sqrt(var_state_pred - np.matmul(np.matmul(var_state_temp, wgt_meas), var_state_pred),
                              np.matmul(var_state_temp, var_meas))
    return mean_state_filt, var_state_filt


def filter(mean_state_past, var_state_past, mean_state, wgt_state, var_state,
           x_meas, mean_meas, wgt_meas, var_meas, *args, **kwargs):
    mean_state_pred, var_state_pred = predict(mean_state_past, var_state_past, mean_state, wgt_state, var_state)
    mean_state_filt, var_state_filt = update(mean_state_pred, var_state_pred, x_meas, mean_meas, wgt_meas, var_meas)
    return mean_state_pred, var_state_pred, mean_state_filt, var_state_filt


def _smooth(var_state_filt, var_state_pred, wgt_state):
    """square_root.py:170-175."""
    variance_state_filt = np.matmul(var_state_filt, _T(var_state_filt))
    inter = _solve_tri(var_state_pred, wgt_state, lower=True)
    inter = np.matmul(inter, variance_state_filt)
    return _T(_solve_tri(_T(var_state_pred), inter, lower=False))


def _J(G, wgt_state):
    return np.eye(G.shape[-1]) - np.matmul(G, wgt_state)


def smooth_mv(mean_state_next, var_state_next, mean_state_filt, var_state_filt,
              mean_state_pred, var_state_pred, wgt_state, var_state, *args, **kwargs):
    """square_root.py:209-219."""
    G = _smooth(var_state_filt, var_state_pred, wgt_state)
    mean_state_smooth = mean_state_filt + _mv(G, mean_state_next - mean_state_pred)
    lead = np.broadcast_shapes(var_state_next.shape[:-2], np.shape(var_state)[:-2])
    both = np.concatenate([np.broadcast_to(var_state_next, lead + var_state_next.shape[-2:]),
                           np.broadcast_to(var_state, lead + np.shape(var_state)[-2:])], axis=-1)
    var_state_smooth = add_sqrt(np.matmul(G, both), np.matmul(_J(G, wgt_state), var_state_filt))
    return mean_state_smooth, var_state_smooth


def smooth_sim(x_state_next, mean_state_filt, var_state_filt, mean_state_pred, var_state_pred,
               wgt_state, var_state, *args, **kwargs):
    """square_root.py:252-261."""
    G = _smooth(var_state_filt, var_state_pred, wgt_state)
    mean_state_sim = mean_state_filt + _mv(G, x_state_next - mean_state_pred)
    var_state_sim = add_sqrt(np.matmul(G, var_state), np.matmul(_J(G, wgt_state), var_state_filt))
    return mean_state_sim, var_state_sim


def smooth(x_state_next, mean_state_next, var_state_next, mean_state_filt, var_state_filt,
           mean_state_pred, var_state_pred, wgt_state, var_state, *args, **kwargs):
    """square_root.py:297-314."""
    ms, vs = smooth_sim(x_state_next, mean_state_filt, var_state_filt, mean_state_pred, var_state_pred,
                        wgt_state, var_state)
    mm, vm = smooth_mv(mean_state_next, var_state_next, mean_state_filt, var_state_filt,
                       mean_state_pred, var_state_pred, wgt_state, var_state)
    return ms, vs, mm, vm


def forecast(mean_state_pred, var_state_pred, mean_meas, wgt_meas, var_meas, *args, **kwargs):
    """square_root.py:342-345 -- NB returns the full variance, not a factor."""
    mean_fore = _mv(wgt_meas, mean_state_pred) + mean_meas
    f = add_sqrt(np.matmul(wgt_meas, var_state_pred), var_meas)
    return mean_fore, np.matmul(f, _T(f))


def smooth_cond(mean_state_filt, var_state_filt, mean_state_pred, var_state_pred, wgt_state, var_state,
                *args, **kwargs):
    """square_root.py:376-385."""
    G = _smooth(var_state_filt, var_state_pred, wgt_state)
    mean_state_cond = mean_state_filt - _mv(G, mean_state_pred)
    var_state_cond = add_sqrt(np.matmul(G, var_state), np.matmul(_J(G, wgt_state), var_state_filt))
    return G, mean_state_cond, var_state_cond
