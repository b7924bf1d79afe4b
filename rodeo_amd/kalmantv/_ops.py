"""
Shared implementation of ``rodeo_amd.kalmantv.standard`` and ``.square_root``: marshal keyword arrays (any leading
batch dims, NumPy broadcasting = the reference's ``jax.vmap``) into the batch-minor device layout and call the
``rk_kalman_*_batched`` entry points (rodeo_amd/csrc/kalman_batched.hip).  No arithmetic happens on the host.
"""
import ctypes as C
import numpy as np
from .. import _lib
from ..device import default_device


def _lead(arrs_nd):
    shapes = [np.shape(a)[:np.ndim(a) - nd] for a, nd in arrs_nd if a is not None]
    return np.broadcast_shapes(*shapes) if shapes else ()


def _up(dev, a, trailing, lead):
    """Broadcast to lead + trailing, flatten lead to n, move n last, upload.  None -> NULL pointer."""
    if a is None:
        return None
    a = np.broadcast_to(np.asarray(a, dtype=np.float64), lead + trailing)
    n = int(np.prod(lead, dtype=np.int64)) if lead else 1
    a = a.reshape((n,) + trailing)
    return dev.to_device(np.ascontiguousarray(np.moveaxis(a, 0, -1)))


def _down(arr, trailing, lead):
    return np.moveaxis(arr.to_host(), -1, 0).reshape(lead + trailing)


def call(fname, kalman_type, n_state, n_meas, inputs, outputs):
    """
    inputs : list of (array-or-None, trailing shape); outputs: list of trailing shapes.
    Returns a tuple of host arrays of shape lead + trailing.
    """
    dev = default_device()
    lead = _lead([(a, len(tr)) for a, tr in inputs])
    n = int(np.prod(lead, dtype=np.int64)) if lead else 1
    cfg = _lib.OpCfg(n=n, n_state=n_state, n_meas=n_meas, kalman_type=kalman_type)
    din = [_up(dev, a, tuple(tr), lead) for a, tr in inputs]
    dout = [dev.empty(tuple(tr) + (n,)) for tr in outputs]
    args = [d.ptr if d is not None else None for d in din] + [d.ptr for d in dout]
    _lib.check(getattr(dev.lib, fname)(dev.h, C.byref(cfg), *args))
    return tuple(_down(d, tuple(tr), lead) for d, tr in zip(dout, outputs))


def make_module_functions(kalman_type, require_var_state):
    """Build the nine ops for one kalman_type; returns a dict name -> function."""
    def _dims_state(a):
        return np.shape(a)[-1]

    def predict(mean_state_past, var_state_past, mean_state, wgt_state, var_state, *args, **kwargs):
        p = _dims_state(mean_state_past)
        return call("rk_kalman_predict_batched", kalman_type, p, 0,
                    [(mean_state_past, (p,)), (var_state_past, (p, p)), (mean_state, (p,)), (wgt_state, (p, p)),
                     (var_state, (p, p))], [(p,), (p, p)])

    def update(mean_state_pred, var_state_pred, x_meas, mean_meas, wgt_meas, var_meas, *args, **kwargs):
        p = _dims_state(mean_state_pred)
        m = np.shape(wgt_meas)[-2]
        return call("rk_kalman_update_batched", kalman_type, p, m,
                    [(mean_state_pred, (p,)), (var_state_pred, (p, p)), (x_meas, (m,)), (mean_meas, (m,)),
                     (wgt_meas, (m, p)), (var_meas, (m, m))], [(p,), (p, p)])

    def filter(mean_state_past, var_state_past, mean_state, wgt_state, var_state, x_meas, mean_meas, wgt_meas,
               var_meas, *args, **kwargs):
        p = _dims_state(mean_state_past)
        m = np.shape(wgt_meas)[-2]
        return call("rk_kalman_filter_batched", kalman_type, p, m,
                    [(mean_state_past, (p,)), (var_state_past, (p, p)), (mean_state, (p,)), (wgt_state, (p, p)),
                     (var_state, (p, p)), (x_meas, (m,)), (mean_meas, (m,)), (wgt_meas, (m, p)), (var_meas, (m, m))],
                    [(p,), (p, p), (p,), (p, p)])

    def _vs(var_state, fname):
        if require_var_state and var_state is None:
            # src/rodeo/kalmantv/square_root.py:185,228: var_state is a required argument of the sqrt smoothers
            raise TypeError(f"{fname}() missing required argument: 'var_state'")
        return var_state

    def smooth_mv(mean_state_next, var_state_next, mean_state_filt, var_state_filt, mean_state_pred, var_state_pred,
                  wgt_state, var_state=None, *args, **kwargs):
        p = _dims_state(mean_state_filt)
        return call("rk_kalman_smooth_mv_batched", kalman_type, p, 0,
                    [(mean_state_next, (p,)), (var_state_next, (p, p)), (mean_state_filt, (p,)),
                     (var_state_filt, (p, p)), (mean_state_pred, (p,)), (var_state_pred, (p, p)), (wgt_state, (p, p)),
                     (_vs(var_state, "smooth_mv"), (p, p))], [(p,), (p, p)])

    def smooth_sim(x_state_next, mean_state_filt, var_state_filt, mean_state_pred, var_state_pred, wgt_state,
                   var_state=None, *args, **kwargs):
        p = _dims_state(mean_state_filt)
        return call("rk_kalman_smooth_sim_batched", kalman_type, p, 0,
                    [(x_state_next, (p,)), (mean_state_filt, (p,)), (var_state_filt, (p, p)), (mean_state_pred, (p,)),
                     (var_state_pred, (p, p)), (wgt_state, (p, p)), (_vs(var_state, "smooth_sim"), (p, p))],
                    [(p,), (p, p)])

    def smooth(x_state_next, mean_state_next, var_state_next, mean_state_filt, var_state_filt, mean_state_pred,
               var_state_pred, wgt_state, var_state=None, *args, **kwargs):
        p = _dims_state(mean_state_filt)
        return call("rk_kalman_smooth_batched", kalman_type, p, 0,
                    [(x_state_next, (p,)), (mean_state_next, (p,)), (var_state_next, (p, p)), (mean_state_filt, (p,)),
                     (var_state_filt, (p, p)), (mean_state_pred, (p,)), (var_state_pred, (p, p)), (wgt_state, (p, p)),
                     (_vs(var_state, "smooth"), (p, p))], [(p,), (p, p), (p,), (p, p)])

    def forecast(mean_state_pred, var_state_pred, mean_meas, wgt_meas, var_meas, *args, **kwargs):
        p = _dims_state(mean_state_pred)
        m = np.shape(wgt_meas)[-2]
        return call("rk_kalman_forecast_batched", kalman_type, p, m,
                    [(mean_state_pred, (p,)), (var_state_pred, (p, p)), (mean_meas, (m,)), (wgt_meas, (m, p)),
                     (var_meas, (m, m))], [(m,), (m, m)])

    def smooth_cond(mean_state_filt, var_state_filt, mean_state_pred, var_state_pred, wgt_state, var_state=None,
                    *args, **kwargs):
        p = _dims_state(mean_state_filt)
        return call("rk_kalman_smooth_cond_batched", kalman_type, p, 0,
                    [(mean_state_filt, (p,)), (var_state_filt, (p, p)), (mean_state_pred, (p,)),
                     (var_state_pred, (p, p)), (wgt_state, (p, p)), (_vs(var_state, "smooth_cond"), (p, p))],
                    [(p, p), (p,), (p, p)])

    return dict(predict=predict, update=update, filter=filter, smooth_mv=smooth_mv, smooth_sim=smooth_sim,
                smooth=smooth, forecast=forecast, smooth_cond=smooth_cond)
