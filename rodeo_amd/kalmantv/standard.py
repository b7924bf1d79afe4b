"""
Time-varying Kalman filtering and smoothing, covariance form -- the drop-in for ``rodeo.kalmantv.standard``
(src/rodeo/kalmantv/standard.py): ``predict, update, filter, smooth_mv, smooth_sim, smooth, forecast, smooth_cond``
with the reference's keyword names.  Every array may carry arbitrary leading batch dims (broadcast like
``jax.vmap``); the arithmetic runs on the GPU (``rk_kalman_*_batched``).  Like the reference, the ops swallow
unknown extra arguments (standard.py:36), and the smoothers ignore ``var_state`` (solve.py:177,271).
"""
from .. import _lib
from ._ops import make_module_functions

KALMAN_TYPE = "standard"
globals().update(make_module_functions(_lib.KALMAN_STANDARD, require_var_state=False))
