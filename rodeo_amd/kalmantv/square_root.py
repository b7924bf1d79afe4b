"""
Square-root Kalman filtering and smoothing -- the drop-in for ``rodeo.kalmantv.square_root``
(src/rodeo/kalmantv/square_root.py): the same nine names; every ``var_*`` argument / return value is a lower
square-root factor (``forecast`` returns the full variance, square_root.py:343-344) and the smoothers REQUIRE
``var_state`` (square_root.py:185,228).  ``add_sqrt`` (src/rodeo/utils.py:10-24) is a Householder QR on the device.
"""
from .. import _lib
from ._ops import make_module_functions

KALMAN_TYPE = "square-root"
globals().update(make_module_functions(_lib.KALMAN_SQRT, require_var_state=True))
