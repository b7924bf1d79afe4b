"""
rodeo_amd -- MI355X-native Kalman filter / smoother time-stepping core with rodeo's API surface
(``rodeo.solve_mv``, ``rodeo.solve_sim``, ``rodeo.kalmantv``, ``rodeo.interrogate``, ``rodeo.prior``;
src/rodeo/__init__.py:3-6).  Host code is plain Python calling hand-written HIP kernels through the C ABI of
include/rodeo_kalman.h (ctypes); there is no CPU fallback.
"""
__version__ = "0.1.0"
from . import interrogate
from . import prior
from . import kalmantv
from . import ode
from . import utils
from . import inference
from .solve import solve_sim, solve_mv, SolvePlan
from .prior.ibm import ibm_init
from .prior.indep_init import indep_init
from .device import Device, DeviceArray, default_device
