"""
Callers of the hot path (SURVEY.md section 8f "next-1"): ``basic`` (src/rodeo/inference/basic.py) and the device-side
Gaussian observation log-posterior reduction used by pseudo-marginal log-posteriors
(docs/examples/parameter.md:188-210, 331-354), and ``pseudo_marginal``: the random-walk Rosenbluth-Metropolis-Hastings
kernels of src/rodeo/inference/pseudo_marginal.py for many chains in lock-step (SURVEY.md section 8f "next-2").
``fenrir``: the Fenrir likelihood (src/rodeo/inference/fenrir.py:261-327, "next-4").  dalton / magi are out of scope.
"""
from .basic import basic
from .logpost import gauss_obs_logpost, obs_index, sim_logpost, stage_upars
from . import pseudo_marginal
from .fenrir import fenrir
