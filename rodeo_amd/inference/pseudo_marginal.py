"""
Random-walk Rosenbluth-Metropolis-Hastings kernels with auxiliary variables -- the host mirror of
``src/rodeo/inference/pseudo_marginal.py`` (itself a copy of ``blackjax.mcmc.random_walk`` whose log-density returns
``(logdensity, auxdata)``), run for MANY CHAINS IN LOCK-STEP: every array carries a leading chain axis C, one
``step`` is one batched log-density evaluation (on the device: ``SolvePlan.update`` + ``sim`` + the log-posterior
reduction, docs/examples/parameter.md:331-354, 383-390) plus C accept/reject decisions on the host.

Same names as the reference: ``RWAState``, ``RWAInfo``, ``init``, ``normal``, ``build_additive_step``,
``additive_step_random_walk``, ``normal_random_walk``, ``build_irmh``, ``irmh_as_top_level_api``, ``build_rmh``,
``rmh_as_top_level_api``, ``build_rmh_transition_energy``, ``rmh_proposal``.

Differences, all forced by the absence of JAX:
* ``rng_key`` is an integer; ``split(key, n)`` derives sub-keys with the library's Philox4x32-10 counter stream
  (oracle/counter_rng.py is the test mirror).  Proposal and acceptance draws therefore have blackjax's law, not its
  bit-stream -- **parity with blackjax is unpinned** (SURVEY.md section 8c); the tests pin this module against a plain
  restatement with injected draws and against exact targets.
* the acceptance rule is blackjax's ``compute_asymmetric_acceptance_ratio`` + ``static_binomial_sampling``
  (pinned versions in the reference's pyproject: blackjax <= 1.2.3 / unpinned): ``log p = E(prev -> new) -
  E(new -> prev)`` with NaN mapped to -inf, ``p = min(1, exp(log p))``, accept iff ``u < p`` for a uniform ``u``.
"""
from typing import Callable, NamedTuple, Optional
import numpy as np


class RWAState(NamedTuple):
    """State of the chains: position (C, dim), logdensity (C,), auxdata (anything indexable by chain, or None)."""
    position: np.ndarray
    logdensity: np.ndarray
    auxdata: object = None


class RWAInfo(NamedTuple):
    """pseudo_marginal.py:119-132.  ``proposal`` is what the reference's kernel puts there -- the state the chains are in
    AFTER the accept / reject step (pseudo_marginal.py:377: ``RWAInfo(p_accept, do_accept, new_state)``); the state that
    was proposed is kept in the additional last field ``proposed``."""
    acceptance_rate: np.ndarray
    is_accepted: np.ndarray
    proposal: RWAState
    proposed: Optional[RWAState] = None


# ---- keys: Philox4x32-10, the same generator the device uses for its draws (rodeo_amd/csrc/philox.hpp) ----
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def _philox(c, k0, k1):
    """c: (..., 4) uint64 holding 32-bit words; returns (..., 4)."""
    c0, c1, c2, c3 = (c[..., i].astype(np.uint64) for i in range(4))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(_M0) * c0, np.uint64(_M1) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask, p0 & mask
        k0, k1 = (k0 + np.uint64(_W0)) & mask, (k1 + np.uint64(_W1)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1)


def _words(key, purpose, n):
    """n x 4 random 32-bit words for (key, purpose)."""
    key = int(key) & 0xFFFFFFFFFFFFFFFF
    c = np.zeros((n, 4), dtype=np.uint64)
    c[:, 0] = np.arange(n, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    c[:, 1] = np.arange(n, dtype=np.uint64) >> np.uint64(32)
    c[:, 2] = np.uint64(purpose)
    return _philox(c, key & 0xFFFFFFFF, key >> 32)


def split(key, num=2):
    """``jax.random.split`` stand-in: ``num`` 64-bit sub-keys, a pure function of ``key``."""
    w = _words(key, 0x5EED, num)
    return [int(w[i, 0] | (w[i, 1] << np.uint64(32))) for i in range(num)]


def _u53(hi, lo):
    return (((hi << np.uint64(21)) | (lo >> np.uint64(11))).astype(np.float64) + 0.5) / 9007199254740992.0


def uniform(key, shape):
    n = int(np.prod(shape))
    w = _words(key, 0x0F1, (n + 1) // 2)
    u = np.stack([_u53(w[:, 0], w[:, 1]), _u53(w[:, 2], w[:, 3])], axis=-1).reshape(-1)[:n]
    return u.reshape(shape)


def standard_normal(key, shape):
    n = int(np.prod(shape))
    w = _words(key, 0x0A1, (n + 1) // 2)
    u1, u2 = _u53(w[:, 0], w[:, 1]), _u53(w[:, 2], w[:, 3])
    rad = np.sqrt(-2.0 * np.log(u1))
    z = np.stack([rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)], axis=-1).reshape(-1)[:n]
    return z.reshape(shape)


def normal(sigma) -> Callable:
    """``blackjax.mcmc.random_walk.normal``: a random step N(0, sigma) -- ``sigma`` (dim,) is a vector of standard
    deviations, ``sigma`` (dim, dim) a square-root factor applied as ``sigma @ z``."""
    sigma = np.asarray(sigma, dtype=np.float64)
    if sigma.ndim > 2:
        raise ValueError("sigma must be a vector of scales or a square factor")

    def propose(rng_key, position):
        z = standard_normal(rng_key, np.shape(position))
        return z * sigma if sigma.ndim < 2 else z @ sigma.T

    return propose


def init(position, logdensity_fn: Callable, rng_key) -> RWAState:
    """pseudo_marginal.py:135-149."""
    logdensity, auxdata = logdensity_fn(position, rng_key)
    return RWAState(np.asarray(position, dtype=np.float64), np.asarray(logdensity, dtype=np.float64), auxdata)


def compute_asymmetric_acceptance_ratio(transition_energy_fn: Callable) -> Callable:
    """blackjax.mcmc.proposal.compute_asymmetric_acceptance_ratio (restated; parity unpinned)."""
    def ratio(initial_state, state):
        initial_energy = transition_energy_fn(state, initial_state)
        new_energy = transition_energy_fn(initial_state, state)
        delta = np.asarray(initial_energy - new_energy, dtype=np.float64)
        return np.where(np.isnan(delta), -np.inf, delta)
    return ratio


def static_binomial_sampling(rng_key, log_p_accept, proposal: RWAState, new_proposal: RWAState):
    """blackjax.mcmc.proposal.static_binomial_sampling, per chain (restated; parity unpinned)."""
    with np.errstate(over="ignore"):
        p_accept = np.minimum(np.exp(log_p_accept), 1.0)
    do_accept = uniform(rng_key, p_accept.shape) < p_accept
    sel = lambda a, b: None if a is None else np.where(do_accept.reshape((-1,) + (1,) * (np.ndim(a) - 1)), b, a)
    state = RWAState(sel(proposal.position, new_proposal.position), sel(proposal.logdensity, new_proposal.logdensity),
                     _select_aux(do_accept, proposal.auxdata, new_proposal.auxdata))
    return state, (do_accept, p_accept, None)


def _select_aux(mask, old, new):
    if old is None or new is None:
        return new if old is None else old
    if isinstance(old, dict):
        return {k: _select_aux(mask, old[k], new[k]) for k in old}
    if isinstance(old, (tuple, list)):
        return type(old)(_select_aux(mask, a, b) for a, b in zip(old, new))
    a, b = np.asarray(old), np.asarray(new)
    return np.where(mask.reshape((-1,) + (1,) * (a.ndim - 1)), b, a)


def build_rmh_transition_energy(proposal_logdensity_fn: Optional[Callable]) -> Callable:
    """pseudo_marginal.py:438-449."""
    if proposal_logdensity_fn is None:
        def transition_energy(prev_state, new_state):
            return -new_state.logdensity
    else:
        def transition_energy(prev_state, new_state):
            return -new_state.logdensity - proposal_logdensity_fn(new_state, prev_state)
    return transition_energy


def rmh_proposal(logdensity_fn: Callable, transition_distribution: Callable, compute_acceptance_ratio: Callable,
                 sample_proposal: Callable = static_binomial_sampling) -> Callable:
    """pseudo_marginal.py:452-483: ``generate(rng_key, previous_state) -> (accepted_state, do_accept, p_accept)``.  The
    proposed state of the last call is kept on the function (``generate.last_proposed``) for ``RWAInfo.proposed``."""
    def generate(rng_key, previous_state: RWAState):
        key_proposal, key_accept, key_logdensity = split(rng_key, 3)
        position = previous_state.position
        new_position = transition_distribution(key_proposal, position)
        new_logdensity, new_auxdata = logdensity_fn(new_position, key_logdensity)
        proposed_state = RWAState(np.asarray(new_position, dtype=np.float64),
                                  np.asarray(new_logdensity, dtype=np.float64), new_auxdata)
        log_p_accept = compute_acceptance_ratio(previous_state, proposed_state)
        accepted_state, info = sample_proposal(key_accept, log_p_accept, previous_state, proposed_state)
        do_accept, p_accept, _ = info
        generate.last_proposed = proposed_state
        return accepted_state, do_accept, p_accept
    generate.last_proposed = None
    return generate


def build_rmh():
    """pseudo_marginal.py:332-379."""
    def kernel(rng_key, state: RWAState, logdensity_fn: Callable, transition_generator: Callable,
               proposal_logdensity_fn: Optional[Callable] = None):
        transition_energy = build_rmh_transition_energy(proposal_logdensity_fn)
        ratio = compute_asymmetric_acceptance_ratio(transition_energy)
        generate = rmh_proposal(logdensity_fn, transition_generator, ratio)
        new_state, do_accept, p_accept = generate(rng_key, state)
        return new_state, RWAInfo(p_accept, do_accept, new_state, generate.last_proposed)
    return kernel


def build_additive_step():
    """pseudo_marginal.py:152-172."""
    def kernel(rng_key, state: RWAState, logdensity_fn: Callable, random_step: Callable):
        def proposal_generator(key_proposal, position):
            return position + random_step(key_proposal, position)
        return build_rmh()(rng_key, state, logdensity_fn, proposal_generator)
    return kernel


class SamplingAlgorithm(NamedTuple):
    init: Callable
    step: Callable


def additive_step_random_walk(logdensity_fn: Callable, random_step: Callable) -> SamplingAlgorithm:
    """pseudo_marginal.py:192-232."""
    kernel = build_additive_step()
    return SamplingAlgorithm(lambda position, rng_key=None: init(position, logdensity_fn, rng_key),
                             lambda rng_key, state: kernel(rng_key, state, logdensity_fn, random_step))


def normal_random_walk(logdensity_fn: Callable, sigma) -> SamplingAlgorithm:
    """pseudo_marginal.py:175-189."""
    return additive_step_random_walk(logdensity_fn, normal(sigma))


def build_irmh() -> Callable:
    """pseudo_marginal.py:235-274: independent proposals."""
    def kernel(rng_key, state: RWAState, logdensity_fn: Callable, proposal_distribution: Callable,
               proposal_logdensity_fn: Optional[Callable] = None):
        def proposal_generator(key, position):
            return proposal_distribution(key)
        return build_rmh()(rng_key, state, logdensity_fn, proposal_generator, proposal_logdensity_fn)
    return kernel


def irmh_as_top_level_api(logdensity_fn: Callable, proposal_distribution: Callable,
                          proposal_logdensity_fn: Optional[Callable] = None) -> SamplingAlgorithm:
    """pseudo_marginal.py:277-329."""
    kernel = build_irmh()
    return SamplingAlgorithm(lambda position, rng_key=None: init(position, logdensity_fn, rng_key),
                             lambda rng_key, state: kernel(rng_key, state, logdensity_fn, proposal_distribution,
                                                           proposal_logdensity_fn))


def rmh_as_top_level_api(logdensity_fn: Callable, proposal_generator: Callable,
                         proposal_logdensity_fn: Optional[Callable] = None) -> SamplingAlgorithm:
    """pseudo_marginal.py:382-435."""
    kernel = build_rmh()
    return SamplingAlgorithm(lambda position, rng_key=None: init(position, logdensity_fn, rng_key),
                             lambda rng_key, state: kernel(rng_key, state, logdensity_fn, proposal_generator,
                                                           proposal_logdensity_fn))


irmh = irmh_as_top_level_api
rmh = rmh_as_top_level_api


# ---- the device-side log-density of docs/examples/parameter.md:331-354 for the built-in FitzHugh-Nagumo ODE ----
class FitzLogPosterior:
    """
    ``logdensity_fn(upars (C, 7), key) -> (logpost (C,), None)`` of the pseudo-marginal example: constrain
    (theta = exp(u[:3]), x0 = u[3:5], sigma = u[5:7]; parameter.md:227-236), ``first_order_pad`` initial value,
    ``ibm_init`` prior, ``solve_sim`` with ``interrogate_chkrebtii`` on the device, Gaussian observation log-likelihood
    at ``searchsorted`` indices plus the N(0, prior_sd^2) prior on the first five components -- all C chains in one launch;
    only C doubles come back.  (BASELINE.json config 4.)
    """

    def __init__(self, obs_data, obs_times, t_min, t_max, n_steps, noise_sd, n_chains, prior_sd=10.0, device=None,
                 traj_offset=0):
        import functools
        import rodeo_amd as ra
        from .logpost import obs_index
        self.ra, self.C, self.N = ra, int(n_chains), int(n_steps)
        self.t_min, self.t_max = float(t_min), float(t_max)
        self.obs = np.asarray(obs_data, dtype=np.float64)
        self.ind = obs_index(t_min, t_max, n_steps, obs_times)
        self.noise_sd, self.prior_sd = float(noise_sd), float(prior_sd)
        self.W, self._init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
        u0 = np.zeros((self.C, 7)); u0[:, 5:] = 0.1
        theta, x0, prior = self._constrain(u0)
        g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard")
        self.plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, self.W, x0, t_min, t_max, n_steps, g, prior, device=device,
                                 traj_offset=traj_offset, theta=theta)

    def _constrain(self, upars):
        theta, x0v, sigma = np.exp(upars[:, :3]), upars[:, 3:5], upars[:, 5:7]
        x0 = self._init(x0v, self.t_min, theta=theta)
        prior = self.ra.ibm_init((self.t_max - self.t_min) / self.N, 3, sigma)
        return theta, x0, prior

    def device(self, upars, key):
        """The log-posteriors as a DEVICE array of shape (C,) -- what a sharded caller hands to the RCCL all-gather without a
        round trip through the host (``shard.gather_scalars_device``).  One of four buffers of the plan, overwritten by the
        fourth call after this one."""
        from .logpost import sim_logpost
        upars = np.asarray(upars, dtype=np.float64)
        if upars.shape != (self.C, 7):
            raise ValueError(f"upars must have shape ({self.C}, 7)")
        theta, x0, prior = self._constrain(upars)
        self.plan.update(ode_init=x0, prior_pars=prior, theta=theta)
        # sampler and reduction as one device call (rk_solve_sim_logpost): no path is stored, only C doubles come out
        return sim_logpost(self.plan, key, self.obs, self.ind, self.noise_sd, upars=upars, prior_sd=self.prior_sd, n_prior=5)

    def __call__(self, upars, key):
        return self.device(upars, key).to_host(), None
