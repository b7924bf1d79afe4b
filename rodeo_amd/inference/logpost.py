"""
Gaussian observation log-likelihood (+ optional normal log-prior) per trajectory, reduced on the device from a solver
output that is still resident in HBM -- the tail of rodeo's user-level log-posteriors:

    obs_ind  = searchsorted(sim_times, obs_times)                          docs/examples/parameter.md:149
    loglik   = sum norm.logpdf(obs, loc = Xt[obs_ind, :, 0], scale = noise_sd)    parameter.md:197-210
    logprior = sum norm.logpdf(upars[:n_prior], 0, prior_sd)               parameter.md:188-194

Only B doubles leave the GPU instead of the (B, N+1, d, p) path.
"""
import ctypes as C
import numpy as np
from .. import _lib


def obs_index(t_min, t_max, n_steps, obs_times):
    """Indices of the solver grid closest-from-the-right to the observation times (``jnp.searchsorted``)."""
    sim_times = np.linspace(t_min, t_max, n_steps + 1)
    return np.searchsorted(sim_times, np.asarray(obs_times, dtype=np.float64)).astype(np.int32)


def gauss_obs_logpost(plan, obs_data, obs_ind, noise_sd, upars=None, prior_sd=10.0, n_prior=None, which="auto",
                      reuse_out=False):
    """
    ``plan``: a ``SolvePlan`` whose ``mv()`` / ``sim()`` has been launched.  ``obs_data`` (n_obs, d), ``obs_ind``
    (n_obs,) int; ``upars`` (B, k) optional unconstrained parameters whose first ``n_prior`` entries get a
    N(0, prior_sd^2) prior.  Returns a DeviceArray of shape (B,) (call ``.to_host()``): a fresh one, or -- with
    ``reuse_out=True`` -- one of four buffers owned by the plan that later calls overwrite in turn.
    """
    dev = plan.dev
    obs = np.ascontiguousarray(obs_data, dtype=np.float64)
    ind = np.ascontiguousarray(obs_ind, dtype=np.int32)
    if obs.shape != (ind.shape[0], plan.d):
        raise ValueError(f"obs_data must have shape (n_obs, {plan.d})")
    if ind.size and (ind.min() < 0 or ind.max() > plan.N):
        raise ValueError("obs_ind outside the solver grid")
    if which == "auto":
        which = "x" if plan.x_state is not None and plan.last_mode == _lib.MODE_SIM else "mean"
    if which == "x":
        state, layout = plan.x_state, _lib.LAYOUT_BATCH_MINOR
    else:
        layout = plan.layout
        state = plan.var_state if layout in (_lib.LAYOUT_TILE3, _lib.LAYOUT_TILE4, _lib.LAYOUT_TILEP) else plan.mean_state
    # observations / indices / output live on the plan and are re-uploaded only when they change (a pseudo-marginal
    # chain calls this once per step with the same data): per call one upload (upars) and one kernel
    cache = plan.__dict__.setdefault("_logpost_cache", {})
    sig = (obs.shape, obs.tobytes(), ind.tobytes())
    if cache.get("sig") != sig:
        cache["sig"], cache["obs"], cache["ind"] = sig, dev.to_device(obs), dev.to_device(ind)
    d_obs, d_ind = cache["obs"], cache["ind"]
    d_up, k = None, 0
    if upars is not None:
        up = np.asarray(upars, dtype=np.float64)
        k = up.shape[1] if n_prior is None else int(n_prior)
        upt = np.ascontiguousarray(up[:, :k].T)
        d_up = cache.get("up")
        if d_up is None or tuple(d_up.shape) != upt.shape:
            d_up = cache["up"] = dev.to_device(upt)
        else:
            d_up.upload(upt)
    # The result is a fresh device array unless the caller opts into `reuse_out`: then it comes from a ring of four buffers on
    # the plan (a device allocation per call cost more than the kernel: 0.08 of C4's 0.32 ms per evaluation) and is
    # overwritten by the FOURTH reusing call after it -- for callers that read the result at once (FitzLogPosterior, basic).
    if reuse_out:
        ring = cache.setdefault("out_ring", [])
        if ring and tuple(ring[0].shape) != (plan.B,):
            ring.clear()
            cache["out_calls"] = 0
        n_call = cache.get("out_calls", 0)
        cache["out_calls"] = n_call + 1
        if len(ring) < 4:
            ring.append(dev.empty((plan.B,)))
        out = ring[n_call % 4]                  # call 5 reuses the buffer of call 1, call 6 that of call 2, ...
    else:
        out = dev.empty((plan.B,))
    _lib.check(dev.lib.rk_gauss_obs_logpost(dev.h, plan.B, plan.N, plan.d, plan.p, layout, state.ptr, d_obs.ptr,
                                            d_ind.ptr, ind.shape[0], float(noise_sd),
                                            d_up.ptr if d_up is not None else None, k, float(prior_sd), out.ptr))
    return out
