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


def _staged(plan, obs_data, obs_ind, upars, n_prior):
    """Device copies of the observations / indices (cached on the plan while unchanged) and of the transposed parameters."""
    dev = plan.dev
    obs = np.ascontiguousarray(obs_data, dtype=np.float64)
    ind = np.ascontiguousarray(obs_ind, dtype=np.int32)
    if obs.shape != (ind.shape[0], plan.d):
        raise ValueError(f"obs_data must have shape (n_obs, {plan.d})")
    if ind.size and (ind.min() < 0 or ind.max() > plan.N):
        raise ValueError("obs_ind outside the solver grid")
    cache = plan.__dict__.setdefault("_logpost_cache", {})
    sig = (obs.shape, obs.tobytes(), ind.tobytes())
    if cache.get("sig") != sig:
        cache["sig"], cache["obs"], cache["ind"] = sig, dev.to_device(obs), dev.to_device(ind)
    d_up, k = None, 0
    if upars is not None and hasattr(upars, "ptr"):           # already staged: a DeviceArray (n_prior, B) from stage_upars()
        d_up, k = upars, int(upars.shape[0])
    elif upars is not None:
        d_up = stage_upars(plan, upars, n_prior)
        k = int(d_up.shape[0])
    return cache, cache["obs"], cache["ind"], ind.shape[0], d_up, k


def stage_upars(plan, upars, n_prior=None):
    """The first ``n_prior`` unconstrained parameters of every trajectory as a device array (n_prior, B), batch-minor: what the
    reduction reads.  Upload it together with the plan's other inputs (``SolvePlan.update``), BEFORE the kernels are launched."""
    cache = plan.__dict__.setdefault("_logpost_cache", {})
    up = np.asarray(upars, dtype=np.float64)
    k = up.shape[1] if n_prior is None else int(n_prior)
    upt = np.ascontiguousarray(up[:, :k].T)
    d_up = cache.get("up")
    if d_up is None or tuple(d_up.shape) != upt.shape:
        d_up = cache["up"] = plan.dev.to_device(upt)
    else:
        d_up.upload(upt)
    return d_up


def sim_logpost(plan, key, obs_data, obs_ind, noise_sd, upars=None, prior_sd=10.0, n_prior=None, keep_path=False):
    """
    ``plan.sim(key)`` followed by ``gauss_obs_logpost`` on its path as ONE device call (``rk_solve_sim_logpost``): the body of
    the log-posterior of docs/examples/parameter.md:331-354.  Everything the kernels read is uploaded BEFORE the launch (an
    upload between sampler and reduction left the GPU idle for 30 of C4's 275 us per evaluation); on the n_bstate = 3 tile
    path the backward sampler reduces the log-posterior itself and, unless ``keep_path``, stores no path at all.
    Returns a DeviceArray (B,) from a ring of four buffers owned by the plan (overwritten by the fourth call after this one).
    """
    from ..solve import _seed
    dev = plan.dev
    cache, d_obs, d_ind, n_obs, d_up, k = _staged(plan, obs_data, obs_ind, upars, n_prior)
    ring = cache.setdefault("out_ring", [])
    if ring and tuple(ring[0].shape) != (plan.B,):
        ring.clear()
        cache["out_calls"] = 0
    n_call = cache.get("out_calls", 0)
    cache["out_calls"] = n_call + 1
    if len(ring) < 4:
        ring.append(dev.empty((plan.B,)))
    out = ring[n_call % 4]
    fused = _fused_supported(plan, n_obs)
    plan._no_path = fused and not keep_path and plan.x_state is None
    try:
        plan.generation += 1
        plan._prepare_out(_lib.MODE_SIM)
    finally:
        plan._no_path = False
    plan.last_mode = _lib.MODE_SIM
    plan.cfg.seed = _seed(key)
    so = plan._out
    if fused and not keep_path:
        so = _lib.SolveOut(workspace=so.workspace, workspace_bytes=so.workspace_bytes, mean_state=so.mean_state,
                           var_state=so.var_state, mean_pred=so.mean_pred, var_pred=so.var_pred, x_state=None)
    _lib.check(dev.lib.rk_solve_sim_logpost(dev.h, C.byref(plan.cfg), C.byref(plan.inp), C.byref(so), d_obs.ptr, d_ind.ptr,
                                            n_obs, float(noise_sd), d_up.ptr if d_up is not None else None, k,
                                            float(prior_sd), out.ptr))
    return out


def _fused_supported(plan, n_obs):
    """Mirror of tile3_sim_logpost_supported (csrc/solve_tile3.hip): the configurations whose sampler reduces the log-posterior."""
    lay = C.c_int32(0)
    _lib.check(plan.dev.lib.rk_solve_layout(C.byref(plan.cfg), _lib.MODE_SIM, C.byref(lay)))
    return lay.value == _lib.LAYOUT_TILE3 and plan.d in (1, 2, 4) and n_obs <= 512 and n_obs * plan.d <= 1024
