"""
``rodeo.inference.fenrir`` (src/rodeo/inference/fenrir.py:261-327): the Fenrir approximate log-likelihood
log p(Y_{0:M} | Z_{1:N}) -- forward filter (``_solve_filter``, storing the predicted moments), then the backward
Markov chain of ``smooth_cond`` run as a Kalman filter backwards in time that conditions on the observations
(``_backward``, fenrir.py:86-259).  Both passes run on the device (``rk_solve_filter`` + ``rk_fenrir_backward``: on
the MFMA-tile kernels at n_bstate = 3, else lane-per-trajectory with ``RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR``);
only B doubles come back.

Same signature as the reference.  Extension: a leading batch axis on ``ode_init`` / ``prior_pars`` / ``**params``
(observations are shared) returns an array (B,).  Observations per block: n_bobs = 1 .. 3 (``obs_data`` (n_obs, n_block,
n_bobs), ``obs_weight`` (n_obs, n_block, n_bobs, n_bstate), ``obs_var`` (n_obs, n_block, n_bobs, n_bobs), fenrir.py:106-122);
vector observations run on the lane-per-trajectory kernels (LU update, eigendecomposition log-density of utils.py:60-78).
``kalman_type="square-root"`` (fenrir.py:292-296, 421-426): ``prior_pars[1]`` and ``obs_var`` are lower factors, the forward
pass is the square-root filter and every backward step map comes from ``square_root.py``; its ``forecast`` squares the
factor before the log-density sees it (square_root.py:343-344), so the value is the same log-likelihood as in covariance
form for ``L L^T`` inputs (tests).  Lane-per-trajectory kernels (``fenrir_sqrt.hip``), n_bstate 2 .. 8.
"""
import ctypes as C
import numpy as np
from .. import _lib
from ..solve import cached_plan
from .logpost import obs_index


def fenrir(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
           obs_data, obs_times, obs_weight, obs_var, kalman_type="standard", **params):
    if kalman_type not in ("standard", "square-root"):
        raise NotImplementedError                                   # fenrir.py:293-298
    obs, D, Om, n_bobs = _check_obs(obs_data, obs_weight, obs_var)
    ind = obs_index(t_min, t_max, n_steps, obs_times)             # fenrir.py:118-120
    if np.any(np.diff(ind) < 0):
        raise ValueError("obs_times must be ascending")
    if kalman_type == "square-root":
        plan = cached_plan(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type, **params)
    else:
        plan = _plan_for(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type, params,
                         tiles_ok=n_bobs == 1)
    if D.shape[1:] != (plan.d, n_bobs, plan.p):
        raise ValueError(f"obs_weight must have shape (n_obs, {plan.d}, n_bobs, {plan.p})")
    plan.filter(key)
    dev = plan.dev
    # observations live on the plan and are uploaded again only when they change (a sampler calls this once per step)
    cache = plan.__dict__.setdefault("_fenrir_obs", {})
    sig = (obs.tobytes(), D.tobytes(), Om.tobytes(), ind.tobytes())
    if cache.get("sig") != sig:
        cache["sig"] = sig
        cache["dev"] = tuple(dev.to_device(np.ascontiguousarray(a)) for a in (obs, D, Om, ind.astype(np.int32)))
    d_obs, d_w, d_v, d_ind = cache["dev"]
    out = dev.empty((plan.B,))
    _lib.check(dev.lib.rk_fenrir_backward(dev.h, C.byref(plan.cfg), C.byref(plan.inp), C.byref(plan._out), d_obs.ptr,
                                          d_w.ptr, d_v.ptr, d_ind.ptr, int(ind.shape[0]), n_bobs, out.ptr))
    ll = out.to_host()
    return ll if plan.batched else float(ll[0])


def _check_obs(obs_data, obs_weight, obs_var):
    """(obs (n_obs, d, n_bobs), D (n_obs, d, n_bobs, p), Omega (n_obs, d, n_bobs, n_bobs), n_bobs) as contiguous float64."""
    obs = np.ascontiguousarray(obs_data, dtype=np.float64)
    D = np.ascontiguousarray(obs_weight, dtype=np.float64)
    Om = np.ascontiguousarray(obs_var, dtype=np.float64)
    if D.ndim != 4:
        raise ValueError("fenrir: obs_weight must have shape (n_obs, n_block, n_bobs, n_bstate)")
    n_bobs = D.shape[2]
    if not 1 <= n_bobs <= 3:
        raise NotImplementedError("fenrir on the device: n_bobs (observations per block) in 1..3")
    if Om.shape != D.shape[:2] + (n_bobs, n_bobs) or obs.shape != D.shape[:2] + (n_bobs,):
        raise ValueError("fenrir: obs_data (n_obs, n_block, n_bobs), obs_weight (n_obs, n_block, n_bobs, n_bstate), obs_var "
                         "(n_obs, n_block, n_bobs, n_bobs)")
    return obs, D, Om, n_bobs


def _plan_for(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type, params,
              tiles_ok=True, tiles_blocked_ok=True):
    """A (cached, solve.cached_plan) plan on the MFMA-tile forward kernels when the configuration has them (n_bstate = 3:
    the backward pass then re-evaluates the predicted moments from the filtered tiles), else on the lane-per-trajectory
    kernels with stored predictions."""
    args = (ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type)
    plan = cached_plan(*args, **params)
    lay = C.c_int32(0)
    _lib.check(plan.dev.lib.rk_solve_layout(C.byref(plan.cfg), _lib.MODE_FILTER, C.byref(lay)))
    # n_bstate = 3: the tile kernels (one observation per block); n_bstate = 4 .. 8: the blocked tile forward pass and a
    # lane-per-block backward filter on its records (any n_bobs); else the lane kernels with stored predictions
    on_tiles = (lay.value == _lib.LAYOUT_TILE3 and tiles_ok) or \
               (lay.value in (_lib.LAYOUT_TILE4, _lib.LAYOUT_TILEP) and kalman_type == "standard" and tiles_blocked_ok)
    if not on_tiles:
        plan = cached_plan(*args, store_pred=True, batch_minor=True, **params)
    return plan


def solve_mv(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
             obs_data, obs_times, obs_weight, obs_var, kalman_type="standard", **params):
    """
    Fenrir's data-adaptive solver (src/rodeo/inference/fenrir.py:405-457): mean and variance of
    p(X_{0:N} | Z_{1:N}, Y_{0:M}) -- forward filter, backward filter through the observations, smoothing pass over the
    backward filter (``_smooth_mv``, fenrir.py:333-402).  Same arguments as ``fenrir``; returns ``(mean (N+1, d, p), var
    (N+1, d, p, p))`` with a leading batch axis for batched inputs.  Lane-per-trajectory kernels (``rk_fenrir_solve_mv``).
    """
    if kalman_type not in ("standard", "square-root"):
        raise NotImplementedError                                   # fenrir.py:421-426
    obs, D, Om, n_bobs = _check_obs(obs_data, obs_weight, obs_var)
    ind = obs_index(t_min, t_max, n_steps, obs_times)
    if np.any(np.diff(ind) < 0):
        raise ValueError("obs_times must be ascending")
    from ..solve import SolvePlan
    sq = kalman_type == "square-root"                               # (its forward pass is batch-minor and keeps no predictions)
    if not sq:
        # n_bstate = 4 .. 8: forward pass on the blocked MFMA tiles, backward filter and smoothing pass on its records
        # (rk_fenrir_solve_mv_tiles: no stored predictions, no batch-minor copy of the filter's output)
        tplan = cached_plan(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type, **params)
        lay = C.c_int32(0)
        _lib.check(tplan.dev.lib.rk_solve_layout(C.byref(tplan.cfg), _lib.MODE_FILTER, C.byref(lay)))
        if lay.value in (_lib.LAYOUT_TILE4, _lib.LAYOUT_TILEP) and 4 <= tplan.p <= 8:
            return _solve_mv_tiles(tplan, key, obs, D, Om, ind, n_bobs)
    plan = SolvePlan(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type,
                     store_pred=not sq, batch_minor=not sq, **params)
    if D.shape[1:] != (plan.d, n_bobs, plan.p):
        raise ValueError(f"obs_weight must have shape (n_obs, {plan.d}, n_bobs, {plan.p})")
    plan.filter(key)
    dev = plan.dev
    d_obs, d_w, d_v, d_ind = (dev.to_device(np.ascontiguousarray(a)) for a in (obs, D, Om, ind.astype(np.int32)))
    nbytes = C.c_size_t(0)
    _lib.check(dev.lib.rk_fenrir_workspace_bytes(C.byref(plan.cfg), C.byref(nbytes)))
    ws = dev.empty((nbytes.value // 8,))
    _lib.check(dev.lib.rk_fenrir_solve_mv(dev.h, C.byref(plan.cfg), C.byref(plan.inp), C.byref(plan._out), d_obs.ptr,
                                          d_w.ptr, d_v.ptr, d_ind.ptr, int(ind.shape[0]), n_bobs, ws.ptr))
    return plan.state_host()


def _solve_mv_tiles(plan, key, obs, D, Om, ind, n_bobs):
    """``solve_mv`` on a plan whose forward pass runs on the blocked tiles (n_bstate 4 .. 8)."""
    if D.shape[1:] != (plan.d, n_bobs, plan.p):
        raise ValueError(f"obs_weight must have shape (n_obs, {plan.d}, n_bobs, {plan.p})")
    plan.filter(key)
    dev = plan.dev
    d_obs, d_w, d_v, d_ind = (dev.to_device(np.ascontiguousarray(a)) for a in (obs, D, Om, ind.astype(np.int32)))
    nbytes = C.c_size_t(0)
    _lib.check(dev.lib.rk_fenrir_workspace_bytes(C.byref(plan.cfg), C.byref(nbytes)))
    ws = dev.empty((nbytes.value // 8,))
    N, d, p, B = plan.cfg.n_steps, plan.d, plan.p, plan.cfg.n_traj
    mean, var = dev.empty((N + 1, d, p, B)), dev.empty((N + 1, d, p, p, B))
    _lib.check(dev.lib.rk_fenrir_solve_mv_tiles(dev.h, C.byref(plan.cfg), C.byref(plan.inp), C.byref(plan._out), d_obs.ptr,
                                                d_w.ptr, d_v.ptr, d_ind.ptr, int(ind.shape[0]), n_bobs, ws.ptr, mean.ptr, var.ptr))
    m, v = mean.batch_first(), var.batch_first()
    return (m, v) if plan.batched else (m[0], v[0])
