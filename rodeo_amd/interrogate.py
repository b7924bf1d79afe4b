"""
Interrogation methods -- the drop-in for ``rodeo.interrogate`` (src/rodeo/interrogate.py): same four names and the
same keyword protocol ``interrogate(key=, ode_fun=, ode_weight=, t=, mean_state_pred=, var_state_pred=, **params)
-> (wgt_meas (d, m, p), mean_meas (d, m), var_meas (d, m, m))`` (src/rodeo/solve.py:70-78).

Two uses:
  * passed to ``solve_mv`` / ``solve_sim`` they are recognised *by identity* and select the interrogation that is
    compiled into the fused forward kernel (rodeo_amd/csrc/solve_small.hip ``interrogate_traj``);
  * called directly they evaluate one interrogation for a batch on the GPU (``rk_interrogate_batched``) -- used by the
    parity tests; ``mean_state_pred`` / ``var_state_pred`` / parameters may carry a leading batch axis.

For ``interrogate_chkrebtii`` the ``key`` is an int seed, or ``(seed, step)`` / ``(seed, step, traj_offset)`` to
address the Philox stream exactly as the fused solver does at a given step.
"""
import ctypes as C
import numpy as np
from . import _lib
from .device import default_device, batch_minor
from .ode import DeviceODE


def _run(itg_id, key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, params, sqrt=False):
    if not isinstance(ode_fun, DeviceODE):
        raise TypeError("ode_fun must be a rodeo_amd.ode.DeviceODE (device code for the right-hand side)")
    dev = default_device()
    W = np.asarray(ode_weight, dtype=np.float64)
    mp = np.asarray(mean_state_pred, dtype=np.float64)
    vp = np.asarray(var_state_pred, dtype=np.float64)
    d, m, p = W.shape[-3:]
    if m != 1 and sqrt:
        raise NotImplementedError("standalone interrogate_chkrebtii(kalman_type='square-root') with n_bmeas > 1 is only "
                                  "available fused into solve_mv / solve_sim")
    theta, Bt = ode_fun.pack_params(params)
    batched = mp.ndim == 3 or vp.ndim == 4 or Bt is not None or W.ndim == 4
    B = next(s for s in ([mp.shape[0]] if mp.ndim == 3 else []) + ([vp.shape[0]] if vp.ndim == 4 else []) +
             ([Bt] if Bt is not None else []) + ([W.shape[0]] if W.ndim == 4 else []) + [1])
    mp = np.broadcast_to(mp, (B, d, p))
    vp = np.broadcast_to(vp, (B, d, p, p))
    seed, step, off = 0, 0, 0
    if key is not None:
        if isinstance(key, tuple):
            seed, step = int(key[0]), int(key[1])
            off = int(key[2]) if len(key) > 2 else 0
        else:
            seed = int(key)
    cfg = _lib.SolveCfg(n_traj=B, n_steps=1, n_block=d, n_bstate=p, n_bmeas=m, rhs_id=ode_fun.rhs_id,
                        interrogate=itg_id, kalman_type=_lib.KALMAN_SQRT if sqrt else _lib.KALMAN_STANDARD, n_theta=ode_fun.n_theta, flags=0,
                        t_min=0.0, t_max=1.0, seed=seed & 0xFFFFFFFFFFFFFFFF, traj_offset=off)
    dW = dev.to_device(batch_minor(W, W.ndim == 4))
    dth = dev.to_device(batch_minor(theta, Bt is not None)) if theta.size else None
    inp = _lib.SolveIn(ode_weight=dW.ptr, ode_weight_batched=int(W.ndim == 4), ode_init=None, ode_init_batched=0,
                       prior_weight=None, prior_weight_batched=0, prior_var=None, prior_var_batched=0,
                       theta=dth.ptr if dth is not None else None, theta_batched=int(Bt is not None))
    dmp, dvp = dev.to_device(batch_minor(mp, True)), dev.to_device(batch_minor(vp, True))
    wm, mm, vm = dev.empty((d, m, p, B)), dev.empty((d, m, B)), dev.empty((d, m, p if sqrt else m, B))
    _lib.check(dev.lib.rk_interrogate_batched(dev.h, C.byref(cfg), C.byref(inp), float(t), int(step),
                                              dmp.ptr, dvp.ptr, wm.ptr, mm.ptr, vm.ptr))
    out = wm.batch_first(), mm.batch_first(), vm.batch_first()
    return out if batched else tuple(o[0] for o in out)


def interrogate_rodeo(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, **params):
    """src/rodeo/interrogate.py:87-115: wgt_meas = 0, mean_meas = -f(mu-), var_meas = W Sigma- W^T."""
    return _run(_lib.INTERROGATE_RODEO, key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, params)


def interrogate_schober(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, **params):
    """src/rodeo/interrogate.py:50-62: wgt_meas = 0, mean_meas = -f(mu-), var_meas = 0."""
    return _run(_lib.INTERROGATE_SCHOBER, key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, params)


def interrogate_kramer(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, **params):
    """src/rodeo/interrogate.py:65-84: wgt_meas = -J, mean_meas = -f(mu-) + J mu-, var_meas = 0 (J block-diagonal)."""
    return _run(_lib.INTERROGATE_KRAMER, key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, params)


def interrogate_chkrebtii(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, kalman_type, **params):
    """src/rodeo/interrogate.py:13-47: "standard": x ~ N(mu-, Sigma-), mean_meas = -f(x), var_meas = W Sigma- W^T.
    "square-root" (interrogate.py:35-42): var_state_pred is the factor L-, var_meas = W L- of shape (d, m, p) and
    x = mu- + (W L-) z (one scalar per block added to every entry, as the reference's broadcast does); n_bmeas = 1."""
    if kalman_type not in ("standard", "square-root"):
        raise NotImplementedError                       # src/rodeo/interrogate.py:43-44
    return _run(_lib.INTERROGATE_CHKREBTII, key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, params,
                sqrt=kalman_type == "square-root")
