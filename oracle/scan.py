"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  NumPy restatement of the solver scans of src/rodeo/solve.py,
vectorised over blocks (the reference's ``jax.vmap``) and over a leading batch axis B of independent
trajectories (the reference leaves that to a user-level ``vmap``).

Indexing conventions followed exactly:
  * forward, n = 0..N-1: predict from filt[n]; t = t_min + (t_max - t_min)(n+1)/N (solve.py:74);
    W_meas = ode_weight + wgt_meas (solve.py:79); x_meas = 0 and mean_state = 0 (solve.py:51-52);
    index 0 of pred and filt is (ode_init, 0) (solve.py:114-121)
  * backward, n = N-1..1 with filt[1:N], pred[2:N+1] (solve.py:189-195, 284-289); out[0] = (ode_init, 0),
    out[N] = filt[N] (solve_mv) or the terminal draw (solve_sim).

Batch handling: if none of ode_init / prior_pars / ode_weight / params carries a batch axis the functions
return the reference's shapes (N+1, d, p) / (N+1, d, p, p); otherwise (B, N+1, d, p) / (B, N+1, d, p, p).
``key`` is an integer seed (or None); draws come from oracle/counter_rng.py keyed by the *global* trajectory
index ``traj_offset + b``.  The factor used for draws is ``psd_factor`` (see interrogations.py); the
reference's SVD / Cholesky factors (solve.py:179, interrogate.py:30) give the same law but another bit-stream.
"""
import numpy as np
from . import kalman_ops, sqrt_ops, counter_rng
from .interrogations import StepKey, psd_factor


def _funs(kalman_type):
    if kalman_type == "standard":
        return kalman_ops
    if kalman_type == "square-root":
        return sqrt_ops
    raise NotImplementedError


def _batch_size(ode_weight, ode_init, prior_weight, prior_var, params, batched_params):
    B = None
    def upd(n):
        nonlocal B
        if B is None or B == 1:
            B = n
        elif n not in (1, B):
            raise ValueError("inconsistent batch sizes")
    if np.ndim(ode_weight) == 4:
        upd(np.shape(ode_weight)[0])
    if np.ndim(ode_init) == 3:
        upd(np.shape(ode_init)[0])
    if np.ndim(prior_weight) == 4:
        upd(np.shape(prior_weight)[0])
    if np.ndim(prior_var) == 4:
        upd(np.shape(prior_var)[0])
    for k in batched_params:
        upd(np.shape(params[k])[0])
    return B


def _default_batched_params(params):
    """A parameter is treated as batched if it is a >= 2-D array (e.g. theta of shape (B, n_theta))."""
    return [k for k, v in params.items() if isinstance(v, np.ndarray) and v.ndim >= 2]


def solve_filter(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate,
                 prior_weight, prior_var, kalman_funs=kalman_ops, traj_offset=0, batched_params=None, **params):
    """solve.py:47-122.  Returns dict with 'state_pred' and 'state_filt' = (mean, var), batch axis first."""
    if batched_params is None:
        batched_params = _default_batched_params(params)
    B = _batch_size(ode_weight, ode_init, prior_weight, prior_var, params, batched_params)
    squeeze = B is None
    B = 1 if squeeze else B
    ode_weight = np.asarray(ode_weight, dtype=np.float64)
    n_block, n_bmeas, n_bstate = ode_weight.shape[-3:]
    mean = np.broadcast_to(np.asarray(ode_init, dtype=np.float64), (B, n_block, n_bstate)).copy()
    var = np.zeros((B, n_block, n_bstate, n_bstate))
    x_meas = np.zeros((B, n_block, n_bmeas))
    mean_state = np.zeros((B, n_block, n_bstate))
    if squeeze:  # give batched-looking params to the ODE anyway
        params = dict(params)

    mp = np.empty((B, n_steps + 1, n_block, n_bstate)); vp = np.empty((B, n_steps + 1, n_block, n_bstate, n_bstate))
    mf = np.empty_like(mp); vf = np.empty_like(vp)
    mp[:, 0] = mean; mf[:, 0] = mean; vp[:, 0] = 0.0; vf[:, 0] = 0.0
    traj = traj_offset + np.arange(B)
    for n in range(n_steps):
        mean_pred, var_pred = kalman_funs.predict(
            mean_state_past=mean, var_state_past=var, mean_state=mean_state,
            wgt_state=prior_weight, var_state=prior_var)
        step_key = None if key is None else StepKey(key, traj, n)
        wgt_meas, mean_meas, var_meas = interrogate(
            key=step_key, ode_fun=ode_fun, ode_weight=ode_weight,
            t=t_min + (t_max - t_min) * (n + 1) / n_steps,
            mean_state_pred=mean_pred, var_state_pred=var_pred, **params)
        W_meas = ode_weight + wgt_meas
        mean, var = kalman_funs.update(
            mean_state_pred=mean_pred, var_state_pred=var_pred, x_meas=x_meas,
            mean_meas=mean_meas, wgt_meas=W_meas, var_meas=var_meas)
        mp[:, n + 1] = mean_pred; vp[:, n + 1] = var_pred
        mf[:, n + 1] = mean; vf[:, n + 1] = var
    out = {"state_pred": (mp, vp), "state_filt": (mf, vf)}
    if squeeze:
        out = {k: (v[0][0], v[1][0]) for k, v in out.items()}
    return out


def _prep(filt_out, squeeze):
    (mp, vp), (mf, vf) = filt_out["state_pred"], filt_out["state_filt"]
    if squeeze:
        mp, vp, mf, vf = mp[None], vp[None], mf[None], vf[None]
    return mp, vp, mf, vf


def solve_mv(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
             kalman_type="standard", traj_offset=0, batched_params=None, **params):
    """solve.py:236-302."""
    kalman_funs = _funs(kalman_type)
    prior_weight, prior_var = prior_pars
    filt_out = solve_filter(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate,
                            prior_weight, prior_var, kalman_funs, traj_offset, batched_params, **params)
    squeeze = filt_out["state_filt"][0].ndim == 3
    mp, vp, mf, vf = _prep(filt_out, squeeze)
    ms = np.empty_like(mf); vs = np.empty_like(vf)
    ms[:, n_steps] = mf[:, n_steps]; vs[:, n_steps] = vf[:, n_steps]
    mean_next, var_next = mf[:, n_steps], vf[:, n_steps]
    for n in range(n_steps - 1, 0, -1):
        mean_next, var_next = kalman_funs.smooth_mv(
            mean_state_next=mean_next, var_state_next=var_next, wgt_state=prior_weight,
            mean_state_filt=mf[:, n], var_state_filt=vf[:, n],
            mean_state_pred=mp[:, n + 1], var_state_pred=vp[:, n + 1], var_state=prior_var)
        ms[:, n] = mean_next; vs[:, n] = var_next
    ms[:, 0] = mf[:, 0]; vs[:, 0] = 0.0
    return (ms[0], vs[0]) if squeeze else (ms, vs)


def solve_sim(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
              kalman_type="standard", traj_offset=0, batched_params=None, z_smooth=None, **params):
    """
    solve.py:137-205.  ``z_smooth`` (B, N+1, d, p) optionally injects the standard normals of the backward
    draws (index n = time index; index 0 unused); otherwise they come from the Philox stream.
    """
    kalman_funs = _funs(kalman_type)
    prior_weight, prior_var = prior_pars
    filt_out = solve_filter(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate,
                            prior_weight, prior_var, kalman_funs, traj_offset, batched_params, **params)
    squeeze = filt_out["state_filt"][0].ndim == 3
    mp, vp, mf, vf = _prep(filt_out, squeeze)
    B, _, n_block, n_bstate = mf.shape
    traj = traj_offset + np.arange(B)

    def draw(n, mean, var):
        if z_smooth is not None:
            z = np.asarray(z_smooth).reshape(B, n_steps + 1, n_block, n_bstate)[:, n]
        else:
            z = counter_rng.normals(key, traj, n, n_block, n_bstate, counter_rng.PURPOSE_SMOOTH)
        if kalman_type == "square-root":
            # already a factor; column signs normalised (diag >= 0) -- see interrogations.interrogate_chkrebtii
            F = var * np.where(np.einsum("...ii->...i", var) < 0, -1.0, 1.0)[..., None, :]
        else:
            F = psd_factor(var)
        return mean + np.matmul(F, z[..., None])[..., 0]

    xs = np.empty_like(mf)
    x_next = draw(n_steps, mf[:, n_steps], vf[:, n_steps])
    xs[:, n_steps] = x_next
    for n in range(n_steps - 1, 0, -1):
        mean_sim, var_sim = kalman_funs.smooth_sim(
            x_state_next=x_next, wgt_state=prior_weight,
            mean_state_filt=mf[:, n], var_state_filt=vf[:, n],
            mean_state_pred=mp[:, n + 1], var_state_pred=vp[:, n + 1], var_state=prior_var)
        x_next = draw(n, mean_sim, var_sim)
        xs[:, n] = x_next
    xs[:, 0] = mf[:, 0]
    return xs[0] if squeeze else xs
