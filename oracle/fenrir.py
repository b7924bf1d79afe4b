"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  Restatement of the Fenrir log-likelihood
(src/rodeo/inference/fenrir.py:40-81 ``_forecast_update``, :86-259 ``_backward``, :261-327 ``fenrir``, :333-457 ``_smooth_mv`` /
``solve_mv``) and of the
eigendecomposition log-density it uses (src/rodeo/utils.py:60-78), for one trajectory or a leading batch axis.

The reference's tests do not cover fenrir (tests/ holds nothing for it), so this restatement is pinned by what the
algorithm must equal: for a linear ODE with an exact interrogation the Fenrir value is the exact Gaussian
log-likelihood log p(y_{0:M} | z_{1:N} = 0), which tests/test_oracle_fenrir.py computes by dense conditioning of
the joint Gaussian (the K1 construction of the reference's tests/gauss_markov.py applied to this model).
"""
import numpy as np
from . import kalman_ops, sqrt_ops, scan


def multivariate_normal_logpdf(x, mean, cov):
    """utils.py:60-78: eigendecomposition, eigenvalues with |w| <= 1e-8 (jnp.isclose(w, 0, rtol=1e-300)) dropped."""
    w, v = np.linalg.eigh(cov)
    z = v.T @ (np.asarray(x) - np.asarray(mean))
    iw = ~np.isclose(w, 0, rtol=1e-300)
    w = np.where(iw, w, 1.0)
    val = z ** 2 / w + np.log(w)
    return -0.5 * np.sum(np.where(iw, val, 0.0)) - np.sum(iw) * 0.5 * np.log(2 * np.pi)


def _forecast_update(mean_state_pred, var_state_pred, x_meas, mean_meas, wgt_meas, var_meas, funs=kalman_ops):
    """fenrir.py:40-81, all blocks at once (arrays (d, ...)); returns (sum of block log-densities, mean, var).
    ``funs`` = kalman_ops or sqrt_ops (fenrir.py:292-296): in square-root form every ``var`` is a lower factor, and
    ``forecast`` still returns the full forecast VARIANCE (square_root.py:343-344), so the value is a log-density."""
    mean_fore, var_fore = funs.forecast(mean_state_pred=mean_state_pred, var_state_pred=var_state_pred,
                                        mean_meas=mean_meas, wgt_meas=wgt_meas, var_meas=var_meas)
    logp = sum(multivariate_normal_logpdf(x_meas[b], mean_fore[b], var_fore[b]) for b in range(len(x_meas)))
    mean_filt, var_filt = funs.update(mean_state_pred=mean_state_pred, var_state_pred=var_state_pred,
                                      x_meas=x_meas, mean_meas=mean_meas, wgt_meas=wgt_meas, var_meas=var_meas)
    return logp, mean_filt, var_filt


def backward(mean_state_filt, var_state_filt, mean_state_pred, var_state_pred, prior_weight, prior_var,
             t_min, t_max, n_steps, obs_data, obs_times, obs_weight, obs_var, return_states=False, funs=kalman_ops):
    """fenrir.py:86-259 for one trajectory: arrays (N+1, d, p[, p]); returns the log-density (and, on request, the
    ``state_par`` dictionary of fenrir.py:236-258: backward-filter predictions / updates for n = 0..N, Markov weights and
    variances for n = 0..N-1)."""
    n_obs, n_block, n_bobs, n_bstate = np.shape(obs_weight)
    sim_times = np.linspace(t_min, t_max, n_steps + 1)
    obs_ind = np.searchsorted(sim_times, obs_times)
    obs_mean = np.zeros((n_block, n_bobs))
    i = n_obs - 1
    logdens = 0.0
    bmean, bvar = mean_state_filt[n_steps], var_state_filt[n_steps]
    N1 = n_steps + 1
    pm, pv = np.zeros(np.shape(mean_state_filt)), np.zeros(np.shape(var_state_filt))        # backward predictions
    fm, fv = np.zeros(np.shape(mean_state_filt)), np.zeros(np.shape(var_state_filt))        # backward updates
    wA, wC = np.zeros((n_steps,) + np.shape(var_state_filt)[1:]), np.zeros((n_steps,) + np.shape(var_state_filt)[1:])
    pm[n_steps], pv[n_steps] = bmean, bvar                                    # fenrir.py:226-232: the terminal point
    if obs_ind[i] >= n_steps:                                                 # fenrir.py:189-209
        logp, bmean, bvar = _forecast_update(bmean, bvar, obs_data[i], obs_mean, obs_weight[i], obs_var[i], funs)
        logdens += logp
        i -= 1
    fm[n_steps], fv[n_steps] = bmean, bvar                                    # fenrir.py:233-238
    for t in range(n_steps - 1, -1, -1):                                       # reverse scan, fenrir.py:131-184, 217-222
        A, b, C = funs.smooth_cond(mean_state_filt=mean_state_filt[t], var_state_filt=var_state_filt[t],
                                         mean_state_pred=mean_state_pred[t + 1], var_state_pred=var_state_pred[t + 1],
                                         wgt_state=prior_weight, var_state=prior_var)
        bmp, bvp = funs.predict(mean_state_past=bmean, var_state_past=bvar, mean_state=b, wgt_state=A, var_state=C)
        if i >= 0 and obs_ind[i] == t:       # (i = -1 indexes the last observation in JAX; it can only match t = N again)
            logp, bmean, bvar = _forecast_update(bmp, bvp, obs_data[i], obs_mean, obs_weight[i], obs_var[i], funs)
            logdens += logp
            i -= 1
        else:
            bmean, bvar = bmp, bvp
        pm[t], pv[t], fm[t], fv[t], wA[t], wC[t] = bmp, bvp, bmean, bvar, A, C
    if return_states:
        return logdens, {"state_pred": (pm, pv), "state_filt": (fm, fv), "wgt_state": wA, "var_state": wC}
    assert N1 == len(pm)
    return logdens


def _smooth_mv(state_par, funs=kalman_ops):
    """fenrir.py:333-402: the smoothing pass over the backward filter (a forward sweep in time; x_0 and x_1 keep the
    backward filter's own estimates, which already condition on all the data)."""
    (pm, pv), (fm, fv) = state_par["state_pred"], state_par["state_filt"]
    wA, wC = state_par["wgt_state"], state_par["var_state"]
    n_tot = pm.shape[0]
    sm, sv = fm.copy(), fv.copy()
    cm, cv = fm[1], fv[1]                                                      # scan_init, fenrir.py:379-382
    for k in range(n_tot - 2):                                                 # filt[2:], pred[1:n_tot-1], wgt/var_state[1:n_tot]
        cm, cv = funs.smooth_mv(mean_state_next=cm, var_state_next=cv, wgt_state=wA[k + 1],
                                      mean_state_filt=fm[k + 2], var_state_filt=fv[k + 2],
                                      mean_state_pred=pm[k + 1], var_state_pred=pv[k + 1], var_state=wC[k + 1])
        sm[k + 2], sv[k + 2] = cm, cv
    return sm, sv


def _funs(kalman_type):
    if kalman_type == "standard":
        return kalman_ops
    if kalman_type == "square-root":
        return sqrt_ops
    raise NotImplementedError                                                  # fenrir.py:292-296, 421-426


def solve_mv(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
             obs_data, obs_times, obs_weight, obs_var, kalman_type="standard", **params):
    """fenrir.py:405-457 for one trajectory: mean and variance (square-root form: lower factor) of
    p(X_{0:N} | Z_{1:N}, Y_{0:M}).  In square-root form ``prior_pars[1]`` and ``obs_var`` are lower factors."""
    funs = _funs(kalman_type)
    prior_weight, prior_var = prior_pars
    filt = scan.solve_filter(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate,
                             prior_weight, prior_var, kalman_funs=funs, **params)
    (mp, vp), (mf, vf) = filt["state_pred"], filt["state_filt"]
    _, state_par = backward(mf, vf, mp, vp, np.asarray(prior_weight, dtype=float), np.asarray(prior_var, dtype=float),
                            t_min, t_max, n_steps, obs_data, obs_times, obs_weight, obs_var, return_states=True, funs=funs)
    return _smooth_mv(state_par, funs)


def fenrir(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
           obs_data, obs_times, obs_weight, obs_var, kalman_type="standard", **params):
    """fenrir.py:261-327.  Leading batch axis on ode_init / params / prior_pars allowed -> array of log-densities.
    kalman_type = "square-root": ``prior_pars[1]`` and ``obs_var`` are lower factors (fenrir.py:292-296)."""
    funs = _funs(kalman_type)
    prior_weight, prior_var = prior_pars
    filt = scan.solve_filter(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate,
                             prior_weight, prior_var, kalman_funs=funs, **params)
    (mp, vp), (mf, vf) = filt["state_pred"], filt["state_filt"]
    Q, R = np.asarray(prior_weight, dtype=float), np.asarray(prior_var, dtype=float)
    if mf.ndim == 3:
        return backward(mf, vf, mp, vp, Q, R, t_min, t_max, n_steps, obs_data, obs_times, obs_weight, obs_var, funs=funs)
    out = np.empty(mf.shape[0])
    for b in range(mf.shape[0]):
        Qb = Q[b] if Q.ndim == 4 else Q
        Rb = R[b] if R.ndim == 4 else R
        out[b] = backward(mf[b], vf[b], mp[b], vp[b], Qb, Rb, t_min, t_max, n_steps, obs_data, obs_times, obs_weight, obs_var,
                          funs=funs)
    return out
