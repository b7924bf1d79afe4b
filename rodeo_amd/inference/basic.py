"""
``rodeo.inference.basic`` (src/rodeo/inference/basic.py:16-62): approximate log-likelihood
``sum_i log p(Y_i | X_n(i) = mu_{n(i)|N})`` from the solver's posterior mean.

Same signature as the reference.  ``obs_loglik`` may be
  * ``GaussianObsLoglik(noise_sd)`` -- reduced on the device (``rk_gauss_obs_logpost``), only B doubles come back;
  * any Python callable ``obs_loglik(obs_data, ode_data, **params)`` with the reference's meaning -- then only the
    ``n_obs`` needed time slices are downloaded and the callable runs on the host per trajectory.
Returns ``(loglik, Xt)`` like the reference's code (basic.py:62; its docstring mentions only the first): the
log-likelihood -- a float, or an array of shape (B,) for batched inputs -- and the posterior mean ``Xt`` of ``solve_mv``.
``Xt`` is a ``LazyMean``: the megabytes stay on the device unless it is actually used (``np.asarray(Xt)``, indexing);
the device buffers are reused by the next call with the same configuration, after which an unread ``Xt`` is stale.
"""
import numpy as np
from .. import _lib
from ..solve import cached_plan
from .logpost import gauss_obs_logpost, obs_index


class LazyMean:
    """The posterior mean of the solve behind a ``basic`` call, downloaded on first use."""

    def __init__(self, plan):
        self._plan, self._gen, self._val = plan, plan.generation, None

    def _get(self):
        if self._val is None:
            if self._plan.generation != self._gen:     # any launch or update() on the (cached, shared) plan since then
                raise RuntimeError("this Xt belongs to an earlier call: its device buffers have been reused (by basic, "
                                   "fenrir or any other call with the same configuration); read it (np.asarray(Xt)) "
                                   "before the next call")
            self._val = np.ascontiguousarray(self._plan.state_host()[0])
        return self._val

    def __array__(self, dtype=None, copy=None):
        a = self._get()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, idx):
        return self._get()[idx]

    @property
    def shape(self):
        p = self._plan
        return ((p.B,) if p.batched else ()) + (p.N + 1, p.d, p.p)


class GaussianObsLoglik:
    """``sum norm.logpdf(obs_data, loc=ode_data[:, :, 0], scale=noise_sd)`` (docs/examples/parameter.md:197-210)."""

    def __init__(self, noise_sd):
        self.noise_sd = float(noise_sd)

    def __call__(self, obs_data, ode_data, **params):          # host form (reference semantics), used by tests
        z = (np.asarray(obs_data) - np.asarray(ode_data)[..., 0]) / self.noise_sd
        return np.sum(-0.5 * z * z - np.log(self.noise_sd) - 0.5 * np.log(2 * np.pi), axis=(-1, -2))


def basic(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
          obs_data, obs_times, obs_loglik, kalman_type="standard", **params):
    plan = cached_plan(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type,
                       **params)        # device buffers are reused across calls with the same configuration
    plan.mv(key)
    Xt = LazyMean(plan)                    # valid until the plan's next launch / update (SolvePlan.generation)
    ind = obs_index(t_min, t_max, n_steps, obs_times)             # basic.py:61-62
    if isinstance(obs_loglik, GaussianObsLoglik):
        ll = gauss_obs_logpost(plan, obs_data, ind, obs_loglik.noise_sd, reuse_out=True).to_host()
        return (ll if plan.batched else float(ll[0])), Xt
    # generic callable: fetch only the observed time slices
    rows = []
    for n in ind:
        if plan.layout == _lib.LAYOUT_TILE3:
            rows.append(plan.var_state.slice0_host(int(n))[..., 3])            # (B, d, 3) means
        elif plan.layout == _lib.LAYOUT_TILE4:
            rows.append(plan.var_state.slice0_host(int(n))[..., 16:])          # (B, d, 4) means
        elif plan.layout == _lib.LAYOUT_TILEP:
            rows.append(plan.var_state.slice0_host(int(n))[..., plan.p * plan.p:])   # (B, d, p) means
        elif plan.layout == _lib.LAYOUT_TRAJ_MAJOR:
            rows.append(plan.mean_state.to_host()[:, int(n)])
        else:
            rows.append(np.moveaxis(plan.mean_state.slice0_host(int(n)), -1, 0))   # (d, p, B) -> (B, d, p)
    ode_data = np.stack(rows, axis=1)                              # (B, n_obs, d, p)
    out = np.array([obs_loglik(obs_data, ode_data[b], **{k: (np.asarray(v)[b] if np.ndim(v) >= 2 else v)
                                                          for k, v in params.items()}) for b in range(plan.B)])
    return (out if plan.batched else out[0]), Xt
