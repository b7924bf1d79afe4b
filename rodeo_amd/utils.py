"""Host-side helpers mirroring src/rodeo/utils.py (only what the solver path needs)."""
import numpy as np


def first_order_pad(ode_fun, n_vars, n_deriv):
    """
    ``rodeo.utils.first_order_pad`` (src/rodeo/utils.py:80-102): returns ``W`` of shape (n_vars, 1, n_deriv) with
    ``W[:, :, 1] = 1`` and ``ode_init(x0, t, **params) = [x0, f(x0, t), 0, ...]`` of shape (n_vars, n_deriv).
    ``x0`` may carry a leading batch axis (B, n_vars) together with batched parameters.
    """
    from .ode import DeviceODE
    fun = ode_fun
    if not isinstance(ode_fun, DeviceODE) and callable(ode_fun):
        # a plain Python right-hand side is written for ONE trajectory (X of shape (n_vars, n_deriv)): evaluate it per
        # batch element instead of letting its indexing run over the batch axis
        from .trace import _host_twin
        fun = _host_twin(ode_fun, n_vars)

    def ode_init(x0, t, **params):
        x0 = np.asarray(x0, dtype=np.float64)[..., :, None]
        f0 = fun(x0, t, **params)
        lead = np.broadcast_shapes(x0.shape[:-2], f0.shape[:-2])
        x0 = np.broadcast_to(x0, lead + x0.shape[-2:])
        f0 = np.broadcast_to(f0, lead + f0.shape[-2:])
        return np.concatenate([x0, f0, np.zeros(lead + (n_vars, n_deriv - 2))], axis=-1)

    W = np.zeros((n_vars, 1, n_deriv))
    W[:, :, 1] = 1.0
    return W, ode_init
