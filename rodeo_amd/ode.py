"""
ODE right-hand sides usable on the device.

In rodeo the ODE is an arbitrary JAX-traceable Python callable ``ode_fun(X, t, **params)`` evaluated inside the
scan (src/rodeo/solve.py:70-78) and differentiated with ``jax.jacfwd`` for ``interrogate_kramer``
(src/rodeo/interrogate.py:76).  Here the whole time loop runs inside one GPU kernel, so the right-hand side must
exist as device code: a ``DeviceODE`` couples

  * ``rhs_id``      -- the device implementation (rodeo_amd/csrc/rhs.hpp) compiled into the fused kernels,
  * ``__call__``    -- the same function in NumPy with rodeo's calling convention, used host-side for input
                       preparation only (``first_order_pad(...)[1]``, src/rodeo/utils.py:96-98),
  * ``param_spec``  -- how ``**params`` are packed into the per-trajectory parameter vector ``theta``.

A plain Python callable (written with NumPy like the reference's JAX functions) is TRACED into such device code:
``from_python`` below / automatically inside ``solve_mv`` / ``solve_sim`` (rodeo_amd/trace.py).  There is no CPU fallback.
"""
import numpy as np
from . import _lib


class DeviceODE:
    def __init__(self, name, rhs_id, n_block, n_bmeas, param_spec, host_fun):
        self.name, self.rhs_id, self.n_block, self.n_bmeas = name, rhs_id, n_block, n_bmeas
        self.param_spec = tuple(param_spec)          # ((kwarg name, size), ...)
        self._host_fun = host_fun

    def __call__(self, X, t, **params):
        return self._host_fun(np.asarray(X, dtype=np.float64), t, **params)

    @property
    def n_theta(self):
        return sum(s for _, s in self.param_spec)

    def pack_params(self, params):
        """-> (theta, batch): theta (n_theta,) if no parameter is batched else (B, n_theta); batch = B or None."""
        parts, B = [], None
        for name, size in self.param_spec:
            if name not in params:
                raise TypeError(f"ODE '{self.name}' needs the parameter '{name}' (size {size})")
            v = np.asarray(params[name], dtype=np.float64)
            if v.ndim == 0 and size == 1:
                v = v[None]
            if v.ndim == 0 or v.shape[-1] != size or v.ndim > 2:
                raise ValueError(f"parameter '{name}' must have shape ({size},) or (B, {size}), got {v.shape}")
            if v.ndim == 2:
                if B not in (None, v.shape[0]):
                    raise ValueError("inconsistent batch sizes among ODE parameters")
                B = v.shape[0]
            parts.append(v)
        extra = set(params) - {n for n, _ in self.param_spec}
        if extra:
            raise TypeError(f"ODE '{self.name}' got unexpected parameters {sorted(extra)}")
        if not parts:
            return np.zeros(0), None
        if B is not None:
            parts = [np.broadcast_to(v, (B, v.shape[-1])) for v in parts]
        return np.concatenate(parts, axis=-1), B


def _fitz(X, t, theta):
    theta = np.asarray(theta, dtype=np.float64)
    a, b, c = theta[..., 0], theta[..., 1], theta[..., 2]
    V, R = X[..., 0, 0], X[..., 1, 0]
    return np.stack([c * (V - V * V * V / 3 + R), -1 / c * (V - a + b * R)], axis=-1)[..., None]


def _lorenz(X, t, theta):
    theta = np.asarray(theta, dtype=np.float64)
    rho, sigma, beta = theta[..., 0], theta[..., 1], theta[..., 2]
    x, y, z = X[..., 0, 0], X[..., 1, 0], X[..., 2, 0]
    return np.stack([-sigma * x + sigma * y, rho * x - y - x * z, -beta * z + x * y], axis=-1)[..., None]


def _higher(X, t):
    return (np.sin(2 * t) - X[..., 0, 0])[..., None, None]


#: FitzHugh-Nagumo (README.md:92-99); ``theta = (a, b, c)``
fitzhugh_nagumo = DeviceODE("fitzhugh_nagumo", _lib.RHS_FITZHUGH_NAGUMO, 2, 1, (("theta", 3),), _fitz)
#: Lorenz63 (docs/examples/lorenz.md:85-92); ``theta = (rho, sigma, beta)``
lorenz63 = DeviceODE("lorenz63", _lib.RHS_LORENZ63, 3, 1, (("theta", 3),), _lorenz)
#: x'' = sin(2t) - x (docs/examples/higher_order.md:47-59); no parameters
higher_order = DeviceODE("higher_order", _lib.RHS_HIGHER_ORDER, 1, 1, (), _higher)


def linear_dense(n_vars, n_deriv):
    """
    Linear ODE x' = A x in the dense single-block ("non-block") form produced by ``prior.indep_init``: the state is
    X of shape (1, n_vars * n_deriv), variable-major (X[0, v * n_deriv + k] = k-th derivative of x_v), the output
    has shape (1, n_vars).  The matrix is passed as the parameter ``A`` (n_vars, n_vars) [or (B, n_vars, n_vars)].
    Mirrors how examples/solve_nb.py / examples/timings.py run rodeo without variable blocking.
    """
    idx0 = np.arange(n_vars) * n_deriv

    def host(X, t, A):
        A = np.asarray(A, dtype=np.float64)
        if A.shape[-1] != n_vars:                          # packed (.., n_vars^2) form
            A = A.reshape(A.shape[:-1] + (n_vars, n_vars))
        x = X[..., 0, idx0]
        return np.matmul(A, x[..., None])[..., 0][..., None, :]

    ode = DeviceODE("linear_dense", _lib.RHS_LINEAR_DENSE, 1, n_vars, (("A", n_vars * n_vars),), host)
    _pack = ode.pack_params

    def pack(params):                                   # accept A as (n, n) or (B, n, n)
        params = dict(params)
        if "A" in params:
            A = np.asarray(params["A"], dtype=np.float64)
            params["A"] = A.reshape(A.shape[:-2] + (n_vars * n_vars,)) if A.ndim >= 2 and A.shape[-1] == n_vars and \
                A.shape[-2] == n_vars else A
        return _pack(params)

    ode.pack_params = pack
    return ode


def from_source(type_name, source, n_block, param_spec=(), host_fun=None, name=None, n_bmeas=1):
    """
    Register an ODE that is not built in, from HIP source (compiled with hiprtc on first use).

    ``source`` defines, inside ``namespace rk``, either a struct with the full interface of
    rodeo_amd/csrc/rhs.hpp (``D``, ``NTHETA``, ``NDEP``, ``f<P>``, ``fjac<P>``) -- then ``type_name`` is its name -- or a
    struct with a scalar-generic ``rhs<T, P>`` -- then pass ``type_name="AutoJac<Name>"`` and the block-diagonal
    Jacobian that ``interrogate_kramer`` needs (``jax.jacfwd`` in src/rodeo/interrogate.py:76-79) comes from
    forward-mode dual numbers (rodeo_amd/csrc/dual.hpp).  ``param_spec`` = ((kwarg name, size), ...) fixes how
    ``**params`` are packed into ``th[]``; ``host_fun(X, t, **params)`` is the NumPy twin used only for host-side input
    preparation (``first_order_pad``).  ``n_bmeas`` > 1 (several measurements per block, src/rodeo/solve.py:48-51): the
    source follows csrc/solve_small_m_kernels.hpp (``static constexpr int M``, ``rhs`` writing ``out[D][M]``, wrapped as
    ``AutoJacM<Name>``) and runs on the lane-per-trajectory kernels with the standard filter.
    """
    import ctypes as C
    lib = _lib.load()
    rid = C.c_int32(0)
    n_theta = sum(s for _, s in param_spec)
    _lib.check(lib.rk_register_rhs_source_m(type_name.encode(), source.encode(), int(n_block), int(n_bmeas),
                                            max(int(n_theta), 0), C.byref(rid)))

    def _no_host(X, t, **params):
        raise TypeError(f"ODE '{name or type_name}' was registered without a host_fun; it cannot be evaluated on the host")

    return DeviceODE(name or type_name, rid.value, int(n_block), int(n_bmeas), param_spec, host_fun or _no_host)


def from_python(fun, n_vars, n_deriv_used=2, name=None, **param_sizes):
    """``DeviceODE`` from an ordinary Python ``fun(X, t, **params)``; see rodeo_amd/trace.py."""
    from .trace import from_python as _fp
    return _fp(fun, n_vars, n_deriv_used, name, **param_sizes)


def compile_check(ode, n_bstate, interrogate_id=_lib.INTERROGATE_KRAMER):
    """Compile the kernels for a ``from_source`` ODE without a GPU; raises RodeoKalmanError with the compiler log."""
    _lib.check(_lib.load().rk_rhs_compile_check(int(ode.rhs_id), int(n_bstate), int(interrogate_id)))
