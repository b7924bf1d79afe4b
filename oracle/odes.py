"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  ODE right-hand sides in rodeo's calling convention
``ode_fun(X, t, **params) -> (n_block, n_bmeas)`` with ``X`` of shape ``(n_block, n_bstate)``
(src/rodeo/solve.py:218, src/rodeo/interrogate.py:95), extended with arbitrary leading batch dims on ``X`` and
on every parameter.  Each ODE also carries the analytic *block-diagonal* Jacobian
``jac(X, t, **params)[..., b, :, :] = d f_b / d X[b, :]`` that ``interrogate_kramer`` extracts from
``jax.jacfwd`` (src/rodeo/interrogate.py:76-79); ``complex_step_blockjac`` recomputes it numerically so the tests
can pin the analytic forms.
"""
import numpy as np


class ODE:
    def __init__(self, name, n_block, n_bmeas, f, jac):
        self.name, self.n_block, self.n_bmeas, self._f, self._jac = name, n_block, n_bmeas, f, jac

    def __call__(self, X, t, **params):
        return self._f(np.asarray(X), t, **params)

    def jac(self, X, t, **params):
        return self._jac(np.asarray(X), t, **params)


def complex_step_blockjac(ode, X, t, h=1e-30, **params):
    """d f_b / d X[b, j] by complex-step differentiation (exact to rounding for analytic f)."""
    X = np.asarray(X, dtype=np.float64)
    d, p = X.shape[-2:]
    f0 = ode(X, t, **params)
    m = f0.shape[-1]
    J = np.zeros(X.shape[:-2] + (d, m, p))
    for b in range(d):
        for j in range(p):
            Xc = X.astype(np.complex128)
            Xc[..., b, j] += 1j * h
            J[..., b, :, j] = np.imag(ode(Xc, t, **params)[..., b, :]) / h
    return J


# --- FitzHugh-Nagumo (README.md:92-99; docs/examples/parameter.md:60-68) -------------------------------------
def _fitz_f(X, t, theta, **_):
    theta = np.asarray(theta)
    a, b, c = theta[..., 0], theta[..., 1], theta[..., 2]
    V, R = X[..., 0, 0], X[..., 1, 0]
    out = np.stack([c * (V - V * V * V / 3 + R), -1 / c * (V - a + b * R)], axis=-1)
    return out[..., None]


def _fitz_jac(X, t, theta, **_):
    theta = np.asarray(theta, dtype=np.float64)
    b, c = theta[..., 1], theta[..., 2]
    V = X[..., 0, 0]
    J = np.zeros(np.broadcast_shapes(X.shape[:-2], theta.shape[:-1]) + (2, 1, X.shape[-1]))
    J[..., 0, 0, 0] = c * (1.0 - V * V)
    J[..., 1, 0, 0] = -b / c
    return J


fitzhugh_nagumo = ODE("fitzhugh_nagumo", 2, 1, _fitz_f, _fitz_jac)


# --- Lorenz63 (docs/examples/lorenz.md:85-92); theta = (rho, sigma, beta) -----------------------------------
def _lorenz_f(X, t, theta, **_):
    theta = np.asarray(theta)
    rho, sigma, beta = theta[..., 0], theta[..., 1], theta[..., 2]
    x, y, z = X[..., 0, 0], X[..., 1, 0], X[..., 2, 0]
    out = np.stack([-sigma * x + sigma * y, rho * x - y - x * z, -beta * z + x * y], axis=-1)
    return out[..., None]


def _lorenz_jac(X, t, theta, **_):
    theta = np.asarray(theta, dtype=np.float64)
    sigma, beta = theta[..., 1], theta[..., 2]
    J = np.zeros(np.broadcast_shapes(X.shape[:-2], theta.shape[:-1]) + (3, 1, X.shape[-1]))
    J[..., 0, 0, 0] = -sigma
    J[..., 1, 0, 0] = -1.0
    J[..., 2, 0, 0] = -beta
    return J


lorenz63 = ODE("lorenz63", 3, 1, _lorenz_f, _lorenz_jac)


# --- second-order ODE of Chkrebtii et al: x'' = sin(2t) - x (docs/examples/higher_order.md:47-59) -----------
def _higher_f(X, t, **_):
    return (np.sin(2 * t) - X[..., 0, 0])[..., None, None]


def _higher_jac(X, t, **_):
    J = np.zeros(X.shape[:-2] + (1, 1, X.shape[-1]))
    J[..., 0, 0, 0] = -1.0
    return J


higher_order = ODE("higher_order", 1, 1, _higher_f, _higher_jac)


def higher_order_exact(t):
    """docs/examples/higher_order.md:26-28."""
    t = np.asarray(t)
    return (2 * np.sin(t) - 3 * np.cos(t) - np.sin(2 * t)) / 3


# --- linear ODE x' = A x in the *dense single block* form of indep_init -----------
# X has shape (..., 1, n_vars * n_deriv); state layout is variable-major: X[.., 0, v * n_deriv + k] = x_v^(k).
def make_linear_dense(A, n_deriv):
    A = np.asarray(A, dtype=np.float64)
    n_vars = A.shape[0]
    idx0 = np.arange(n_vars) * n_deriv

    def f(X, t, **_):
        x = X[..., 0, idx0]
        return np.matmul(A, x[..., None])[..., 0][..., None, :]          # (..., 1, n_vars)

    def jac(X, t, **_):
        J = np.zeros(X.shape[:-2] + (1, n_vars, X.shape[-1]))
        Jv = J[..., 0, :, :]            # view (..., n_vars, n_vars*n_deriv)
        Jv[..., :, idx0] = A
        return J

    return ODE("linear_dense", 1, n_vars, f, jac)


# --- linear ODE x' = A x in block form (n_vars blocks); kramer keeps only diag(A) ---------------------------
def make_linear_block(A):
    A = np.asarray(A, dtype=np.float64)
    n_vars = A.shape[0]

    def f(X, t, **_):
        return np.matmul(A, X[..., :, 0][..., None])                       # (..., n_vars, 1)

    def jac(X, t, **_):
        J = np.zeros(X.shape[:-2] + (n_vars, 1, X.shape[-1]))
        J[..., :, 0, 0] = np.diag(A)
        return J

    return ODE("linear_block", n_vars, 1, f, jac)
