"""
q-times integrated Brownian motion prior -- host-side input generator (microseconds, stays on the CPU), mirroring
``rodeo.prior.ibm_init`` / ``ibm_state`` (src/rodeo/prior/ibm.py:37-88):

    Q_ij = 1[i<=j] dt^(j-i) / (j-i)!        R_ij = sigma^2 dt^(2q+1-i-j) / ((2q+1-i-j) (q-i)! (q-j)!)

Factorials are exact integers by default.  The reference evaluates ``exp(gammaln(x + 1))`` (src/rodeo/prior/ibm.py:21-34),
which differs from the integer in the last bit or two; ``reference_factorial=True`` takes that route (``scipy.special.gammaln``,
same operation order as ibm.py:54-61: ``dt**mesh / f``, ``sigma**2 * num / (mesh * f * f)``) for callers who compare Q, R
against reference outputs bit by bit.  (JAX's own ``gammaln`` is XLA's Lanczos series, so even this is the reference's
formula rather than a guarantee of its last bit -- there is no JAX here to compare with.)
"""
import math
import numpy as np


def _gamma_factorial(x):
    """src/rodeo/prior/ibm.py:21-34: exp(gammaln(x + 1)), elementwise (SciPy's gammaln when present, else math.lgamma)."""
    x = np.asarray(x, dtype=np.float64) + 1.0
    try:
        from scipy.special import gammaln
        lg = gammaln(x)
    except ImportError:                                  # pragma: no cover
        lg = np.vectorize(math.lgamma, otypes=[np.float64])(x)
    return np.exp(lg)


def ibm_state(dt, q, sigma, reference_factorial=False):
    """(Q, R) of shape (q+1, q+1) for one block."""
    dt = float(dt)
    if reference_factorial:
        # the reference's array expressions, term for term (ibm.py:54-61); gammaln of a negative integer is +inf, so the
        # lower triangle of Q comes out as dt**k / inf = 0 like the reference's nan_to_num'ed quotient
        row, col = np.arange(q + 1)[:, None], np.arange(q + 1)[None, :]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            Q = np.nan_to_num(dt ** (col - row) / _gamma_factorial(col - row), nan=0.0)
        e = (2.0 * q + 1.0) - row - col
        R = sigma ** 2 * dt ** e / (e * _gamma_factorial(q - row) * _gamma_factorial(q - col))
        return Q, R
    Q = np.zeros((q + 1, q + 1))
    R = np.zeros((q + 1, q + 1))
    for i in range(q + 1):
        for j in range(q + 1):
            if j >= i:
                Q[i, j] = dt ** (j - i) / math.factorial(j - i)
            e = 2 * q + 1 - i - j
            R[i, j] = sigma ** 2 * dt ** e / (e * math.factorial(q - i) * math.factorial(q - j))
    return Q, R


def ibm_init(dt, n_deriv, sigma, reference_factorial=False):
    """
    ``sigma`` (n_block,) -> ``(wgt_state, var_state)`` of shape (n_block, p, p) each; a batched ``sigma`` of shape
    (B, n_block) gives ``var_state`` of shape (B, n_block, p, p) (``wgt_state`` does not depend on sigma).
    """
    sigma = np.asarray(sigma, dtype=np.float64)
    Q1, R1 = ibm_state(dt, n_deriv - 1, 1.0, reference_factorial)
    n_block = sigma.shape[-1]
    wgt_state = np.repeat(Q1[None], n_block, axis=0)
    var_state = (sigma ** 2)[..., None, None] * R1
    return wgt_state, var_state
