"""
q-times integrated Brownian motion prior -- host-side input generator (microseconds, stays on the CPU), mirroring
``rodeo.prior.ibm_init`` / ``ibm_state`` (src/rodeo/prior/ibm.py:37-88):

    Q_ij = 1[i<=j] dt^(j-i) / (j-i)!        R_ij = sigma^2 dt^(2q+1-i-j) / ((2q+1-i-j) (q-i)! (q-j)!)

Factorials are exact integers here (the reference evaluates exp(gammaln(.)), src/rodeo/prior/ibm.py:21-34, which
differs from the integer in the last bit or two).
"""
import math
import numpy as np


def ibm_state(dt, q, sigma):
    """(Q, R) of shape (q+1, q+1) for one block."""
    dt = float(dt)
    Q = np.zeros((q + 1, q + 1))
    R = np.zeros((q + 1, q + 1))
    for i in range(q + 1):
        for j in range(q + 1):
            if j >= i:
                Q[i, j] = dt ** (j - i) / math.factorial(j - i)
            e = 2 * q + 1 - i - j
            R[i, j] = sigma ** 2 * dt ** e / (e * math.factorial(q - i) * math.factorial(q - j))
    return Q, R


def ibm_init(dt, n_deriv, sigma):
    """
    ``sigma`` (n_block,) -> ``(wgt_state, var_state)`` of shape (n_block, p, p) each; a batched ``sigma`` of shape
    (B, n_block) gives ``var_state`` of shape (B, n_block, p, p) (``wgt_state`` does not depend on sigma).
    """
    sigma = np.asarray(sigma, dtype=np.float64)
    Q1, R1 = ibm_state(dt, n_deriv - 1, 1.0)
    n_block = sigma.shape[-1]
    wgt_state = np.repeat(Q1[None], n_block, axis=0)
    var_state = (sigma ** 2)[..., None, None] * R1
    return wgt_state, var_state
