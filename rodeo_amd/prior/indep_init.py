"""``rodeo.prior.indep_init`` (src/rodeo/prior/indep_init.py:8-23): merge all blocks into one dense block."""
import numpy as np


def _block_diag(mats):
    n = sum(m.shape[0] for m in mats)
    out = np.zeros((n, n))
    o = 0
    for m in mats:
        k = m.shape[0]
        out[o:o + k, o:o + k] = m
        o += k
    return out


def indep_init(prior_pars):
    prior_weight, prior_var = prior_pars
    return _block_diag(list(np.asarray(prior_weight)))[None, :], _block_diag(list(np.asarray(prior_var)))[None, :]
