"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  Restatement of the reference's own *test oracle* (K1):
brute-force conditioning of the joint Gaussian of a short state-space model.

  gauss_markov_mv  <- tests/gauss_markov.py:30-70     mean / variance of a Gaussian Markov chain
  kalman2gm        <- tests/gauss_markov.py:73-125    state-space model -> Gaussian Markov chain
  mvncond          <- src/rodeo/utils.py:27-57        A, b, V of y[!icond] | y[icond]
  kalman_theta     <- tests/utils.py:24-63            theta_{m|n} = (E, var)(x_m | y_{0:n}) from the joint
  rel_err          <- tests/utils.py:11-18            NB signed denominator 0.1 + x1 (reference quirk, kept)

Valid for *any* inputs, so the tests pin oracle/kalman_ops.py (and then the HIP ops) with their own seeds;
the reference's PRNGKey(0) inputs cannot be regenerated without JAX.
"""
import numpy as np


def rel_err(X1, X2):
    x1 = np.ravel(X1) * 1.0
    x2 = np.ravel(X2) * 1.0
    return np.max(np.abs((x1 - x2) / (0.1 + x1)))


def _semi_chol(X):
    """tests/gauss_markov.py:127-144: Cholesky restricted to the non-zero rows/cols of a PSD matrix."""
    ind = np.any(X, axis=1)
    jind = np.nonzero(ind)[0]
    out = np.zeros_like(X)
    if len(jind):
        out[np.ix_(jind, jind)] = np.linalg.cholesky(X[np.ix_(jind, jind)])
    return out


def gauss_markov_mv(A, b, C):
    """Y_0 = b_0 + C_0 e_0, Y_n = b_n + A_n Y_{n-1} + C_n e_n  ->  mean (n_tot, n_dim), var (n_tot,n_dim,n_tot,n_dim)."""
    n_tot, n_dim = b.shape
    AA = np.zeros((n_tot, n_tot, n_dim, n_dim))
    for m in range(n_tot):
        for n in range(n_tot):
            AA[n, m] = np.eye(n_dim) if n <= m else A[n - 1].dot(AA[n - 1, m])
    L = np.zeros((n_tot, n_dim, n_tot, n_dim))
    for m in range(n_tot):
        for n in range(m, n_tot):
            L[n, :, m, :] = AA[n, m].dot(C[m])
    u = np.zeros((n_tot, n_dim))
    for n in range(n_tot):
        for m in range(n + 1):
            u[n] += AA[n, m].dot(b[m])
    L = L.reshape(n_tot * n_dim, n_tot * n_dim)
    return u, L.dot(L.T).reshape(n_tot, n_dim, n_tot, n_dim)


def kalman2gm(wgt_state, mean_state, var_state, wgt_meas, mean_meas, var_meas):
    """Stack (x_n, y_n) into one Markov chain of dimension n_state + n_meas."""
    n_tot, n_meas, n_state = wgt_meas.shape
    n_dim = n_state + n_meas
    wgt_state = np.concatenate([np.zeros((1, n_state, n_state)), wgt_state])
    zero_sm, zero_mm = np.zeros((n_state, n_meas)), np.zeros((n_meas, n_meas))
    mean_gm = np.zeros((n_tot, n_dim))
    chol_gm = np.zeros((n_tot, n_dim, n_dim))
    trans_gm = np.zeros((n_tot, n_dim, n_dim))
    for i in range(n_tot):
        mean_gm[i] = np.concatenate([mean_state[i], mean_meas[i] + wgt_meas[i].dot(mean_state[i])])
        if i > 0:
            trans_gm[i] = np.block([[wgt_state[i], zero_sm], [wgt_meas[i].dot(wgt_state[i]), zero_mm]])
        cs, cm = _semi_chol(var_state[i]), _semi_chol(var_meas[i])
        chol_gm[i] = np.block([[cs, zero_sm], [wgt_meas[i].dot(cs), cm]])
    return trans_gm[1:], mean_gm, chol_gm


def mvncond(mu, Sigma, icond):
    """y[~icond] | y[icond] ~ N(A y[icond] + b, V); the inverse is an LU solve as in utils.py:51-54."""
    icond = np.asarray(icond, dtype=bool)
    f, t = np.nonzero(~icond)[0], np.nonzero(icond)[0]
    A = Sigma[np.ix_(f, t)].dot(np.linalg.solve(Sigma[np.ix_(t, t)], np.eye(len(t))))
    b = mu[~icond] - A.dot(mu[icond])
    V = Sigma[np.ix_(f, f)] - A.dot(Sigma[np.ix_(t, f)])
    return A, b, V


def kalman_theta(m, y, mu, Sigma):
    """theta_{m|n}: condition the joint on y_{0:n} (n+1 = len(y)) and read off the x_m marginal."""
    n_tot, n_dim = mu.shape
    n_y, n_meas = y.shape
    n_state = n_dim - n_meas
    m = np.atleast_1d(m)
    n_x = len(m)
    icond = np.full((n_tot, n_dim), False)
    icond[:n_y, n_state:n_dim] = True
    imarg = np.full((n_tot, n_dim), False)
    imarg[np.ix_(m, np.arange(n_state))] = True
    imarg = np.ravel(imarg)[~np.ravel(icond)]
    A, b, V = mvncond(np.ravel(mu), Sigma.reshape(n_tot * n_dim, n_tot * n_dim), np.ravel(icond))
    mu_mn = (A.dot(np.ravel(y)) + b)[imarg].reshape(n_x, n_state)
    V_mn = V[np.ix_(imarg, imarg)].reshape(n_x, n_state, n_x, n_state)
    if n_x == 1:
        mu_mn, V_mn = mu_mn[0], V_mn[0, :, 0, :]
    return mu_mn, V_mn


def random_model(rng, n_meas=None, n_state=None, n_tot=3):
    """A random 3-step state-space model shaped like tests/utils.py:117-145 (own seed; wgt_state scaled 0.01)."""
    if n_meas is None:
        n_meas = int(rng.integers(1, 4))
    if n_state is None:
        n_state = n_meas + int(rng.integers(1, 5))
    def spd(k):
        a = rng.standard_normal((n_tot, k, k))
        return np.matmul(a, np.swapaxes(a, -1, -2))
    mdl = dict(
        n_meas=n_meas, n_state=n_state, n_tot=n_tot,
        mean_state=rng.standard_normal((n_tot, n_state)),
        var_state=spd(n_state),
        wgt_state=0.01 * rng.standard_normal((n_tot - 1, n_state, n_state)),
        mean_meas=rng.standard_normal((n_tot, n_meas)),
        var_meas=spd(n_meas),
        wgt_meas=rng.standard_normal((n_tot, n_meas, n_state)),
        x_meas=rng.standard_normal((n_tot, n_meas)),
        x_state_next=rng.standard_normal(n_state),
    )
    A, b, C = kalman2gm(mdl["wgt_state"], mdl["mean_state"], mdl["var_state"],
                        mdl["wgt_meas"], mdl["mean_meas"], mdl["var_meas"])
    mdl["mean_gm"], mdl["var_gm"] = gauss_markov_mv(A, b, C)
    return mdl


def filter_targets(mdl, step):
    """tests/utils.py:157-191: (theta_{s-1|s-1}, theta_{s|s-1}, theta_{s|s}) for step s = 1 or 2."""
    y, mu, S = mdl["x_meas"], mdl["mean_gm"], mdl["var_gm"]
    past = kalman_theta(step - 1, y[:step], mu, S)
    pred = kalman_theta(step, y[:step], mu, S)
    filt = kalman_theta(step, y[:step + 1], mu, S)
    return past, pred, filt


def smooth_targets(mdl):
    """tests/utils.py:194-215."""
    y, mu, S = mdl["x_meas"], mdl["mean_gm"], mdl["var_gm"]
    n_state = mdl["n_state"]
    nxt = kalman_theta(1, y, mu, S)
    filt = kalman_theta(0, y[:1], mu, S)
    pred = kalman_theta(1, y[:1], mu, S)
    mean_sm, var_sm = kalman_theta([0, 1], y, mu, S)
    A, b, V = mvncond(mean_sm.ravel(), var_sm.reshape(2 * n_state, 2 * n_state),
                      np.array([False] * n_state + [True] * n_state))
    return dict(next=nxt, filt=filt, pred=pred, smooth=(mean_sm[0], var_sm[0, :, 0, :]),
                sim=(A.dot(mdl["x_state_next"]) + b, V), cond=(A, b, V))
