"""
Pins oracle/fenrir.py: for a LINEAR ODE with the first-order (kramer) interrogation the solver's model is exactly
linear Gaussian (z_n = W X_n - f(mu-) - J (X_n - mu-) = W X_n - f(X_n)), so Fenrir's value must equal the exact log p(y_{0:M} | z_{1:N} = 0), computed here by brute force from the
joint Gaussian of all states (the construction of the reference's tests/gauss_markov.py, applied to this model).
The reference has no fenrir test of its own.
"""
import numpy as np
import pytest
from scipy.stats import multivariate_normal
from oracle import fenrir as ofen, odes, priors, interrogations as oi


def _exact_loglik(W, x0, Q, R, N, t_min, t_max, forcing, obs_ind, D, Om, y):
    """One block, state dim p; z_n = W X_n - (a^T X_n + forcing(t_n)) = 0 exactly; X_n = Q X_{n-1} + N(0, R)."""
    p = len(x0)
    # joint of (X_1..X_N): mean and covariance by propagation
    mean = np.zeros((N + 1, p)); mean[0] = x0
    cov = np.zeros((N + 1, N + 1, p, p))
    for n in range(1, N + 1):
        mean[n] = Q @ mean[n - 1]
        cov[n, n] = Q @ cov[n - 1, n - 1] @ Q.T + R
        for k in range(n):
            cov[n, k] = Q @ cov[n - 1, k]
            cov[k, n] = cov[n, k].T
    mu = mean[1:].reshape(-1)
    S = np.block([[cov[i, j] for j in range(1, N + 1)] for i in range(1, N + 1)])
    ts = t_min + (t_max - t_min) * np.arange(1, N + 1) / N
    # linear measurements z_n = Hz X_n - f_n with z = 0, and observations y_m = D X_m + noise
    Hz = np.zeros((N, N * p)); fz = np.zeros(N)
    for n in range(N):
        Hz[n, n * p:(n + 1) * p] = W - forcing["a"]
        fz[n] = forcing["f"](ts[n])
    Hy = np.zeros((len(obs_ind), N * p))
    for m, n in enumerate(obs_ind):
        Hy[m, (n - 1) * p:n * p] = D
    # condition on z = 0: X | z
    Szz = Hz @ S @ Hz.T
    K = S @ Hz.T @ np.linalg.inv(Szz)
    mu_c = mu + K @ (fz - Hz @ mu)            # z = Hz X - fz = 0  <=>  Hz X = fz
    S_c = S - K @ Hz @ S
    my, Sy = Hy @ mu_c, Hy @ S_c @ Hy.T + Om * np.eye(len(obs_ind))
    return multivariate_normal.logpdf(y, my, Sy, allow_singular=True)


@pytest.mark.parametrize("itg", ["kramer"])
def test_fenrir_equals_exact_gaussian_loglik_for_a_linear_ode(itg):
    # x'' = sin(2t) - x  (docs/examples/higher_order.md), p = 3: W = [0, 0, 1], f = -X_0 + sin 2t
    N, t_min, t_max, p = 10, 0.0, 1.0, 3
    W = np.array([[[0.0, 0.0, 1.0]]])
    x0 = np.array([[-1.0, 0.0, 1.0]])
    Q, R = priors.ibm_init((t_max - t_min) / N, p, np.array([0.5]))
    obs_times = np.array([0.2, 0.5, 1.0])
    obs_ind = np.searchsorted(np.linspace(t_min, t_max, N + 1), obs_times)
    rng = np.random.default_rng(0)
    y = rng.standard_normal((3, 1, 1)) * 0.3 - 0.5
    D = np.array([1.0, 0.0, 0.0])
    obs_weight = np.tile(D[None, None, None, :], (3, 1, 1, 1))
    Om = 0.05
    obs_var = np.full((3, 1, 1, 1), Om)
    g = {"schober": oi.interrogate_schober, "kramer": oi.interrogate_kramer}[itg]
    val = ofen.fenrir(None, odes.higher_order, W, x0, t_min, t_max, N, g, (Q, R), y, obs_times, obs_weight, obs_var)
    ref = _exact_loglik(W[0, 0], x0[0], Q[0], R[0], N, t_min, t_max,
                        {"a": np.array([-1.0, 0.0, 0.0]), "f": lambda t: np.sin(2 * t)}, obs_ind, D, Om, y[:, 0, 0])
    assert abs(val - ref) < 1e-8 * max(1.0, abs(ref)), (val, ref)


def test_logpdf_drops_null_directions():
    cov = np.diag([2.0, 0.0, 0.5])
    x, m = np.array([1.0, 5.0, -1.0]), np.zeros(3)
    val = ofen.multivariate_normal_logpdf(x, m, cov)
    ref = multivariate_normal.logpdf([1.0, -1.0], [0, 0], np.diag([2.0, 0.5]))
    assert abs(val - ref) < 1e-12


def _exact_posterior(W, x0, Q, R, N, t_min, t_max, forcing, obs_ind, D, Om, y):
    """Mean and marginal covariances of X_1..X_N given z_{1:N} = 0 AND the observations (same model as _exact_loglik)."""
    p = len(x0)
    mean = np.zeros((N + 1, p)); mean[0] = x0
    cov = np.zeros((N + 1, N + 1, p, p))
    for n in range(1, N + 1):
        mean[n] = Q @ mean[n - 1]
        cov[n, n] = Q @ cov[n - 1, n - 1] @ Q.T + R
        for k in range(n):
            cov[n, k] = Q @ cov[n - 1, k]
            cov[k, n] = cov[n, k].T
    mu = mean[1:].reshape(-1)
    S = np.block([[cov[i, j] for j in range(1, N + 1)] for i in range(1, N + 1)])
    ts = t_min + (t_max - t_min) * np.arange(1, N + 1) / N
    H = np.zeros((N + len(obs_ind), N * p)); rhs = np.zeros(N + len(obs_ind)); noise = np.zeros(N + len(obs_ind))
    for n in range(N):
        H[n, n * p:(n + 1) * p] = W - forcing["a"]
        rhs[n] = forcing["f"](ts[n])
    for m, n in enumerate(obs_ind):
        H[N + m, (n - 1) * p:n * p] = D
        rhs[N + m] = y[m]; noise[N + m] = Om
    Sh = H @ S @ H.T + np.diag(noise)
    K = S @ H.T @ np.linalg.pinv(Sh, hermitian=True)
    mu_c = mu + K @ (rhs - H @ mu)
    S_c = S - K @ H @ S
    return mu_c.reshape(N, p), np.stack([S_c[n * p:(n + 1) * p, n * p:(n + 1) * p] for n in range(N)])


def test_fenrir_solve_mv_equals_exact_gaussian_posterior_for_a_linear_ode():
    """oracle/fenrir.solve_mv (fenrir.py:333-457) against the exact posterior p(X_n | z_{1:N} = 0, y) of the same model."""
    N, t_min, t_max, p = 10, 0.0, 1.0, 3
    W = np.array([[[0.0, 0.0, 1.0]]])
    x0 = np.array([[-1.0, 0.0, 1.0]])
    Q, R = priors.ibm_init((t_max - t_min) / N, p, np.array([0.5]))
    obs_times = np.array([0.2, 0.5, 1.0])
    obs_ind = np.searchsorted(np.linspace(t_min, t_max, N + 1), obs_times)
    rng = np.random.default_rng(0)
    y = rng.standard_normal((3, 1, 1)) * 0.3 - 0.5
    D = np.array([1.0, 0.0, 0.0])
    obs_weight = np.tile(D[None, None, None, :], (3, 1, 1, 1))
    Om = 0.05
    obs_var = np.full((3, 1, 1, 1), Om)
    m, v = ofen.solve_mv(None, odes.higher_order, W, x0, t_min, t_max, N, oi.interrogate_kramer, (Q, R), y, obs_times,
                         obs_weight, obs_var)
    assert m.shape == (N + 1, 1, p) and v.shape == (N + 1, 1, p, p)
    me, ve = _exact_posterior(W[0, 0], x0[0], Q[0], R[0], N, t_min, t_max,
                              {"a": np.array([-1.0, 0.0, 0.0]), "f": lambda t: np.sin(2 * t)}, obs_ind, D, Om, y[:, 0, 0])
    np.testing.assert_allclose(m[0, 0], x0[0]); assert np.all(v[0] == 0)
    assert np.max(np.abs(m[1:, 0] - me)) < 1e-8 and np.max(np.abs(v[1:, 0] - ve)) < 1e-8


def test_fenrir_vector_observations_equal_exact_gaussian_loglik():
    """n_bobs = 2 (fenrir.py:106-122: obs_weight (n_obs, n_block, n_bobs, n_bstate), a full 2 x 2 obs_var): the restatement
    against the exact Gaussian log-likelihood of the same linear model."""
    from scipy.linalg import block_diag
    N, t_min, t_max, p = 10, 0.0, 1.0, 3
    W = np.array([[[0.0, 0.0, 1.0]]])
    x0 = np.array([[-1.0, 0.0, 1.0]])
    Q, R = priors.ibm_init((t_max - t_min) / N, p, np.array([0.5]))
    obs_times = np.array([0.2, 0.5, 1.0])
    obs_ind = np.searchsorted(np.linspace(t_min, t_max, N + 1), obs_times)
    rng = np.random.default_rng(1)
    y = rng.standard_normal((3, 1, 2)) * 0.3
    D2 = np.array([[1.0, 0.0, 0.0], [0.3, 1.0, 0.0]])
    obs_weight = np.tile(D2[None, None], (3, 1, 1, 1))
    Om2 = np.array([[0.05, 0.01], [0.01, 0.2]])
    obs_var = np.tile(Om2[None, None], (3, 1, 1, 1))
    val = ofen.fenrir(None, odes.higher_order, W, x0, t_min, t_max, N, oi.interrogate_kramer, (Q, R), y, obs_times,
                      obs_weight, obs_var)
    # exact: joint of X_1..X_N, conditioned on z = 0, then the 6 observations jointly
    Qm, Rm = Q[0], R[0]
    mean = np.zeros((N + 1, p)); mean[0] = x0[0]
    cov = np.zeros((N + 1, N + 1, p, p))
    for n in range(1, N + 1):
        mean[n] = Qm @ mean[n - 1]
        cov[n, n] = Qm @ cov[n - 1, n - 1] @ Qm.T + Rm
        for k in range(n):
            cov[n, k] = Qm @ cov[n - 1, k]; cov[k, n] = cov[n, k].T
    mu = mean[1:].reshape(-1)
    S = np.block([[cov[i, j] for j in range(1, N + 1)] for i in range(1, N + 1)])
    ts = t_min + (t_max - t_min) * np.arange(1, N + 1) / N
    Hz = np.zeros((N, N * p)); fz = np.sin(2 * ts)
    for n in range(N):
        Hz[n, n * p:(n + 1) * p] = W[0, 0] - np.array([-1.0, 0.0, 0.0])
    K = S @ Hz.T @ np.linalg.inv(Hz @ S @ Hz.T)
    mu_c, S_c = mu + K @ (fz - Hz @ mu), S - K @ Hz @ S
    Hy = np.zeros((6, N * p))
    for m, n in enumerate(obs_ind):
        Hy[2 * m:2 * m + 2, (n - 1) * p:n * p] = D2
    ref = multivariate_normal.logpdf(y.reshape(-1), Hy @ mu_c, Hy @ S_c @ Hy.T + block_diag(Om2, Om2, Om2), allow_singular=True)
    assert abs(val - ref) < 1e-8 * max(1.0, abs(ref)), (val, ref)


def _sqrt_problem(n_bobs):
    N, t_min, t_max, p = 10, 0.0, 1.0, 3
    W = np.array([[[0.0, 0.0, 1.0]]])
    x0 = np.array([[-1.0, 0.0, 1.0]])
    Q, R = priors.ibm_init((t_max - t_min) / N, p, np.array([0.5]))
    obs_times = np.array([0.2, 0.5, 1.0])
    rng = np.random.default_rng(n_bobs)
    if n_bobs == 1:
        y = rng.standard_normal((3, 1, 1)) * 0.3 - 0.5
        D = np.array([[1.0, 0.0, 0.0]])
        Om = np.array([[0.05]])
    else:
        y = rng.standard_normal((3, 1, 2)) * 0.3
        D = np.array([[1.0, 0.0, 0.0], [0.3, 1.0, 0.0]])
        Om = np.array([[0.05, 0.01], [0.01, 0.2]])
    obs_weight = np.tile(D[None, None], (3, 1, 1, 1))
    obs_var = np.tile(Om[None, None], (3, 1, 1, 1))
    return dict(N=N, t_min=t_min, t_max=t_max, W=W, x0=x0, Q=Q, R=R, obs_times=obs_times, y=y, obs_weight=obs_weight,
                obs_var=obs_var, obs_fac=np.linalg.cholesky(obs_var), Rh=np.linalg.cholesky(R))


@pytest.mark.parametrize("n_bobs", [1, 2])
def test_fenrir_square_root_equals_the_standard_value(n_bobs):
    """kalman_type = "square-root" (fenrir.py:292-296 with square_root.py's forecast, which squares its factor before it is
    handed to the log-density, square_root.py:343-344): with L L^T inputs (prior_pars[1] = chol R, obs_var = chol Omega) the
    value is the same log-likelihood -- already pinned to the exact Gaussian value above for the standard form."""
    s = _sqrt_problem(n_bobs)
    args = (None, odes.higher_order, s["W"], s["x0"], s["t_min"], s["t_max"], s["N"], oi.interrogate_kramer)
    std = ofen.fenrir(*args, (s["Q"], s["R"]), s["y"], s["obs_times"], s["obs_weight"], s["obs_var"])
    sq = ofen.fenrir(*args, (s["Q"], s["Rh"]), s["y"], s["obs_times"], s["obs_weight"], s["obs_fac"], kalman_type="square-root")
    assert abs(sq - std) < 1e-8 * max(1.0, abs(std)), (sq, std)
    with pytest.raises(NotImplementedError):
        ofen.fenrir(*args, (s["Q"], s["R"]), s["y"], s["obs_times"], s["obs_weight"], s["obs_var"], kalman_type="other")


def _fixed_noise_interrogation(sd):
    """Schober's interrogation with a constant measurement noise: var_meas = sd^2 I in covariance form, sd I as the factor in
    square-root form -- the SAME model in both forms (interrogate_rodeo is not: in square-root mode the reference hands
    W L- W^T to the update as if it were a factor, interrogate.py:110-113, which changes the model)."""
    def make(factor):
        def itg(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, **params):
            wgt, mean_meas, var_meas = oi.interrogate_schober(key, ode_fun, ode_weight, t, mean_state_pred, var_state_pred, **params)
            m = var_meas.shape[-1]
            return wgt, mean_meas, var_meas + (sd if factor else sd * sd) * np.eye(m)
        return itg
    return make(False), make(True)


def test_fenrir_solve_mv_square_root_equals_the_standard_form():
    """fenrir.solve_mv and fenrir in square-root form (fenrir.py:421-426, 292-296) against the standard form, which is pinned
    to the exact Gaussian posterior above.  With an EXACT measurement (kramer, var_meas = 0) the backward filter's predicted
    covariance is rank deficient, its factor singular, and square_root.py:172-174's triangular solve with it diverges (1e14 in
    this restatement, as it must in the reference; the covariance form's LU happens to survive) -- so the comparison uses a
    measurement noise that is the same model in both forms; means and L L^T to 1e-8."""
    s = _sqrt_problem(1)
    N = s["N"]
    itg_std, itg_sqrt = _fixed_noise_interrogation(0.03)
    args = (None, odes.higher_order, s["W"], s["x0"], s["t_min"], s["t_max"], N)
    m, L = ofen.solve_mv(*args, itg_sqrt, (s["Q"], s["Rh"]), s["y"], s["obs_times"], s["obs_weight"], s["obs_fac"],
                         kalman_type="square-root")
    m2, v2 = ofen.solve_mv(*args, itg_std, (s["Q"], s["R"]), s["y"], s["obs_times"], s["obs_weight"], s["obs_var"])
    v = L @ np.swapaxes(L, -1, -2)
    assert np.max(np.abs(m - m2)) < 1e-8 and np.max(np.abs(v - v2)) < 1e-8
    ll = ofen.fenrir(*args, itg_sqrt, (s["Q"], s["Rh"]), s["y"], s["obs_times"], s["obs_weight"], s["obs_fac"], kalman_type="square-root")
    ll2 = ofen.fenrir(*args, itg_std, (s["Q"], s["R"]), s["y"], s["obs_times"], s["obs_weight"], s["obs_var"])
    assert abs(ll - ll2) < 1e-8 * max(1.0, abs(ll2))
