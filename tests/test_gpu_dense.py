"""
GPU parity of the dense large-block path (BASELINE config 5's shape: a stiff linear ODE x' = A x solved WITHOUT
variable blocking -- prior.indep_init merges the IBM blocks into one dense block, ode_weight = block_diag(W)[None],
as examples/timings.py:209 and examples/solve_nb.py do) against the NumPy oracle on the same inputs.
"""
import numpy as np
import pytest
from scipy.linalg import block_diag, expm
from oracle import scan, odes, priors, interrogations as oi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rodeo_amd
    return rodeo_amd


def dense_problem(ra, n_vars, n_deriv, n_steps, t_max, B=None, seed=0):
    rng = np.random.default_rng(20243 + seed)
    lam = np.linspace(0.5, 2.0, n_vars)         # non-stiff: the covariance-form recursion amplifies rounding 8x per step
                                                # on stiff spectra (lambda dt ~ 0.4), which makes trajectory parity meaningless
    A = -np.diag(lam) + 0.1 * rng.standard_normal((n_vars, n_vars)) / np.sqrt(n_vars)
    Wb, _ = ra.utils.first_order_pad(lambda x, t: x, n_vars, n_deriv)          # (n_vars, 1, n_deriv)
    W = block_diag(*[w for w in Wb])[None]                                       # (1, n_vars, n_vars * n_deriv)
    prior = ra.indep_init(ra.ibm_init(t_max / n_steps, n_deriv, np.ones(n_vars)))
    x0v = np.ones(n_vars) if B is None else 1.0 + 0.1 * rng.standard_normal((B, n_vars))
    X0 = np.zeros(x0v.shape[:-1] + (n_vars, n_deriv))
    X0[..., 0] = x0v
    X0[..., 1] = x0v @ A.T
    X0 = X0.reshape(x0v.shape[:-1] + (1, n_vars * n_deriv))
    return dict(A=A, W=W, prior=prior, x0=X0, x0v=x0v)


# The covariance-form recursion of the reference is numerically fragile on this non-block problem: with an exact
# measurement (var_meas = 0: kramer, schober) a 1e-15 perturbation of the prior variance grows ~10x per step in the
# weakly observed high-derivative components *of the oracle itself* (x stays accurate), and at n_deriv = 5, N = 100 the
# oracle's LU meets an exactly singular matrix.  Trajectory parity is therefore asserted over long horizons only for
# interrogate_rodeo (var_meas = W Sigma- W^T regularises S), and over 5 steps for the exact-measurement variants.
@pytest.mark.parametrize("n_vars,n_deriv,itg,N,tol", [
    (4, 3, "rodeo", 24, 1e-8), (6, 3, "rodeo", 24, 1e-8), (24, 3, "rodeo", 8, 1e-8),        # p = 12, 18, 72
    (4, 3, "kramer", 5, 1e-7), (4, 3, "schober", 5, 1e-7), (6, 5, "kramer", 5, 1e-6), (14, 5, "kramer", 4, 1e-6)])
def test_dense_parity(ra, n_vars, n_deriv, itg, N, tol):
    from rodeo_amd import _lib
    t_max, B = N / 24.0, 3
    s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B)
    ode_d = ra.ode.linear_dense(n_vars, n_deriv)
    ode_o = odes.make_linear_dense(s["A"], n_deriv)
    g, o = getattr(ra.interrogate, "interrogate_" + itg), getattr(oi, "interrogate_" + itg)
    plan = ra.SolvePlan(ode_d, s["W"], s["x0"], 0.0, t_max, N, g, s["prior"], A=s["A"])
    plan.mv(None)
    assert plan.layout == _lib.LAYOUT_TRAJ_MAJOR
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(None, ode_o, s["W"], s["x0"], 0.0, t_max, N, o, s["prior"])
    p = n_vars * n_deriv
    assert m.shape == (B, N + 1, 1, p) and v.shape == (B, N + 1, 1, p, p)
    scale_m = np.max(np.abs(mo), axis=(0, 1, 2))                    # per state component (derivatives grow)
    assert np.max(np.abs(m - mo) / np.maximum(scale_m, 1e-300)) < tol
    dv = np.sqrt(np.abs(np.einsum("bnkii->bnki", vo)).max(axis=(0, 1, 2)))     # per-component sd scale
    assert np.max(np.abs(v - vo) / (dv[:, None] * dv[None, :] + 1e-300)) < 100 * tol
    np.testing.assert_array_equal(m[:, 0], s["x0"]); assert np.all(v[:, 0] == 0)
    # filter only
    plan.filter(None)
    mf, vf = plan.state_host()
    fo = scan.solve_filter(None, ode_o, s["W"], s["x0"], 0.0, t_max, N, o, *s["prior"])
    assert np.max(np.abs(mf - fo["state_filt"][0]) / np.maximum(scale_m, 1e-300)) < tol


def test_dense_matches_exact_solution_and_api(ra):
    """solve_mv drop-in call with the dense ODE converges to expm(A t) x0; unbatched shapes like the reference."""
    n_vars, n_deriv, N, t_max = 5, 3, 200, 1.0
    s = dense_problem(ra, n_vars, n_deriv, N, t_max)
    m, v = ra.solve_mv(None, ra.ode.linear_dense(n_vars, n_deriv), s["W"], s["x0"], 0.0, t_max, N,
                       ra.interrogate.interrogate_rodeo, s["prior"], A=s["A"])
    assert m.shape == (N + 1, 1, n_vars * n_deriv) and v.shape == (N + 1, 1, n_vars * n_deriv, n_vars * n_deriv)
    exact = expm(s["A"] * t_max) @ s["x0v"]
    assert np.max(np.abs(m[-1, 0, ::n_deriv] - exact)) < 1e-3
    # the same posterior through the square-root form (tests/test_gpu_dense_sqrt.py holds its parity tests)
    Q, R = s["prior"]
    m2, L2 = ra.solve_mv(None, ra.ode.linear_dense(n_vars, n_deriv), s["W"], s["x0"], 0.0, t_max, N,
                         ra.interrogate.interrogate_kramer, (Q, np.linalg.cholesky(R)), kalman_type="square-root", A=s["A"])
    assert m2.shape == m.shape and L2.shape == v.shape
    assert np.max(np.abs(m2[-1, 0, ::n_deriv] - exact)) < 1e-3


def test_config5_full_size_properties(ra):
    """
    BASELINE.json config 5 at full size (stiff linear ODE, n_vars=32, n_deriv=5 -> one dense 160-dim block, m=32,
    N=2000, B=256: 105 GB of results on the device).  With the exact-measurement interrogations the reference's
    covariance-form recursion is itself unstable on this stiff problem (the NumPy oracle's error against expm(A t) x0
    reaches 1e8 as well), so long-horizon properties are asserted with interrogate_rodeo and interrogate_kramer (the
    combination the config is timed with) is compared with the oracle over the first filter steps:
      * trajectory 0 against the oracle over all 2000 steps, and against the exact solution expm(A t) x0;
      * the last trajectory equals the same trajectory solved alone, bit for bit (no cross-workgroup leakage);
      * end conditions (solve.py:295-301).
    """
    from rodeo_amd import _lib
    n_vars, n_deriv, N, B = 32, 5, 2000, 256
    p = n_vars * n_deriv
    rng = np.random.default_rng(20243)
    lam = np.logspace(0, 3, n_vars)
    A = -np.diag(lam) + 0.1 * rng.standard_normal((n_vars, n_vars)) / np.sqrt(n_vars)
    Wb, _ = ra.utils.first_order_pad(lambda x, t: x, n_vars, n_deriv)
    W = block_diag(*[w for w in Wb])[None]
    prior = ra.indep_init(ra.ibm_init(1.0 / N, n_deriv, np.ones(n_vars)))
    x0v = 1.0 + 0.01 * rng.standard_normal((B, n_vars))
    X0 = np.zeros((B, n_vars, n_deriv)); X0[..., 0] = x0v; X0[..., 1] = x0v @ A.T
    X0 = X0.reshape(B, 1, p)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(A, n_deriv)
    plan = ra.SolvePlan(ode_d, W, X0, 0.0, 1.0, N, ra.interrogate.interrogate_rodeo, prior, A=A)
    plan.mv(None)
    assert plan.layout == _lib.LAYOUT_TRAJ_MAJOR
    m0, v0 = plan.mean_state.slice0_host(0), plan.var_state.slice0_host(0)          # (N+1, 1, p), (N+1, 1, p, p)
    mL, vL = plan.mean_state.slice0_host(B - 1), plan.var_state.slice0_host(B - 1)
    assert np.all(np.isfinite(m0)) and np.all(np.isfinite(v0))
    np.testing.assert_array_equal(m0[0], X0[0]); assert np.all(v0[0] == 0)
    for n in (N // 2, N):
        assert np.max(np.abs(m0[n, 0, ::n_deriv] - expm(A * n / N) @ x0v[0])) < 1e-5
    mo, vo = scan.solve_mv(None, ode_o, W, X0[0], 0.0, 1.0, N, oi.interrogate_rodeo, prior)
    scale_m = np.max(np.abs(mo), axis=(0, 1))
    assert np.max(np.abs(m0 - mo) / scale_m) < 1e-8
    dv = np.sqrt(np.abs(np.einsum("nkii->nki", vo)).max(axis=(0, 1)))
    assert np.max(np.abs(v0 - vo) / (dv[:, None] * dv[None, :])) < 1e-6
    del plan
    p1 = ra.SolvePlan(ode_d, W, X0[B - 1:], 0.0, 1.0, N, ra.interrogate.interrogate_rodeo, prior, A=A)
    p1.mv(None)
    np.testing.assert_array_equal(p1.mean_state.slice0_host(0), mL)
    np.testing.assert_array_equal(p1.var_state.slice0_host(0), vL)
    # the timed combination, first filter steps (same dt: t_max = 4 / N on 4 steps)
    pk = ra.SolvePlan(ode_d, W, X0[:3], 0.0, 4.0 / N, 4, ra.interrogate.interrogate_kramer, prior, A=A)
    pk.filter(None)
    mf, _ = pk.state_host()
    fo = scan.solve_filter(None, ode_o, W, X0[:3], 0.0, 4.0 / N, 4, oi.interrogate_kramer, *prior)
    mfo = fo["state_filt"][0]
    assert np.max(np.abs(mf - mfo) / np.maximum(np.max(np.abs(mfo), axis=(0, 1, 2)), 1e-300)) < 1e-6


def test_config5_timed_combination_full_horizon(ra):
    """
    The combination BASELINE config 5 is TIMED with (solve_mv + interrogate_kramer, stiff spectrum, N = 2000) over the whole
    horizon.  The reference's covariance-form recursion is unstable here in fp64 -- the oracle's own answer moves by
    orders of magnitude under a 1-ulp perturbation of the prior variance and its error against expm(A t) x0 reaches 1e8
    -- so "agrees with the oracle to 1e-8" cannot hold at late steps for ANY second implementation.  What can be asserted
    at every one of the 2000 steps, and is: the device's deviation from the oracle stays within a fixed factor of the
    oracle's own sensitivity, d_n <= 1000 max(s_n, 1e-12), where (per step, relative to the running scale of each state
    component)  d_n = |device - oracle|  and  s_n = |oracle(R (1 + 1e-15)) - oracle(R)|.
    The numbers of the run are printed (pytest -s) and quoted in DESIGN.md section 2.
    """
    n_vars, n_deriv, N, B = 32, 5, 2000, 2
    p = n_vars * n_deriv
    rng = np.random.default_rng(20243)
    lam = np.logspace(0, 3, n_vars)
    A = -np.diag(lam) + 0.1 * rng.standard_normal((n_vars, n_vars)) / np.sqrt(n_vars)
    Wb, _ = ra.utils.first_order_pad(lambda x, t: x, n_vars, n_deriv)
    W = block_diag(*[w for w in Wb])[None]
    prior = ra.indep_init(ra.ibm_init(1.0 / N, n_deriv, np.ones(n_vars)))
    x0v = 1.0 + 0.01 * rng.standard_normal((256, n_vars))[:B]
    X0 = np.zeros((B, n_vars, n_deriv)); X0[..., 0] = x0v; X0[..., 1] = x0v @ A.T
    X0 = X0.reshape(B, 1, p)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(A, n_deriv)
    plan = ra.SolvePlan(ode_d, W, X0, 0.0, 1.0, N, ra.interrogate.interrogate_kramer, prior, A=A)
    plan.filter(None)
    mf = plan.mean_state.slice0_host(0)[:, 0]                                       # (N+1, p) filtered means, trajectory 0
    with np.errstate(all="ignore"):
        fo = scan.solve_filter(None, ode_o, W, X0[0], 0.0, 1.0, N, oi.interrogate_kramer, *prior)["state_filt"][0][:, 0]
        fp = scan.solve_filter(None, ode_o, W, X0[0], 0.0, 1.0, N, oi.interrogate_kramer, prior[0],
                               prior[1] * (1.0 + 1e-15))["state_filt"][0][:, 0]
    scale = np.maximum.accumulate(np.abs(fo), axis=0) + 1e-300                      # running per-component scale
    ok = np.all(np.isfinite(fo), axis=1) & np.all(np.isfinite(fp), axis=1)
    d = np.max(np.abs(mf - fo) / scale, axis=1)
    s_ = np.max(np.abs(fp - fo) / scale, axis=1)
    n_ok = int(np.argmin(ok)) if not ok.all() else N + 1                           # first step at which the oracle is no longer finite
    first_bad = lambda x, tol: int(np.argmax(x[:n_ok] > tol)) if np.any(x[:n_ok] > tol) else n_ok
    print(f"\nC5 + kramer, filter, trajectory 0: oracle finite up to step {n_ok - 1}; device-oracle deviation exceeds 1e-8 "
          f"from step {first_bad(d, 1e-8)}, 1e-4 from step {first_bad(d, 1e-4)}; the oracle's own 1e-15-perturbation sensitivity "
          f"exceeds 1e-8 from step {first_bad(s_, 1e-8)}, 1e-4 from step {first_bad(s_, 1e-4)}; max d_n / max(s_n, 1e-12) = "
          f"{np.max(d[:n_ok] / np.maximum(s_[:n_ok], 1e-12)):.3g}; |x - expm| at t = 1: device "
          f"{np.max(np.abs(mf[N, ::n_deriv] - expm(A) @ x0v[0])):.3g}, oracle {np.max(np.abs(fo[N, ::n_deriv] - expm(A) @ x0v[0])):.3g}")
    assert n_ok >= 5
    assert np.all(d[:n_ok] <= 1000.0 * np.maximum(s_[:n_ok], 1e-12))
    assert np.all(d[:5] < 1e-6)


@pytest.mark.parametrize("n_vars,n_deriv,N", [(40, 5, 3), (66, 5, 2), (44, 4, 3)])
def test_dense_large_blocks(ra, n_vars, n_deriv, N):
    """
    p = 200 and 176 (GEMM macro-blocks beyond 160 x 160, LU panels with five register rows per lane) and p = 330 (LU
    beyond the blocked solver's limit: unblocked fallback) against the oracle, interrogate_rodeo, a few steps.
    """
    t_max, B = N / 24.0, 2
    s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(s["A"], n_deriv)
    plan = ra.SolvePlan(ode_d, s["W"], s["x0"], 0.0, t_max, N, ra.interrogate.interrogate_rodeo, s["prior"], A=s["A"])
    plan.mv(None)
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(None, ode_o, s["W"], s["x0"], 0.0, t_max, N, oi.interrogate_rodeo, s["prior"])
    scale_m = np.max(np.abs(mo), axis=(0, 1, 2))
    assert np.max(np.abs(m - mo) / np.maximum(scale_m, 1e-300)) < 1e-8
    dv = np.sqrt(np.abs(np.einsum("bnkii->bnki", vo)).max(axis=(0, 1, 2)))
    assert np.max(np.abs(v - vo) / (dv[:, None] * dv[None, :] + 1e-300)) < 1e-6


def test_dense_general_prior_weight(ra):
    """A prior weight matrix that is NOT block diagonal takes the dense-GEMM predict path (no skipped terms)."""
    n_vars, n_deriv, N = 6, 3, 12
    t_max, B = N / 24.0, 3
    s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B)
    rng = np.random.default_rng(5)
    Q, R = s["prior"]
    Q = Q + 1e-3 * rng.standard_normal(Q.shape)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(s["A"], n_deriv)
    plan = ra.SolvePlan(ode_d, s["W"], s["x0"], 0.0, t_max, N, ra.interrogate.interrogate_rodeo, (Q, R), A=s["A"])
    plan.mv(None)
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(None, ode_o, s["W"], s["x0"], 0.0, t_max, N, oi.interrogate_rodeo, (Q, R))
    scale_m = np.max(np.abs(mo), axis=(0, 1, 2))
    assert np.max(np.abs(m - mo) / np.maximum(scale_m, 1e-300)) < 1e-8
    dv = np.sqrt(np.abs(np.einsum("bnkii->bnki", vo)).max(axis=(0, 1, 2)))
    assert np.max(np.abs(v - vo) / (dv[:, None] * dv[None, :] + 1e-300)) < 1e-6


def test_dense_batched_ode_matrix(ra):
    """A different matrix A per trajectory (parameter ``A`` of shape (B, n_vars, n_vars)): kramer over a few steps and
    rodeo over a longer horizon against the oracle, trajectory by trajectory."""
    n_vars, n_deriv, B = 6, 3, 4
    for itg, N, tol in (("rodeo", 20, 1e-8), ("kramer", 5, 1e-7)):
        t_max = N / 24.0
        s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B)
        rng = np.random.default_rng(11)
        A = s["A"][None] + 0.05 * rng.standard_normal((B, n_vars, n_vars))
        X0 = s["x0"].copy().reshape(B, n_vars, n_deriv)
        X0[..., 1] = np.einsum("bij,bj->bi", A, X0[..., 0])
        X0 = X0.reshape(B, 1, -1)
        g, o = getattr(ra.interrogate, "interrogate_" + itg), getattr(oi, "interrogate_" + itg)
        m, v = ra.solve_mv(None, ra.ode.linear_dense(n_vars, n_deriv), s["W"], X0, 0.0, t_max, N, g, s["prior"], A=A)
        for b in range(B):
            mo, vo = scan.solve_mv(None, odes.make_linear_dense(A[b], n_deriv), s["W"], X0[b], 0.0, t_max, N, o, s["prior"])
            scale_m = np.max(np.abs(mo), axis=(0, 1))
            assert np.max(np.abs(m[b] - mo) / np.maximum(scale_m, 1e-300)) < tol


def _ring_problem(ra, n_vars, n_deriv, N, t_max, B, seed=5):
    """A nonlinear coupled ring x_i' = -k0 x_i + k1 sin(x_{i+1}) - 0.1 x_i^3 + 0.05 cos t in the reference's NON-BLOCK
    form: one block of n_vars * n_deriv states, ode_weight (1, n_vars, p), an ordinary Python ode_fun returning
    (1, n_vars) -- what examples/solve_nb.py does with its own ODEs."""
    rng = np.random.default_rng(seed)
    p = n_vars * n_deriv
    idx0 = np.arange(n_vars) * n_deriv

    def fun(X, t, kc):
        x = X[0, ::n_deriv]
        return np.array([[-kc[0] * x[i] + kc[1] * np.sin(x[(i + 1) % n_vars]) - 0.1 * x[i] ** 3 + 0.05 * np.cos(t)
                          for i in range(n_vars)]])

    def host(X, t, kc):
        kc = np.asarray(kc, dtype=np.float64)
        x = X[..., 0, idx0]
        k0, k1 = kc[..., 0:1], kc[..., 1:2]
        return (-k0 * x + k1 * np.sin(np.roll(x, -1, axis=-1)) - 0.1 * x ** 3 + 0.05 * np.cos(t))[..., None, :]

    def jac(X, t, kc):
        kc = np.asarray(kc, dtype=np.float64)
        x = X[..., 0, idx0]
        J = np.zeros(X.shape[:-2] + (1, n_vars, p))
        for i in range(n_vars):
            J[..., 0, i, idx0[i]] = -kc[..., 0] - 0.3 * x[..., i] ** 2
            J[..., 0, i, idx0[(i + 1) % n_vars]] += kc[..., 1] * np.cos(x[..., (i + 1) % n_vars])
        return J
    Wb, _ = ra.utils.first_order_pad(lambda x, t: x, n_vars, n_deriv)
    W = block_diag(*[w for w in Wb])[None]
    prior = ra.indep_init(ra.ibm_init(t_max / N, n_deriv, 0.5 * np.ones(n_vars)))
    kc = np.array([0.8, 0.6]) * np.exp(0.1 * rng.standard_normal((B, 2)))
    xv = 0.5 * rng.standard_normal((B, n_vars))
    X0 = np.zeros((B, n_vars, n_deriv))
    X0[..., 0] = xv
    X0 = X0.reshape(B, 1, p)
    X0[:, 0, idx0 + 1] = host(X0, 0.0, kc)[:, 0, :]               # x' = f(x, 0)
    return dict(fun=fun, o_ode=odes.ODE("ring_nb", 1, n_vars, host, jac), W=W, prior=prior, kc=kc, x0=X0)


@pytest.mark.parametrize("n_vars,n_deriv,itg,N", [(8, 3, "rodeo", 30), (8, 3, "kramer", 4), (8, 3, "schober", 4),
                                                   (5, 2, "rodeo", 30), (20, 4, "rodeo", 12), (20, 4, "kramer", 3)])
def test_dense_path_takes_any_traced_ode_fun(ra, n_vars, n_deriv, itg, N):
    """The non-block form with an ordinary (nonlinear, time-dependent, parametrised) Python ode_fun beyond the sizes of
    the lane kernels: traced, its dense Jacobian by forward-mode duals (one direction per thread), run on the dense MFMA
    path with the step cut at the interrogation (solve_dense_itg_kernels.hpp); solve_mv and the filter against the
    oracle with the analytic Jacobian.  Exact-measurement interrogations over a few steps only (see test_dense_parity)."""
    from rodeo_amd import _lib
    B, t_max = 3, N / 30.0
    s = _ring_problem(ra, n_vars, n_deriv, N, t_max, B)
    g, o = getattr(ra.interrogate, "interrogate_" + itg), getattr(oi, "interrogate_" + itg)
    plan = ra.SolvePlan(s["fun"], s["W"], s["x0"], 0.0, t_max, N, g, s["prior"], kc=s["kc"])
    plan.mv(None)
    assert plan.layout == _lib.LAYOUT_TRAJ_MAJOR
    plan.dev.profile_enable(True)
    plan.mv(None)
    assert [k for k, _ in plan.dev.profile_last()] == ["dense_fwd_kernel<user, stepwise>", "dense_bwd_mv_kernel"]
    plan.dev.profile_enable(False)
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(None, s["o_ode"], s["W"], s["x0"], 0.0, t_max, N, o, s["prior"], kc=s["kc"])
    p = n_vars * n_deriv
    assert m.shape == (B, N + 1, 1, p) and v.shape == (B, N + 1, 1, p, p)
    scale_m = np.maximum(np.max(np.abs(mo), axis=(0, 1, 2)), 1e-3)
    assert np.max(np.abs(m - mo) / scale_m) < 1e-7
    dv = np.sqrt(np.abs(np.einsum("bnkii->bnki", vo)).max(axis=(0, 1, 2)))
    assert np.max(np.abs(v - vo) / (dv[:, None] * dv[None, :] + 1e-300)) < 1e-5
    plan.filter(None)
    mf, _ = plan.state_host()
    fo = scan.solve_filter(None, s["o_ode"], s["W"], s["x0"], 0.0, t_max, N, o, *s["prior"], kc=s["kc"])
    assert np.max(np.abs(mf - fo["state_filt"][0]) / scale_m) < 1e-7


@pytest.mark.parametrize("n_vars,n_deriv,N", [(4, 3, 16), (6, 3, 12), (24, 3, 6), (14, 5, 5)])
def test_dense_chkrebtii_and_solve_sim(ra, n_vars, n_deriv, N):
    """interrogate_chkrebtii (x ~ N(mu-, Sigma-) through the PSD-safe Cholesky factor, interrogate.py:22-34) and solve_sim
    (solve.py:137-205: terminal draw, backward draws from N(mu_f + G (x - mu-), Sigma_f - G T^T)) on the dense path, built-in
    linear right-hand side, against the oracle with the shared Philox stream."""
    import functools
    t_max, B = N / 24.0, 3
    s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(s["A"], n_deriv)
    p = n_vars * n_deriv
    g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard")
    o = functools.partial(oi.interrogate_chkrebtii, kalman_type="standard")
    plan = ra.SolvePlan(ode_d, s["W"], s["x0"], 0.0, t_max, N, g, s["prior"], A=s["A"])
    plan.mv(7)
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(7, ode_o, s["W"], s["x0"], 0.0, t_max, N, o, s["prior"])
    scale_m = np.maximum(np.max(np.abs(mo), axis=(0, 1, 2)), 1e-3)
    assert np.max(np.abs(m - mo) / scale_m) < 1e-7
    dv = np.sqrt(np.abs(np.einsum("bnkii->bnki", vo)).max(axis=(0, 1, 2)))
    assert np.max(np.abs(v - vo) / (dv[:, None] * dv[None, :] + 1e-300)) < 1e-5
    plan.mv(8)
    assert np.max(np.abs(plan.state_host()[0] - m)) > 0          # another key, another path
    for gg, oo, key in ((ra.interrogate.interrogate_rodeo, oi.interrogate_rodeo, 3), (g, o, 4)):
        x = ra.solve_sim(key, ode_d, s["W"], s["x0"], 0.0, t_max, N, gg, s["prior"], A=s["A"])
        xo = scan.solve_sim(key, ode_o, s["W"], s["x0"], 0.0, t_max, N, oo, s["prior"])
        assert x.shape == (B, N + 1, 1, p)
        np.testing.assert_array_equal(x[:, 0], s["x0"])
        sx = np.maximum(np.max(np.abs(xo), axis=(0, 1, 2)), 1e-3)
        assert np.max(np.abs(x - xo) / sx) < 1e-6


def test_dense_traced_ode_fun_chkrebtii_and_solve_sim(ra):
    """The same two on the dense path with a traced nonlinear ode_fun (the run-time-built interrogation kernel evaluates f at
    the draw)."""
    import functools
    n_vars, n_deriv, N, B = 8, 3, 16, 3
    t_max = N / 30.0
    s = _ring_problem(ra, n_vars, n_deriv, N, t_max, B)
    g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard")
    o = functools.partial(oi.interrogate_chkrebtii, kalman_type="standard")
    args = (s["W"], s["x0"], 0.0, t_max, N)
    m, v = ra.solve_mv(5, s["fun"], *args, g, s["prior"], kc=s["kc"])
    mo, vo = scan.solve_mv(5, s["o_ode"], *args, o, s["prior"], kc=s["kc"])
    scale_m = np.maximum(np.max(np.abs(mo), axis=(0, 1, 2)), 1e-3)
    assert np.max(np.abs(m - mo) / scale_m) < 1e-7
    x = ra.solve_sim(6, s["fun"], *args, g, s["prior"], kc=s["kc"])
    xo = scan.solve_sim(6, s["o_ode"], *args, o, s["prior"], kc=s["kc"])
    assert np.max(np.abs(x - xo) / np.maximum(np.max(np.abs(xo), axis=(0, 1, 2)), 1e-3)) < 1e-6


def test_dense_solve_filter_returns_the_predictions(ra):
    """_solve_filter (solve.py:31-122: state_pred and state_filt, index 0 = (ode_init, 0) in both) on the dense path, for the
    built-in linear right-hand side and a traced one."""
    n_vars, n_deriv, N, B = 6, 3, 10, 3
    t_max = N / 24.0
    s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B)
    out = ra.solve._solve_filter(None, ra.ode.linear_dense(n_vars, n_deriv), s["W"], s["x0"], 0.0, t_max, N,
                                 ra.interrogate.interrogate_rodeo, *s["prior"], A=s["A"])
    ref = scan.solve_filter(None, odes.make_linear_dense(s["A"], n_deriv), s["W"], s["x0"], 0.0, t_max, N,
                            oi.interrogate_rodeo, *s["prior"])
    r = _ring_problem(ra, 8, 3, N, N / 30.0, B)
    out2 = ra.solve._solve_filter(None, r["fun"], r["W"], r["x0"], 0.0, N / 30.0, N, ra.interrogate.interrogate_rodeo,
                                  *r["prior"], kc=r["kc"])
    ref2 = scan.solve_filter(None, r["o_ode"], r["W"], r["x0"], 0.0, N / 30.0, N, oi.interrogate_rodeo, *r["prior"], kc=r["kc"])
    for o, f, x0 in ((out, ref, s["x0"]), (out2, ref2, r["x0"])):
        for k in ("state_pred", "state_filt"):
            m, v = o[k]
            mo, vo = f[k]
            assert m.shape == mo.shape and v.shape == vo.shape
            sm = np.maximum(np.max(np.abs(mo), axis=(0, 1, 2)), 1e-3)
            assert np.max(np.abs(m - mo) / sm) < 1e-8, k
            dv = np.sqrt(np.abs(np.einsum("bnkii->bnki", vo)).max(axis=(0, 1, 2)))
            assert np.max(np.abs(v - vo) / (dv[:, None] * dv[None, :] + 1e-300)) < 1e-6, k
            np.testing.assert_array_equal(m[:, 0], x0)
            assert np.all(v[:, 0] == 0)


@pytest.mark.parametrize("n_vars,n_deriv,mode", [(32, 5, "mv"), (20, 5, "mv"), (26, 3, "mv"), (17, 4, "mv"), (13, 5, "sim"),
                                                  (14, 5, "mv")])
def test_dense_lu_register_resident_equals_panel_loop(ra, n_vars, n_deriv, mode, monkeypatch):
    """The register-resident forward elimination (solve_dense_lu_regs.hpp: [Sigma- | T^T] in the registers of the eight
    waves, pivoting without row movement) replaces the panel loop over global memory of wg_lu_solve with the SAME
    operations in the same order -- same pivots (LAPACK's, ties included), same substitutions, same four MFMAs per tile --
    so the smoothed moments must have the same BITS (p = 160, 100, 78, 68, 65 [, 70: n_bstate <= 64 keeps the old
    path, both legs equal trivially]; RK_DENSE_LU=global selects the panel loop)."""
    N, B = 6, 5
    t_max = N / 24.0
    s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B, seed=n_vars)
    ode_d = ra.ode.linear_dense(n_vars, n_deriv)
    out = {}
    for leg in ("regs", "global"):
        monkeypatch.setenv("RK_DENSE_LU", leg)
        plan = ra.SolvePlan(ode_d, s["W"], s["x0"], 0.0, t_max, N, ra.interrogate.interrogate_rodeo, s["prior"], A=s["A"])
        if mode == "mv":
            plan.mv(None)
            out[leg] = plan.state_host()
        else:
            plan.sim(7)
            out[leg] = (np.array(plan.x_host()),)
    for a_, b_ in zip(out["regs"], out["global"]):
        assert np.all(np.isfinite(a_))
        np.testing.assert_array_equal(a_, b_)
