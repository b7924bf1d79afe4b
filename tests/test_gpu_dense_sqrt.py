"""
GPU parity of the DENSE square-root filter / smoother / sampler (kalman_type="square-root" on the non-block path of
prior.indep_init: src/rodeo/kalmantv/square_root.py:30-261, src/rodeo/utils.py:10-24 inside src/rodeo/solve.py) against
the NumPy oracle's square-root scan on the same inputs.  Factors are compared as L L^T (a QR determines R only up to the
signs of its rows), means directly.  This is the form in which BASELINE config 5 is numerically meaningful: its
covariance-form recursion ends 6e8 away from expm(A) x0, the square-root form 2e-8.
"""
import numpy as np
import pytest
from scipy.linalg import block_diag, expm
from threadpoolctl import threadpool_limits
from oracle import scan, odes, interrogations as oi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rodeo_amd
    return rodeo_amd


def _sq(L):
    return L @ np.swapaxes(L, -1, -2)


def dense_problem(ra, n_vars, n_deriv, n_steps, t_max, B=None, seed=0, stiff=False):
    rng = np.random.default_rng(20243 + seed)
    lam = np.logspace(0, 3, n_vars) if stiff else np.linspace(0.5, 2.0, n_vars)
    A = -np.diag(lam) + 0.1 * rng.standard_normal((n_vars, n_vars)) / np.sqrt(n_vars)
    Wb, _ = ra.utils.first_order_pad(lambda x, t: x, n_vars, n_deriv)
    W = block_diag(*[w for w in Wb])[None]
    Q, R = ra.indep_init(ra.ibm_init(t_max / n_steps, n_deriv, np.ones(n_vars)))
    x0v = np.ones(n_vars) if B is None else 1.0 + 0.1 * rng.standard_normal((B, n_vars))
    X0 = np.zeros(x0v.shape[:-1] + (n_vars, n_deriv))
    X0[..., 0] = x0v
    X0[..., 1] = x0v @ A.T
    X0 = X0.reshape(x0v.shape[:-1] + (1, n_vars * n_deriv))
    return dict(A=A, W=W, prior=(Q, np.linalg.cholesky(R)), prior_cov=(Q, R), x0=X0, x0v=x0v)


def _check(m, L, mo, Lo, tol_m, tol_v, Rh):
    """means relative to each state component's scale; L L^T entrywise relative to sd_i sd_j, sd = the component's largest
    posterior standard deviation -- floored at 1e-4 of its one-step prior standard deviation: a component that an exact
    measurement pins (schober: the measured derivative) has variance 0 + rounding noise and no scale of its own."""
    scale_m = np.maximum(np.max(np.abs(mo), axis=tuple(range(mo.ndim - 1))), 1e-300)      # per state component
    em = np.max(np.abs(m - mo) / scale_m)
    vo = _sq(Lo)
    dv = np.sqrt(np.abs(np.einsum("...ii->...i", vo)).max(axis=tuple(range(vo.ndim - 2))))
    dv = np.maximum(dv, 1e-4 * np.sqrt(np.einsum("...ii->...i", _sq(Rh))).reshape(-1))
    ev = np.max(np.abs(_sq(L) - vo) / (dv[:, None] * dv[None, :]))
    assert em < tol_m and ev < tol_v, (em, ev)
    return em, ev


@pytest.mark.parametrize("n_vars,n_deriv,itg,N", [
    (4, 3, "rodeo", 24), (4, 3, "kramer", 24), (4, 3, "schober", 24),            # p = 12: one QR panel, one row block
    (6, 3, "kramer", 24), (7, 5, "kramer", 12), (7, 5, "rodeo", 12),              # p = 18, 35: ragged panels
    (24, 3, "kramer", 8), (24, 3, "rodeo", 8),                                     # p = 72, m = 24
    (16, 3, "kramer", 8), (16, 4, "schober", 8), (16, 5, "rodeo", 8),              # p = 48, 64, 80: the structured predict QR
    (32, 5, "kramer", 6)])                                                         # config 5's shape (p = 160, m = 32)
def test_dense_sqrt_parity(ra, n_vars, n_deriv, itg, N):
    """solve_mv and the filter in square-root form against the oracle, and (exact measurement) against the covariance form."""
    from rodeo_amd import _lib
    t_max, B = N / 24.0, 3
    s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(s["A"], n_deriv)
    g, o = getattr(ra.interrogate, "interrogate_" + itg), getattr(oi, "interrogate_" + itg)
    p = n_vars * n_deriv
    plan = ra.SolvePlan(ode_d, s["W"], s["x0"], 0.0, t_max, N, g, s["prior"], kalman_type="square-root", A=s["A"])
    plan.mv(None)
    assert plan.layout == _lib.LAYOUT_TRAJ_MAJOR
    m, L = plan.state_host()
    assert m.shape == (B, N + 1, 1, p) and L.shape == (B, N + 1, 1, p, p)
    assert np.all(np.isfinite(m)) and np.all(np.isfinite(L))
    assert np.all(np.triu(L, 1) == 0.0)                                     # lower factors, exact zeros above the diagonal
    with threadpool_limits(limits=1):
        mo, Lo = scan.solve_mv(None, ode_o, s["W"], s["x0"], 0.0, t_max, N, o, s["prior"], kalman_type="square-root")
    _check(m, L, mo, Lo, 1e-9, 1e-8, s["prior"][1])
    np.testing.assert_array_equal(m[:, 0], s["x0"]); assert np.all(L[:, 0] == 0)
    plan.filter(None)
    mf, Lf = plan.state_host()
    with threadpool_limits(limits=1):
        fo = scan.solve_filter(None, ode_o, s["W"], s["x0"], 0.0, t_max, N, o, *s["prior"], kalman_funs=scan.sqrt_ops)
    _check(mf, Lf, fo["state_filt"][0], fo["state_filt"][1], 1e-9, 1e-8, s["prior"][1])


def test_dense_sqrt_filter_predictions(ra):
    """_solve_filter's four outputs in square-root form (RK_FLAG_STORE_PRED: the predicted factors are the caller's array)."""
    n_vars, n_deriv, N = 6, 4, 10
    s = dense_problem(ra, n_vars, n_deriv, N, 0.4, B=2)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(s["A"], n_deriv)
    out = ra.solve._solve_filter(None, ode_d, s["W"], s["x0"], 0.0, 0.4, N, ra.interrogate.interrogate_kramer, *s["prior"],
                                 kalman_funs=ra.kalmantv.square_root, A=s["A"])
    fo = scan.solve_filter(None, ode_o, s["W"], s["x0"], 0.0, 0.4, N, oi.interrogate_kramer, *s["prior"], kalman_funs=scan.sqrt_ops)
    for k in ("state_pred", "state_filt"):
        _check(out[k][0], out[k][1], fo[k][0], fo[k][1], 1e-9, 1e-8, s["prior"][1])


def test_dense_sqrt_sample_paths(ra):
    """solve_sim in square-root form on the shared Philox stream (interrogate_rodeo: var_meas > 0 keeps every factor of
    full rank, so a factor is unique once its diagonal signs are fixed and device and oracle draw the same path)."""
    n_vars, n_deriv, N = 8, 3, 12
    s = dense_problem(ra, n_vars, n_deriv, N, 0.5, B=3)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(s["A"], n_deriv)
    x = ra.solve_sim(11, ode_d, s["W"], s["x0"], 0.0, 0.5, N, ra.interrogate.interrogate_rodeo, s["prior"],
                     kalman_type="square-root", A=s["A"])
    xo = scan.solve_sim(11, ode_o, s["W"], s["x0"], 0.0, 0.5, N, oi.interrogate_rodeo, s["prior"], kalman_type="square-root")
    assert x.shape == xo.shape == (3, N + 1, 1, n_vars * n_deriv)
    np.testing.assert_array_equal(x[:, 0], s["x0"])
    scale = np.maximum(np.max(np.abs(xo), axis=(0, 1, 2)), 1e-300)
    assert np.max(np.abs(x - xo) / scale) < 1e-7


@pytest.mark.parametrize("n_vars,n_deriv", [(16, 3), (16, 5), (32, 5)])
def test_dense_sqrt_structured_predict_qr(ra, n_vars, n_deriv, monkeypatch):
    """The predict step's stack [R^(1/2)T ; (Q L)^T] of an indep_init prior is upper triangular over block upper triangular;
    wg_qr_r (rodeo_amd/csrc/solve_dense_sqrt.hpp) leaves the structural zeros out of its panels.  Same factors as the generic
    QR of the same stack (RK_DENSE_STRUCTURED=0) up to rounding, and a prior WITHOUT the structure (Q with an entry outside
    its diagonal blocks; a full square root of R) takes the generic path by itself and still matches the oracle."""
    N, B = 6, 2
    t_max = N / 24.0
    s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(s["A"], n_deriv)
    p = n_vars * n_deriv
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RK_DENSE_STRUCTURED", flag)
        plan = ra.SolvePlan(ode_d, s["W"], s["x0"], 0.0, t_max, N, ra.interrogate.interrogate_kramer, s["prior"],
                            kalman_type="square-root", A=s["A"])
        plan.mv(None)
        res[flag] = plan.state_host()
    monkeypatch.delenv("RK_DENSE_STRUCTURED")
    _check(res["1"][0], res["1"][1], res["0"][0], res["0"][1], 1e-9, 1e-9, s["prior"][1])      # (2e-11 / 8e-14 measured at p = 160)
    assert np.all(np.triu(res["1"][1], 1) == 0.0)
    # a prior without the structure: Q coupled across two blocks, R's symmetric square root instead of its Cholesky factor
    Q, Rh = s["prior"]                                   # (1, p, p): one block
    Q2 = Q.copy(); Q2[0, 0, n_deriv] = 0.01 * Q[0, 0, 1]
    w_, V_ = np.linalg.eigh(Rh[0] @ Rh[0].T)
    Rsym = ((V_ * np.sqrt(w_)) @ V_.T)[None]
    for prior in ((Q2, Rh), (Q, Rsym)):
        plan = ra.SolvePlan(ode_d, s["W"], s["x0"], 0.0, t_max, N, ra.interrogate.interrogate_kramer, prior,
                            kalman_type="square-root", A=s["A"])
        plan.mv(None)
        m, L = plan.state_host()
        with threadpool_limits(limits=1):
            mo, Lo = scan.solve_mv(None, ode_o, s["W"], s["x0"], 0.0, t_max, N, oi.interrogate_kramer, prior, kalman_type="square-root")
        _check(m, L, mo, Lo, 1e-9, 1e-8, Rh)


@pytest.mark.parametrize("n_vars,n_deriv,N", [(36, 5, 3), (44, 4, 3)])
def test_dense_sqrt_large_blocks(ra, n_vars, n_deriv, N):
    """p = 180, 176: the smoother's 3p x p stack exceeds the LDS panel (unblocked Householder fallback), the triangular
    solves exceed the register-resident form; a few steps against the oracle."""
    t_max, B = N / 24.0, 2
    s = dense_problem(ra, n_vars, n_deriv, N, t_max, B=B)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(s["A"], n_deriv)
    m, L = ra.solve_mv(None, ode_d, s["W"], s["x0"], 0.0, t_max, N, ra.interrogate.interrogate_kramer, s["prior"],
                       kalman_type="square-root", A=s["A"])
    with threadpool_limits(limits=1):
        mo, Lo = scan.solve_mv(None, ode_o, s["W"], s["x0"], 0.0, t_max, N, oi.interrogate_kramer, s["prior"],
                               kalman_type="square-root")
    _check(m, L, mo, Lo, 1e-9, 1e-8, s["prior"][1])


def test_dense_sqrt_traced_python_rhs(ra):
    """The square-root dense path with an ordinary (traced, nonlinear, time-dependent, parametrised) Python right-hand side
    in the non-block form: the interrogation kernel built by hiprtc sits between the two halves of the square-root
    forward step; the oracle uses the analytic Jacobian."""
    from test_gpu_dense import _ring_problem
    n_vars, n_deriv, N, B = 12, 3, 8, 2
    t_max = N / 30.0
    s = _ring_problem(ra, n_vars, n_deriv, N, t_max, B)
    pr = (s["prior"][0], np.linalg.cholesky(s["prior"][1]))
    for itg in ("kramer", "rodeo"):
        g, o = getattr(ra.interrogate, "interrogate_" + itg), getattr(oi, "interrogate_" + itg)
        m, L = ra.solve_mv(None, s["fun"], s["W"], s["x0"], 0.0, t_max, N, g, pr, kalman_type="square-root", kc=s["kc"])
        mo, Lo = scan.solve_mv(None, s["o_ode"], s["W"], s["x0"], 0.0, t_max, N, o, pr, kalman_type="square-root", kc=s["kc"])
        _check(m, L, mo, Lo, 1e-8, 1e-7, pr[1])


def test_dense_sqrt_chkrebtii_is_refused_like_the_reference(ra):
    """interrogate.py:36-42: in square-root mode the draw is mu- + (W L-) z, (n_bmeas,) + (n_bstate,): no broadcast for the
    dense shapes -- the reference raises, so does the library (with its own message)."""
    import functools
    from rodeo_amd._lib import RodeoKalmanError
    s = dense_problem(ra, 4, 3, 4, 0.2)
    g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="square-root")
    with pytest.raises(RodeoKalmanError, match="does not broadcast"):
        ra.solve_mv(1, ra.ode.linear_dense(4, 3), s["W"], s["x0"], 0.0, 0.2, 4, g, s["prior"], kalman_type="square-root", A=s["A"])


def test_config5_square_root_full_horizon(ra):
    """
    BASELINE config 5's problem (stiff linear ODE, n_vars = 32, n_deriv = 5 -> p = 160, m = 32, N = 2000, interrogate_kramer,
    solve_mv) in square-root form, trajectory 0 and the last of the 256: device against the oracle's square-root scan over
    ALL 2000 steps (means to 1e-6 of each component's scale, L L^T to 1e-6), and against the exact solution expm(A t) x0
    to 1e-5 -- where the covariance form of the same reference ends 6e8 away (test_gpu_dense.py).
    """
    n_vars, n_deriv, N = 32, 5, 2000
    p = n_vars * n_deriv
    rng = np.random.default_rng(20243)
    lam = np.logspace(0, 3, n_vars)
    A = -np.diag(lam) + 0.1 * rng.standard_normal((n_vars, n_vars)) / np.sqrt(n_vars)
    Wb, _ = ra.utils.first_order_pad(lambda x, t: x, n_vars, n_deriv)
    W = block_diag(*[w for w in Wb])[None]
    Q, R = ra.indep_init(ra.ibm_init(1.0 / N, n_deriv, np.ones(n_vars)))
    pr = (Q, np.linalg.cholesky(R))
    x0v = (1.0 + 0.01 * rng.standard_normal((256, n_vars)))[[0, 255]]
    X0 = np.zeros((2, n_vars, n_deriv)); X0[..., 0] = x0v; X0[..., 1] = x0v @ A.T
    X0 = X0.reshape(2, 1, p)
    ode_d, ode_o = ra.ode.linear_dense(n_vars, n_deriv), odes.make_linear_dense(A, n_deriv)
    plan = ra.SolvePlan(ode_d, W, X0, 0.0, 1.0, N, ra.interrogate.interrogate_kramer, pr, kalman_type="square-root", A=A)
    plan.mv(None)
    m, L = plan.state_host()
    assert np.all(np.isfinite(m)) and np.all(np.isfinite(L))
    for b in range(2):
        for n in (N // 10, N // 2, N):
            assert np.max(np.abs(m[b, n, 0, ::n_deriv] - expm(A * n / N) @ x0v[b])) < 1e-5
    with threadpool_limits(limits=1):
        mo, Lo = scan.solve_mv(None, ode_o, W, X0, 0.0, 1.0, N, oi.interrogate_kramer, pr, kalman_type="square-root")
    em, ev = _check(m, L, mo, Lo, 1e-6, 1e-6, pr[1])
    print(f"\nC5 square-root, N = {N}: max |mean - oracle| / scale = {em:.3g}, max |L L^T - oracle| / (sd sd) = {ev:.3g}; "
          f"|x - expm(A) x0| at t = 1: device {np.max(np.abs(m[0, N, 0, ::n_deriv] - expm(A) @ x0v[0])):.3g}, "
          f"oracle {np.max(np.abs(mo[0, N, 0, ::n_deriv] - expm(A) @ x0v[0])):.3g}")
