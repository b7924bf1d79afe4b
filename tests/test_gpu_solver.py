"""
GPU parity of the whole-solve boundary (rk_solve_filter / rk_solve_mv / rk_solve_sim through rodeo_amd.solve) against
the NumPy oracle on identical seeded inputs, plus the reference's own solver-level oracles evaluated directly on the
HIP path (K3 odeint, K4 analytic), plus size-independent properties at BASELINE.json's full sizes.

Tolerances (fp64).  Single-step maps agree to ~1e-13 relative; over a trajectory the IBM covariances are
ill-conditioned (cond(Sigma_pred) ~ 5e9 for FN at dt = 0.01, SURVEY.md hard part 3) so LU-solve rounding differences
are amplified; the stated per-trajectory bounds are:
    FitzHugh-Nagumo / higher-order:  |mean - oracle| <= 1e-9 absolute,  var: 1e-9 relative to max|var|
    Lorenz63 (chaotic, t <= 2):      |mean - oracle| <= 1e-6 absolute
and everything is far inside the reference's own 5e-8 rel_err criterion on the non-chaotic problems.
"""
import functools
import numpy as np
import pytest
from scipy.integrate import odeint
from oracle import scan, odes, priors, joint_gaussian as jg, interrogations as oi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rodeo_amd
    return rodeo_amd


def _itg(ra, name):
    return getattr(ra.interrogate, "interrogate_" + name), getattr(oi, "interrogate_" + name)


def fitz_problem(ra, N=200, t_max=10.0, sigma=.001, p=3, B=None, seed=0):
    theta = np.array([0.2, 0.2, 3.0])
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0v = np.array([-1., 1.])
    if B is not None:
        rng = np.random.default_rng(seed)
        theta = theta * np.exp(0.1 * rng.standard_normal((B, 3)))
        x0v = x0v + 0.1 * rng.standard_normal((B, 2))
    x0 = init(x0v, 0.0, theta=theta)
    prior = ra.ibm_init(t_max / N, p, np.array([sigma] * 2))
    return dict(W=W, x0=x0, theta=theta, prior=prior, t_max=t_max, N=N)


def _vclose(v, vo, rtol):
    assert np.max(np.abs(v - vo)) <= rtol * np.max(np.abs(vo))


@pytest.mark.parametrize("name", ["rodeo", "schober", "kramer"])
def test_fitz_mv_and_filter_parity(ra, name):
    g, o = _itg(ra, name)
    s = fitz_problem(ra)
    args = (s["W"], s["x0"], 0.0, s["t_max"], s["N"])
    m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    assert m.shape == (201, 2, 3) and v.shape == (201, 2, 3, 3)
    assert np.max(np.abs(m - mo)) < 1e-9
    _vclose(v, vo, 1e-9)
    assert jg.rel_err(mo, m) < 5e-8 and jg.rel_err(vo, v) < 5e-8          # the reference's own criterion
    np.testing.assert_array_equal(m[0], s["x0"]); assert np.all(v[0] == 0)
    from rodeo_amd.solve import _solve_filter
    f = _solve_filter(None, ra.ode.fitzhugh_nagumo, *args, g, *s["prior"], theta=s["theta"])
    fo = scan.solve_filter(None, odes.fitzhugh_nagumo, *args, o, *s["prior"], theta=s["theta"])
    for k in ("state_pred", "state_filt"):
        assert np.max(np.abs(f[k][0] - fo[k][0])) < 1e-9
        _vclose(f[k][1], fo[k][1], 1e-9)
    np.testing.assert_allclose(m[-1], f["state_filt"][0][-1], rtol=1e-9, atol=1e-9)   # solve.py:295-301 (two different kernels)


def test_positional_and_keyword_calls(ra):
    """solve_mv is called positionally (docs/examples/lorenz.md:143-146) and by keyword (README.md:139-152)."""
    s = fitz_problem(ra, N=50, t_max=2.5)
    a = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0.0, 2.5, 50, ra.interrogate.interrogate_kramer,
                    s["prior"], theta=s["theta"])
    b = ra.solve_mv(key=None, ode_fun=ra.ode.fitzhugh_nagumo, ode_weight=s["W"], ode_init=s["x0"], t_min=0.0, t_max=2.5,
                    n_steps=50, interrogate=ra.interrogate.interrogate_kramer, prior_pars=s["prior"],
                    kalman_type="standard", theta=s["theta"])
    np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])


def test_errors(ra):
    s = fitz_problem(ra, N=10, t_max=1.0)
    args = (s["W"], s["x0"], 0.0, 1.0, 10)
    with pytest.raises(NotImplementedError):                                  # solve.py:142-143
        ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, s["prior"],
                    kalman_type="nope", theta=s["theta"])
    with pytest.raises(TypeError):                                            # neither device code nor a function
        ra.solve_mv(None, "fitz", *args, ra.interrogate.interrogate_kramer, s["prior"], theta=s["theta"])
    with pytest.raises(ValueError):                                           # a Python function is traced: wrong output shape
        ra.solve_mv(None, lambda X, t, **p: X, *args, ra.interrogate.interrogate_kramer, s["prior"], theta=s["theta"])
    with pytest.raises(TypeError):
        ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, lambda **k: None, s["prior"], theta=s["theta"])
    with pytest.raises(TypeError):
        ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, s["prior"])   # no theta
    with pytest.raises(ValueError):
        ra.solve_mv(None, ra.ode.lorenz63, *args, ra.interrogate.interrogate_kramer, s["prior"], theta=s["theta"])


@pytest.mark.parametrize("B", [1, 7, 64, 100])
def test_batched_ragged_sizes(ra, B):
    """Batch sizes that are not multiples of the 64-lane wavefront, batched theta + x0, shared prior."""
    s = fitz_problem(ra, N=60, t_max=3.0, sigma=.1, B=B, seed=B)
    args = (s["W"], s["x0"], 0.0, 3.0, 60)
    m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, s["prior"], theta=s["theta"])
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, oi.interrogate_kramer, s["prior"], theta=s["theta"])
    assert m.shape == (B, 61, 2, 3)
    assert np.max(np.abs(m - mo)) < 1e-9
    _vclose(v, vo, 1e-9)


def test_batched_prior_sigma(ra):
    """Per-trajectory sigma -> batched prior_var (what docs/examples/parameter.md:230-235 does per draw)."""
    B = 5
    s = fitz_problem(ra, N=40, t_max=2.0, B=B)
    sig = 0.05 + 0.1 * np.random.default_rng(1).random((B, 2))
    Q, R = ra.ibm_init(2.0 / 40, 3, sig)
    assert R.shape == (B, 2, 3, 3)
    args = (s["W"], s["x0"], 0.0, 2.0, 40)
    m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_rodeo, (Q, R), theta=s["theta"])
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, oi.interrogate_rodeo, (Q, R), theta=s["theta"])
    assert np.max(np.abs(m - mo)) < 1e-9
    _vclose(v, vo, 1e-9)


@pytest.mark.parametrize("p", [2, 3, 4, 5])
def test_n_deriv_range(ra, p):
    s = fitz_problem(ra, N=50, t_max=2.5, sigma=.1, p=p)
    args = (s["W"], s["x0"], 0.0, 2.5, 50)
    m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, s["prior"], theta=s["theta"])
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, oi.interrogate_kramer, s["prior"], theta=s["theta"])
    assert np.max(np.abs(m - mo)) < 1e-8 * max(1.0, np.max(np.abs(mo)))
    _vclose(v, vo, 1e-7)


def test_lorenz_parity_short_horizon(ra):
    """C3's problem (docs/examples/lorenz.md:56-105) on a short horizon where chaos has not amplified rounding."""
    theta = np.array([28., 10., 8. / 3.])
    W, init = ra.utils.first_order_pad(ra.ode.lorenz63, 3, 4)
    x0 = init(np.array([-12., -5., 38.]), 0.0, theta=theta)
    N, t_max = 2000, 2.0
    prior = ra.ibm_init(t_max / N, 4, np.array([5e7] * 3))
    m, v = ra.solve_mv(None, ra.ode.lorenz63, W, x0, 0.0, t_max, N, ra.interrogate.interrogate_kramer, prior, theta=theta)
    mo, vo = scan.solve_mv(None, odes.lorenz63, W, x0, 0.0, t_max, N, oi.interrogate_kramer, prior, theta=theta)
    assert np.max(np.abs(m[:, :, 0] - mo[:, :, 0])) < 1e-6
    assert np.all(np.isfinite(v))

    def lor(X, t):
        x, y, z = X
        return [-10 * x + 10 * y, 28 * x - y - x * z, -8. / 3. * z + x * y]
    exact = odeint(lor, [-12., -5., 38.], np.linspace(0, t_max, N + 1), rtol=1e-11, atol=1e-11)
    assert np.max(np.abs(m[:, :, 0] - exact)) < 0.1            # solver truncation error at dt = 1e-3 (0.045), not parity


def test_higher_order_analytic(ra):
    """K4 directly on the HIP path: O(h^2) convergence to the analytic solution, and parity with the oracle."""
    W = np.array([[[0., 0., 1., 0.]]]); x0 = np.array([[-1., 0., 1., 0.]])
    errs = []
    for N in (50, 100, 200, 400):
        prior = ra.ibm_init(10.0 / N, 4, np.array([.001]))
        m, v = ra.solve_mv(None, ra.ode.higher_order, W, x0, 0.0, 10.0, N, ra.interrogate.interrogate_kramer, prior)
        errs.append(np.max(np.abs(m[:, 0, 0] - odes.higher_order_exact(np.linspace(0, 10, N + 1)))))
        if N == 100:
            mo, vo = scan.solve_mv(None, odes.higher_order, W, x0, 0.0, 10.0, N, oi.interrogate_kramer, prior)
            assert np.max(np.abs(m - mo)) < 1e-9
    np.testing.assert_allclose(errs, [6.74e-3, 1.67e-3, 4.17e-4, 1.04e-4], rtol=0.02)


def test_fitz_vs_odeint_on_gpu(ra):
    """K3 (tests/test_fitz.py:17-29) evaluated on the HIP path."""
    s = fitz_problem(ra)
    exact = odeint(lambda X, t: [3 * (X[0] - X[0] ** 3 / 3 + X[1]), -(X[0] - .2 + .2 * X[1]) / 3], [-1., 1.],
                   np.linspace(0, 10, 201), rtol=1e-10, atol=1e-10)
    args = (s["W"], s["x0"], 0.0, 10.0, 200)
    m, _ = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_rodeo, s["prior"], theta=s["theta"])
    assert jg.rel_err(m[:, :, 0], exact) <= 5.0
    x = ra.solve_sim(0, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_rodeo, s["prior"], theta=s["theta"])
    assert x.shape == (201, 2, 3) and jg.rel_err(x[:, :, 0], exact) <= 5.0


@pytest.mark.parametrize("name", ["rodeo", "kramer"])
def test_solve_sim_parity(ra, name):
    """Same Philox stream on both sides -> sample paths agree to rounding (draw parity with JAX: unpinned)."""
    g, o = _itg(ra, name)
    B = 9
    s = fitz_problem(ra, N=80, t_max=4.0, sigma=.1, B=B)
    args = (s["W"], s["x0"], 0.0, 4.0, 80)
    x = ra.solve_sim(42, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    xo = scan.solve_sim(42, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    assert x.shape == (B, 81, 2, 3)
    np.testing.assert_array_equal(x[:, 0], s["x0"])                         # solve.py:196-204
    assert np.max(np.abs(x - xo)) < 1e-7
    x2 = ra.solve_sim(43, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    assert np.max(np.abs(x - x2)) > 1e-6                                    # another seed, another path


def test_chkrebtii_parity_and_sharding(ra):
    """C4's combination: solve_sim + interrogate_chkrebtii (docs/examples/parameter.md:331-351)."""
    g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard")
    o = functools.partial(oi.interrogate_chkrebtii, kalman_type="standard")
    B = 12
    s = fitz_problem(ra, N=100, t_max=5.0, sigma=.1, B=B)
    args = (s["W"], s["x0"], 0.0, 5.0, 100)
    x = ra.solve_sim(7, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    xo = scan.solve_sim(7, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    assert np.max(np.abs(x - xo)) < 1e-7
    # sharding invariance: trajectories 4..7 solved alone with traj_offset = 4 give the same draws
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, s["W"], s["x0"][4:8], 0.0, 5.0, 100, g, s["prior"], traj_offset=4,
                        theta=s["theta"][4:8])
    plan.sim(7)
    np.testing.assert_array_equal(plan.x_host(), x[4:8])
    with pytest.raises(TypeError):                                          # kalman_type must be bound
        ra.solve_sim(7, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_chkrebtii, s["prior"], theta=s["theta"])


def test_solve_sim_law(ra):
    """Sample mean / variance over many draws match solve_mv's posterior (distributional check of the draws)."""
    B = 4096
    s = fitz_problem(ra, N=50, t_max=2.5, sigma=.5)
    x0 = np.broadcast_to(s["x0"], (B, 2, 3)).copy()
    args = (s["W"], x0, 0.0, 2.5, 50)
    xs = ra.solve_sim(11, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_rodeo, s["prior"], theta=s["theta"])
    m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0.0, 2.5, 50, ra.interrogate.interrogate_rodeo,
                       s["prior"], theta=s["theta"])
    var = np.einsum("nbii->nbi", v)
    z = (xs.mean(0) - m)[1:] / np.sqrt(var[1:] / B)
    assert np.max(np.abs(z)) < 5.0
    ratio = xs.var(0)[1:] / var[1:]
    assert 0.85 < ratio.min() and ratio.max() < 1.15


def test_headline_config_full_size_properties(ra):
    """
    BASELINE.json config 2 at full size (FN, p=3, N=4000, B=1024, kramer, solve_mv): size-independent properties --
    trajectory 0 (the unperturbed README problem) matches the oracle and odeint; every trajectory's b-th result equals
    the same trajectory solved alone (no cross-lane leakage); variances symmetric-PSD and finite; end conditions hold.
    """
    B, N = 1024, 4000
    rng = np.random.default_rng(20240)
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.1 * rng.standard_normal((B, 3)))
    x0v = np.array([-1., 1.]) + 0.1 * rng.standard_normal((B, 2))
    theta[0] = [0.2, 0.2, 3.0]; x0v[0] = [-1., 1.]
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    x0 = init(x0v, 0.0, theta=theta)
    prior = ra.ibm_init(0.01, 3, np.array([.1, .1]))
    m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, W, x0, 0.0, 40.0, N, ra.interrogate.interrogate_kramer, prior, theta=theta)
    assert m.shape == (B, N + 1, 2, 3) and np.all(np.isfinite(m)) and np.all(np.isfinite(v))
    np.testing.assert_array_equal(m[:, 0], x0); assert np.all(v[:, 0] == 0)
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, W, x0[0], 0.0, 40.0, N, oi.interrogate_kramer, prior, theta=theta[0])
    assert np.max(np.abs(m[0] - mo)) < 1e-9
    _vclose(v[0], vo, 1e-8)
    exact = odeint(lambda X, t: [3 * (X[0] - X[0] ** 3 / 3 + X[1]), -(X[0] - .2 + .2 * X[1]) / 3], [-1., 1.],
                   np.linspace(0, 40, N + 1), rtol=1e-10, atol=1e-10)
    assert np.max(np.abs(m[0, :, :, 0] - exact)) < 1e-4
    for b in (1, 63, 64, 777, 1023):
        m1, v1 = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, W, x0[b], 0.0, 40.0, N, ra.interrogate.interrogate_kramer,
                             prior, theta=theta[b])
        np.testing.assert_array_equal(m[b], m1); np.testing.assert_array_equal(v[b], v1)
    asym = np.max(np.abs(v - np.swapaxes(v, -1, -2)))
    assert asym < 1e-12 * np.max(np.abs(v))
    assert np.einsum("bnkii->bnki", v).min() >= -1e-18


@pytest.mark.parametrize("name", ["rodeo", "schober", "kramer"])
@pytest.mark.parametrize("B", [1, 2, 3, 37])
def test_tile_path_equals_batch_minor_path_and_oracle(ra, name, B):
    """
    The MFMA-tile kernels (p = 3) against the lane-per-trajectory kernels and the oracle, including tile counts that
    do not fill a wave (B*d not a multiple of 4) and N-1 not a multiple of the 16-step hand-off chunk.
    """
    from rodeo_amd import _lib
    g, o = _itg(ra, name)
    N = 77
    s = fitz_problem(ra, N=N, t_max=3.85, sigma=.1, B=B, seed=100 + B)
    args = (s["W"], s["x0"], 0.0, 3.85, N)
    p_tile = ra.SolvePlan(ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    p_tile.mv(None)
    assert p_tile.layout == _lib.LAYOUT_TILE3
    m, v = p_tile.state_host()
    p_bm = ra.SolvePlan(ra.ode.fitzhugh_nagumo, *args, g, s["prior"], batch_minor=True, theta=s["theta"])
    p_bm.mv(None)
    assert p_bm.layout == _lib.LAYOUT_BATCH_MINOR
    m2, v2 = p_bm.state_host()
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    assert m.shape == mo.shape and v.shape == vo.shape
    for mm, vv in ((m, v), (m2, v2)):
        assert np.max(np.abs(mm - mo)) < 1e-9
        _vclose(vv, vo, 1e-9)
    # filter-only through the tile path
    p_tile.filter(None)
    mf, vf = p_tile.state_host()
    fo = scan.solve_filter(None, odes.fitzhugh_nagumo, *args, o, *s["prior"], theta=s["theta"])
    assert np.max(np.abs(mf - fo["state_filt"][0])) < 1e-9
    _vclose(vf, fo["state_filt"][1], 1e-9)


@pytest.mark.parametrize("N", [1, 2, 3, 15, 16, 17, 18, 33, 49])
def test_tile_path_short_horizons(ra, N):
    """
    Step counts around the backward kernels' 16-step hand-off chunk and three-stage pipeline (no chunk, one partial
    chunk, exactly one, one + 1, ...), solve_mv and solve_sim through the p = 3 tile kernels, B*d not a multiple of 4.
    """
    from rodeo_amd import _lib
    B = 5
    s = fitz_problem(ra, N=N, t_max=0.05 * N, sigma=.1, B=B, seed=300 + N)
    args = (s["W"], s["x0"], 0.0, 0.05 * N, N)
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, s["prior"], theta=s["theta"])
    plan.mv(None)
    assert plan.layout == _lib.LAYOUT_TILE3
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, oi.interrogate_kramer, s["prior"], theta=s["theta"])
    assert m.shape == mo.shape == (B, N + 1, 2, 3)
    assert np.max(np.abs(m - mo)) < 1e-10
    _vclose(v, vo, 1e-9)
    x = ra.solve_sim(7, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_rodeo, s["prior"], theta=s["theta"])
    xo = scan.solve_sim(7, odes.fitzhugh_nagumo, *args, oi.interrogate_rodeo, s["prior"], theta=s["theta"])
    assert x.shape == xo.shape == (B, N + 1, 2, 3)
    assert np.max(np.abs(x - xo)) < 1e-8


def test_tile_path_huge_batch_64bit_offsets(ra):
    """
    A batch so large that one 16-step chunk of time rows spans more than 2 GiB (the backward consumer's buffer-store
    window no longer fits and it takes its pointer path; every row offset needs 64 bits): 760,000 trajectories x 17
    steps, checked on a few hundred trajectories at three time slices downloaded individually.
    """
    B, N, t_max = 760000, 17, 0.17
    rng = np.random.default_rng(1)
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.1 * rng.standard_normal((B, 3)))
    x0v = np.array([-1., 1.]) + 0.1 * rng.standard_normal((B, 2))
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    x0 = init(x0v, 0.0, theta=theta)
    prior = ra.ibm_init(t_max / N, 3, np.array([.1, .1]))
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, t_max, N, ra.interrogate.interrogate_kramer, prior, theta=theta)
    assert 15 * 2 * B * 96 + 384 > 2 ** 31
    plan.mv(None)
    idx = np.concatenate([np.arange(0, 40), np.arange(B - 40, B), rng.integers(0, B, 120)])
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, W, x0[idx], 0.0, t_max, N, oi.interrogate_kramer, prior, theta=theta[idx])
    for n in (1, 9, N):
        t = plan.var_state.slice0_host(n)                      # (B, d, 3, 4) tiles of time n
        assert np.max(np.abs(t[idx][..., 3] - mo[:, n])) < 1e-10
        assert np.max(np.abs(t[idx][..., :3] - vo[:, n])) <= 1e-9 * np.max(np.abs(vo[:, n]))


def test_tile_path_higher_order_single_block(ra):
    """n_block = 1 through the tile path (4 trajectories per wave), p = 3: x'' = sin 2t - x with W = [0, 0, 1]."""
    from rodeo_amd import _lib
    W = np.array([[[0., 0., 1.]]])
    B = 6
    x0 = np.tile(np.array([[-1., 0., 1.]]), (B, 1, 1)) + 0.01 * np.random.default_rng(0).standard_normal((B, 1, 3))
    prior = ra.ibm_init(0.05, 3, np.array([.01]))
    plan = ra.SolvePlan(ra.ode.higher_order, W, x0, 0.0, 5.0, 100, ra.interrogate.interrogate_kramer, prior)
    plan.mv(None)
    assert plan.layout == _lib.LAYOUT_TILE3
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(None, odes.higher_order, W, x0, 0.0, 5.0, 100, oi.interrogate_kramer, prior)
    assert np.max(np.abs(m - mo)) < 1e-9
    _vclose(v, vo, 1e-9)


def test_rccl_single_rank_comm(ra):
    """RCCL path of the C ABI with one rank: communicator init, all-gather and barrier on the device stream."""
    import ctypes as C
    from rodeo_amd import _lib
    from rodeo_amd.shard import RcclComm
    dev = ra.Device(0)
    comm = RcclComm(dev, 0, 1, bcast=lambda b, src=0: b)
    send = dev.to_device(np.arange(8, dtype=np.float64))
    recv = dev.zeros((8,))
    comm.allgather(send, recv, 8)
    comm.barrier()
    np.testing.assert_array_equal(recv.to_host(), np.arange(8.0))
    comm.close()
    dev.close()


@pytest.mark.parametrize("prob,B,N", [("lorenz", 1, 50), ("lorenz", 5, 83), ("fitz4", 3, 40), ("fitz4", 9, 97),
                                      ("higher", 6, 66), ("higher", 1, 17)])
@pytest.mark.parametrize("name", ["kramer", "rodeo"])
def test_tile4_path_equals_batch_minor_and_oracle(ra, prob, B, N, name):
    """p = 4 MFMA-tile kernels (n_block = 3, 2, 1; ragged tile counts and chunk tails) vs lane-per-trajectory vs oracle."""
    from rodeo_amd import _lib
    g, o = _itg(ra, name)
    rng = np.random.default_rng(B * 1000 + N)
    if prob == "lorenz":
        theta = np.array([28., 10., 8. / 3.]) * np.exp(0.01 * rng.standard_normal((B, 3)))
        W, init = ra.utils.first_order_pad(ra.ode.lorenz63, 3, 4)
        x0 = init(np.array([-12., -5., 38.]) + 0.1 * rng.standard_normal((B, 3)), 0.0, theta=theta)
        t_max = N * 1e-3
        prior = ra.ibm_init(1e-3, 4, np.array([5e7] * 3))
        dode, oode, kw = ra.ode.lorenz63, odes.lorenz63, dict(theta=theta)
    elif prob == "fitz4":
        s = fitz_problem(ra, N=N, t_max=N * 0.05, sigma=.1, p=4, B=B, seed=N)
        W, x0, prior, t_max = s["W"], s["x0"], s["prior"], s["t_max"]
        dode, oode, kw = ra.ode.fitzhugh_nagumo, odes.fitzhugh_nagumo, dict(theta=s["theta"])
    else:
        W = np.array([[[0., 0., 1., 0.]]])
        x0 = np.tile(np.array([[-1., 0., 1., 0.]]), (B, 1, 1)) + 0.01 * rng.standard_normal((B, 1, 4))
        t_max = N * 0.05
        prior = ra.ibm_init(0.05, 4, np.array([.01]))
        dode, oode, kw = ra.ode.higher_order, odes.higher_order, {}
    args = (W, x0, 0.0, t_max, N)
    pt = ra.SolvePlan(dode, *args, g, prior, **kw)
    pt.mv(None)
    assert pt.layout == _lib.LAYOUT_TILE4
    m, v = pt.state_host()
    pb = ra.SolvePlan(dode, *args, g, prior, batch_minor=True, **kw)
    pb.mv(None)
    m2, v2 = pb.state_host()
    mo, vo = scan.solve_mv(None, oode, *args, o, prior, **kw)
    assert m.shape == mo.shape and v.shape == vo.shape
    sm = np.max(np.abs(mo), axis=(0, 1, 2))                           # per derivative order
    sd = np.sqrt(np.max(np.abs(np.einsum("bnkii->bnki", vo)), axis=(0, 1, 2)))   # per-component sd scale
    sv = sd[:, None] * sd[None, :]                                    # covariance scale (scale-invariant metric)
    for mm, vv in ((m, v), (m2, v2)):
        assert np.max(np.abs(mm - mo) / sm) < 1e-8
        assert np.max(np.abs(vv - vo) / sv) < 1e-7
    pt.filter(None)
    mf, vf = pt.state_host()
    fo = scan.solve_filter(None, oode, *args, o, *prior, **kw)
    assert np.max(np.abs(mf - fo["state_filt"][0]) / sm) < 1e-8


def _chol_prior(prior):
    return prior[0], np.linalg.cholesky(prior[1])


def _sq(L):
    return L @ np.swapaxes(L, -1, -2)


@pytest.mark.parametrize("name", ["kramer", "schober", "rodeo", "chkrebtii"])
def test_square_root_solver_parity(ra, name):
    """kalman_type='square-root' (docs/examples/higher_order.md:106-142): factors compared as L L^T against the oracle's
    square-root scan, and against the standard filter's variances (same posterior in exact arithmetic)."""
    B, N = 5, 60
    s = fitz_problem(ra, N=N, t_max=3.0, sigma=.1, B=B, seed=9)
    pr = _chol_prior(s["prior"])
    if name == "chkrebtii":
        g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="square-root")
        o = functools.partial(oi.interrogate_chkrebtii, kalman_type="square-root")
    else:
        g, o = _itg(ra, name)
    args = (s["W"], s["x0"], 0.0, 3.0, N)
    m, L = ra.solve_mv(5, ra.ode.fitzhugh_nagumo, *args, g, pr, kalman_type="square-root", theta=s["theta"])
    mo, Lo = scan.solve_mv(5, odes.fitzhugh_nagumo, *args, o, pr, kalman_type="square-root", theta=s["theta"])
    assert m.shape == (B, N + 1, 2, 3) and L.shape == (B, N + 1, 2, 3, 3)
    assert np.max(np.abs(m - mo)) < 1e-8
    _vclose(_sq(L), _sq(Lo), 1e-7)
    if name in ("kramer", "schober"):                   # var_meas = 0: identical posterior to the standard filter
        m2, v2 = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
        assert np.max(np.abs(m - m2)) < 1e-7
        _vclose(_sq(L), v2, 1e-6)
    # _solve_filter's four outputs in square-root form: predicted AND filtered means / factors (solve.py:114-121)
    from rodeo_amd.solve import _solve_filter
    f = _solve_filter(5, ra.ode.fitzhugh_nagumo, *args, g, *pr, kalman_funs=ra.kalmantv.square_root, theta=s["theta"])
    fo = scan.solve_filter(5, odes.fitzhugh_nagumo, *args, o, *pr, kalman_funs=scan.sqrt_ops, theta=s["theta"])
    for k in ("state_pred", "state_filt"):
        assert np.max(np.abs(f[k][0] - fo[k][0])) < 1e-8
        _vclose(_sq(f[k][1]), _sq(fo[k][1]), 1e-7)
    np.testing.assert_array_equal(f["state_pred"][0][:, 0], s["x0"]); assert np.all(f["state_pred"][1][:, 0] == 0)
    x = ra.solve_sim(5, ra.ode.fitzhugh_nagumo, *args, g, pr, kalman_type="square-root", theta=s["theta"])
    xo = scan.solve_sim(5, odes.fitzhugh_nagumo, *args, o, pr, kalman_type="square-root", theta=s["theta"])
    assert x.shape == (B, N + 1, 2, 3) and np.all(np.isfinite(x))
    np.testing.assert_array_equal(x[:, 0], s["x0"])
    if name in ("rodeo", "chkrebtii"):
        # var_meas > 0: full-rank factors are unique once their diagonal signs are fixed -> the same sample path
        assert np.max(np.abs(x - xo)) < 1e-6
    else:
        # exact measurement (var_meas = 0): the filtered factor is rank deficient, a lower factor is then not unique and
        # two QR implementations draw different (equally valid) paths; check the draw against the smoothed moments
        sd = np.sqrt(np.einsum("bnkii->bnki", _sq(L)))
        assert np.max(np.abs(x - m)[:, 1:] / (sd[:, 1:] + 1e-12)) < 8.0
        if name == "schober":
            # ... and a sharp property that every valid path shares: an exact measurement leaves no variance along W, in the
            # filtered state (W Sigma_f = 0) and hence in the smoothing gain (W G = W Sigma_f Q^T S-^{-1} = 0) and in every
            # conditional draw, so W x_n = W mu_f,n at every step -- the sampled path satisfies the filter's constraint
            mf = f["state_filt"][0]
            res = np.einsum("kp,bnkp->bnk", s["W"][:, 0, :], x - mf)[:, 1:]
            scale = np.abs(np.einsum("kp,bnkp->bnk", np.abs(s["W"][:, 0, :]), np.abs(mf)))[:, 1:] + 1.0
            assert np.max(np.abs(res) / scale) < 1e-9, np.max(np.abs(res) / scale)


@pytest.mark.parametrize("p", [3, 5, 8])
def test_square_root_backward_forms_agree(ra, p, monkeypatch):
    """solve_mv in square-root form on small blocks: the two-kernel backward pass (carry-independent part of every step
    time-parallel into a workspace, then the chain; rodeo_amd/csrc/solve_sqrt.hip) and the one-kernel pass that runs without a
    workspace do the same arithmetic in the same order: the same bits.  RK_SQRT_BWD is read per call."""
    rng = np.random.default_rng(11 + p)
    B, N = 5, 23
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.1 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0 = init(np.array([-1., 1.]) + 0.1 * rng.standard_normal((B, 2)), 0.0, theta=theta)
    Q, R = ra.ibm_init(1.0 / N, p, np.array([.1, .1]))
    if p >= 7:                                           # (the IBM variance matrix is singular in fp64 there)
        Lr = np.tril(0.1 * rng.standard_normal((2, p, p))) + np.eye(p) * 0.5
    else:
        Lr = np.linalg.cholesky(R)
    res = {}
    for form in ("two", "single"):
        monkeypatch.setenv("RK_SQRT_BWD", form)
        plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, 1.0, N, ra.interrogate.interrogate_kramer, (Q, Lr),
                            kalman_type="square-root", theta=theta)
        plan.dev.profile_enable(True)
        plan.mv(None)
        names = [k for k, _ in plan.dev.profile_last()]
        plan.dev.profile_enable(False)
        assert ("bwd_sqrt_chain_kernel" in names) == (form == "two"), names
        res[form] = plan.state_host()
        plan.dev.profile_enable(True)
        plan.sim(77)
        names = [k for k, _ in plan.dev.profile_last()]
        plan.dev.profile_enable(False)
        assert ("bwd_sqrt_sim_chain_kernel" in names) == (form == "two"), names
        res[form + "_x"] = plan.x_host()
    assert np.all(np.isfinite(res["two"][0])) and np.all(np.isfinite(res["two"][1]))
    np.testing.assert_array_equal(res["two"][0], res["single"][0])
    np.testing.assert_array_equal(res["two"][1], res["single"][1])
    # solve_sim: same normals, same factors; the draw's L z arrives summed in the two-kernel form (rounding-level difference)
    assert np.all(np.isfinite(res["two_x"]))
    np.testing.assert_allclose(res["two_x"], res["single_x"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("name", ["kramer", "chkrebtii", "rodeo"])
def test_square_root_three_blocks_ragged_waves(ra, name):
    """Square-root filter with THREE blocks per trajectory (Lorenz63): the forward kernel puts 64 // 3 = 21 trajectories into a
    wave (one lane idle) and exchanges the blocks' interrogation points between neighbouring lanes; 25 trajectories = one
    full wave and a ragged one.  solve_mv and solve_sim against the oracle."""
    B, N, t_max, p = 25, 40, 0.4, 3
    rng = np.random.default_rng(31)
    theta = np.array([28., 10., 8. / 3.]) * np.exp(0.02 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.lorenz63, 3, p)
    x0 = init(np.array([-12., -5., 38.]) + 0.1 * rng.standard_normal((B, 3)), 0.0, theta=theta)
    Q, R = ra.ibm_init(t_max / N, p, np.array([10.0] * 3))
    pr = (Q, np.linalg.cholesky(R))
    if name == "chkrebtii":
        g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="square-root")
        o = functools.partial(oi.interrogate_chkrebtii, kalman_type="square-root")
    else:
        g, o = _itg(ra, name)
    args = (W, x0, 0.0, t_max, N)
    m, L = ra.solve_mv(3, ra.ode.lorenz63, *args, g, pr, kalman_type="square-root", theta=theta)
    mo, Lo = scan.solve_mv(3, odes.lorenz63, *args, o, pr, kalman_type="square-root", theta=theta)
    assert m.shape == (B, N + 1, 3, p) and np.all(np.isfinite(L))
    scale = np.maximum(np.max(np.abs(mo), axis=(0, 1)), 1.0)
    assert np.max(np.abs(m - mo) / scale) < 1e-8
    _vclose(_sq(L), _sq(Lo), 1e-7)
    if name != "kramer":                                 # (exact measurement: no unique sample path, test_square_root_solver_parity)
        x = ra.solve_sim(3, ra.ode.lorenz63, *args, g, pr, kalman_type="square-root", theta=theta)
        xo = scan.solve_sim(3, odes.lorenz63, *args, o, pr, kalman_type="square-root", theta=theta)
        assert np.max(np.abs(x - xo) / scale) < 1e-6


def test_square_root_higher_order_example(ra):
    """The docs' square-root run (higher_order.md:109-125): n_deriv = 4 second-order ODE, vs the analytic solution."""
    W = np.array([[[0., 0., 1., 0.]]]); x0 = np.array([[-1., 0., 1., 0.]])
    N = 400
    Q, R = ra.ibm_init(10.0 / N, 4, np.array([.001]))
    m, L = ra.solve_mv(None, ra.ode.higher_order, W, x0, 0.0, 10.0, N, ra.interrogate.interrogate_kramer,
                       (Q, np.linalg.cholesky(R)), kalman_type="square-root")
    assert m.shape == (N + 1, 1, 4) and np.all(np.isfinite(L))
    assert abs(np.max(np.abs(m[:, 0, 0] - odes.higher_order_exact(np.linspace(0, 10, N + 1)))) - 1.04e-4) < 2e-5
    mo, Lo = scan.solve_mv(None, odes.higher_order, W, x0, 0.0, 10.0, N, oi.interrogate_kramer,
                           (Q, np.linalg.cholesky(R)), kalman_type="square-root")
    assert np.max(np.abs(m - mo)) < 1e-8


def test_config3_full_size_properties(ra):
    """
    BASELINE.json config 3 at full size (Lorenz63, p=4, N=20000, B=512, kramer, solve_mv; MFMA tile4 kernels):
    sampled trajectories against the plain-C oracle over the whole (chaotic) horizon, the same trajectory solved alone
    (no cross-tile leakage, bit-exact), end conditions, finite symmetric variances.
    """
    from oracle import c_port
    from rodeo_amd import _lib
    B, N, p = 512, 20000, 4
    rng = np.random.default_rng(20241)
    theta = np.array([28., 10., 8. / 3.])
    W, init = ra.utils.first_order_pad(ra.ode.lorenz63, 3, p)
    x0 = init(np.array([-12., -5., 38.]) + 1e-3 * rng.standard_normal((B, 3)), 0.0, theta=theta)
    prior = ra.ibm_init(20.0 / N, p, np.array([5e7] * 3))
    plan = ra.SolvePlan(ra.ode.lorenz63, W, x0, 0.0, 20.0, N, ra.interrogate.interrogate_kramer, prior, theta=theta)
    plan.mv(None)
    assert plan.layout == _lib.LAYOUT_TILE4
    m, v = plan.state_host()
    assert m.shape == (B, N + 1, 3, 4) and v.shape == (B, N + 1, 3, 4, 4)
    np.testing.assert_array_equal(m[:, 0], x0); assert np.all(v[:, 0] == 0)
    sample = [0, 1, 170, 511]
    mo, vo = c_port.solve_mv("lorenz63", "kramer", W, x0[sample], 0.0, 20.0, N, prior, theta=theta)
    sm = np.max(np.abs(mo), axis=(0, 1, 2))
    # chaotic: rounding differences grow like exp(0.9 t) -- measured 1.1e-10 of the scale up to t = 10, 6.5e-8 at
    # t = 15, 6.7e-6 at t = 20 (the covariances do not depend on the path: 8e-10 throughout)
    assert np.max(np.abs(m[sample][:, :N // 2] - mo[:, :N // 2]) / sm) < 1e-8
    assert np.max(np.abs(m[sample] - mo) / sm) < 1e-3
    sd = np.sqrt(np.max(np.abs(np.einsum("bnkii->bnki", vo)), axis=(0, 1, 2)))
    assert np.max(np.abs(v[sample] - vo) / (sd[:, None] * sd[None, :])) < 1e-7
    for b in (1, 511):
        m1, v1 = ra.solve_mv(None, ra.ode.lorenz63, W, x0[b], 0.0, 20.0, N, ra.interrogate.interrogate_kramer, prior, theta=theta)
        np.testing.assert_array_equal(m[b], m1); np.testing.assert_array_equal(v[b], v1)
    for lo in range(0, B, 64):                       # whole-batch checks in slabs (the full result is 4.9 GB)
        ms, vs = m[lo:lo + 64], v[lo:lo + 64]
        assert np.all(np.isfinite(ms)) and np.all(np.isfinite(vs))
        assert np.max(np.abs(vs - np.swapaxes(vs, -1, -2))) <= 1e-12 * np.max(np.abs(vs))
        assert np.max(np.abs(ms[:, :, :, 0])) < 100.0                     # on the attractor


def test_integration_md_stub_runs(ra):
    """The ctypes stub printed in INTEGRATION.md (section 2), executed as it stands (only the library path is filled in),
    gives the numbers of rodeo_amd.solve_mv: the documented binding is a working one."""
    import os, re
    from rodeo_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C, numpy as np.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("librodeo_kalman.so")', f'C.CDLL({os.path.join(root, "rodeo_amd", "librodeo_kalman.so")!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    B = 5
    s = fitz_problem(ra, N=40, t_max=2.0, sigma=.1, B=B)
    m, v = ns["solve_mv"](None, _lib.RHS_FITZHUGH_NAGUMO, s["W"], s["x0"], 0.0, 2.0, 40, _lib.INTERROGATE_KRAMER,
                          s["prior"], s["theta"])
    m2, v2 = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0.0, 2.0, 40, ra.interrogate.interrogate_kramer,
                         s["prior"], theta=s["theta"])
    assert m.shape == m2.shape and np.max(np.abs(m - m2)) < 1e-9 and np.max(np.abs(v - v2)) < 1e-9 * np.max(np.abs(v2))


@pytest.mark.parametrize("name", ["kramer", "rodeo", "schober"])
def test_two_state_blocks_run_on_the_three_state_tiles(ra, name):
    """n_deriv = 2 through solve_mv / solve_sim: padded with a decoupled third component and run on the MFMA tiles --
    against the oracle at n_deriv = 2 and against the lane-per-trajectory n_bstate = 2 kernels (SolvePlan)."""
    g, o = _itg(ra, name)
    B = 6
    s = fitz_problem(ra, N=120, t_max=3.0, sigma=.5, p=2, B=B)
    args = (s["W"], s["x0"], 0.0, 3.0, 120)
    m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    assert m.shape == (B, 121, 2, 2) and v.shape == (B, 121, 2, 2, 2)
    assert np.max(np.abs(m - mo)) < 1e-9 * max(1.0, np.max(np.abs(mo)))
    _vclose(v, vo, 1e-8)
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    plan.mv(None)
    ml, vl = plan.state_host()
    assert np.max(np.abs(m - ml)) < 1e-10 * max(1.0, np.max(np.abs(ml)))
    x = ra.solve_sim(5, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    xo = scan.solve_sim(5, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    assert x.shape == (B, 121, 2, 2) and np.max(np.abs(x - xo)) < 1e-7


@pytest.mark.parametrize("p", [3, 5, 6, 7, 8])
def test_square_root_solver_traced_python_rhs_and_higher_n_deriv(ra, p):
    """kalman_type='square-root' with an ordinary Python ode_fun (traced, compiled by hiprtc: the forward square-root
    kernel is a template over the right-hand side like the standard one) and at n_deriv = 5, 6 -- where the square-root
    form is what a user reaches for (src/rodeo/kalmantv/square_root.py; the covariance form loses definiteness first), up to
    n_deriv = 8 (the blocked-tile covariance kernels' range; the square-root lane kernels spill there: a functional path)."""
    def fitz(X, t, theta):
        a, b, c = theta
        V, R = X[0, 0], X[1, 0]
        return np.array([[c * (V - V * V * V / 3 + R)], [-1 / c * (V - a + b * R)]])
    B, N, t_max = 4, 50, 1.0
    s = fitz_problem(ra, N=N, t_max=t_max, sigma=.1, B=B, seed=p, p=p)
    if p <= 6:
        pr = _chol_prior(s["prior"])
    else:
        # n_deriv = 7, 8: the IBM variance matrix is singular in fp64 whatever the step (cond 2e21 / 9e24: a Hilbert-type
        # matrix), so its numerical Cholesky factor -- the INPUT of this mode -- is meaningless.  The kernels' sizes are
        # exercised with the IBM weight matrix and a well-conditioned lower factor in its place.
        rng = np.random.default_rng(p)
        pr = (s["prior"][0], 0.05 * (np.eye(p) + 0.2 * np.tril(rng.standard_normal((2, p, p)), -1)))
    args = (s["W"], s["x0"], 0.0, t_max, N)
    for name in ("kramer", "rodeo"):
        g, o = _itg(ra, name)
        mo, Lo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, o, pr, kalman_type="square-root", theta=s["theta"])
        scale = np.maximum(np.max(np.abs(mo), axis=(0, 1, 2)), 1.0)
        for fun in (fitz, ra.ode.fitzhugh_nagumo):
            m, L = ra.solve_mv(None, fun, *args, g, pr, kalman_type="square-root", theta=s["theta"])
            assert m.shape == (B, N + 1, 2, p) and L.shape == (B, N + 1, 2, p, p)
            assert np.max(np.abs(m - mo) / scale) < (1e-8 if p <= 6 else 1e-7)
            _vclose(_sq(L), _sq(Lo), 1e-6 if p <= 6 else 1e-5)
    x = ra.solve_sim(3, fitz, *args, ra.interrogate.interrogate_rodeo, pr, kalman_type="square-root", theta=s["theta"])
    xo = scan.solve_sim(3, odes.fitzhugh_nagumo, *args, oi.interrogate_rodeo, pr, kalman_type="square-root", theta=s["theta"])
    assert np.max(np.abs(x - xo) / np.maximum(np.max(np.abs(xo), axis=(0, 1, 2)), 1.0)) < 1e-6


def test_tile4_backward_chunk_boundaries_and_ragged_waves(ra):
    """bwd_mv_tile4_kernel around every special case of its chunked hand-off (chunks of 12 steps with four tiles per wave,
    16 with three; the last two chunks flushed by the consumer itself, the others by the stage-3 producer wave; a ragged
    last chunk stored element by element; row images with pieces masked for tiles past the end): tile path against the
    lane-per-trajectory kernels on the same inputs, many (n_block, batch, steps)."""
    from rodeo_amd import _lib
    g = ra.interrogate.interrogate_rodeo
    worst = 0.0
    for prob, Bs, Ns in (("fitz4", (1, 2, 3, 5, 7), (2, 3, 12, 13, 14, 24, 25, 26, 36, 37, 38, 49, 61)),
                         ("higher", (1, 3, 5, 9), (2, 13, 25, 37, 50)),
                         ("lorenz", (1, 2, 5), (2, 3, 16, 17, 18, 32, 33, 34, 48, 49, 50, 65, 81))):
        for B in Bs:
            for N in Ns:
                rng = np.random.default_rng(B * 1000 + N)
                if prob == "lorenz":
                    theta = np.array([28., 10., 8. / 3.]) * np.exp(0.01 * rng.standard_normal((B, 3)))
                    W, init = ra.utils.first_order_pad(ra.ode.lorenz63, 3, 4)
                    x0 = init(np.array([-12., -5., 38.]) + 0.1 * rng.standard_normal((B, 3)), 0.0, theta=theta)
                    t_max, prior, dode, kw = N * 1e-3, ra.ibm_init(1e-3, 4, np.array([5e7] * 3)), ra.ode.lorenz63, dict(theta=theta)
                elif prob == "fitz4":
                    s = fitz_problem(ra, N=N, t_max=N * 0.05, sigma=.1, p=4, B=B, seed=N)
                    W, x0, prior, t_max, dode, kw = s["W"], s["x0"], s["prior"], s["t_max"], ra.ode.fitzhugh_nagumo, dict(theta=s["theta"])
                else:
                    W = np.array([[[0., 0., 1., 0.]]])
                    x0 = np.tile(np.array([[-1., 0., 1., 0.]]), (B, 1, 1)) + 0.01 * rng.standard_normal((B, 1, 4))
                    t_max, prior, dode, kw = N * 0.05, ra.ibm_init(0.05, 4, np.array([.01])), ra.ode.higher_order, {}
                args = (W, x0, 0.0, t_max, N)
                pt = ra.SolvePlan(dode, *args, g, prior, **kw)
                pt.mv(None)
                assert pt.layout == _lib.LAYOUT_TILE4
                m, v = pt.state_host()
                pb = ra.SolvePlan(dode, *args, g, prior, batch_minor=True, **kw)
                pb.mv(None)
                m2, v2 = pb.state_host()
                sm = np.maximum(np.max(np.abs(m2), axis=(0, 1, 2)), 1e-12)
                sd = np.sqrt(np.max(np.abs(np.einsum("bnkii->bnki", v2)), axis=(0, 1, 2))) + 1e-300
                em, ev = np.max(np.abs(m - m2) / sm), np.max(np.abs(v - v2) / (sd[:, None] * sd[None, :]))
                assert np.all(np.isfinite(m)) and np.all(np.isfinite(v)) and em < 1e-8 and ev < 1e-7, (prob, B, N, em, ev)
                worst = max(worst, em, ev)
                del pt, pb
    assert worst < 1e-7


def test_square_root_sim_law_is_L_Lt(ra):
    """The law of solve_sim(kalman_type='square-root') (MIGRATION.md): draws are x = mean + L z, i.e. N(mean, L L^T).
    The backward sampler's marginal at time n is then N(mu^s_n, L^s_n L^s_n^T) with the smoothed moments of solve_mv in the
    same mode: checked by the first two empirical moments of 4096 independent draws of ONE problem (interrogate_rodeo: a
    deterministic filter), entry by entry against 6 standard errors.  (The reference hands L to multivariate_normal as if it
    were the covariance, src/rodeo/solve.py:179,182-186 -- that law, N(mean, L), is rejected by the same numbers.)"""
    B, N = 4096, 30
    s = fitz_problem(ra, N=N, t_max=1.5, sigma=.1, B=None, seed=3)
    pr = _chol_prior(s["prior"])
    g, _ = _itg(ra, "rodeo")
    x0 = np.broadcast_to(s["x0"], (B,) + s["x0"].shape).copy()
    args = (s["W"], x0, 0.0, 1.5, N)
    x = ra.solve_sim(11, ra.ode.fitzhugh_nagumo, *args, g, pr, kalman_type="square-root", theta=s["theta"])
    m, L = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0.0, 1.5, N, g, pr, kalman_type="square-root",
                       theta=s["theta"])
    worst = 0.0
    for n in (N // 3, N // 2, N - 1):
        for blk in range(2):
            C = L[n, blk] @ L[n, blk].T
            xs = x[:, n, blk, :]
            d = xs - m[n, blk]
            se_mean = np.sqrt(np.diag(C) / B)
            assert np.all(np.abs(d.mean(axis=0)) < 6 * se_mean + 1e-300)
            Chat = d.T @ d / B
            se_cov = np.sqrt((np.outer(np.diag(C), np.diag(C)) + C ** 2) / B)
            z = np.abs(Chat - C) / (se_cov + 1e-300)
            worst = max(worst, z.max())
            assert z.max() < 6.0
            # ... and the factor itself in the covariance slot is NOT what is sampled
            assert np.max(np.abs(Chat - L[n, blk]) / (se_cov + 1e-300)) > 20.0
    assert worst > 0.0
