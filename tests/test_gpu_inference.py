"""
GPU parity of the callers of the hot path: ``inference.basic`` (src/rodeo/inference/basic.py:47-62) and the
pseudo-marginal log-posterior of docs/examples/parameter.md:331-354 (BASELINE config 4's per-draw quantity:
constrain -> first_order_pad -> ibm_init(sigma per draw) -> solve_sim + interrogate_chkrebtii -> gather at the
observation times -> Gaussian log-likelihood + N(0, 10^2) log-prior), against the same computation on the oracle.
"""
import functools
import numpy as np
import pytest
from scipy.stats import norm
from oracle import scan, odes, priors, interrogations as oi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rodeo_amd
    return rodeo_amd


def _setup(ra, n_steps=80, t_max=4.0, n_obs=5):
    theta = np.array([0.2, 0.2, 3.0])
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    x0 = init(np.array([-1., 1.]), 0.0, theta=theta)
    prior = ra.ibm_init(t_max / n_steps, 3, np.array([.1, .1]))
    obs_times = np.linspace(0, t_max, n_obs)
    rng = np.random.default_rng(3)
    m, _ = scan.solve_mv(None, odes.fitzhugh_nagumo, W, x0, 0., t_max, n_steps, oi.interrogate_kramer, prior, theta=theta)
    ind = np.searchsorted(np.linspace(0, t_max, n_steps + 1), obs_times)
    Y = m[ind, :, 0] + np.sqrt(0.005) * rng.standard_normal((n_obs, 2))
    return dict(theta=theta, W=W, x0=x0, prior=prior, obs_times=obs_times, Y=Y, ind=ind, N=n_steps, t_max=t_max)


def test_basic_gaussian_and_callable(ra):
    from rodeo_amd.inference.basic import GaussianObsLoglik
    s = _setup(ra)
    sd = np.sqrt(0.005)
    args = (None, ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0., s["t_max"], s["N"], ra.interrogate.interrogate_kramer,
            s["prior"], s["Y"], s["obs_times"])
    ll_dev, Xt = ra.inference.basic(*args, GaussianObsLoglik(sd), theta=s["theta"])     # (loglik, Xt): basic.py:62
    mo, _ = scan.solve_mv(None, odes.fitzhugh_nagumo, s["W"], s["x0"], 0., s["t_max"], s["N"], oi.interrogate_kramer,
                          s["prior"], theta=s["theta"])
    ll_ref = np.sum(norm.logpdf(s["Y"], loc=mo[s["ind"], :, 0], scale=sd))
    assert isinstance(ll_dev, float) and abs(ll_dev - ll_ref) < 1e-7 * max(1.0, abs(ll_ref))
    # an arbitrary Python obs_loglik with the reference's signature
    def my_loglik(obs_data, ode_data, **params):
        return np.sum(norm.logpdf(obs_data, loc=ode_data[:, :, 0], scale=sd))
    assert Xt.shape == mo.shape and np.max(np.abs(np.asarray(Xt) - mo)) < 1e-9 and np.max(np.abs(Xt[-1] - mo[-1])) < 1e-9
    ll_call, _ = ra.inference.basic(*args, my_loglik, theta=s["theta"])
    assert abs(ll_call - ll_ref) < 1e-7 * max(1.0, abs(ll_ref))
    # batched (tile layout inside): B trajectories
    B = 5
    th = s["theta"] * np.exp(0.05 * np.random.default_rng(0).standard_normal((B, 3)))
    _, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    x0 = init(np.tile([-1., 1.], (B, 1)), 0., theta=th)
    llb, XtB = ra.inference.basic(None, ra.ode.fitzhugh_nagumo, s["W"], x0, 0., s["t_max"], s["N"],
                             ra.interrogate.interrogate_kramer, s["prior"], s["Y"], s["obs_times"],
                             GaussianObsLoglik(sd), theta=th)
    mB, _ = scan.solve_mv(None, odes.fitzhugh_nagumo, s["W"], x0, 0., s["t_max"], s["N"], oi.interrogate_kramer,
                          s["prior"], theta=th)
    ref = np.array([np.sum(norm.logpdf(s["Y"], loc=mB[b][s["ind"], :, 0], scale=sd)) for b in range(B)])
    np.testing.assert_allclose(llb, ref, rtol=1e-7, atol=1e-7)
    assert XtB.shape == mB.shape and np.max(np.abs(np.asarray(XtB) - mB)) < 1e-9
    with pytest.raises(RuntimeError):                                        # Xt of the first call: its buffers were reused
        ra.inference.basic(*args, GaussianObsLoglik(sd), theta=s["theta"])
        ra.inference.basic(*args, GaussianObsLoglik(sd), theta=s["theta"])[1]
        stale = ra.inference.basic(*args, GaussianObsLoglik(sd), theta=s["theta"])[1]
        ra.inference.basic(*args, GaussianObsLoglik(sd), theta=s["theta"])
        np.asarray(stale)


def test_chkrebtii_pseudo_marginal_logposterior(ra):
    """docs/examples/parameter.md:227-236, 331-354 for a batch of parameter draws with per-draw sigma."""
    from rodeo_amd.inference import gauss_obs_logpost, obs_index
    s = _setup(ra, n_steps=100, t_max=5.0, n_obs=6)
    B, N, t_max = 16, s["N"], s["t_max"]
    rng = np.random.default_rng(20242)
    u0 = np.concatenate([np.log(s["theta"]), [-1., 1.], [.1, .1]])
    scale = np.array([0.01, 0.1, 0.01, 0.01, 0.01, 0.01, 0.01])
    upars = u0 + scale * rng.standard_normal((B, 7))
    theta, x0v, sigma = np.exp(upars[:, :3]), upars[:, 3:5], upars[:, 5:]
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    X0 = init(x0v, 0., theta=theta)
    prior = ra.ibm_init(t_max / N, 3, sigma)                                   # batched prior_var (B, 2, 3, 3)
    g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard")
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, X0, 0., t_max, N, g, prior, theta=theta)
    plan.sim(99)
    ind = obs_index(0., t_max, N, s["obs_times"])
    np.testing.assert_array_equal(ind, s["ind"])
    sd = np.sqrt(0.005)
    lp = gauss_obs_logpost(plan, s["Y"], ind, sd, upars=upars, prior_sd=10.0, n_prior=5).to_host()
    o = functools.partial(oi.interrogate_chkrebtii, kalman_type="standard")
    xo = scan.solve_sim(99, odes.fitzhugh_nagumo, W, X0, 0., t_max, N, o, priors.ibm_init(t_max / N, 3, sigma[0])
                        if False else prior, theta=theta)
    ref = np.array([np.sum(norm.logpdf(s["Y"], loc=xo[b][ind, :, 0], scale=sd)) +
                    np.sum(norm.logpdf(upars[b, :5], 0., 10.)) for b in range(B)])
    np.testing.assert_allclose(lp, ref, rtol=1e-6, atol=1e-5)
    # the same reduction from the downloaded path
    x = plan.x_host()
    ref2 = np.array([np.sum(norm.logpdf(s["Y"], loc=x[b][ind, :, 0], scale=sd)) +
                     np.sum(norm.logpdf(upars[b, :5], 0., 10.)) for b in range(B)])
    np.testing.assert_allclose(lp, ref2, rtol=1e-12, atol=1e-9)
    with pytest.raises(ValueError):
        gauss_obs_logpost(plan, s["Y"], ind + 1000, sd)
    # Ownership of the result.  Default: a fresh device array per call -- four kept results are still intact after a fifth
    # call.  reuse_out=True: a ring of four buffers on the plan, call 5 reuses the buffer of call 1 (and only that one).
    ups = [upars + 0.01 * k for k in range(5)]
    want = [gauss_obs_logpost(plan, s["Y"], ind, sd, upars=u, n_prior=5).to_host() for u in ups]
    kept = [gauss_obs_logpost(plan, s["Y"], ind, sd, upars=u, n_prior=5) for u in ups]
    for k in range(5):
        np.testing.assert_array_equal(kept[k].to_host(), want[k])
    ring = [gauss_obs_logpost(plan, s["Y"], ind, sd, upars=u, n_prior=5, reuse_out=True) for u in ups]
    assert ring[4].ptr.value == ring[0].ptr.value and len({r.ptr.value for r in ring[:4]}) == 4
    for k in (1, 2, 3, 4):
        np.testing.assert_array_equal(ring[k].to_host(), want[k])
    np.testing.assert_array_equal(ring[0].to_host(), want[4])              # overwritten by the fifth call, as documented
    # Sampler and reduction as ONE device call (rk_solve_sim_logpost): the backward sampler's consumer wave adds the observation
    # terms itself and stores NO path -- same draws (same key), same value up to the order of the sum; observations at the first
    # and at the last grid point included (x_0 = ode_init, the terminal draw)
    from rodeo_amd.inference import sim_logpost
    plan2 = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, X0, 0., t_max, N, g, prior, theta=theta)
    lp_f = sim_logpost(plan2, 99, s["Y"], ind, sd, upars=upars, prior_sd=10.0, n_prior=5).to_host()
    assert plan2.x_state is None                                           # nothing of the path was stored
    np.testing.assert_allclose(lp_f, lp, rtol=1e-13, atol=1e-10)
    np.testing.assert_allclose(lp_f, ref, rtol=1e-6, atol=1e-5)
    ind_e = np.array([0, 3, 3, N // 2, N], dtype=np.int32)                 # ends of the grid, a repeated index
    Ye = np.random.default_rng(5).standard_normal((5, 2))
    lp_e = sim_logpost(plan2, 99, Ye, ind_e, sd, upars=None).to_host()
    ref_e = np.array([np.sum(norm.logpdf(Ye, loc=x[b][ind_e, :, 0], scale=sd)) for b in range(B)])
    np.testing.assert_allclose(lp_e, ref_e, rtol=1e-12, atol=1e-9)
    lp_k = sim_logpost(plan2, 99, s["Y"], ind, sd, upars=upars, n_prior=5, keep_path=True).to_host()     # ... and with the path kept
    np.testing.assert_allclose(lp_k, lp_f, rtol=0, atol=0)
    np.testing.assert_array_equal(plan2.x_host(), x)


def test_lockstep_pseudo_marginal_chain_on_device(ra):
    """
    docs/examples/parameter.md:372-390 for C chains in lock-step: the device log-posterior (SolvePlan.update + solve_sim
    with interrogate_chkrebtii + reduction) inside rodeo_amd.inference.pseudo_marginal's random-walk kernel, against
    the same chain driven by the oracle's log-posterior with the same keys (draws share the Philox stream, so the two
    chains make the same decisions and stay equal to rounding).
    """
    from rodeo_amd.inference import pseudo_marginal as pm
    s = _setup(ra, n_steps=60, t_max=3.0, n_obs=4)
    C, N, t_max, sd = 12, s["N"], s["t_max"], np.sqrt(0.005)
    logpost = pm.FitzLogPosterior(s["Y"], s["obs_times"], 0.0, t_max, N, sd, n_chains=C)
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    o = functools.partial(oi.interrogate_chkrebtii, kalman_type="standard")

    def oracle_logpost(upars, key):
        theta, x0v, sigma = np.exp(upars[:, :3]), upars[:, 3:5], upars[:, 5:7]
        X0 = init(x0v, 0.0, theta=theta)
        pq = [priors.ibm_init(t_max / N, 3, sigma[c]) for c in range(C)]       # per-chain prior scales
        prior = (np.stack([q for q, _ in pq]), np.stack([r for _, r in pq]))
        xo = scan.solve_sim(key, odes.fitzhugh_nagumo, W, X0, 0.0, t_max, N, o, prior, theta=theta)
        return np.array([np.sum(norm.logpdf(s["Y"], loc=xo[b][s["ind"], :, 0], scale=sd)) +
                         np.sum(norm.logpdf(upars[b, :5], 0.0, 10.0)) for b in range(C)]), None

    u0 = np.concatenate([np.log(s["theta"]), [-1.0, 1.0], [0.1, 0.1]])
    start = u0 + np.array([0.01, 0.1, 0.01, 0.01, 0.01, 0.005, 0.005]) * np.random.default_rng(5).standard_normal((C, 7))
    rw_sd = np.array([0.01, 0.1, 0.01, 0.01, 0.01, 0.01, 0.01])                 # parameter.md:372-373
    dev_alg, ora_alg = pm.normal_random_walk(logpost, rw_sd), pm.normal_random_walk(oracle_logpost, rw_sd)
    sd_state, so_state = dev_alg.init(start, 77), ora_alg.init(start, 77)
    np.testing.assert_allclose(sd_state.logdensity, so_state.logdensity, rtol=1e-6, atol=1e-5)
    n_acc = 0
    for k in range(6):
        sd_state, info_d = dev_alg.step(1000 + k, sd_state)
        so_state, info_o = ora_alg.step(1000 + k, so_state)
        np.testing.assert_array_equal(info_d.is_accepted, info_o.is_accepted)
        np.testing.assert_array_equal(sd_state.position, so_state.position)
        np.testing.assert_allclose(sd_state.logdensity, so_state.logdensity, rtol=1e-6, atol=1e-5)
        n_acc += int(info_d.is_accepted.sum())
    assert 0 < n_acc < 6 * C and np.all(np.isfinite(sd_state.logdensity))


@pytest.mark.parametrize("name", ["kramer", "rodeo"])
def test_fenrir_parity(ra, name):
    """src/rodeo/inference/fenrir.py:261-327 on the device against the oracle restatement (itself pinned by the exact
    Gaussian likelihood of a linear model, tests/test_oracle_fenrir.py): single trajectory and a batch, an observation
    at the terminal time, one at time 0, several in between."""
    from oracle import fenrir as ofen
    s = _setup(ra, n_steps=60, t_max=3.0, n_obs=5)
    g = {"kramer": ra.interrogate.interrogate_kramer, "rodeo": ra.interrogate.interrogate_rodeo}[name]
    o = {"kramer": oi.interrogate_kramer, "rodeo": oi.interrogate_rodeo}[name]
    n_obs = len(s["obs_times"])
    y = s["Y"][:, :, None]                                                     # (n_obs, d, 1)
    Dw = np.zeros((n_obs, 2, 1, 3)); Dw[..., 0] = 1.0
    Om = np.full((n_obs, 2, 1, 1), 0.005)
    args = (s["W"], s["x0"], 0.0, s["t_max"], s["N"])
    val = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], y, s["obs_times"], Dw, Om,
                              theta=s["theta"])
    ref = ofen.fenrir(None, odes.fitzhugh_nagumo, *args, o, s["prior"], y, s["obs_times"], Dw, Om, theta=s["theta"])
    assert isinstance(val, float) and abs(val - ref) < 1e-7 * max(1.0, abs(ref)), (val, ref)
    B = 7
    th = s["theta"] * np.exp(0.05 * np.random.default_rng(1).standard_normal((B, 3)))
    _, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    x0 = init(np.tile([-1., 1.], (B, 1)) + 0.05 * np.random.default_rng(2).standard_normal((B, 2)), 0., theta=th)
    vb = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, s["W"], x0, 0.0, s["t_max"], s["N"], g, s["prior"], y,
                             s["obs_times"], Dw, Om, theta=th)
    rb = ofen.fenrir(None, odes.fitzhugh_nagumo, s["W"], x0, 0.0, s["t_max"], s["N"], o, s["prior"], y, s["obs_times"],
                     Dw, Om, theta=th)
    assert vb.shape == (B,)
    np.testing.assert_allclose(vb, rb, rtol=1e-7, atol=1e-7)
    with pytest.raises(NotImplementedError):
        ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], y, s["obs_times"], Dw, Om,
                            kalman_type="cubature", theta=s["theta"])


def test_lazy_mean_goes_stale_when_fenrir_reuses_the_cached_plan(ra):
    """basic() and fenrir() with the same configuration share one cached SolvePlan (solve.cached_plan): fenrir's forward
    filter overwrites the smoothed moments basic() left in HBM, so an Xt that was not read before must refuse to hand out
    numbers (SolvePlan.generation), and one that was read keeps its values."""
    from rodeo_amd.inference.basic import GaussianObsLoglik
    s = _setup(ra, n_steps=60, t_max=3.0, n_obs=5)
    sd = np.sqrt(0.005)
    n_obs = len(s["obs_times"])
    y = s["Y"][:, :, None]
    Dw = np.zeros((n_obs, 2, 1, 3)); Dw[..., 0] = 1.0
    Om = np.full((n_obs, 2, 1, 1), 0.005)
    args = (None, ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0., s["t_max"], s["N"], ra.interrogate.interrogate_kramer,
            s["prior"])
    _, read = ra.inference.basic(*args, s["Y"], s["obs_times"], GaussianObsLoglik(sd), theta=s["theta"])
    kept = np.asarray(read).copy()
    _, unread = ra.inference.basic(*args, s["Y"], s["obs_times"], GaussianObsLoglik(sd), theta=s["theta"])
    ra.inference.fenrir(*args, y, s["obs_times"], Dw, Om, theta=s["theta"])
    np.testing.assert_array_equal(np.asarray(read), kept)
    with pytest.raises(RuntimeError):
        np.asarray(unread)


def test_config4_full_size_per_gpu(ra):
    """
    BASELINE.json config 4 at its per-GPU size (1024 parameter draws, FN, N=800, solve_sim + interrogate_chkrebtii,
    41 observations, log-posterior on the device): sampled draws against the oracle with the shared Philox stream
    (global draw index = traj_offset + b, as on rank 3 of 8), the download-and-reduce value for every draw, and
    invariance to the batch the draw sits in.
    """
    from rodeo_amd.inference import gauss_obs_logpost, obs_index
    B, N, off = 1024, 800, 3 * 1024
    rng = np.random.default_rng(20242)
    u0 = np.concatenate([np.log([.2, .2, 3.]), [-1., 1.], [.1, .1]])
    upars = u0 + np.array([0.01, 0.1, 0.01, 0.01, 0.01, 0.01, 0.01]) * rng.standard_normal((B, 7))
    theta, x0v, sigma = np.exp(upars[:, :3]), upars[:, 3:5], upars[:, 5:]
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    X0 = init(x0v, 0., theta=theta)
    prior = ra.ibm_init(40.0 / N, 3, sigma)
    g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard")
    o = functools.partial(oi.interrogate_chkrebtii, kalman_type="standard")
    obs_t = np.linspace(0, 40, 41)
    ind = obs_index(0., 40., N, obs_t)
    Y = np.array([-1., 1.]) + rng.standard_normal((41, 2))
    sd = np.sqrt(0.005)
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, X0, 0., 40., N, g, prior, traj_offset=off, theta=theta)
    plan.sim(20242)
    lp = gauss_obs_logpost(plan, Y, ind, sd, upars=upars, prior_sd=10.0, n_prior=5).to_host()
    assert lp.shape == (B,) and np.all(np.isfinite(lp))
    x = plan.x_host()
    assert x.shape == (B, N + 1, 2, 3)
    ref = np.sum(norm.logpdf(Y[None], loc=x[:, ind][..., 0], scale=sd), axis=(1, 2)) + np.sum(norm.logpdf(upars[:, :5], 0., 10.), axis=1)
    np.testing.assert_allclose(lp, ref, rtol=1e-12, atol=1e-8)
    for b in (0, 511, 1023):
        sl = slice(b, b + 1)
        xo = scan.solve_sim(20242, odes.fitzhugh_nagumo, W, X0[sl], 0., 40., N, o, (prior[0], prior[1][sl]),
                            traj_offset=off + b, theta=theta[sl])
        assert np.max(np.abs(x[b] - xo[0])) < 1e-6
        p1 = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, X0[sl], 0., 40., N, g, (prior[0], prior[1][sl]), traj_offset=off + b,
                          theta=theta[sl])
        p1.sim(20242)
        np.testing.assert_array_equal(p1.x_host()[0], x[b])


@pytest.mark.parametrize("p", [3, 4, 5, 6, 7])        # (n_bstate = 8 on this grid: the fp64 ORACLE overflows)
def test_fenrir_long_horizon_sparse_observations(ra, p):
    """Fenrir with few observations on a long grid: at n_bstate = 3 the MFMA-tile kernels (forward tiles, backward filter
    with whole 16-step chunks between observations); at n_bstate = 4 .. 8 the blocked-tile forward pass and the
    lane-per-block backward filter on its records (predictions re-evaluated).  Batch of 5 against the oracle."""
    from oracle import fenrir as ofen
    N, t_max, n_obs, B = 200, 10.0, 4, 5
    rng = np.random.default_rng(7)
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.05 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0 = init(np.array([-1., 1.]) + 0.05 * rng.standard_normal((B, 2)), 0.0, theta=theta)
    prior = ra.ibm_init(t_max / N, p, np.array([.1, .1]))
    obs_times = np.array([0.0, 3.35, 7.5, 10.0])
    y = rng.standard_normal((n_obs, 2, 1))
    Dw = np.zeros((n_obs, 2, 1, p)); Dw[..., 0] = 1.0; Dw[..., 1] = 0.3
    Om = np.full((n_obs, 2, 1, 1), 0.05)
    args = (W, x0, 0.0, t_max, N)
    val = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, prior, y, obs_times,
                              Dw, Om, theta=theta)
    ref = ofen.fenrir(None, odes.fitzhugh_nagumo, *args, oi.interrogate_kramer, prior, y, obs_times, Dw, Om, theta=theta)
    assert val.shape == (B,)
    tol = {3: 1e-7, 4: 1e-7, 5: 1e-5, 6: 1e-5, 7: 1e-4}[p]                                   # (conditioning, test_gpu_tilen.py)
    np.testing.assert_allclose(val, ref, rtol=tol, atol=tol)
    # which kernels ran: the forward pass on tiles at every n_bstate here, no lane-per-trajectory filter
    from rodeo_amd.solve import cached_plan
    plan = cached_plan(ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, prior, "standard", theta=theta)
    plan.dev.profile_enable(True)
    ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, prior, y, obs_times, Dw, Om, theta=theta)
    names = [k for k, _ in plan.dev.profile_last()]
    plan.dev.profile_enable(False)
    assert any(k.startswith("fwd_tile") for k in names), names
    assert any("fenrir" in k and ("tile" in k) for k in names), names


@pytest.mark.parametrize("N", [1, 2, 16, 17, 18, 33])
def test_fenrir_tile_short_horizons(ra, N):
    """Chunk / pipeline boundaries of the tile backward filter: horizons around the 16-step chunk size, observations at
    both ends and (when there is room) in between, three trajectories (ragged tile-waves)."""
    from oracle import fenrir as ofen
    t_max, B = 0.05 * N, 3
    rng = np.random.default_rng(100 + N)
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.05 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    x0 = init(np.array([-1., 1.]) + 0.05 * rng.standard_normal((B, 2)), 0.0, theta=theta)
    prior = ra.ibm_init(t_max / N, 3, np.array([.1, .1]))
    obs_times = np.unique(np.array([0.0, 0.05 * (N // 2), t_max]))
    n_obs = len(obs_times)
    y = rng.standard_normal((n_obs, 2, 1))
    Dw = np.zeros((n_obs, 2, 1, 3)); Dw[..., 0] = 1.0
    Om = np.full((n_obs, 2, 1, 1), 0.01)
    args = (W, x0, 0.0, t_max, N)
    val = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, prior, y, obs_times,
                              Dw, Om, theta=theta)
    ref = ofen.fenrir(None, odes.fitzhugh_nagumo, *args, oi.interrogate_kramer, prior, y, obs_times, Dw, Om, theta=theta)
    np.testing.assert_allclose(val, ref, rtol=1e-7, atol=1e-7)


def test_fenrir_lorenz_three_blocks(ra):
    """docs/examples/lorenz.md's setting (Lorenz63, n_deriv = 3, three variables) through fenrir: tiles of one trajectory
    spread over tile-waves (three blocks, four tiles per wave), built-in functor and the traced Python function."""
    from oracle import fenrir as ofen

    def lorenz(X, t, **params):
        rho, sigma, beta = params["theta"]
        x, y, z = X[:, 0]
        return np.array([[-sigma * x + sigma * y], [rho * x - y - x * z], [-beta * z + x * y]])
    N, t_max, B = 120, 0.6, 6
    rng = np.random.default_rng(5)
    theta = np.array([28., 10., 8. / 3.]) * np.exp(0.01 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.lorenz63, 3, 3)
    x0 = init(np.array([-12., -5., 38.]) + 0.1 * rng.standard_normal((B, 3)), 0.0, theta=theta)
    prior = ra.ibm_init(t_max / N, 3, np.array([5e7] * 3))
    obs_times = np.linspace(0, t_max, 7)
    n_obs = len(obs_times)
    y = np.array([-12., -5., 38.])[None, :, None] + rng.standard_normal((n_obs, 3, 1))
    Dw = np.zeros((n_obs, 3, 1, 3)); Dw[..., 0] = 1.0
    Om = np.full((n_obs, 3, 1, 1), 0.25)
    args = (W, x0, 0.0, t_max, N)
    ref = ofen.fenrir(None, odes.lorenz63, *args, oi.interrogate_kramer, prior, y, obs_times, Dw, Om, theta=theta)
    for fun in (ra.ode.lorenz63, lorenz):
        val = ra.inference.fenrir(None, fun, *args, ra.interrogate.interrogate_kramer, prior, y, obs_times, Dw, Om, theta=theta)
        np.testing.assert_allclose(val, ref, rtol=1e-7, atol=1e-6)


@pytest.mark.parametrize("p", [3, 4])
def test_fenrir_solve_mv_parity(ra, p):
    """inference.fenrir.solve_mv (fenrir.py:405-457, the data-adaptive solver of docs/examples/lorenz.md) against the oracle
    restatement (itself pinned by the exact Gaussian posterior of a linear model): single trajectory and a batch,
    observations at both ends and in between."""
    from oracle import fenrir as ofen
    from rodeo_amd.inference.fenrir import solve_mv as fsolve            # as docs/examples/lorenz.md:39 imports it
    N, t_max, B = 60, 3.0, 4
    rng = np.random.default_rng(21)
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.05 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0 = init(np.array([-1., 1.]) + 0.05 * rng.standard_normal((B, 2)), 0.0, theta=theta)
    prior = ra.ibm_init(t_max / N, p, np.array([.1, .1]))
    obs_times = np.array([0.0, 0.75, 1.5, 2.25, 3.0])
    n_obs = len(obs_times)
    y = np.array([-1., 1.])[None, :, None] + 0.3 * rng.standard_normal((n_obs, 2, 1))
    Dw = np.zeros((n_obs, 2, 1, p)); Dw[..., 0] = 1.0
    Om = np.full((n_obs, 2, 1, 1), 0.02)
    m, v = fsolve(None, ra.ode.fitzhugh_nagumo, W, x0, 0.0, t_max, N, ra.interrogate.interrogate_kramer, prior,
                         y, obs_times, Dw, Om, theta=theta)
    assert m.shape == (B, N + 1, 2, p) and v.shape == (B, N + 1, 2, p, p)
    for b in range(B):
        mo, vo = ofen.solve_mv(None, odes.fitzhugh_nagumo, W, x0[b], 0.0, t_max, N, oi.interrogate_kramer, prior, y,
                               obs_times, Dw, Om, theta=theta[b])
        assert np.max(np.abs(m[b] - mo)) < 1e-8 * max(1.0, np.max(np.abs(mo)))
        assert np.max(np.abs(v[b] - vo)) < 1e-7 * np.max(np.abs(vo))
    m1, v1 = fsolve(None, ra.ode.fitzhugh_nagumo, W, x0[0], 0.0, t_max, N, ra.interrogate.interrogate_kramer, prior,
                           y, obs_times, Dw, Om, theta=theta[0])
    assert m1.shape == (N + 1, 2, p) and np.max(np.abs(m1 - m[0])) < 1e-12
    # the observations pull the solution: at an observed time the posterior mean sits between solver and data
    m0, _ = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, W, x0[0], 0.0, t_max, N, ra.interrogate.interrogate_kramer, prior,
                        theta=theta[0])
    assert np.max(np.abs(m1 - m0)) > 1e-9


def test_lorenz_md_example_with_fenrir_solver():
    """examples/lorenz_fenrir.py = docs/examples/lorenz.md (its Python lorenz function, settings and data): the data-free
    solver is accurate early and has left the chaotic trajectory by t = 10; the Fenrir solver passes through the
    observations (the document's own conclusion: only dalton recovers the truth in between)."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "lorenz_fenrir.py")
    spec = importlib.util.spec_from_file_location("lorenz_fenrir", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    err_early, at_obs_r, at_obs_f = mod.main()
    assert err_early < 3.0 and at_obs_r > 5.0 and at_obs_f < 0.05


@pytest.mark.parametrize("N", [1, 2, 3])
def test_fenrir_solve_mv_tiny_horizons(ra, N):
    """fenrir.solve_mv at N = 1, 2, 3 (the smoothing pass starts at time 2) and with a single observation at the end."""
    from oracle import fenrir as ofen
    from rodeo_amd.inference.fenrir import solve_mv as fsolve
    theta = np.array([0.2, 0.2, 3.0])
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    x0 = init(np.array([-1., 1.]), 0.0, theta=theta)
    t_max = 0.05 * N
    prior = ra.ibm_init(0.05, 3, np.array([.1, .1]))
    obs_times = np.array([t_max])
    y = np.array([[[-0.9], [0.95]]])
    Dw = np.zeros((1, 2, 1, 3)); Dw[..., 0] = 1.0
    Om = np.full((1, 2, 1, 1), 0.01)
    args = (W, x0, 0.0, t_max, N)
    m, v = fsolve(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, prior, y, obs_times, Dw, Om, theta=theta)
    mo, vo = ofen.solve_mv(None, odes.fitzhugh_nagumo, *args, oi.interrogate_kramer, prior, y, obs_times, Dw, Om, theta=theta)
    assert m.shape == (N + 1, 2, 3) and np.max(np.abs(m - mo)) < 1e-9 and np.max(np.abs(v - vo)) < 1e-9 * max(np.max(np.abs(vo)), 1e-300)


@pytest.mark.parametrize("p,n_bobs", [(3, 2), (3, 3), (4, 2), (5, 3)])
def test_fenrir_vector_observations(ra, p, n_bobs):
    """n_bobs > 1 (src/rodeo/inference/fenrir.py:106-122): several observed linear combinations per block with a full
    observation covariance -- LU update (utils.py:119) and the eigendecomposition log-density of utils.py:60-78 on the
    device -- for inference.fenrir and for fenrir.solve_mv, single trajectory and batch, against the oracle (pinned by the
    exact Gaussian likelihood, tests/test_oracle_fenrir.py)."""
    from oracle import fenrir as ofen
    from rodeo_amd.inference.fenrir import solve_mv as fsolve
    N, t_max, B = 48, 2.4, 5
    rng = np.random.default_rng(100 * p + n_bobs)
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.05 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0 = init(np.array([-1., 1.]) + 0.05 * rng.standard_normal((B, 2)), 0.0, theta=theta)
    prior = ra.ibm_init(t_max / N, p, np.array([.1, .1]))
    obs_times = np.array([0.0, 0.6, 1.2, 1.8, 2.4])
    n_obs = len(obs_times)
    Dw = 0.3 * rng.standard_normal((n_obs, 2, n_bobs, p)); Dw[:, :, 0, 0] += 1.0; Dw[:, :, 1, 1] += 1.0
    a = 0.2 * rng.standard_normal((n_obs, 2, n_bobs, n_bobs))
    Om = a @ np.swapaxes(a, -1, -2) + 0.05 * np.eye(n_bobs)
    y = rng.standard_normal((n_obs, 2, n_bobs))
    args = (W, x0, 0.0, t_max, N)
    g, o = ra.interrogate.interrogate_kramer, oi.interrogate_kramer
    val = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, g, prior, y, obs_times, Dw, Om, theta=theta)
    ref = ofen.fenrir(None, odes.fitzhugh_nagumo, *args, o, prior, y, obs_times, Dw, Om, theta=theta)
    assert val.shape == (B,)
    np.testing.assert_allclose(val, ref, rtol=1e-7, atol=1e-6)
    one = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, W, x0[1], 0.0, t_max, N, g, prior, y, obs_times, Dw, Om, theta=theta[1])
    assert isinstance(one, float) and abs(one - ref[1]) < 1e-6 * max(1.0, abs(ref[1]))
    m, v = fsolve(None, ra.ode.fitzhugh_nagumo, *args, g, prior, y, obs_times, Dw, Om, theta=theta)
    for b in (0, B - 1):
        mo, vo = ofen.solve_mv(None, odes.fitzhugh_nagumo, W, x0[b], 0.0, t_max, N, o, prior, y, obs_times, Dw, Om, theta=theta[b])
        scale = np.maximum(np.max(np.abs(mo), axis=(0, 1)), 1.0)
        assert np.max(np.abs(m[b] - mo) / scale) < 1e-7
        assert np.max(np.abs(v[b] - vo)) < 1e-6 * np.max(np.abs(vo))
    with pytest.raises(NotImplementedError):
        ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, g, prior, np.zeros((n_obs, 2, 4)), obs_times,
                            np.zeros((n_obs, 2, 4, p)), np.zeros((n_obs, 2, 4, 4)), theta=theta)


@pytest.mark.parametrize("name,p,n_bobs", [("kramer", 3, 1), ("rodeo", 3, 1), ("kramer", 4, 1), ("kramer", 3, 2), ("rodeo", 5, 2),
                                            ("kramer", 2, 1), ("kramer", 6, 3), ("kramer", 7, 1), ("rodeo", 8, 2)])
def test_fenrir_parity_square_root(ra, name, p, n_bobs):
    """fenrir with kalman_type="square-root" (src/rodeo/inference/fenrir.py:292-296 with square_root.py:30-345: the forecast
    returns the full variance, square_root.py:343-344): prior_pars[1] and obs_var are lower factors.  Device against the
    oracle's square-root branch (1e-7), and -- exact measurement, the same model in both forms -- against the covariance
    form's value for the L L^T inputs (1e-6); a batch, observations at both ends."""
    from oracle import fenrir as ofen
    N, t_max, B = 60, 3.0, 5
    rng = np.random.default_rng(40 + p + n_bobs)
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.05 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0 = init(np.array([-1., 1.]) + 0.05 * rng.standard_normal((B, 2)), 0.0, theta=theta)
    Q, R = ra.ibm_init(t_max / N, p, np.array([.1, .1]))
    if p >= 7:                                           # (the IBM variance matrix is singular in fp64 there: a well-conditioned factor)
        Rh = np.tril(0.02 * rng.standard_normal((2, p, p))) + np.eye(p) * 0.1
    else:
        Rh = np.linalg.cholesky(R)
    obs_times = np.array([0.0, 0.75, 1.5, 2.2, 3.0])
    n_obs = len(obs_times)
    y = rng.standard_normal((n_obs, 2, n_bobs))
    Dw = 0.3 * rng.standard_normal((n_obs, 2, n_bobs, p)); Dw[..., 0, 0] = 1.0
    A = rng.standard_normal((n_obs, 2, n_bobs, n_bobs))
    Om = 0.02 * (A @ np.swapaxes(A, -1, -2) + np.eye(n_bobs))
    Oh = np.linalg.cholesky(Om)
    g, o = getattr(ra.interrogate, "interrogate_" + name), getattr(oi, "interrogate_" + name)
    args = (W, x0, 0.0, t_max, N)
    val = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, g, (Q, Rh), y, obs_times, Dw, Oh, kalman_type="square-root",
                              theta=theta)
    ref = ofen.fenrir(None, odes.fitzhugh_nagumo, *args, o, (Q, Rh), y, obs_times, Dw, Oh, kalman_type="square-root", theta=theta)
    assert val.shape == (B,)
    tol = 1e-7 if p <= 4 else (1e-5 if p <= 6 else 1e-4)
    np.testing.assert_allclose(val, ref, rtol=tol, atol=tol)
    if name == "kramer" and p <= 4:
        std = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, g, (Q, R), y, obs_times, Dw, Om, theta=theta)
        np.testing.assert_allclose(val, std, rtol=1e-6, atol=1e-6)
    one = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, W, x0[0], 0.0, t_max, N, g, (Q, Rh), y, obs_times, Dw, Oh,
                              kalman_type="square-root", theta=theta[0])
    assert isinstance(one, float) and abs(one - ref[0]) < tol * max(1.0, abs(ref[0]))


@pytest.mark.parametrize("p,n_bobs", [(3, 1), (4, 2)])
def test_fenrir_solve_mv_square_root(ra, p, n_bobs):
    """fenrir.solve_mv with kalman_type="square-root" (fenrir.py:421-426, square_root.py:209-219 in the smoothing sweep) against
    the oracle: means, and the factors as L L^T.  interrogate_rodeo: with an exact measurement the backward filter's predicted
    factor is singular and square_root.py:172-174 divides by its zero diagonal -- in the reference as in the oracle
    (tests/test_oracle_fenrir.py)."""
    from oracle import fenrir as ofen
    N, t_max, B = 40, 2.0, 3
    rng = np.random.default_rng(60 + p)
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.05 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0 = init(np.array([-1., 1.]) + 0.05 * rng.standard_normal((B, 2)), 0.0, theta=theta)
    Q, R = ra.ibm_init(t_max / N, p, np.array([.1, .1]))
    Rh = np.linalg.cholesky(R)
    obs_times = np.array([0.0, 0.5, 1.25, 2.0])
    n_obs = len(obs_times)
    y = rng.standard_normal((n_obs, 2, n_bobs))
    Dw = 0.3 * rng.standard_normal((n_obs, 2, n_bobs, p)); Dw[..., 0, 0] = 1.0
    Oh = np.tile(np.linalg.cholesky(0.05 * np.eye(n_bobs) + 0.01), (n_obs, 2, 1, 1))
    from rodeo_amd.inference.fenrir import solve_mv as fenrir_solve_mv
    m, L = fenrir_solve_mv(None, ra.ode.fitzhugh_nagumo, W, x0, 0.0, t_max, N, ra.interrogate.interrogate_rodeo, (Q, Rh), y,
                           obs_times, Dw, Oh, kalman_type="square-root", theta=theta)
    assert m.shape == (B, N + 1, 2, p) and L.shape == (B, N + 1, 2, p, p)
    for b in range(B):
        mo, Lo = ofen.solve_mv(None, odes.fitzhugh_nagumo, W, x0[b], 0.0, t_max, N, oi.interrogate_rodeo, (Q, Rh), y, obs_times,
                               Dw, Oh, kalman_type="square-root", theta=theta[b])
        scale = np.maximum(np.max(np.abs(mo), axis=(0, 1)), 1.0)
        assert np.max(np.abs(m[b] - mo) / scale) < 1e-7
        vo, v = Lo @ np.swapaxes(Lo, -1, -2), L[b] @ np.swapaxes(L[b], -1, -2)
        dv = np.sqrt(np.abs(np.einsum("nkii->nki", vo)).max(axis=(0, 1)))
        assert np.max(np.abs(v - vo) / (dv[:, None] * dv[None, :])) < 1e-6


@pytest.mark.parametrize("p", [4, 5, 6])          # (n_bstate = 7 on this grid: the fp64 ORACLE meets a singular matrix in its smoothing pass)
def test_fenrir_solve_mv_on_tile_records(ra, p):
    """fenrir.solve_mv at n_bstate = 4 .. 8 (fenrir.py:333-457): forward pass on the blocked MFMA tiles, backward filter and
    smoothing pass on its records (rk_fenrir_solve_mv_tiles: predictions re-evaluated, no lane-per-trajectory filter, no stored
    predictions) -- against the oracle, and against the lane-per-trajectory path of rk_fenrir_solve_mv where that exists (p <= 6)."""
    from oracle import fenrir as ofen
    from rodeo_amd.inference.fenrir import solve_mv as fsolve
    from rodeo_amd.solve import SolvePlan
    from rodeo_amd import _lib
    import ctypes as C
    N, t_max, B = 80, 4.0, 5
    rng = np.random.default_rng(40 + p)
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.05 * rng.standard_normal((B, 3)))
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0 = init(np.array([-1., 1.]) + 0.05 * rng.standard_normal((B, 2)), 0.0, theta=theta)
    prior = ra.ibm_init(t_max / N, p, np.array([.1, .1]))
    obs_times = np.array([0.0, 1.0, 2.0, 3.0, 4.0])
    n_obs = len(obs_times)
    y = np.array([-1., 1.])[None, :, None] + 0.3 * rng.standard_normal((n_obs, 2, 1))
    Dw = np.zeros((n_obs, 2, 1, p)); Dw[..., 0] = 1.0
    Om = np.full((n_obs, 2, 1, 1), 0.02)
    args = (W, x0, 0.0, t_max, N)
    g = ra.interrogate.interrogate_kramer
    dev = ra.device.default_device()
    dev.profile_enable(True)
    m, v = fsolve(None, ra.ode.fitzhugh_nagumo, *args, g, prior, y, obs_times, Dw, Om, theta=theta)
    names = [k for k, _ in dev.profile_last()]
    dev.profile_enable(False)
    assert any(k.startswith("fwd_tile") for k in names) and "fenrir_bwd_kernel<tiles>" in names and "fenrir_smooth_kernel" in names, names
    assert not any(k == "fwd_kernel" for k in names), names
    assert m.shape == (B, N + 1, 2, p) and v.shape == (B, N + 1, 2, p, p)
    tol_m, tol_v = {4: (1e-8, 1e-7), 5: (1e-7, 1e-6), 6: (1e-6, 1e-5)}[p]                       # (conditioning, test_gpu_tilen.py)
    for b in (0, B - 1):
        mo, vo = ofen.solve_mv(None, odes.fitzhugh_nagumo, W, x0[b], 0.0, t_max, N, oi.interrogate_kramer, prior, y,
                               obs_times, Dw, Om, theta=theta[b])
        scale = np.maximum(np.max(np.abs(mo), axis=(0, 1)), 1.0)
        assert np.max(np.abs(m[b] - mo) / scale) < tol_m
        assert np.max(np.abs(v[b] - vo)) < tol_v * np.max(np.abs(vo))
    m1, _ = fsolve(None, ra.ode.fitzhugh_nagumo, W, x0[0], 0.0, t_max, N, g, prior, y, obs_times, Dw, Om, theta=theta[0])
    assert m1.shape == (N + 1, 2, p) and np.max(np.abs(m1 - m[0])) < 1e-12
    if p <= 6:
        # the lane-per-trajectory path (stored predictions) on the same problem
        from rodeo_amd.inference.fenrir import _check_obs
        from rodeo_amd.inference.logpost import obs_index
        obs, D, Omc, n_bobs = _check_obs(y, Dw, Om)
        ind = obs_index(0.0, t_max, N, obs_times)
        plan = SolvePlan(ra.ode.fitzhugh_nagumo, *args, g, prior, "standard", store_pred=True, batch_minor=True, theta=theta)
        plan.filter(None)
        d_obs, d_w, d_v, d_ind = (dev.to_device(np.ascontiguousarray(a)) for a in (obs, D, Omc, ind.astype(np.int32)))
        nbytes = C.c_size_t(0)
        _lib.check(dev.lib.rk_fenrir_workspace_bytes(C.byref(plan.cfg), C.byref(nbytes)))
        ws = dev.empty((nbytes.value // 8,))
        _lib.check(dev.lib.rk_fenrir_solve_mv(dev.h, C.byref(plan.cfg), C.byref(plan.inp), C.byref(plan._out), d_obs.ptr, d_w.ptr,
                                              d_v.ptr, d_ind.ptr, int(ind.shape[0]), n_bobs, ws.ptr))
        ml, vl = plan.state_host()
        scale = np.maximum(np.max(np.abs(ml), axis=(0, 1, 2)), 1.0)
        assert np.max(np.abs(m - ml) / scale) < 10 * tol_m
