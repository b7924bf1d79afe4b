"""
GPU parity of the per-step operator boundary (rk_kalman_*_batched through rodeo_amd.kalmantv.*):
  * against the reference's own test oracle K1 (joint-Gaussian conditioning), exactly as tests/test_standard.py /
    tests/test_square_root.py of the reference do, tolerance 5e-8 in the reference's rel_err metric;
  * against oracle/kalman_ops.py / sqrt_ops.py on random batches (single-step maps on identical inputs): tight.
"""
import numpy as np
import pytest
from oracle import kalman_ops as oktv, sqrt_ops as osq, joint_gaussian as jg

pytestmark = pytest.mark.gpu
TOL = 5e-8


@pytest.fixture(scope="module")
def ktv():
    import rodeo_amd.kalmantv.standard as m
    return m


@pytest.fixture(scope="module")
def sq():
    import rodeo_amd.kalmantv.square_root as m
    return m


@pytest.fixture(params=range(6))
def mdl(request):
    return jg.random_model(np.random.default_rng(3000 + request.param))


def _close(a, b, tol=TOL):
    assert np.shape(a) == np.shape(b)
    assert jg.rel_err(a, b) < tol


def test_standard_vs_joint_gaussian(ktv, mdl):
    for step in (1, 2):
        past, pred, filt = jg.filter_targets(mdl, step)
        kw = dict(mean_state=mdl["mean_state"][step], wgt_state=mdl["wgt_state"][step - 1],
                  var_state=mdl["var_state"][step])
        kwm = dict(x_meas=mdl["x_meas"][step], mean_meas=mdl["mean_meas"][step], wgt_meas=mdl["wgt_meas"][step],
                   var_meas=mdl["var_meas"][step])
        mp, vp = ktv.predict(mean_state_past=past[0], var_state_past=past[1], **kw)
        _close(pred[0], mp); _close(pred[1], vp)
        mf, vf = ktv.update(mean_state_pred=pred[0], var_state_pred=pred[1], **kwm)
        _close(filt[0], mf); _close(filt[1], vf)
        out = ktv.filter(mean_state_past=past[0], var_state_past=past[1], **kw, **kwm)
        for a, b in zip((pred[0], pred[1], filt[0], filt[1]), out):
            _close(a, b)
    t = jg.smooth_targets(mdl)
    sm = dict(mean_state_filt=t["filt"][0], var_state_filt=t["filt"][1], mean_state_pred=t["pred"][0],
              var_state_pred=t["pred"][1], wgt_state=mdl["wgt_state"][0])
    m2, v2 = ktv.smooth_mv(mean_state_next=t["next"][0], var_state_next=t["next"][1], **sm)
    _close(t["smooth"][0], m2); _close(t["smooth"][1], v2)
    m3, v3 = ktv.smooth_sim(x_state_next=mdl["x_state_next"], **sm)
    _close(t["sim"][0], m3); _close(t["sim"][1], v3)
    o = ktv.smooth(x_state_next=mdl["x_state_next"], mean_state_next=t["next"][0], var_state_next=t["next"][1], **sm)
    _close(t["sim"][0], o[0]); _close(t["sim"][1], o[1]); _close(t["smooth"][0], o[2]); _close(t["smooth"][1], o[3])
    A2, b2, V2 = ktv.smooth_cond(**sm)
    A, b, V = t["cond"]
    _close(A, A2); _close(b, b2); _close(V, V2)


def _rand_batch(rng, n, p, m):
    def spd(k, lead):
        a = rng.standard_normal(lead + (k, k))
        return a @ np.swapaxes(a, -1, -2) + 0.1 * np.eye(k)
    return dict(mu=rng.standard_normal((n, p)), S=spd(p, (n,)), Q=0.3 * rng.standard_normal((n, p, p)),
                R=spd(p, (n,)), c=rng.standard_normal((n, p)), W=rng.standard_normal((n, m, p)), V=spd(m, (n,)),
                a=rng.standard_normal((n, m)), x=rng.standard_normal((n, m)), xn=rng.standard_normal((n, p)),
                mn=rng.standard_normal((n, p)), Sn=spd(p, (n,)))


@pytest.mark.parametrize("p,m", [(2, 1), (3, 1), (4, 2), (6, 3), (8, 1), (12, 4), (16, 2)])
def test_standard_vs_oracle_batches(ktv, p, m):
    r = _rand_batch(np.random.default_rng(p * 10 + m), 130, p, m)           # 130: not a multiple of the wave size
    mp, vp = ktv.predict(r["mu"], r["S"], r["c"], r["Q"], r["R"])
    omp, ovp = oktv.predict(r["mu"], r["S"], r["c"], r["Q"], r["R"])
    np.testing.assert_allclose(mp, omp, rtol=1e-12, atol=1e-12); np.testing.assert_allclose(vp, ovp, rtol=1e-12, atol=1e-12)
    mf, vf = ktv.update(omp, ovp, r["x"], r["a"], r["W"], r["V"])
    omf, ovf = oktv.update(omp, ovp, r["x"], r["a"], r["W"], r["V"])
    np.testing.assert_allclose(mf, omf, rtol=1e-9, atol=1e-10); np.testing.assert_allclose(vf, ovf, rtol=1e-9, atol=1e-10)
    fo = ktv.forecast(omp, ovp, r["a"], r["W"], r["V"])
    ofo = oktv.forecast(omp, ovp, r["a"], r["W"], r["V"])
    np.testing.assert_allclose(fo[0], ofo[0], rtol=1e-12, atol=1e-12); np.testing.assert_allclose(fo[1], ofo[1], rtol=1e-12, atol=1e-12)
    sm = ktv.smooth(r["xn"], r["mn"], r["Sn"], omf, ovf, omp, ovp, r["Q"])
    osm = oktv.smooth(r["xn"], r["mn"], r["Sn"], omf, ovf, omp, ovp, r["Q"])
    for g, o in zip(sm, osm):
        np.testing.assert_allclose(g, o, rtol=1e-8, atol=1e-9)
    sc = ktv.smooth_cond(omf, ovf, omp, ovp, r["Q"])
    osc = oktv.smooth_cond(omf, ovf, omp, ovp, r["Q"])
    for g, o in zip(sc, osc):
        np.testing.assert_allclose(g, o, rtol=1e-8, atol=1e-9)


def test_broadcast_and_kwargs(ktv):
    """Leading dims broadcast like vmap; unknown kwargs are swallowed (standard.py:36)."""
    rng = np.random.default_rng(0)
    p = 3
    mu = rng.standard_normal((2, 5, p)); a = rng.standard_normal((5, p, p)); S = a @ np.swapaxes(a, -1, -2)
    Q = rng.standard_normal((p, p)); R = np.eye(p)
    mp, vp = ktv.predict(mean_state_past=mu, var_state_past=S, mean_state=np.zeros(p), wgt_state=Q, var_state=R,
                         extra_kwarg=1)
    assert mp.shape == (2, 5, p) and vp.shape == (2, 5, p, p)
    o = oktv.predict(mu, S, np.zeros(p), Q, R)
    np.testing.assert_allclose(mp, o[0], rtol=1e-12, atol=1e-13); np.testing.assert_allclose(vp, np.broadcast_to(o[1], vp.shape), rtol=1e-12, atol=1e-13)
    # unbatched call returns unbatched shapes
    m1, v1 = ktv.predict(mu[0, 0], S[0], np.zeros(p), Q, R)
    assert m1.shape == (p,) and v1.shape == (p, p)


def test_errors(ktv, sq):
    from rodeo_amd._lib import RodeoKalmanError
    with pytest.raises(RodeoKalmanError):
        ktv.predict(np.zeros(800), np.eye(800), np.zeros(800), np.eye(800), np.eye(800))      # beyond the dense blocks' 768
    with pytest.raises(TypeError):
        sq.smooth_mv(np.zeros(2), np.eye(2), np.zeros(2), np.eye(2), np.zeros(2), np.eye(2), np.eye(2))  # var_state required


def _chol(A):
    return np.linalg.cholesky(A)


def _sqr(L):
    return L @ np.swapaxes(L, -1, -2)


def test_square_root_vs_joint_gaussian(sq, mdl):
    past, pred, filt = jg.filter_targets(mdl, 1)
    m2, v2 = sq.predict(past[0], _chol(past[1]), mdl["mean_state"][1], mdl["wgt_state"][0], _chol(mdl["var_state"][1]))
    _close(pred[0], m2); _close(pred[1], _sqr(v2))
    m3, v3 = sq.update(pred[0], _chol(pred[1]), mdl["x_meas"][1], mdl["mean_meas"][1], mdl["wgt_meas"][1],
                       _chol(mdl["var_meas"][1]))
    _close(filt[0], m3); _close(filt[1], _sqr(v3))
    t = jg.smooth_targets(mdl)
    args = dict(mean_state_filt=t["filt"][0], var_state_filt=_chol(t["filt"][1]), mean_state_pred=t["pred"][0],
                var_state_pred=_chol(t["pred"][1]), wgt_state=mdl["wgt_state"][0], var_state=_chol(mdl["var_state"][1]))
    mm, vm = sq.smooth_mv(mean_state_next=t["next"][0], var_state_next=_chol(t["next"][1]), **args)
    _close(t["smooth"][0], mm); _close(t["smooth"][1], _sqr(vm))
    ms, vs = sq.smooth_sim(x_state_next=mdl["x_state_next"], **args)
    _close(t["sim"][0], ms); _close(t["sim"][1], _sqr(vs))
    A2, b2, V2 = sq.smooth_cond(**args)
    A, b, V = t["cond"]
    _close(A, A2); _close(b, b2); _close(V, _sqr(V2))
    f = sq.forecast(pred[0], _chol(pred[1]), mdl["mean_meas"][1], mdl["wgt_meas"][1], _chol(mdl["var_meas"][1]))
    of = oktv.forecast(pred[0], pred[1], mdl["mean_meas"][1], mdl["wgt_meas"][1], mdl["var_meas"][1])
    _close(of[0], f[0]); _close(of[1], f[1])


def test_square_root_vs_oracle_batch(sq):
    r = _rand_batch(np.random.default_rng(77), 70, 4, 2)
    Ls, LR, LV = _chol(r["S"]), _chol(r["R"]), _chol(r["V"])
    mp, vp = sq.predict(r["mu"], Ls, r["c"], r["Q"], LR)
    omp, ovp = osq.predict(r["mu"], Ls, r["c"], r["Q"], LR)
    np.testing.assert_allclose(mp, omp, rtol=1e-12, atol=1e-12); np.testing.assert_allclose(_sqr(vp), _sqr(ovp), rtol=1e-10, atol=1e-11)
    mf, vf = sq.update(omp, ovp, r["x"], r["a"], r["W"], LV)
    omf, ovf = osq.update(omp, ovp, r["x"], r["a"], r["W"], LV)
    np.testing.assert_allclose(mf, omf, rtol=1e-9, atol=1e-10); np.testing.assert_allclose(_sqr(vf), _sqr(ovf), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("name", ["rodeo", "schober", "kramer", "chkrebtii"])
@pytest.mark.parametrize("rhs", ["fitzhugh_nagumo", "lorenz63"])
def test_standalone_interrogations_builtin_rhs(name, rhs):
    """rk_interrogate_batched for each of the four interrogations with a built-in right-hand side (src/rodeo/
    interrogate.py:13-115): all three outputs (wgt_meas, mean_meas, var_meas) against the oracle on a random batch; a
    single (un-batched) call returns the reference's shapes.  The chkrebtii draw goes through the shared Philox stream,
    addressed by (seed, step, traj_offset) exactly as the fused solver addresses it."""
    import functools
    import rodeo_amd as ra
    from oracle import odes, interrogations as oi
    rng = np.random.default_rng(77)
    d, p, B = (2, 3, 6) if rhs == "fitzhugh_nagumo" else (3, 4, 5)
    fun, ofun = getattr(ra.ode, rhs), getattr(odes, rhs)
    theta0 = np.array([0.2, 0.2, 3.0]) if rhs == "fitzhugh_nagumo" else np.array([28.0, 10.0, 8.0 / 3.0])
    theta = theta0 * np.exp(0.05 * rng.standard_normal((B, 3)))
    W, _ = ra.utils.first_order_pad(fun, d, p)
    mp = rng.standard_normal((B, d, p))
    a = rng.standard_normal((B, d, p, p))
    vp = a @ np.swapaxes(a, -1, -2) + 0.1 * np.eye(p)
    g = getattr(ra.interrogate, "interrogate_" + name)
    o = getattr(oi, "interrogate_" + name)
    if name == "chkrebtii":
        seed, step, off = 1234, 7, 3
        g = functools.partial(g, kalman_type="standard")
        got = g((seed, step, off), fun, W, 0.3, mp, vp, theta=theta)
        ref = o(oi.StepKey(seed, off + np.arange(B), step), ofun, W, 0.3, mp, vp, kalman_type="standard", theta=theta)
        tol = 1e-9                                      # sqrt / Box-Muller on the device vs NumPy
    else:
        got = g(None, fun, W, 0.3, mp, vp, theta=theta)
        ref = o(None, ofun, W, 0.3, mp, vp, theta=theta)
        tol = 1e-12
    for x, y, shape in zip(got, ref, [(B, d, 1, p), (B, d, 1), (B, d, 1, 1)]):
        assert x.shape == shape
        np.testing.assert_allclose(x, np.broadcast_to(y, shape), rtol=tol, atol=tol)
    if name != "chkrebtii":
        one = g(None, fun, W, 0.3, mp[0], vp[0], theta=theta[0])
        assert [x.shape for x in one] == [(d, 1, p), (d, 1), (d, 1, 1)]
        for x, y in zip(one, got):
            np.testing.assert_array_equal(x, y[0])


# ---- blocks beyond the lane-per-item kernels (n_state > 16): one workgroup per item on the dense building blocks ----------
@pytest.mark.parametrize("p,m", [(20, 3), (48, 16), (160, 32), (176, 8)])
def test_standard_large_blocks_vs_oracle(ktv, p, m):
    """The operators are size-agnostic in the reference (standard.py:31-60); at n_state > 16 they run on the dense solver's
    GEMM / pivoted LU blocks (solve_dense_ops.hpp), e.g. BASELINE config 5's 160 x 160 state with 32 measurements."""
    rng = np.random.default_rng(p + m)
    r = _rand_batch(rng, 3, p, m)
    r["Q"] = r["Q"] / np.sqrt(p)                                              # keep |Q| ~ 1 at any size
    mp, vp = ktv.predict(r["mu"], r["S"], r["c"], r["Q"], r["R"])
    omp, ovp = oktv.predict(r["mu"], r["S"], r["c"], r["Q"], r["R"])
    sv = np.abs(ovp).max()
    np.testing.assert_allclose(mp, omp, rtol=1e-11, atol=1e-11); np.testing.assert_allclose(vp, ovp, rtol=0, atol=1e-12 * sv)
    mf, vf = ktv.update(omp, ovp, r["x"], r["a"], r["W"], r["V"])
    omf, ovf = oktv.update(omp, ovp, r["x"], r["a"], r["W"], r["V"])
    np.testing.assert_allclose(mf, omf, rtol=0, atol=1e-9 * max(1.0, np.abs(omf).max()))
    np.testing.assert_allclose(vf, ovf, rtol=0, atol=1e-9 * sv)
    out = ktv.filter(r["mu"], r["S"], r["c"], r["Q"], r["R"], r["x"], r["a"], r["W"], r["V"])
    for g, o in zip(out, (omp, ovp, omf, ovf)):
        np.testing.assert_allclose(g, o, rtol=0, atol=1e-9 * max(1.0, np.abs(o).max()))
    fo = ktv.forecast(omp, ovp, r["a"], r["W"], r["V"])
    ofo = oktv.forecast(omp, ovp, r["a"], r["W"], r["V"])
    np.testing.assert_allclose(fo[0], ofo[0], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(fo[1], ofo[1], rtol=0, atol=1e-11 * np.abs(ofo[1]).max())
    sm = ktv.smooth(r["xn"], r["mn"], r["Sn"], omf, ovf, omp, ovp, r["Q"])
    osm = oktv.smooth(r["xn"], r["mn"], r["Sn"], omf, ovf, omp, ovp, r["Q"])
    for g, o in zip(sm, osm):
        np.testing.assert_allclose(g, o, rtol=0, atol=1e-8 * max(1.0, np.abs(o).max()))
    m2 = ktv.smooth_mv(r["mn"], r["Sn"], omf, ovf, omp, ovp, r["Q"])
    np.testing.assert_allclose(m2[0], osm[2], rtol=0, atol=1e-8 * max(1.0, np.abs(osm[2]).max()))
    m3 = ktv.smooth_sim(r["xn"], omf, ovf, omp, ovp, r["Q"])
    np.testing.assert_allclose(m3[1], osm[1], rtol=0, atol=1e-8 * max(1.0, np.abs(osm[1]).max()))
    sc = ktv.smooth_cond(omf, ovf, omp, ovp, r["Q"])
    osc = oktv.smooth_cond(omf, ovf, omp, ovp, r["Q"])
    for g, o in zip(sc, osc):
        np.testing.assert_allclose(g, o, rtol=0, atol=1e-8 * max(1.0, np.abs(o).max()))


@pytest.mark.parametrize("p,m", [(20, 3), (48, 16), (160, 32)])
def test_square_root_large_blocks_vs_oracle(sq, p, m):
    """Square-root operators at n_state > 16 (blocked Householder QR, triangular solves; solve_dense_sqrt.hpp): factors compared
    as L L^T, forecast as the full variance (square_root.py:343-344)."""
    rng = np.random.default_rng(100 + p + m)
    r = _rand_batch(rng, 2, p, m)
    r["Q"] = r["Q"] / np.sqrt(p)
    LS, LR, LV, LSn = _chol(r["S"]), _chol(r["R"]), _chol(r["V"]), _chol(r["Sn"])
    def close_f(L, Lo, tol):
        vo = _sqr(Lo)
        np.testing.assert_allclose(_sqr(L), vo, rtol=0, atol=tol * np.abs(vo).max())
    mp, Lp = sq.predict(r["mu"], LS, r["c"], r["Q"], LR)
    omp, oLp = osq.predict(r["mu"], LS, r["c"], r["Q"], LR)
    np.testing.assert_allclose(mp, omp, rtol=1e-11, atol=1e-11); close_f(Lp, oLp, 1e-11)
    assert np.all(np.triu(Lp, 1) == 0.0)
    mf, Lf = sq.update(omp, oLp, r["x"], r["a"], r["W"], LV)
    omf, oLf = osq.update(omp, oLp, r["x"], r["a"], r["W"], LV)
    np.testing.assert_allclose(mf, omf, rtol=0, atol=1e-9 * max(1.0, np.abs(omf).max())); close_f(Lf, oLf, 1e-9)
    fo = sq.forecast(omp, oLp, r["a"], r["W"], LV)
    ofo = osq.forecast(omp, oLp, r["a"], r["W"], LV)
    np.testing.assert_allclose(fo[0], ofo[0], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(fo[1], ofo[1], rtol=0, atol=1e-10 * np.abs(ofo[1]).max())
    sm = sq.smooth(r["xn"], r["mn"], LSn, omf, oLf, omp, oLp, r["Q"], LR)
    osm = osq.smooth(r["xn"], r["mn"], LSn, omf, oLf, omp, oLp, r["Q"], LR)
    for k in (0, 2):
        np.testing.assert_allclose(sm[k], osm[k], rtol=0, atol=1e-8 * max(1.0, np.abs(osm[k]).max()))
    close_f(sm[1], osm[1], 1e-8); close_f(sm[3], osm[3], 1e-8)
    A, bb, C = sq.smooth_cond(omf, oLf, omp, oLp, r["Q"], LR)
    oA, obb, oC = osq.smooth_cond(omf, oLf, omp, oLp, r["Q"], LR)
    np.testing.assert_allclose(A, oA, rtol=0, atol=1e-8 * max(1.0, np.abs(oA).max()))
    np.testing.assert_allclose(bb, obb, rtol=0, atol=1e-8 * max(1.0, np.abs(obb).max())); close_f(C, oC, 1e-8)
