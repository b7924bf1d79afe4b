"""
Pins oracle/kalman_ops.py with the reference's own test oracle K1 (joint-Gaussian conditioning), the way
tests/test_standard.py:11-200 pins src/rodeo/kalmantv/standard.py.  Tolerance: the reference asserts
assertAlmostEqual(rel_err, 0.0) = 7 places (5e-8) in its rel_err metric (tests/utils.py:11-18).
"""
import numpy as np
import pytest
from oracle import kalman_ops as ktv
from oracle import joint_gaussian as jg

TOL = 5e-8
SEEDS = [0, 1, 2, 3, 4, 5, 6, 7]


@pytest.fixture(params=SEEDS)
def mdl(request):
    return jg.random_model(np.random.default_rng(1000 + request.param))


def _close(a, b):
    assert jg.rel_err(a, b) < TOL


def test_predict(mdl):
    past, pred, _ = jg.filter_targets(mdl, 1)
    m2, v2 = ktv.predict(mean_state_past=past[0], var_state_past=past[1], mean_state=mdl["mean_state"][1],
                         wgt_state=mdl["wgt_state"][0], var_state=mdl["var_state"][1])
    _close(pred[0], m2); _close(pred[1], v2)


def test_update(mdl):
    _, pred, filt = jg.filter_targets(mdl, 1)
    m2, v2 = ktv.update(mean_state_pred=pred[0], var_state_pred=pred[1], x_meas=mdl["x_meas"][1],
                        mean_meas=mdl["mean_meas"][1], wgt_meas=mdl["wgt_meas"][1], var_meas=mdl["var_meas"][1])
    _close(filt[0], m2); _close(filt[1], v2)


@pytest.mark.parametrize("step", [1, 2])
def test_filter(mdl, step):
    past, pred, filt = jg.filter_targets(mdl, step)
    mp, vp, mf, vf = ktv.filter(mean_state_past=past[0], var_state_past=past[1],
                                mean_state=mdl["mean_state"][step], wgt_state=mdl["wgt_state"][step - 1],
                                var_state=mdl["var_state"][step], x_meas=mdl["x_meas"][step],
                                mean_meas=mdl["mean_meas"][step], wgt_meas=mdl["wgt_meas"][step],
                                var_meas=mdl["var_meas"][step])
    _close(pred[0], mp); _close(pred[1], vp); _close(filt[0], mf); _close(filt[1], vf)


def test_forecast(mdl):
    # y_1 | y_0 from the joint: condition on y_0, read the y_1 marginal
    _, pred, _ = jg.filter_targets(mdl, 1)
    mf, vf = ktv.forecast(mean_state_pred=pred[0], var_state_pred=pred[1], mean_meas=mdl["mean_meas"][1],
                          wgt_meas=mdl["wgt_meas"][1], var_meas=mdl["var_meas"][1])
    mu, S = mdl["mean_gm"], mdl["var_gm"]
    n_tot, n_dim = mu.shape
    ns = mdl["n_state"]
    icond = np.zeros((n_tot, n_dim), bool); icond[0, ns:] = True
    A, b, V = jg.mvncond(mu.ravel(), S.reshape(n_tot * n_dim, -1), icond.ravel())
    imarg = np.zeros((n_tot, n_dim), bool); imarg[1, ns:] = True
    imarg = imarg.ravel()[~icond.ravel()]
    _close((A.dot(mdl["x_meas"][0]) + b)[imarg], mf)
    _close(V[np.ix_(imarg, imarg)], vf)


def test_smooth_mv(mdl):
    t = jg.smooth_targets(mdl)
    m2, v2 = ktv.smooth_mv(mean_state_next=t["next"][0], var_state_next=t["next"][1],
                           mean_state_filt=t["filt"][0], var_state_filt=t["filt"][1],
                           mean_state_pred=t["pred"][0], var_state_pred=t["pred"][1],
                           wgt_state=mdl["wgt_state"][0])
    _close(t["smooth"][0], m2); _close(t["smooth"][1], v2)


def test_smooth_sim(mdl):
    t = jg.smooth_targets(mdl)
    m2, v2 = ktv.smooth_sim(x_state_next=mdl["x_state_next"], mean_state_filt=t["filt"][0],
                            var_state_filt=t["filt"][1], mean_state_pred=t["pred"][0],
                            var_state_pred=t["pred"][1], wgt_state=mdl["wgt_state"][0])
    _close(t["sim"][0], m2); _close(t["sim"][1], v2)


def test_smooth(mdl):
    t = jg.smooth_targets(mdl)
    ms, vs, mm, vm = ktv.smooth(x_state_next=mdl["x_state_next"], mean_state_next=t["next"][0],
                                var_state_next=t["next"][1], mean_state_filt=t["filt"][0],
                                var_state_filt=t["filt"][1], mean_state_pred=t["pred"][0],
                                var_state_pred=t["pred"][1], wgt_state=mdl["wgt_state"][0])
    _close(t["sim"][0], ms); _close(t["sim"][1], vs); _close(t["smooth"][0], mm); _close(t["smooth"][1], vm)


def test_smooth_cond(mdl):
    t = jg.smooth_targets(mdl)
    A2, b2, V2 = ktv.smooth_cond(mean_state_filt=t["filt"][0], var_state_filt=t["filt"][1],
                                 mean_state_pred=t["pred"][0], var_state_pred=t["pred"][1],
                                 wgt_state=mdl["wgt_state"][0])
    A, b, V = t["cond"]
    _close(A, A2); _close(b, b2); _close(V, V2)


def test_ops_broadcast_over_batch():
    """Leading batch dims give the same numbers as looping (the reference's vmap semantics)."""
    rng = np.random.default_rng(7)
    B, p, m = 5, 4, 2
    mu = rng.standard_normal((B, p)); a = rng.standard_normal((B, p, p)); S = a @ np.swapaxes(a, -1, -2)
    Q = rng.standard_normal((p, p)); r = rng.standard_normal((p, p)); R = r @ r.T
    W = rng.standard_normal((B, m, p)); v = rng.standard_normal((m, m)); V = v @ v.T
    mp, vp = ktv.predict(mu, S, np.zeros(p), Q, R)
    mf, vf = ktv.update(mp, vp, np.zeros(m), rng.standard_normal((B, m)) * 0 + 1.0, W, V)
    for i in range(B):
        mp1, vp1 = ktv.predict(mu[i], S[i], np.zeros(p), Q, R)
        mf1, vf1 = ktv.update(mp1, vp1, np.zeros(m), np.ones(m), W[i], V)
        np.testing.assert_allclose(mp[i], mp1, rtol=1e-13); np.testing.assert_allclose(vp[i], vp1, rtol=1e-13)
        np.testing.assert_allclose(mf[i], mf1, rtol=1e-12); np.testing.assert_allclose(vf[i], vf1, rtol=1e-10, atol=1e-13)


def test_kwargs_are_swallowed():
    """standard.py:36 -- ops ignore unknown kwargs (solve.py:177,271 rely on it)."""
    p = 3
    out = ktv.smooth_mv(np.zeros(p), np.eye(p), np.zeros(p), np.eye(p), np.zeros(p), 2 * np.eye(p), np.eye(p),
                        var_state=np.eye(p), anything=1)
    assert out[0].shape == (p,)
