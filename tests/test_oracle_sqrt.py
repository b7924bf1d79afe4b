"""
Pins oracle/sqrt_ops.py: K5 (tests/test_add_sqrt.py:11-20: add_sqrt squares to A + B) and K1 with inputs
Cholesky-factored and outputs compared as L L^T (tests/test_square_root.py:11-16).  The reference's own
tolerances there are 7 places for means, and only 2 places for the smooth_sim / smooth / smooth_cond variances
(test_square_root.py:120,138,162); exact arithmetic gives equality, so the tight 5e-8 is used throughout.
"""
import numpy as np
import pytest
from oracle import sqrt_ops as sq
from oracle import joint_gaussian as jg

TOL = 5e-8


def _chol(A):
    return np.linalg.cholesky(A)


def _sq(L):
    return L @ np.swapaxes(L, -1, -2)


@pytest.fixture(params=range(6))
def mdl(request):
    return jg.random_model(np.random.default_rng(2000 + request.param))


def test_add_sqrt():
    rng = np.random.default_rng(0)
    for n in (1, 2, 5):
        a = rng.standard_normal((n, n)); b = rng.standard_normal((n, n))
        A, B = a @ a.T, b @ b.T
        s = sq.add_sqrt(_chol(A), _chol(B))
        np.testing.assert_allclose(s @ s.T, A + B, rtol=1e-12, atol=1e-12)
    # rectangular inputs as used by square_root.py:217
    a = rng.standard_normal((3, 6)); b = rng.standard_normal((3, 3))
    s = sq.add_sqrt(a, b)
    np.testing.assert_allclose(s @ s.T, a @ a.T + b @ b.T, rtol=1e-12, atol=1e-12)


def test_predict_update(mdl):
    past, pred, filt = jg.filter_targets(mdl, 1)
    m2, v2 = sq.predict(past[0], _chol(past[1]), mdl["mean_state"][1], mdl["wgt_state"][0], _chol(mdl["var_state"][1]))
    assert jg.rel_err(pred[0], m2) < TOL and jg.rel_err(pred[1], _sq(v2)) < TOL
    m3, v3 = sq.update(pred[0], _chol(pred[1]), mdl["x_meas"][1], mdl["mean_meas"][1], mdl["wgt_meas"][1],
                       _chol(mdl["var_meas"][1]))
    assert jg.rel_err(filt[0], m3) < TOL and jg.rel_err(filt[1], _sq(v3)) < TOL


def test_smoothers(mdl):
    t = jg.smooth_targets(mdl)
    R = _chol(mdl["var_state"][1])
    args = dict(mean_state_filt=t["filt"][0], var_state_filt=_chol(t["filt"][1]), mean_state_pred=t["pred"][0],
                var_state_pred=_chol(t["pred"][1]), wgt_state=mdl["wgt_state"][0], var_state=R)
    mm, vm = sq.smooth_mv(mean_state_next=t["next"][0], var_state_next=_chol(t["next"][1]), **args)
    assert jg.rel_err(t["smooth"][0], mm) < TOL and jg.rel_err(t["smooth"][1], _sq(vm)) < TOL
    ms, vs = sq.smooth_sim(x_state_next=mdl["x_state_next"], **args)
    assert jg.rel_err(t["sim"][0], ms) < TOL and jg.rel_err(t["sim"][1], _sq(vs)) < TOL
    A2, b2, V2 = sq.smooth_cond(**args)
    A, b, V = t["cond"]
    assert jg.rel_err(A, A2) < TOL and jg.rel_err(b, b2) < TOL and jg.rel_err(V, _sq(V2)) < TOL
    out = sq.smooth(x_state_next=mdl["x_state_next"], mean_state_next=t["next"][0],
                    var_state_next=_chol(t["next"][1]), **args)
    assert jg.rel_err(t["sim"][1], _sq(out[1])) < TOL and jg.rel_err(t["smooth"][1], _sq(out[3])) < TOL


def test_forecast_returns_full_variance(mdl):
    _, pred, _ = jg.filter_targets(mdl, 1)
    from oracle import kalman_ops as ktv
    m1, v1 = ktv.forecast(pred[0], pred[1], mdl["mean_meas"][1], mdl["wgt_meas"][1], mdl["var_meas"][1])
    m2, v2 = sq.forecast(pred[0], _chol(pred[1]), mdl["mean_meas"][1], mdl["wgt_meas"][1], _chol(mdl["var_meas"][1]))
    assert jg.rel_err(m1, m2) < TOL and jg.rel_err(v1, v2) < TOL
