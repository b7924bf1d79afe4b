"""
Committed fixtures (tests/golden/README.md): the published Philox known-answer vectors, the SURVEY's regression
anchors, and a small oracle regression vector -- checked against the oracle on the CPU and, marked gpu, against the
device through the C ABI.
"""
import json, os
import numpy as np
import pytest
from scipy.integrate import odeint
from oracle import scan, odes, priors, interrogations as oi, counter_rng as cr

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = np.load(os.path.join(HERE, "oracle_fn_small.npz"))
ITG = {"kramer": oi.interrogate_kramer, "rodeo": oi.interrogate_rodeo, "schober": oi.interrogate_schober}


def test_philox_known_answer_vectors_from_file():
    kat = json.load(open(os.path.join(HERE, "philox4x32_10_kat.json")))
    for v in kat["vectors"]:
        c, k = [int(x, 16) for x in v["counter"]], [int(x, 16) for x in v["key"]]
        out = cr.philox4x32_10(*[np.uint32(x) for x in c], *[np.uint32(x) for x in k])
        assert [int(x) for x in out] == [int(x, 16) for x in v["expected"]]


@pytest.mark.parametrize("name", ["kramer", "rodeo", "schober"])
def test_oracle_reproduces_the_committed_vector(name):
    args = (G["W"], G["x0"], 0.0, float(G["t_max"]), int(G["N"]))
    m, v = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, ITG[name], (G["Q"], G["R"]), theta=G["theta"])
    np.testing.assert_allclose(m, G[f"mv_mean_{name}"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(v, G[f"mv_var_{name}"], rtol=1e-10, atol=1e-22)
    # the prior of the fixture is the IBM closed form (K6)
    Q, R = priors.ibm_init(float(G["t_max"]) / int(G["N"]), 3, np.array([0.1, 0.1]))
    np.testing.assert_array_equal(Q, G["Q"]); np.testing.assert_array_equal(R, G["R"])


def test_anchors_k4_and_k3():
    anc = json.load(open(os.path.join(HERE, "anchors.json")))
    W = np.array([[[0.0, 0.0, 1.0, 0.0]]])
    x0 = np.array([[-1.0, 0.0, 1.0, 0.0]])
    for n_str, err in anc["K4_higher_order_kramer_sigma_0.001_max_abs_error"].items():
        N = int(n_str)
        prior = priors.ibm_init(10.0 / N, 4, np.array([0.001]))
        m, _ = scan.solve_mv(None, odes.higher_order, W, x0, 0.0, 10.0, N, oi.interrogate_kramer, prior)
        got = np.max(np.abs(m[:, 0, 0] - odes.higher_order_exact(np.linspace(0, 10, N + 1))))
        assert abs(got - err) < 0.02 * err, (N, got, err)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kramer", "rodeo", "schober"])
def test_device_reproduces_the_committed_vector(name):
    import rodeo_amd as ra
    g = {"kramer": ra.interrogate.interrogate_kramer, "rodeo": ra.interrogate.interrogate_rodeo,
         "schober": ra.interrogate.interrogate_schober}[name]
    args = (G["W"], G["x0"], 0.0, float(G["t_max"]), int(G["N"]))
    m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, g, (G["Q"], G["R"]), theta=G["theta"])
    np.testing.assert_allclose(m, G[f"mv_mean_{name}"], rtol=0, atol=1e-10)
    assert np.max(np.abs(v - G[f"mv_var_{name}"])) <= 1e-9 * np.max(np.abs(G[f"mv_var_{name}"]))
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, *args, g, (G["Q"], G["R"]), theta=G["theta"])
    plan.filter(None)
    mf, vf = plan.state_host()
    np.testing.assert_allclose(mf, G[f"filt_mean_{name}"], rtol=0, atol=1e-10)
    assert np.max(np.abs(vf - G[f"filt_var_{name}"])) <= 1e-9 * np.max(np.abs(G[f"filt_var_{name}"]))
    if name == "rodeo":
        x = ra.solve_sim(5, ra.ode.fitzhugh_nagumo, *args, g, (G["Q"], G["R"]), theta=G["theta"])
        np.testing.assert_allclose(x, G["sim_rodeo_seed5"], rtol=0, atol=1e-8)


# ---- round 3 fixture: square-root form, n_deriv = 5, a dense block in both forms, fenrir in both forms ---------------------------
G3 = np.load(os.path.join(HERE, "oracle_round3.npz"))


def _sq(L):
    return L @ np.swapaxes(L, -1, -2)


def test_oracle_reproduces_the_round3_vector():
    from oracle import fenrir as ofen
    args = (G["W"], G["x0"], 0.0, float(G["t_max"]), int(G["N"]))
    cholR = np.linalg.cholesky(G["R"])
    for name in ("kramer", "rodeo"):
        m, L = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, ITG[name], (G["Q"], cholR), kalman_type="square-root", theta=G["theta"])
        np.testing.assert_allclose(m, G3[f"sqrt_mv_mean_{name}"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(_sq(L), G3[f"sqrt_mv_var_{name}"], rtol=1e-9, atol=1e-22)
    m5, v5 = scan.solve_mv(None, odes.fitzhugh_nagumo, G3["W5"], G3["x05"], 0.0, float(G["t_max"]), int(G["N"]), oi.interrogate_kramer,
                           (G3["Q5"], G3["R5"]), theta=G["theta"])
    np.testing.assert_allclose(m5, G3["mv5_mean"], rtol=1e-9, atol=1e-9)
    ode_d = odes.make_linear_dense(G3["dense_A"], 3)
    dargs = (G3["dense_W"], G3["dense_x0"], 0.0, float(G3["dense_t"]), int(G3["dense_N"]))
    md, vd = scan.solve_mv(None, ode_d, *dargs, oi.interrogate_kramer, (G3["dense_Q"], G3["dense_R"]))
    np.testing.assert_allclose(md, G3["dense_mv_mean"], rtol=1e-10, atol=1e-12)
    # the two filter forms of the fixture agree with each other (exact measurement: the same posterior)
    assert np.max(np.abs(G3["dense_mv_var"] - G3["dense_sqrt_mv_var"])) < 1e-6 * np.max(np.abs(G3["dense_mv_var"]))
    ll = ofen.fenrir(None, odes.fitzhugh_nagumo, *args, oi.interrogate_kramer, (G["Q"], G["R"]), G3["fen_y"], G3["fen_obs_t"], G3["fen_D"],
                     G3["fen_Om"], theta=G["theta"])
    assert abs(ll - float(G3["fen_ll_standard"])) < 1e-9 and abs(float(G3["fen_ll_standard"]) - float(G3["fen_ll_sqrt"])) < 1e-8


@pytest.mark.gpu
def test_device_reproduces_the_round3_vector():
    import rodeo_amd as ra
    args = (G["W"], G["x0"], 0.0, float(G["t_max"]), int(G["N"]))
    cholR = np.linalg.cholesky(G["R"])
    for name, g in (("kramer", ra.interrogate.interrogate_kramer), ("rodeo", ra.interrogate.interrogate_rodeo)):
        m, L = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, g, (G["Q"], cholR), kalman_type="square-root", theta=G["theta"])
        np.testing.assert_allclose(m, G3[f"sqrt_mv_mean_{name}"], rtol=0, atol=1e-9)
        assert np.max(np.abs(_sq(L) - G3[f"sqrt_mv_var_{name}"])) <= 1e-7 * np.max(np.abs(G3[f"sqrt_mv_var_{name}"]))
    # n_deriv = 5: blocked tiles, one-kernel backward pass
    m5, v5 = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, G3["W5"], G3["x05"], 0.0, float(G["t_max"]), int(G["N"]),
                         ra.interrogate.interrogate_kramer, (G3["Q5"], G3["R5"]), theta=G["theta"])
    scale = np.maximum(np.max(np.abs(G3["mv5_mean"]), axis=(0, 1)), 1.0)
    assert np.max(np.abs(m5 - G3["mv5_mean"]) / scale) < 1e-8
    assert np.max(np.abs(v5 - G3["mv5_var"])) <= 1e-7 * np.max(np.abs(G3["mv5_var"]))
    # the dense block in both forms
    ode_d = ra.ode.linear_dense(4, 3)
    dargs = (G3["dense_W"], G3["dense_x0"], 0.0, float(G3["dense_t"]), int(G3["dense_N"]))
    md, vd = ra.solve_mv(None, ode_d, *dargs, ra.interrogate.interrogate_kramer, (G3["dense_Q"], G3["dense_R"]), A=G3["dense_A"])
    dscale = np.maximum(np.max(np.abs(G3["dense_mv_mean"]), axis=(0, 1)), 1.0)           # per state component (derivatives grow)
    assert np.max(np.abs(md - G3["dense_mv_mean"]) / dscale) < 1e-9
    assert np.max(np.abs(vd - G3["dense_mv_var"])) <= 1e-7 * np.max(np.abs(G3["dense_mv_var"]))
    ms, Ls = ra.solve_mv(None, ode_d, *dargs, ra.interrogate.interrogate_kramer, (G3["dense_Q"], np.linalg.cholesky(G3["dense_R"])),
                         kalman_type="square-root", A=G3["dense_A"])
    assert np.max(np.abs(ms - G3["dense_sqrt_mv_mean"]) / dscale) < 1e-9
    assert np.max(np.abs(_sq(Ls) - G3["dense_sqrt_mv_var"])) <= 1e-7 * np.max(np.abs(G3["dense_sqrt_mv_var"]))
    # fenrir in both forms
    fa = (G3["fen_y"], G3["fen_obs_t"], G3["fen_D"])
    ll = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, (G["Q"], G["R"]), *fa, G3["fen_Om"],
                             theta=G["theta"])
    assert abs(ll - float(G3["fen_ll_standard"])) < 1e-6
    lls = ra.inference.fenrir(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, (G["Q"], cholR), *fa,
                              np.sqrt(G3["fen_Om"]), kalman_type="square-root", theta=G["theta"])
    assert abs(lls - float(G3["fen_ll_sqrt"])) < 1e-6
