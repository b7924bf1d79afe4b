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
