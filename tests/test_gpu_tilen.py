"""
GPU parity of the BLOCKED MFMA-tile path (rodeo_amd/csrc/solve_tilen*.h*): n_bstate = 4 .. 8 -- the reference treats
n_deriv as a free argument (src/rodeo/solve.py:48, prior/ibm.py:65-88) -- for solve_mv, the filter, solve_sim and all four
interrogations, against the NumPy oracle on identical seeded inputs, through the C ABI.

Tolerances.  The IBM prior's conditioning worsens quickly with n_deriv: a 1e-13 relative perturbation of the prior
variance moves the ORACLE's own smoothed means by 4e-11 (p = 5) .. 5e-10 (p = 8) and its variances by up to 8e-8 (p = 8,
kramer) on the problems below, i.e. rounding differences between two correct implementations are amplified 1e3 .. 1e6
times.  The bounds are therefore stated per n_deriv (TOL_MEAN / TOL_VAR, relative to the scale of each derivative order
/ of the largest variance); with p = 7, 8 the covariance-form recursion of the reference is itself unstable on longer
horizons (the oracle overflows on FitzHugh-Nagumo beyond t = 1 at dt = 0.02), so those use a short horizon.
"""
import functools
import numpy as np
import pytest
from oracle import scan, odes, priors, interrogations as oi

pytestmark = pytest.mark.gpu

TOL_MEAN = {4: 1e-9, 5: 1e-8, 6: 1e-7, 7: 1e-6, 8: 1e-5}
TOL_VAR = {4: 1e-8, 5: 1e-7, 6: 1e-6, 7: 1e-5, 8: 1e-4}


@pytest.fixture(scope="module")
def ra():
    import rodeo_amd
    return rodeo_amd


def _itg(ra, name):
    g, o = getattr(ra.interrogate, "interrogate_" + name), getattr(oi, "interrogate_" + name)
    if name == "chkrebtii":
        g, o = functools.partial(g, kalman_type="standard"), functools.partial(o, kalman_type="standard")
    return g, o


def _fitz(ra, p, B=None, seed=0, N=50, t_max=0.5, sigma=0.1):
    theta = np.array([0.2, 0.2, 3.0])
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)
    x0v = np.array([-1., 1.])
    if B is not None:
        rng = np.random.default_rng(seed)
        theta = theta * np.exp(0.1 * rng.standard_normal((B, 3)))
        x0v = x0v + 0.1 * rng.standard_normal((B, 2))
    x0 = init(x0v, 0.0, theta=theta)
    return dict(W=W, x0=x0, theta=theta, prior=ra.ibm_init(t_max / N, p, np.array([sigma] * 2)), N=N, t_max=t_max)


def _close(m, mo, v, vo, p, R=None, entrywise=True, yard=None):
    """Means relative to the scale of each derivative order.  Variances TWICE: against the largest variance (dominated by the
    highest derivatives) and ENTRY BY ENTRY against sd_i sd_j of the two components involved (sd = the component's largest
    posterior standard deviation over the run) -- so that an error in Sigma_00 or Sigma_01, the entries a user reads, is
    measured on their own scale and cannot hide under the high-derivative variances."""
    scale = np.maximum(np.max(np.abs(mo), axis=tuple(range(mo.ndim - 1))), 1.0)        # per derivative order
    em = np.max(np.abs(m - mo) / scale)
    ev = np.max(np.abs(v - vo)) / np.max(np.abs(vo))
    sd = np.sqrt(np.abs(np.einsum("...ii->...i", vo)).max(axis=tuple(range(vo.ndim - 2))))
    if R is not None:       # a component pinned by an exact measurement (schober: the measured derivative) has variance 0 + rounding
        sd = np.maximum(sd, 1e-4 * np.sqrt(np.einsum("...ii->...i", np.asarray(R)).max(axis=tuple(range(np.ndim(R) - 2)))))
    ee = np.max(np.abs(v - vo) / (sd[:, None] * sd[None, :]))
    assert em < TOL_MEAN[p] and ev < TOL_VAR[p], (p, em, ev, ee)
    if entrywise and not ee < 10 * TOL_VAR[p]:
        # beyond the entry-wise bound: legitimate only if the fp64 ORACLE is as far from extended precision (conditioning)
        assert yard is not None, (p, em, ev, ee)
        ml, vl = yard()
        e_dev, e_orc = np.max(np.abs(v - vl) / (sd[:, None] * sd[None, :])), np.max(np.abs(vo - vl) / (sd[:, None] * sd[None, :]))
        assert e_dev <= 20 * e_orc + 1e-12, (p, ee, float(e_dev), float(e_orc))
    return em, ev


def _longdouble_solve_mv(s, name):
    """solve_mv of the FitzHugh-Nagumo problem `s` restated in np.longdouble (x87 extended precision: 64-bit mantissa): the
    recursion of src/rodeo/solve.py:47-122, 257-301 with standard.py:57-59, 93-102, 175-176, 210-216 and an LU with partial
    pivoting for standard.py:176, all in plain loops -- the yardstick that tells conditioning from defects: a correct fp64
    implementation must sit about as far from it as the fp64 oracle does."""
    LD = np.longdouble
    W, x0, th = s["W"].astype(LD), s["x0"].astype(LD), np.asarray(s["theta"], dtype=LD)
    Q, R = (a.astype(LD) for a in s["prior"])
    N, t_max = s["N"], s["t_max"]
    B, d, p = x0.shape

    def lu_solve(A, Bm):
        A, Bm = A.copy(), Bm.copy()
        n = A.shape[0]
        for k in range(n):
            pk = k + int(np.argmax(np.abs(A[k:, k])))
            if pk != k:
                A[[k, pk]] = A[[pk, k]]; Bm[[k, pk]] = Bm[[pk, k]]
            for i in range(k + 1, n):
                l = A[i, k] / A[k, k]
                A[i, k:] -= l * A[k, k:]; Bm[i] -= l * Bm[k]
        for k in range(n - 1, -1, -1):
            Bm[k] = (Bm[k] - A[k, k + 1:] @ Bm[k + 1:]) / A[k, k]
        return Bm
    ms = np.zeros((B, N + 1, d, p), dtype=LD); vs = np.zeros((B, N + 1, d, p, p), dtype=LD)
    for b in range(B):
        a_, b_, c_ = th[b]
        mu, Sig = x0[b].copy(), np.zeros((d, p, p), dtype=LD)
        mf = np.zeros((N + 1, d, p), dtype=LD); vf = np.zeros((N + 1, d, p, p), dtype=LD)
        mp = np.zeros_like(mf); vp = np.zeros_like(vf)
        mf[0] = mu
        for n in range(N):
            mup = np.stack([Q[k] @ mu[k] for k in range(d)])
            Sp = np.stack([Q[k] @ Sig[k] @ Q[k].T + R[k] for k in range(d)])
            V_, R_ = mup[0, 0], mup[1, 0]
            f = np.array([c_ * (V_ - V_ * V_ * V_ / 3 + R_), -(V_ - a_ + b_ * R_) / c_], dtype=LD)
            J = np.zeros((d, p), dtype=LD)
            if name == "kramer":
                J[0, 0] = c_ * (1 - V_ * V_); J[1, 0] = -b_ / c_
            for k in range(d):
                Wt = W[k, 0] - J[k]
                am = -f[k] + J[k] @ mup[k]
                WS = Wt @ Sp[k]
                Sm = WS @ Wt + (W[k, 0] @ Sp[k] @ W[k, 0] if name == "rodeo" else LD(0))
                K = (Sp[k] @ Wt) / Sm
                mu[k] = mup[k] + K * (LD(0) - (Wt @ mup[k] + am))
                Sig[k] = Sp[k] - np.outer(K, WS)
            mf[n + 1], vf[n + 1], mp[n + 1], vp[n + 1] = mu, Sig, mup, Sp
        ms[b, N], vs[b, N] = mf[N], vf[N]
        cm, cv = mf[N].copy(), vf[N].copy()
        for n in range(N - 1, 0, -1):
            for k in range(d):
                T = vf[n, k] @ Q[k].T
                G = lu_solve(vp[n + 1, k], T.T).T
                cm[k] = mf[n, k] + G @ (cm[k] - mp[n + 1, k])
                cv[k] = vf[n, k] + G @ (cv[k] - vp[n + 1, k]) @ G.T
            ms[b, n], vs[b, n] = cm, cv
        ms[b, 0] = x0[b]
    return ms, vs


@pytest.mark.parametrize("p,name", [(7, "kramer"), (8, "kramer"), (4, "schober"), (5, "schober"), (6, "schober"), (7, "schober"),
                                    (8, "schober")])
def test_gap_to_the_oracle_is_conditioning_not_a_defect(ra, p, name):
    """Where the fp64 bounds are loose -- n_deriv = 7, 8 (1e-6 .. 1e-4), and interrogate_schober at every n_deriv, whose exact
    measurement of x' leaves the smoother with singular filtered covariances (entry-wise errors of 1e-7 .. 1e-5 on the LOW
    derivatives' own scale, invisible next to the largest variance) -- the yardstick is an extended-precision restatement of
    the same recursion: the DEVICE is about as far from it as the fp64 ORACLE is (within a factor 20 plus a floor at rounding
    level), entry by entry on each entry's own scale.  What separates device and oracle is the problem's conditioning, not the
    blocked-tile arithmetic (bmm_tn, the rows chain)."""
    s = _fitz(ra, p, B=2, seed=p) if p >= 7 else _fitz(ra, p, B=2, seed=p, N=60, t_max=1.2)
    args = (s["W"], s["x0"], 0.0, s["t_max"], s["N"])
    g, o = _itg(ra, name)
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    plan.mv(None)
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    ml, vl = _longdouble_solve_mv(s, name)
    scale = np.maximum(np.max(np.abs(mo), axis=(0, 1, 2)), 1.0)
    sd = np.sqrt(np.abs(np.einsum("...ii->...i", vo)).max(axis=(0, 1, 2)))
    sd = np.maximum(sd, 1e-4 * np.sqrt(np.einsum("...ii->...i", s["prior"][1]).max(axis=0)))     # (pinned components)
    sv = sd[:, None] * sd[None, :]
    e_dev_m = float(np.max(np.abs(m - ml) / scale)); e_orc_m = float(np.max(np.abs(mo - ml) / scale))
    e_dev_v = float(np.max(np.abs(v - vl) / sv)); e_orc_v = float(np.max(np.abs(vo - vl) / sv))
    print(f"\nn_deriv = {p}, {name}: means  device-ld {e_dev_m:.2e}  oracle-ld {e_orc_m:.2e};  variances (per entry)  device-ld "
          f"{e_dev_v:.2e}  oracle-ld {e_orc_v:.2e}")
    assert e_dev_m <= 20 * e_orc_m + 1e-12 and e_dev_v <= 20 * e_orc_v + 1e-12, (e_dev_m, e_orc_m, e_dev_v, e_orc_v)


def _plan_layout(ra, p):
    from rodeo_amd import _lib
    return _lib.LAYOUT_TILE4 if p == 4 else _lib.LAYOUT_TILEP


@pytest.mark.parametrize("p", [4, 5, 6, 7, 8])
@pytest.mark.parametrize("name", ["kramer", "rodeo", "schober"])
def test_blocked_tiles_solve_mv_fitzhugh(ra, p, name):
    g, o = _itg(ra, name)
    s = _fitz(ra, p, B=5, seed=p) if p >= 7 else _fitz(ra, p, B=5, seed=p, N=60, t_max=1.2)
    args = (s["W"], s["x0"], 0.0, s["t_max"], s["N"])
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    plan.mv(None)
    m, v = plan.state_host()
    if p >= 5:                                          # p = 4 solve_mv has its own hand-trimmed kernels (solve_tile4.hip)
        assert plan.layout == _plan_layout(ra, p)
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    assert m.shape == (5, s["N"] + 1, 2, p) and v.shape == (5, s["N"] + 1, 2, p, p)
    # (schober's entry-wise errors are conditioning: test_gap_to_the_oracle_is_conditioning_not_a_defect holds its yardstick)
    _close(m, mo, v, vo, p, s["prior"][1], entrywise=name != "schober")
    np.testing.assert_array_equal(m[:, 0], s["x0"])
    assert np.all(v[:, 0] == 0)


@pytest.mark.parametrize("p", [5, 6])
def test_blocked_tiles_kernels_are_the_ones_that_run(ra, p):
    s = _fitz(ra, p, B=3, N=20, t_max=0.4)
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0.0, s["t_max"], s["N"], ra.interrogate.interrogate_kramer,
                        s["prior"], theta=s["theta"])
    plan.dev.profile_enable(True)
    plan.mv(None)
    names = [k for k, _ in plan.dev.profile_last()]
    plan.dev.profile_enable(False)
    assert names[0] == "fwd_tilen_kernel" and all("tilen" in k for k in names[1:]) and len(names) >= 2, names
    # keep=True: the launches of every call since it was switched on (what bench.py reads after its timed region)
    plan.dev.profile_enable(True, keep=True)
    for _ in range(3):
        plan.mv(None)
    kept = plan.dev.profile_last(cap=64)
    plan.dev.profile_enable(False)
    assert [k for k, _ in kept] == names * 3 and all(ms > 0 for _, ms in kept)
    plan.dev.profile_enable(True)
    plan.mv(None)
    assert [k for k, _ in plan.dev.profile_last()] == names
    plan.dev.profile_enable(False)


def test_blocked_tiles_chain_variants_agree(ra):
    """The two forms of the solve_mv chain (element-per-lane accesses; whole rows through LDS) are the same arithmetic:
    bit-identical results at every n_bstate (the library picks one per size; RK_TILEN_CHAIN forces it -- read once per
    process, so this test runs the other form in a child process)."""
    import subprocess, sys, os, json
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r); import rodeo_amd as ra\n"
        "out = {}\n"
        "for p in (4, 5, 6, 7, 8):\n"
        "    import functools\n"
        "    theta = np.array([0.2, 0.2, 3.0]); W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, p)\n"
        "    x0 = init(np.array([[-1., 1.], [-0.9, 1.1], [-1.1, 0.9]]), 0.0, theta=theta)\n"
        "    g = functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type='standard') if p == 4 else ra.interrogate.interrogate_kramer\n"
        "    m, v = ra.solve_mv(3, ra.ode.fitzhugh_nagumo, W, x0, 0.0, 0.5, 37, g, ra.ibm_init(0.5 / 37, p, np.array([.1, .1])), theta=theta)\n"
        "    out[str(p)] = [float(np.sum(m)), float(np.sum(np.abs(v))), float(m[1, 20, 1, 2])]\n"
        "print(json.dumps(out))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for form in ("rows", "scattered"):
        env = dict(os.environ, RK_TILEN_CHAIN=form, RK_TILEN_BWD="split")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[form] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["rows"] == res["scattered"], res


@pytest.mark.parametrize("p", [4, 5, 6, 7, 8])
@pytest.mark.parametrize("B,N", [(5, 37), (1, 70), (2, 2), (3, 6), (2, 9)])
def test_blocked_tiles_gain_variants_agree(ra, p, B, N, monkeypatch):
    """Three forms of solve_mv's backward pass on the blocked tiles (rodeo_amd/csrc/solve_tilen.hip) sum the same terms in the
    same order -- the smoothed means and variances are the same to the bit, at ragged unit counts (10 units = 2.5 waves),
    across the 32-step chunks of the column-per-lane gain kernel and the 4-step chunks of the fused kernel:
      split + lanes : tilen_gain_kernel (one lane per item)   + the chain kernel, records through HBM
      split + cols  : tilen_gain_cols_kernel (a 16-lane DPP row per item, a matrix column per lane) + the chain kernel
      fused         : bwd_mv_tilen_fused_kernel (the chain wave fed through LDS by two gain waves; the default at n_bstate >= 5)
    RK_TILEN_BWD / RK_TILEN_GAIN are read per call."""
    s = _fitz(ra, p, B=B, N=N, t_max=0.01 * N)
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0.0, s["t_max"], s["N"], ra.interrogate.interrogate_kramer,
                        s["prior"], theta=s["theta"])
    res = {}
    for form, env in (("lanes", dict(RK_TILEN_BWD="split", RK_TILEN_GAIN="lanes")),
                      ("cols", dict(RK_TILEN_BWD="split", RK_TILEN_GAIN="cols")), ("fused", dict(RK_TILEN_BWD="fused"))):
        monkeypatch.delenv("RK_TILEN_GAIN", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan.dev.profile_enable(True)
        plan.mv(None)
        m, v = plan.state_host()
        names = [k for k, _ in plan.dev.profile_last()]
        plan.dev.profile_enable(False)
        if N >= 2 and p >= 5:                            # (p = 4 solve_mv: the hand-trimmed kernels of solve_tile4.hip)
            assert ("tilen_gain_cols_kernel" in names) == (form == "cols"), names
            assert ("bwd_mv_tilen_fused_kernel" in names) == (form == "fused"), names
        res[form] = (np.array(m), np.array(v))
    assert np.all(np.isfinite(res["fused"][0])) and np.all(np.isfinite(res["fused"][1]))
    for form in ("cols", "fused"):
        np.testing.assert_array_equal(res["lanes"][0], res[form][0])
        np.testing.assert_array_equal(res["lanes"][1], res[form][1])


@pytest.mark.parametrize("p", [5, 8])
@pytest.mark.parametrize("mode", ["mv", "sim"])
def test_blocked_tiles_per_trajectory_priors(ra, p, mode):
    """Prior matrices that differ between trajectories (the pseudo-marginal sampler's sigma per draw, and here a weight matrix
    per trajectory too): the batched-input indexing of the forward kernel, of the gain items on DPP rows and of the
    lane-per-item gain kernel (solve_sim) against the oracle."""
    B, N, t_max = 6, 40, 0.4
    rng = np.random.default_rng(50 + p)
    s = _fitz(ra, p, B=B, seed=3, N=N, t_max=t_max)
    sig = 0.1 * np.exp(0.3 * rng.standard_normal((B, 2)))
    Qs, Rs = zip(*[ra.ibm_init(t_max / N * (1.0 + 0.05 * b), p, sig[b]) for b in range(B)])
    prior = (np.stack(Qs), np.stack(Rs))                                  # (B, 2, p, p) each
    args = (s["W"], s["x0"], 0.0, t_max, N)
    g, o = _itg(ra, "kramer")
    if mode == "mv":
        m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, g, prior, theta=s["theta"])
        mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, o, prior, theta=s["theta"])
        _close(m, mo, v, vo, p, prior[1][0], entrywise=False)
    else:
        x = ra.solve_sim(4, ra.ode.fitzhugh_nagumo, *args, g, prior, theta=s["theta"])
        xo = scan.solve_sim(4, odes.fitzhugh_nagumo, *args, o, prior, theta=s["theta"])
        scale = np.maximum(np.max(np.abs(xo), axis=(0, 1, 2)), 1.0)
        assert np.max(np.abs(x - xo) / scale) < (1e-6 if p == 5 else 1e-4)


@pytest.mark.parametrize("p,rhs", [(5, "lorenz63"), (6, "lorenz63"), (6, "higher_order"), (8, "higher_order")])
def test_blocked_tiles_other_block_counts(ra, p, rhs):
    """n_block = 3 (one trajectory per wave, three units) and n_block = 1 (four trajectories per wave)."""
    B = 6
    rng = np.random.default_rng(p)
    if rhs == "lorenz63":
        theta = np.array([28., 10., 8. / 3.]) * np.exp(0.01 * rng.standard_normal((B, 3)))
        W, init = ra.utils.first_order_pad(ra.ode.lorenz63, 3, p)
        x0 = init(np.array([-12., -5., 38.]) + 0.1 * rng.standard_normal((B, 3)), 0.0, theta=theta)
        N, t_max, prior = 60, 0.06, ra.ibm_init(1e-3, p, np.array([5e7] * 3))
        fun, ofun, kw = ra.ode.lorenz63, odes.lorenz63, dict(theta=theta)
    else:
        W = np.zeros((1, 1, p)); W[0, 0, 2] = 1.0
        x0 = np.zeros((B, 1, p)); x0[:, 0, :4] = np.array([-1., 0., 1., 0.]) + 0.01 * rng.standard_normal((B, 4))
        N, t_max, prior = 40, 2.0, ra.ibm_init(0.05, p, np.array([.001]))
        fun, ofun, kw = ra.ode.higher_order, odes.higher_order, {}
    for name in ("kramer", "rodeo"):
        g, o = _itg(ra, name)
        m, v = ra.solve_mv(None, fun, W, x0, 0.0, t_max, N, g, prior, **kw)
        mo, vo = scan.solve_mv(None, ofun, W, x0, 0.0, t_max, N, o, prior, **kw)
        _close(m, mo, v, vo, p, prior[1])


@pytest.mark.parametrize("N", [1, 2, 3, 7, 8, 9, 10, 16, 17, 25, 26, 33])
@pytest.mark.parametrize("B,p", [(1, 5), (3, 5), (3, 6), (2, 7)])
def test_blocked_tiles_short_horizons_and_ragged_batches(ra, N, B, p):
    """Around the backward chunk sizes (8-step chunks / rings, three producer stages in flight) and with unit counts that
    do not fill a wave."""
    s = _fitz(ra, p, B=B, seed=N, N=N, t_max=0.01 * N)
    args = (s["W"], s["x0"], 0.0, s["t_max"], N)
    m, v = ra.solve_mv(None, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_kramer, s["prior"], theta=s["theta"])
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, oi.interrogate_kramer, s["prior"], theta=s["theta"])
    _close(m, mo, v, vo, p, s["prior"][1], yard=lambda: _longdouble_solve_mv(s, "kramer"))
    x = ra.solve_sim(5, ra.ode.fitzhugh_nagumo, *args, ra.interrogate.interrogate_rodeo, s["prior"], theta=s["theta"])
    xo = scan.solve_sim(5, odes.fitzhugh_nagumo, *args, oi.interrogate_rodeo, s["prior"], theta=s["theta"])
    assert x.shape == xo.shape
    scale = np.maximum(np.max(np.abs(xo), axis=tuple(range(xo.ndim - 1))), 1.0)
    assert np.max(np.abs(x - xo) / scale) < 1e-6


@pytest.mark.parametrize("p", [4, 5, 6])
@pytest.mark.parametrize("name", ["rodeo", "chkrebtii", "kramer"])
def test_blocked_tiles_solve_sim(ra, p, name):
    """solve_sim (solve.py:125-205): forward pass (with interrogate_chkrebtii's draws inside it), backward sampler.  Device
    and oracle share the Philox stream, so the sample paths agree to rounding (times the amplification above)."""
    g, o = _itg(ra, name)
    s = _fitz(ra, p, B=6, seed=10 + p, N=60, t_max=1.2)
    args = (s["W"], s["x0"], 0.0, s["t_max"], s["N"])
    x = ra.solve_sim(77, ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    xo = scan.solve_sim(77, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    assert x.shape == (6, 61, 2, p)
    scale = np.maximum(np.max(np.abs(xo), axis=(0, 1, 2)), 1.0)
    err = np.max(np.abs(x - xo) / scale)
    assert err < 100 * TOL_MEAN[p], err                # the psd factor of a nearly singular conditional variance
    np.testing.assert_array_equal(x[:, 0], s["x0"])
    # draws are keyed by the global trajectory index: the second half of the batch alone gives the same paths
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, s["W"], s["x0"][3:], 0.0, s["t_max"], s["N"], g, s["prior"], traj_offset=3,
                        theta=s["theta"][3:])
    plan.sim(77)
    np.testing.assert_array_equal(plan.x_host(), x[3:])


@pytest.mark.parametrize("p", [5, 7])
def test_blocked_tiles_filter_and_mv_with_chkrebtii(ra, p):
    """The filter alone (SolvePlan.filter) and solve_mv with interrogate_chkrebtii (draws inside the forward pass)."""
    g, o = _itg(ra, "chkrebtii")
    s = _fitz(ra, p, B=4, seed=p)
    args = (s["W"], s["x0"], 0.0, s["t_max"], s["N"])
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, *args, g, s["prior"], theta=s["theta"])
    plan.filter(9)
    mf, vf = plan.state_host()
    ref = scan.solve_filter(9, odes.fitzhugh_nagumo, *args, o, *s["prior"], theta=s["theta"])
    _close(mf, ref["state_filt"][0], vf, ref["state_filt"][1], p, s["prior"][1])
    plan.mv(9)
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(9, odes.fitzhugh_nagumo, *args, o, s["prior"], theta=s["theta"])
    _close(m, mo, v, vo, p, s["prior"][1])


def test_blocked_tiles_basic_and_logposterior_read_the_tile_layout(ra):
    from scipy.stats import norm
    from rodeo_amd.inference.basic import GaussianObsLoglik
    p = 5
    s = _fitz(ra, p, B=4, seed=1, N=60, t_max=1.2)
    obs_times = np.array([0.0, 0.4, 0.8, 1.2])
    Y = np.array([-1., 1.]) + 0.05 * np.random.default_rng(3).standard_normal((4, 2))
    sd = 0.1
    ll, Xt = ra.inference.basic(None, ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0.0, s["t_max"], s["N"],
                                ra.interrogate.interrogate_kramer, s["prior"], Y, obs_times, GaussianObsLoglik(sd),
                                theta=s["theta"])
    mo, _ = scan.solve_mv(None, odes.fitzhugh_nagumo, s["W"], s["x0"], 0.0, s["t_max"], s["N"], oi.interrogate_kramer,
                          s["prior"], theta=s["theta"])
    ind = np.searchsorted(np.linspace(0, s["t_max"], s["N"] + 1), obs_times)
    ref = np.array([np.sum(norm.logpdf(Y, loc=mo[b][ind, :, 0], scale=sd)) for b in range(4)])
    np.testing.assert_allclose(ll, ref, rtol=1e-7, atol=1e-7)
    scale = np.maximum(np.max(np.abs(mo), axis=(0, 1, 2)), 1.0)                      # per derivative order
    assert np.max(np.abs(np.asarray(Xt) - mo) / scale) < TOL_MEAN[p]
    # an arbitrary Python obs_loglik: only the observed time slices come back
    ll2, _ = ra.inference.basic(None, ra.ode.fitzhugh_nagumo, s["W"], s["x0"], 0.0, s["t_max"], s["N"],
                                ra.interrogate.interrogate_kramer, s["prior"], Y, obs_times,
                                lambda obs, ode, **kw: np.sum(norm.logpdf(obs, loc=ode[:, :, 0], scale=sd)), theta=s["theta"])
    np.testing.assert_allclose(ll2, ref, rtol=1e-7, atol=1e-7)


def test_blocked_tiles_traced_python_rhs(ra):
    """An ordinary Python ode_fun at n_deriv = 5 and 6: traced, compiled with hiprtc, on the blocked tile kernels
    (two variables: one wave carries two trajectories; six variables: a two-wave workgroup per trajectory)."""
    from rodeo_amd import _lib

    def fitz(X, t, theta):
        a, b, c = theta
        V, R = X[0, 0], X[1, 0]
        return np.array([[c * (V - V * V * V / 3 + R)], [-1 / c * (V - a + b * R)]])

    for p in (5, 6):
        s = _fitz(ra, p, B=5, seed=p, N=40, t_max=0.8)
        args = (s["W"], s["x0"], 0.0, s["t_max"], s["N"])
        plan = ra.SolvePlan(fitz, *args, ra.interrogate.interrogate_kramer, s["prior"], theta=s["theta"])
        plan.mv(None)
        assert plan.layout == _lib.LAYOUT_TILEP
        m, v = plan.state_host()
        mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, *args, oi.interrogate_kramer, s["prior"], theta=s["theta"])
        _close(m, mo, v, vo, p, s["prior"][1])
    # six variables (the user ODE of test_gpu_user_rhs.py, Jacobian by duals) at n_deriv = 5
    from test_gpu_user_rhs import SIX_SRC, _six_host
    six = ra.ode.from_source("AutoJac<Six>", SIX_SRC, 6, (("theta", 4),), _six_host, name="six_p5")

    def jac(X, t, theta):
        th = np.asarray(theta, dtype=np.float64)
        b, k, g, d = th[..., 0], th[..., 1], th[..., 2], th[..., 3]
        I, A = X[..., 2, 0], X[..., 4, 0]
        J = np.zeros(np.broadcast_shapes(X.shape[:-2], th.shape[:-1]) + (6, 1, X.shape[-1]))
        J[..., 0, 0, 0] = -b * (I + 0.5 * A)
        J[..., 1, 0, 0] = -k
        J[..., 2, 0, 0] = -(g + d)
        J[..., 4, 0, 0] = -g
        J[..., 5, 0, 0] = -0.2
        return J
    o_ode = odes.ODE("six", 6, 1, lambda X, t, theta: _six_host(X, t, theta), jac)
    p, B, N = 5, 3, 40
    rng = np.random.default_rng(6)
    theta = np.array([2.0, 0.7, 0.3, 0.1]) * np.exp(0.05 * rng.standard_normal((B, 4)))
    W, init = ra.utils.first_order_pad(six, 6, p)
    x0 = init(np.array([0.9, 0.04, 0.03, 0.0, 0.03, 0.0]) + 0.005 * rng.standard_normal((B, 6)), 0.0, theta=theta)
    prior = ra.ibm_init(0.8 / N, p, np.array([.1] * 6))
    plan = ra.SolvePlan(six, W, x0, 0., 0.8, N, ra.interrogate.interrogate_kramer, prior, theta=theta)
    plan.mv(None)
    assert plan.layout == _lib.LAYOUT_TILEP
    m, v = plan.state_host()
    mo, vo = scan.solve_mv(None, o_ode, W, x0, 0., 0.8, N, oi.interrogate_kramer, prior, theta=theta)
    _close(m, mo, v, vo, p, prior[1])
