"""
CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/rodeo_kalman.h declares, the
ctypes table matches the header, the product fails loudly without a device, and the host logic (priors, padding,
parameter packing, interrogate recognition) behaves like the reference's.
"""
import ctypes
import functools
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "rodeo_kalman.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from rodeo_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rodeo_kalman.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    assert lib.rk_version().startswith(b"rodeo_kalman")


def test_struct_layouts_match_header():
    from rodeo_amd import _lib
    assert ctypes.sizeof(_lib.SolveCfg) == 10 * 4 + 2 * 8 + 2 * 8
    assert ctypes.sizeof(_lib.SolveIn) == 5 * 16
    assert ctypes.sizeof(_lib.SolveOut) == 7 * 8
    assert ctypes.sizeof(_lib.OpCfg) == 16


def test_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device the product raises (on the GPU box this test is a no-op)."""
    from rodeo_amd import _lib
    lib = _lib.load()
    n = ctypes.c_int(0)
    rc = lib.rk_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    import rodeo_amd as ra
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    x0 = init(np.array([-1., 1.]), 0.0, theta=np.array([.2, .2, 3.]))
    with pytest.raises(_lib.RodeoKalmanError):
        ra.solve_mv(None, ra.ode.fitzhugh_nagumo, W, x0, 0., 1., 10, ra.interrogate.interrogate_kramer,
                    ra.ibm_init(.1, 3, np.array([.1, .1])), theta=np.array([.2, .2, 3.]))
    with pytest.raises(_lib.RodeoKalmanError):
        ra.kalmantv.standard.predict(np.zeros(2), np.eye(2), np.zeros(2), np.eye(2), np.eye(2))


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rodeo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports oracle"
                assert "librk_oracle" not in src


def test_ibm_init_matches_oracle_and_closed_form():
    import rodeo_amd as ra
    from oracle import priors
    for p in (2, 3, 4, 5):
        Q, R = ra.ibm_init(0.01, p, np.array([0.1, 3.0]))
        Qo, Ro = priors.ibm_init(0.01, p, np.array([0.1, 3.0]))
        np.testing.assert_allclose(Q, Qo, rtol=4e-15); np.testing.assert_allclose(R, Ro, rtol=4e-15)
        # reference_factorial=True: the reference's exp(gammaln(.)) route (ibm.py:21-34, 54-61), bit for bit with its restatement
        Qr, Rr = ra.ibm_init(0.01, p, np.array([0.1, 3.0]), reference_factorial=True)
        np.testing.assert_array_equal(Qr, Qo); np.testing.assert_array_equal(Rr, Ro)
    Q, R = ra.ibm_init(0.05, 3, np.array([[.1, .2], [.3, .4]]))          # batched sigma -> batched R
    assert Q.shape == (2, 3, 3) and R.shape == (2, 2, 3, 3)
    np.testing.assert_allclose(R[1, 0], ra.ibm_init(0.05, 3, np.array([.3]))[1][0], rtol=1e-15)
    Qd, Rd = ra.indep_init(ra.ibm_init(0.1, 3, np.array([1., 2.])))
    Qe, Re = priors.indep_init(priors.ibm_init(0.1, 3, np.array([1., 2.])))
    np.testing.assert_allclose(Qd, Qe, rtol=4e-15); np.testing.assert_allclose(Rd, Re, rtol=4e-15)
    assert Qd.shape == (1, 6, 6)


def test_first_order_pad_and_host_odes():
    import rodeo_amd as ra
    from oracle import odes, priors
    theta = np.array([0.2, 0.2, 3.0])
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, 2, 3)
    Wo, inito = priors.first_order_pad(odes.fitzhugh_nagumo, 2, 3)
    np.testing.assert_array_equal(W, Wo)
    np.testing.assert_allclose(init(np.array([-1., 1.]), 0., theta=theta), [[-1., 1., 0.], [1., 1 / 3, 0.]], rtol=1e-15)
    B = 4
    th = theta * np.exp(0.1 * np.random.default_rng(0).standard_normal((B, 3)))
    x0 = np.random.default_rng(1).standard_normal((B, 2))
    Xb = init(x0, 0., theta=th)
    assert Xb.shape == (B, 2, 3)
    for b in range(B):
        np.testing.assert_allclose(Xb[b], inito(x0[b], 0., theta=th[b]), rtol=1e-15)
    X = np.random.default_rng(2).standard_normal((5, 3, 4))
    thl = np.broadcast_to(np.array([28., 10., 8 / 3]), (5, 3))
    np.testing.assert_allclose(ra.ode.lorenz63(X, 0.1, theta=thl), odes.lorenz63(X, 0.1, theta=thl), rtol=1e-15)
    np.testing.assert_allclose(ra.ode.higher_order(X[:, :1], 0.3), odes.higher_order(X[:, :1], 0.3), rtol=1e-15)


def test_param_packing_and_interrogate_recognition():
    import rodeo_amd as ra
    from rodeo_amd import _lib
    from rodeo_amd.solve import _interrogate_id, _seed
    th, B = ra.ode.fitzhugh_nagumo.pack_params(dict(theta=np.ones((7, 3))))
    assert th.shape == (7, 3) and B == 7
    th, B = ra.ode.fitzhugh_nagumo.pack_params(dict(theta=[.2, .2, 3.]))
    assert th.shape == (3,) and B is None
    with pytest.raises(TypeError):
        ra.ode.fitzhugh_nagumo.pack_params({})
    with pytest.raises(TypeError):
        ra.ode.fitzhugh_nagumo.pack_params(dict(theta=np.ones(3), bogus=1))
    with pytest.raises(ValueError):
        ra.ode.fitzhugh_nagumo.pack_params(dict(theta=np.ones(4)))
    assert _interrogate_id(ra.interrogate.interrogate_kramer)[0] == _lib.INTERROGATE_KRAMER
    pid, bound = _interrogate_id(functools.partial(ra.interrogate.interrogate_chkrebtii, kalman_type="standard"))
    assert pid == _lib.INTERROGATE_CHKREBTII and bound == {"kalman_type": "standard"}
    with pytest.raises(TypeError):
        _interrogate_id(lambda **k: None)
    assert _seed(None) == 0 and _seed(5) == 5 and _seed(np.array([1, 2], dtype=np.uint32)) == (1 << 32) | 2


def test_unknown_kalman_type_raises_before_touching_the_device():
    import rodeo_amd as ra
    with pytest.raises(NotImplementedError):                 # src/rodeo/solve.py:142-143
        ra.solve_mv(None, ra.ode.fitzhugh_nagumo, np.zeros((2, 1, 3)), np.zeros((2, 3)), 0., 1., 5,
                    ra.interrogate.interrogate_kramer, (np.zeros((2, 3, 3)),) * 2, kalman_type="cholesky",
                    theta=np.ones(3))


def test_code_objects_fit_their_launch_assumptions():
    """Every gfx950 kernel in the shipped library: static stack, bounded scratch, LDS <= 160 KB, registers that fit the
    waves its launch bound puts on a SIMD (scripts/check_code_objects.py; the regression guard for the dense path's real
    device-function calls)."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_code_objects", os.path.join(root, "scripts", "check_code_objects.py"))
    cco = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cco)
    ks = cco.kernels_of(os.path.join(root, "rodeo_amd", "librodeo_kalman.so"))
    assert len(ks) > 100 and any("dense_bwd_mv_kernel" in k["name"] for k in ks)
    assert cco.check(ks) == []
    # the checker does flag what it is there for
    base = {"name": "k", "uses_dynamic_stack": False, "private_segment_fixed_size": 0, "group_segment_fixed_size": 0,
            "vgpr_count": 64, "agpr_count": 0, "max_flat_workgroup_size": 256}
    assert cco.check([base]) == []
    bad = [dict(base, uses_dynamic_stack=True), dict(base, private_segment_fixed_size=1 << 20),
           dict(base, vgpr_count=300, max_flat_workgroup_size=512)]
    assert len(cco.check(bad)) == 3
    # two workgroups per CU where a kernel is built for that: 260 registers (or 96 KB of LDS) would halve it silently
    pc = dict(base, name="_ZN2rk19bwd_mv_tile4_kernelILi3EEEvNS_9SolveArgsEPd", group_segment_fixed_size=74816, vgpr_count=256)
    assert cco.workgroups_per_cu(pc) == 2 and cco.check([pc]) == []
    assert cco.workgroups_per_cu(dict(pc, vgpr_count=260)) == 1 and len(cco.check([dict(pc, vgpr_count=260)])) == 1
    assert cco.workgroups_per_cu(dict(pc, group_segment_fixed_size=96256)) == 1
    for k in ks:
        if any(key in k["name"] for key in cco.MIN_WORKGROUPS_PER_CU):
            assert cco.workgroups_per_cu(k) >= 2, k["name"]
