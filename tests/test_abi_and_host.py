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
            # (the 8-wave two-set form of bwd_mv_tile4_kernel, RK_T4_BWD=ds, is one workgroup per CU by design)
            assert cco.workgroups_per_cu(k) >= (2 if k.get("max_flat_workgroup_size", 1024) <= 256 else 1), k["name"]


def test_dpp_hazard_scanner_rules():
    """scripts/check_dpp_hazards.py (run over the disassembly of every shipped code object by check_code_objects.py):
    VGPR write -> DPP read needs two wait states, a VALU write of EXEC five, a branch target resets the history."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_dpp_hazards", os.path.join(ROOT, "scripts", "check_dpp_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dpp = "v_fmac_f64_dpp v[0:1], v[2:3], v[4:5] row_newbcast:1 row_mask:0xf bank_mask:0xf"
    k = "_ZN2rk1kEv:"
    cases = [
        ([k, "v_mul_f64 v[2:3], v[6:7], v[8:9]", dpp], 1),                                  # written right in front
        ([k, "v_mul_f64 v[2:3], v[6:7], v[8:9]", "s_nop 1", dpp], 0),                       # two wait states
        ([k, "v_mul_f64 v[2:3], v[6:7], v[8:9]", "v_mov_b32 v9, v10", dpp], 1),             # one instruction is one wait state
        ([k, "v_mul_f64 v[2:3], v[6:7], v[8:9]", "v_mov_b32 v9, v10", "v_mov_b32 v11, v10", dpp], 0),
        ([k, "v_mul_f64 v[12:13], v[6:7], v[8:9]", dpp], 0),                                # another register
        ([k, "v_cmpx_lt_f64 vcc, v[6:7], v[8:9]", "s_nop 2", dpp], 1),                      # EXEC write: three wait states are not five
        ([k, "v_cmpx_lt_f64 vcc, v[6:7], v[8:9]", "s_nop 4", dpp], 0),
        ([k, "s_nop 7", ".LBB0_1:", dpp], 1),                                               # a branch target right in front
        ([k, ".LBB0_1:", "s_nop 1", dpp], 0),
    ]
    for lines, want in cases:
        n, bad = mod.scan(lines, verbose=False)
        assert n == 1 and bad == want, (lines, bad, want)


def test_optional_workspace_allocation_failure_falls_back():
    """the record workspace of the square-root small-block path's two-kernel backward pass is optional
    (include/rodeo_kalman.h) -- if its allocation fails, SolvePlan leaves workspace = NULL and the library runs the one-kernel
    form; a REQUIRED workspace (dense path) still raises.  No GPU: the plan is put together by hand around a fake device."""
    import ctypes as C
    import numpy as np
    from rodeo_amd import _lib
    from rodeo_amd.solve import SolvePlan
    lib = _lib.load()

    class FakeArr:
        def __init__(self, shape, pad_bytes=0):
            self.shape, self.nbytes, self.ptr = tuple(shape), int(np.prod(shape)) * 8 + pad_bytes, C.c_void_p(0x1000)

    class FakeDev:
        def __init__(self):
            self.lib, self.h, self.refused = lib, None, []

        def empty(self, shape, pad_bytes=0):
            if len(shape) == 1:                                    # the workspace: a flat array of doubles
                self.refused.append(shape[0])
                raise MemoryError("simulated hipErrorOutOfMemory")
            return FakeArr(shape, pad_bytes)

    def plan_for(kalman, n_bstate, n_bmeas, n_block, rhs):
        p = SolvePlan.__new__(SolvePlan)
        p.dev, p.N, p.d, p.p, p.B = FakeDev(), 50, n_block, n_bstate, 8
        p.cfg = _lib.SolveCfg(n_traj=8, n_steps=50, n_block=n_block, n_bstate=n_bstate, n_bmeas=n_bmeas, rhs_id=rhs,
                              interrogate=_lib.INTERROGATE_KRAMER, kalman_type=kalman, n_theta=3, flags=0, t_min=0.0,
                              t_max=1.0, seed=0, traj_offset=0)
        p._store_pred, p._bufs, p._ws, p.layout = False, {}, None, None
        p.mean_state = p.var_state = p.mean_pred = p.var_pred = p.x_state = None
        return p

    p = plan_for(_lib.KALMAN_SQRT, 3, 1, 2, _lib.RHS_FITZHUGH_NAGUMO)
    p._prepare_out(_lib.MODE_MV)
    assert p.dev.refused and p._ws is None and not p._out.workspace and p._out.workspace_bytes == 0
    d = plan_for(_lib.KALMAN_STANDARD, 40, 8, 1, _lib.RHS_LINEAR_DENSE)          # the dense path NEEDS its workspace
    with pytest.raises(MemoryError):
        d._prepare_out(_lib.MODE_MV)
