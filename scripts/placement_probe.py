"""Experiment build only (library compiled with -DRK_PLACEMENT_DEBUG): on which SIMD did every wave of fwd_tile3_kernel run,
in a filter-only stream and right behind bwd_mv_tile3_kernel, at B = 2048 (1024 single-wave workgroups on 1024 SIMDs)?
Prints the histogram of SIMD occupancies next to the kernel time."""
import sys, os, json, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
import bench
W, x0, theta, prior = bench.make_problem(ra, 0)
for B in [int(v) for v in sys.argv[1:]] or [2048]:
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, np.concatenate([x0] * 4)[:B], 0.0, 40.0, 4000, ra.interrogate.interrogate_kramer,
                        prior, theta=np.concatenate([theta] * 4)[:B])
    dev = plan.dev
    n_waves = B * 2 // 4

    def placement():
        import ctypes as C
        from rodeo_amd import _lib
        tail = np.empty(n_waves * 64)                     # the scratch tail behind the tiles (rk_solve_sizes)
        _lib.check(dev.lib.rk_d2h(dev.h, tail.ctypes.data_as(C.c_void_p),
                                  C.c_void_p(plan.var_state.ptr.value + plan.var_state.nbytes), tail.nbytes))
        tail = tail.reshape(n_waves, 64)
        ids, xcc = tail[:, 0].astype(np.int64), tail[:, 1].astype(np.int64) & 0xf
        key = (xcc << 20) | (((ids >> 13) & 7) << 16) | (((ids >> 12) & 1) << 12) | (((ids >> 8) & 0xf) << 4) | ((ids >> 4) & 3)
        cnt = collections.Counter(key.tolist())
        hist = collections.Counter(cnt.values())
        per_cu = collections.Counter((key >> 4).tolist())                   # waves per (xcc, se, sh, cu)
        return {"distinct_simds": len(cnt), "simds_by_waves": dict(sorted(hist.items())),
                "cus_by_waves": dict(sorted(collections.Counter(per_cu.values()).items()))}

    def timed(seq_before):
        out = []
        for _ in range(5):
            seq_before()
            dev.profile_enable(True)
            plan.filter(None)
            ms = dict(dev.profile_last())["fwd_tile3_kernel"]
            dev.profile_enable(False)
            out.append((round(ms, 4), placement()))
        return out
    print(json.dumps({"B": B, "after filter": timed(lambda: plan.filter(None))}))
    print(json.dumps({"B": B, "after solve_mv": timed(lambda: plan.mv(None))}))
    # does a trivial launch of the forward kernel's own shape (same grid, one time step) in between restore the placement?
    one = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, np.concatenate([x0] * 4)[:B], 0.0, 0.01, 1, ra.interrogate.interrogate_kramer,
                       ra.ibm_init(0.01, 3, np.array([0.1, 0.1])), theta=np.concatenate([theta] * 4)[:B])
    small = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0[:4], 0.0, 0.01, 1, ra.interrogate.interrogate_kramer,
                         ra.ibm_init(0.01, 3, np.array([0.1, 0.1])), theta=theta[:4])
    print(json.dumps({"B": B, "after solve_mv + same-grid 1-step filter": timed(lambda: (plan.mv(None), one.filter(None)))}))
    print(json.dumps({"B": B, "after solve_mv + 1-wave 1-step filter": timed(lambda: (plan.mv(None), small.filter(None)))}))
