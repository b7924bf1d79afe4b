import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
import bench
W, x0, theta, prior = bench.make_problem(ra, 0)
for dbg in (0, 8, 10):
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, 40.0, 4000, ra.interrogate.interrogate_kramer, prior, theta=theta)
    plan.cfg.flags |= dbg << 16
    dev = plan.dev
    plan.mv(None); plan.mv(None)
    dev.profile_enable(True)
    acc = {}
    for _ in range(5):
        plan.mv(None)
        for k, ms in dev.profile_last():
            acc.setdefault(k, []).append(ms)
    dev.profile_enable(False)
    print("dbg", dbg, {k: round(float(np.mean(v)), 4) for k, v in acc.items()})
    if dbg in (4, 5, 8, 10):
        import ctypes as C
        from rodeo_amd import _lib
        out = np.zeros(64)
        off = 4001 * 2048 * 96
        _lib.check(dev.lib.rk_d2h(dev.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(plan.var_state.ptr.value + off), 512))
        cyc = out[3]; print("  busy (between barriers) cycles =", out[7], " loads done =", out[11], " chain issued =", out[15])
        print("  producer 0: phase A cycles =", out[20], " phase B cycles =", out[24], " of A waiting for the fetch =", out[28], " fetch issued at", out[32], "(83/84 phases each)")
        ms = float(np.mean(acc["bwd_mv_tile3_kernel"]))
        print("  consumer loop shader cycles =", cyc, " kernel ms =", ms, " => effective clock GHz ~", cyc / (ms * 1e6))
