"""Why does fwd_tile3_kernel run slower inside the solve_mv stream than in a filter-only stream at B = 2048 (VERDICT r1,
weak #7)?  Times the forward kernel (HIP events around the launch) in different neighbourhoods:
  filter     : fwd, fwd, fwd, ...                         (the kernel alone)
  mv         : fwd, bwd, fwd, bwd, ...                    (the solve_mv stream)
  mv + idle  : fwd, bwd, [sync + 2 ms pause], fwd, ...    (does an idle gap restore the filter-only time?)
  mv + sync  : fwd, bwd, [sync only], fwd, ...
  memset     : fwd, [memset of the 1.6 GB output], fwd, ...   (a pure HBM writer before the forward kernel)
  mv 2 bufs  : two plans alternating                     (the forward pass never overwrites what the backward just wrote)
"""
import sys, os, time, json, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
from rodeo_amd import _lib
import bench

W, x0, theta, prior = bench.make_problem(ra, 0)
for B in [int(v) for v in sys.argv[1:]] or [1024, 2048]:
    reps = -(-B // 1024)
    x0b, thb = np.concatenate([x0] * reps)[:B], np.concatenate([theta] * reps)[:B]
    mk = lambda: ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0b, 0.0, 40.0, 4000, ra.interrogate.interrogate_kramer, prior, theta=thb)
    plan, plan2 = mk(), mk()
    dev = plan.dev

    def run(seq, n=8):
        acc = {}
        for f in seq[:2]:
            f()
        dev.sync()
        dev.profile_enable(True)
        for _ in range(n):
            for f in seq:
                f()
                for k, ms in dev.profile_last():
                    acc.setdefault(k, []).append(ms)
        dev.profile_enable(False)
        return {k: round(float(np.median(v)), 4) for k, v in acc.items()}

    def pause():
        dev.sync(); time.sleep(0.002)
    def memset():
        _lib.check(dev.lib.rk_memset(dev.h, plan2.var_state.ptr, 0, plan2.var_state.nbytes))
    plan2.mv(None)
    out = {"B": B,
           "filter": run([lambda: plan.filter(None)]),
           "mv": run([lambda: plan.mv(None)]),
           "mv+idle": run([lambda: plan.mv(None), pause]),
           "mv+sync": run([lambda: plan.mv(None), dev.sync]),
           "memset,filter": run([memset, lambda: plan.filter(None)]),
           "mv two buffers": run([lambda: plan.mv(None), lambda: plan2.mv(None)])}
    print(json.dumps(out), flush=True)
    del plan, plan2
