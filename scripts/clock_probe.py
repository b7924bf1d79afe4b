"""Is the forward kernel's slowdown behind a backward kernel (B = 2048) a core-clock effect?  A 4-trajectory filter launch
(one wave: its time is 4000 x the step's dependent chain = a clock meter) is timed (a) in a stream of its own kind, (b) right
after a B = 2048 solve_mv, (c) right after a B = 2048 filter, (d) after 5 ms of idle."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
import bench
W, x0, theta, prior = bench.make_problem(ra, 0)
mk = lambda B: ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, np.concatenate([x0] * 2)[:B], 0.0, 40.0, 4000,
                            ra.interrogate.interrogate_kramer, prior, theta=np.concatenate([theta] * 2)[:B])
big, small = mk(2048), mk(4)
dev = big.dev
def t_small(before):
    out = []
    for _ in range(6):
        before()
        dev.profile_enable(True)
        small.filter(None)
        out.append(dict(dev.profile_last())["fwd_tile3_kernel"])
        dev.profile_enable(False)
    return round(float(np.median(out)), 4)
res = {"small after small": t_small(lambda: small.filter(None)),
       "small after big solve_mv": t_small(lambda: big.mv(None)),
       "small after big filter": t_small(lambda: big.filter(None)),
       "small after 5 ms idle": t_small(lambda: (dev.sync(), time.sleep(0.005))),
       "small after big solve_mv + 5 ms idle": t_small(lambda: (big.mv(None), dev.sync(), time.sleep(0.005)))}
print(json.dumps(res))
