import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra
from oracle import scan, odes, interrogations as oi
np.set_printoptions(linewidth=200, precision=3)
B, N = 1, 50
theta = np.array([[28., 10., 8. / 3.]])
W, init = ra.utils.first_order_pad(ra.ode.lorenz63, 3, 4)
x0 = init(np.array([[-12., -5., 38.]]), 0.0, theta=theta)
prior = ra.ibm_init(1e-3, 4, np.array([5e7] * 3))
args = (W, x0, 0.0, N * 1e-3, N)
for bm in (False, True):
    pt = ra.SolvePlan(ra.ode.lorenz63, *args, ra.interrogate.interrogate_kramer, prior, batch_minor=bm, theta=theta)
    for mode in ("filter", "mv"):
        getattr(pt, mode)(None)
        m, v = pt.state_host()
        if mode == "filter":
            fo = scan.solve_filter(None, odes.lorenz63, *args, oi.interrogate_kramer, *prior, theta=theta)
            mo, vo = fo["state_filt"]
        else:
            mo, vo = scan.solve_mv(None, odes.lorenz63, *args, oi.interrogate_kramer, prior, theta=theta)
        sd = np.sqrt(np.max(np.abs(np.einsum("bnkii->bnki", vo)), axis=(0, 1, 2)))
        e = np.abs(v - vo) / (sd[:, None] * sd[None, :])
        print("batch_minor", bm, mode, "max scaled var err", e.max(), "at", np.unravel_index(e.argmax(), e.shape), "per step max", e.max(axis=(0, 2, 3, 4))[[1, 2, 5, 10, 25, 49]])
        print("   sd scale", sd, " vo diag at last", np.einsum("bnkii->bnki", vo)[0, -1, 0])
