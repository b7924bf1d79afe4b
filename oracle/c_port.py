"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  ctypes loader for oracle/c/librk_oracle.so, the plain-C restatement
of solve_mv (reference layout, OpenMP over trajectories).  Only tests/ and bench.py's cpu_baseline leg use it.
"""
import ctypes as C
import os
import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "librk_oracle.so")
RHS = {"fitzhugh_nagumo": 1, "lorenz63": 2, "higher_order": 3}
ITG = {"rodeo": 0, "schober": 1, "kramer": 2}
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError(f"{_PATH} not built: run `make -C oracle/c` (or __graft_entry__.build())")
        _lib = C.CDLL(_PATH)
        dp = C.POINTER(C.c_double)
        _lib.rko_solve_mv.restype = C.c_int
        _lib.rko_solve_mv.argtypes = [C.c_int] * 6 + [C.c_double] * 2 + [dp] * 5 + [C.c_int] + [dp] * 2 + [C.c_int]
        _lib.rko_max_threads.restype = C.c_int
    return _lib


def max_threads():
    return load().rko_max_threads()


def solve_mv(rhs, itg, W, x0, t_min, t_max, n_steps, prior_pars, theta=None, nthreads=0):
    """x0 (B, d, p); theta (B, n_theta) or None; W (d,1,p); prior (d,p,p) shared.  Returns mean (B,N+1,d,p), var."""
    lib = load()
    W = np.ascontiguousarray(W, dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    Q = np.ascontiguousarray(prior_pars[0], dtype=np.float64)
    R = np.ascontiguousarray(prior_pars[1], dtype=np.float64)
    B, d, p = x0.shape
    if theta is not None:
        theta = np.ascontiguousarray(np.broadcast_to(theta, (B, np.shape(theta)[-1])), dtype=np.float64)
    mean = np.empty((B, n_steps + 1, d, p))
    var = np.empty((B, n_steps + 1, d, p, p))
    dp = C.POINTER(C.c_double)
    ptr = lambda a: a.ctypes.data_as(dp) if a is not None else None
    rc = lib.rko_solve_mv(RHS[rhs], ITG[itg], B, n_steps, d, p, float(t_min), float(t_max), ptr(W), ptr(x0), ptr(Q),
                          ptr(R), ptr(theta), 0 if theta is None else theta.shape[1], ptr(mean), ptr(var), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"rko_solve_mv failed with {rc}")
    return mean, var
