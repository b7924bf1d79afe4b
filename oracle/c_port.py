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
        _lib.rko_scratch_doubles.restype = C.c_size_t
        _lib.rko_scratch_doubles.argtypes = [C.c_int] * 3
        _lib.rko_first_touch.restype = None
        _lib.rko_first_touch.argtypes = [dp, C.c_size_t, C.c_int, dp, C.c_size_t, C.c_int]
        _lib.rko_solve_mv_ws.restype = C.c_int
        _lib.rko_solve_mv_ws.argtypes = ([C.c_int] * 6 + [C.c_double] * 2 + [dp] * 5 + [C.c_int] + [dp] * 3
                                         + [C.c_int] * 2 + [dp])
    return _lib


def max_threads():
    return load().rko_max_threads()


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


class TimedSolve:
    """
    bench.py's cpu_baseline leg: outputs and per-thread scratch are allocated ONCE here and paged in by the threads that
    will write them (rko_first_touch); `run(reps, nthreads)` then times nothing but the C passes (the time comes from
    omp_get_wtime around the parallel region).  `max_threads` bounds the scratch.
    """

    def __init__(self, rhs, itg, W, x0, t_min, t_max, n_steps, prior_pars, theta=None, max_threads=None):
        self.lib = load()
        self.W = np.ascontiguousarray(W, dtype=np.float64)
        self.x0 = np.ascontiguousarray(x0, dtype=np.float64)
        self.Q = np.ascontiguousarray(prior_pars[0], dtype=np.float64)
        self.R = np.ascontiguousarray(prior_pars[1], dtype=np.float64)
        self.B, self.d, self.p = self.x0.shape
        self.N, self.t_min, self.t_max = int(n_steps), float(t_min), float(t_max)
        self.rhs, self.itg = RHS[rhs], ITG[itg]
        self.theta = None if theta is None else np.ascontiguousarray(
            np.broadcast_to(theta, (self.B, np.shape(theta)[-1])), dtype=np.float64)
        self.max_threads = int(max_threads or max_threads_default())
        self.mean = np.empty((self.B, self.N + 1, self.d, self.p))
        self.var = np.empty((self.B, self.N + 1, self.d, self.p, self.p))
        self.per_thread = int(self.lib.rko_scratch_doubles(self.N, self.d, self.p))
        self.scratch = np.empty(self.max_threads * self.per_thread)
        self._touched_for = None

    def _touch(self, nthreads):
        if self._touched_for == nthreads:
            return
        lib = self.lib
        lib.rko_first_touch(_ptr(self.mean), self.mean[0].size, self.B, _ptr(self.scratch), self.per_thread, nthreads)
        lib.rko_first_touch(_ptr(self.var), self.var[0].size, self.B, None, 0, nthreads)
        self._touched_for = nthreads

    def run(self, reps, nthreads, n_traj=None):
        """`reps` passes over the first `n_traj` trajectories (default: all) on `nthreads` threads; returns the seconds
        the passes took (outputs and scratch are paged in before, outside the timing)."""
        if nthreads > self.max_threads:
            raise ValueError("nthreads exceeds the scratch this object was sized for")
        B = self.B if n_traj is None else int(n_traj)
        if not (1 <= B <= self.B):
            raise ValueError("n_traj out of range")
        self._touch(nthreads)
        sec = C.c_double(0.0)
        rc = self.lib.rko_solve_mv_ws(self.rhs, self.itg, B, self.N, self.d, self.p, self.t_min, self.t_max,
                                      _ptr(self.W), _ptr(self.x0), _ptr(self.Q), _ptr(self.R), _ptr(self.theta),
                                      0 if self.theta is None else self.theta.shape[1], _ptr(self.mean), _ptr(self.var),
                                      _ptr(self.scratch), int(reps), int(nthreads), C.byref(sec))
        if rc != 0:
            raise RuntimeError(f"rko_solve_mv_ws failed with {rc}")
        return sec.value


def max_threads_default():
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
