"""
ctypes binding of librodeo_kalman.so (C ABI: include/rodeo_kalman.h).

There is NO CPU fallback: if the shared library is missing, or a call fails, an exception is raised.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C rodeo_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RK_LIB_PATH: another build of the SAME library (the phase-timing build `make -C rodeo_amd/csrc stamps`); there is still no
# fallback -- a path that does not load raises.
LIB_PATH = os.environ.get("RK_LIB_PATH") or os.path.join(_HERE, "librodeo_kalman.so")

RK_OK = 0
RK_ERR_INVALID, RK_ERR_UNSUPPORTED, RK_ERR_HIP, RK_ERR_RCCL, RK_ERR_NOMEM = -1, -2, -3, -4, -5
KALMAN_STANDARD, KALMAN_SQRT = 0, 1
INTERROGATE_RODEO, INTERROGATE_SCHOBER, INTERROGATE_KRAMER, INTERROGATE_CHKREBTII = 0, 1, 2, 3
RHS_FITZHUGH_NAGUMO, RHS_LORENZ63, RHS_HIGHER_ORDER, RHS_LINEAR_DENSE = 1, 2, 3, 4
RHS_USER_BASE = 1000
FLAG_STORE_PRED = 1
FLAG_BATCH_MINOR = 2
MODE_FILTER, MODE_MV, MODE_SIM = 0, 1, 2
LAYOUT_BATCH_MINOR, LAYOUT_TILE3, LAYOUT_TRAJ_MAJOR, LAYOUT_TILE4, LAYOUT_TILEP = 0, 1, 2, 3, 4
COMM_UID_BYTES = 128


class RodeoKalmanError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librodeo_kalman error {code}: {msg}")
        self.code = code


class SolveCfg(C.Structure):
    _fields_ = [("n_traj", C.c_int32), ("n_steps", C.c_int32), ("n_block", C.c_int32), ("n_bstate", C.c_int32),
                ("n_bmeas", C.c_int32), ("rhs_id", C.c_int32), ("interrogate", C.c_int32),
                ("kalman_type", C.c_int32), ("n_theta", C.c_int32), ("flags", C.c_int32),
                ("t_min", C.c_double), ("t_max", C.c_double), ("seed", C.c_uint64), ("traj_offset", C.c_uint64)]


class SolveIn(C.Structure):
    _fields_ = [("ode_weight", C.c_void_p), ("ode_weight_batched", C.c_int32),
                ("ode_init", C.c_void_p), ("ode_init_batched", C.c_int32),
                ("prior_weight", C.c_void_p), ("prior_weight_batched", C.c_int32),
                ("prior_var", C.c_void_p), ("prior_var_batched", C.c_int32),
                ("theta", C.c_void_p), ("theta_batched", C.c_int32)]


class SolveOut(C.Structure):
    _fields_ = [("mean_state", C.c_void_p), ("var_state", C.c_void_p), ("mean_pred", C.c_void_p),
                ("var_pred", C.c_void_p), ("x_state", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)]


class OpCfg(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_state", C.c_int32), ("n_meas", C.c_int32), ("kalman_type", C.c_int32)]


_H = C.c_void_p          # rk_handle
_P = C.c_void_p          # device / host pointer
_D = C.c_double
_I = C.c_int32

# name -> (restype, argtypes).  Every symbol declared in include/rodeo_kalman.h is listed here; the CPU test-suite
# checks that the header, this table and the built library agree.
SIGNATURES = {
    "rk_create": (C.c_int, [C.c_int, C.POINTER(_H)]),
    "rk_destroy": (C.c_int, [_H]),
    "rk_last_error": (C.c_char_p, []),
    "rk_version": (C.c_char_p, []),
    "rk_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rk_device_name": (C.c_int, [_H, C.c_char_p, C.c_size_t]),
    "rk_alloc": (C.c_int, [_H, C.c_size_t, C.POINTER(_P)]),
    "rk_free": (C.c_int, [_H, _P]),
    "rk_memset": (C.c_int, [_H, _P, C.c_int, C.c_size_t]),
    "rk_h2d": (C.c_int, [_H, _P, _P, C.c_size_t]),
    "rk_d2h": (C.c_int, [_H, _P, _P, C.c_size_t]),
    "rk_d2d": (C.c_int, [_H, _P, _P, C.c_size_t]),
    "rk_sync": (C.c_int, [_H]),
    "rk_timer_start": (C.c_int, [_H]),
    "rk_timer_stop": (C.c_int, [_H, C.POINTER(_D)]),
    "rk_profile_enable": (C.c_int, [_H, C.c_int]),
    "rk_profile_last": (C.c_int, [_H, C.c_int, C.POINTER(C.c_char_p), C.POINTER(_D), C.POINTER(C.c_int)]),
    "rk_register_rhs_source": (C.c_int, [C.c_char_p, C.c_char_p, _I, _I, C.POINTER(_I)]),
    "rk_register_rhs_source_m": (C.c_int, [C.c_char_p, C.c_char_p, _I, _I, _I, C.POINTER(C.c_int32)]),
    "rk_rhs_compile_check": (C.c_int, [_I, _I, _I]),
    "rk_solve_layout": (C.c_int, [C.POINTER(SolveCfg), _I, C.POINTER(_I)]),
    "rk_solve_sizes": (C.c_int, [C.POINTER(SolveCfg), _I, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rk_solve_workspace_bytes": (C.c_int, [C.POINTER(SolveCfg), _I, C.POINTER(C.c_size_t)]),
    "rk_solve_filter": (C.c_int, [_H, C.POINTER(SolveCfg), C.POINTER(SolveIn), C.POINTER(SolveOut)]),
    "rk_solve_mv": (C.c_int, [_H, C.POINTER(SolveCfg), C.POINTER(SolveIn), C.POINTER(SolveOut)]),
    "rk_solve_sim": (C.c_int, [_H, C.POINTER(SolveCfg), C.POINTER(SolveIn), C.POINTER(SolveOut)]),
    "rk_gauss_obs_logpost": (C.c_int, [_H, _I, _I, _I, _I, _I, _P, _P, _P, _I, _D, _P, _I, _D, _P]),
    "rk_solve_sim_logpost": (C.c_int, [_H, C.POINTER(SolveCfg), C.POINTER(SolveIn), C.POINTER(SolveOut), _P, _P, _I, _D, _P, _I,
                                       _D, _P]),
    "rk_fenrir_backward": (C.c_int, [_H, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "rk_fenrir_workspace_bytes": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "rk_fenrir_solve_mv": (C.c_int, [_H, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "rk_fenrir_solve_mv_tiles": (C.c_int, [_H, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "rk_kalman_predict_batched": (C.c_int, [_H, C.POINTER(OpCfg)] + [_P] * 7),
    "rk_kalman_update_batched": (C.c_int, [_H, C.POINTER(OpCfg)] + [_P] * 8),
    "rk_kalman_filter_batched": (C.c_int, [_H, C.POINTER(OpCfg)] + [_P] * 13),
    "rk_kalman_smooth_mv_batched": (C.c_int, [_H, C.POINTER(OpCfg)] + [_P] * 10),
    "rk_kalman_smooth_sim_batched": (C.c_int, [_H, C.POINTER(OpCfg)] + [_P] * 9),
    "rk_kalman_smooth_batched": (C.c_int, [_H, C.POINTER(OpCfg)] + [_P] * 13),
    "rk_kalman_forecast_batched": (C.c_int, [_H, C.POINTER(OpCfg)] + [_P] * 7),
    "rk_kalman_smooth_cond_batched": (C.c_int, [_H, C.POINTER(OpCfg)] + [_P] * 9),
    "rk_interrogate_batched": (C.c_int, [_H, C.POINTER(SolveCfg), C.POINTER(SolveIn), _D, _I] + [_P] * 5),
    "rk_comm_uid": (C.c_int, [_P]),
    "rk_comm_init": (C.c_int, [_H, C.c_int, C.c_int, _P]),
    "rk_comm_destroy": (C.c_int, [_H]),
    "rk_allgather_f64": (C.c_int, [_H, _P, _P, C.c_size_t]),
    "rk_allreduce_max_f64": (C.c_int, [_H, _P, _P, C.c_size_t]),
    "rk_comm_barrier": (C.c_int, [_H]),
}

_lib = None


def load():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library has not been built.  rodeo_amd has no CPU fallback; "
            "run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C rodeo_amd/csrc`.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != RK_OK:
        msg = load().rk_last_error()
        raise RodeoKalmanError(code, msg.decode() if msg else "")
    return code
