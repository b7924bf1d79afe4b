"""
Stochastic block solver for ODE initial value problems on MI355X -- the drop-in for ``rodeo.solve``
(src/rodeo/solve.py): same function names, argument order, keywords and return layouts,

    solve_mv (key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
              kalman_type="standard", **params) -> (mean (N+1, d, p), var (N+1, d, p, p))     solve.py:208-302
    solve_sim(... same ...)                     -> x (N+1, d, p)                              solve.py:125-205

plus ONE extension: ``ode_init``, ``prior_pars`` (either matrix), ``ode_weight`` and every ODE parameter may carry a
leading batch axis B of independent trajectories (what a rodeo user writes as ``jax.vmap`` of the solver); outputs
then have a leading B as well.  The forward scan, the interrogation and the backward scan of ALL trajectories run
inside fused HIP kernels (rodeo_amd/csrc/solve_small.hip); nothing is computed on the host and there is no CPU
fallback.

Differences from the reference that a caller can see:
  * ``ode_fun`` is a ``rodeo_amd.ode.DeviceODE`` (device code for the right-hand side) or an ordinary Python function,
    which is traced into device code on first use (``rodeo_amd/trace.py``), and ``interrogate`` one
    of the four functions of ``rodeo_amd.interrogate`` (recognised by identity, ``functools.partial`` allowed);
  * ``key`` is an integer seed (or ``None``) for a Philox counter stream instead of a JAX threefry key -- draws have
    the reference's law, not its bit-stream ("parity unpinned" for draws).
"""
import collections
import ctypes as C
import functools
import numpy as np
from . import _lib, interrogate as _itg
from .device import default_device, batch_minor as _bm
from .ode import DeviceODE

_KALMAN = {"standard": _lib.KALMAN_STANDARD, "square-root": _lib.KALMAN_SQRT}


def _interrogate_id(interrogate):
    fn = interrogate
    bound = {}
    while isinstance(fn, functools.partial):
        bound.update(fn.keywords or {})
        fn = fn.func
    ids = {_itg.interrogate_rodeo: _lib.INTERROGATE_RODEO, _itg.interrogate_schober: _lib.INTERROGATE_SCHOBER,
           _itg.interrogate_kramer: _lib.INTERROGATE_KRAMER, _itg.interrogate_chkrebtii: _lib.INTERROGATE_CHKREBTII}
    if fn not in ids:
        raise TypeError("interrogate must be one of rodeo_amd.interrogate.interrogate_{rodeo,schober,kramer,chkrebtii} "
                        "(optionally wrapped in functools.partial); arbitrary Python callables cannot run inside the "
                        "GPU time loop")
    return ids[fn], bound


def _seed(key):
    """Integer seed from ``key``: None -> 0; ints pass; a 2-word uint32 array (a JAX-style key) is packed."""
    if key is None:
        return 0
    if isinstance(key, (int, np.integer)):
        return int(key) & 0xFFFFFFFFFFFFFFFF
    k = np.asarray(key).astype(np.uint64).ravel()
    if k.size == 2:
        return int((k[0] << np.uint64(32)) | (k[1] & np.uint64(0xFFFFFFFF)))
    raise TypeError("key must be None, an int seed, or a 2-word uint32 array")


class SolvePlan:
    """
    Device-resident form of one solver call: inputs uploaded once in the batch-minor layout, outputs allocated once.
    ``filter() / mv() / sim()`` only enqueue kernels on the device's stream (asynchronous); results stay in HBM as
    ``DeviceArray`` until ``*_host()`` is called.  This is what bench.py times.
    """

    def __init__(self, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
                 kalman_type="standard", device=None, traj_offset=0, store_pred=False, batch_minor=False, **params):
        if kalman_type not in _KALMAN:
            raise NotImplementedError                    # src/rodeo/solve.py:142-143, 240-241
        if not isinstance(ode_fun, DeviceODE):
            if not callable(ode_fun):
                raise TypeError("ode_fun must be a rodeo_amd.ode.DeviceODE or a traceable Python function: the time loop "
                                "runs on the GPU and needs device code for the right-hand side (see rodeo_amd/ode.py); "
                                "there is no CPU fallback")
            # an ordinary Python right-hand side, like the reference's: traced once into device code (rodeo_amd/trace.py)
            from .trace import from_python
            skip = {"kalman_type"}
            sizes = {k: int(np.shape(v)[-1]) if np.ndim(v) >= 1 else 1 for k, v in params.items() if k not in skip}
            ode_fun = from_python(ode_fun, int(np.shape(ode_weight)[-3]), int(np.shape(ode_weight)[-1]), **sizes)
        self.dev = device if device is not None else default_device()
        self._ode_fun = ode_fun
        itg_id, bound = _interrogate_id(interrogate)
        if itg_id == _lib.INTERROGATE_CHKREBTII:
            kt = bound.get("kalman_type", params.pop("kalman_type", None) if "kalman_type" in params else None)
            if kt is None:
                raise TypeError("interrogate_chkrebtii needs kalman_type bound with functools.partial "
                                "(src/rodeo/interrogate.py:13-15)")
            if kt not in _KALMAN:
                raise NotImplementedError                # src/rodeo/interrogate.py:43-44
            if kt != kalman_type:
                raise NotImplementedError("interrogate_chkrebtii's kalman_type must equal the solver's kalman_type")
        prior_weight, prior_var = prior_pars
        W = np.asarray(ode_weight, dtype=np.float64)
        x0 = np.asarray(ode_init, dtype=np.float64)
        Q = np.asarray(prior_weight, dtype=np.float64)
        R = np.asarray(prior_var, dtype=np.float64)
        if W.ndim not in (3, 4) or x0.ndim not in (2, 3) or Q.ndim not in (3, 4) or R.ndim not in (3, 4):
            raise ValueError("ode_weight (d,m,p), ode_init (d,p), prior_pars (d,p,p) [+ optional leading batch axis]")
        d, m, p = W.shape[-3:]
        if x0.shape[-2:] != (d, p) or Q.shape[-3:] != (d, p, p) or R.shape[-3:] != (d, p, p):
            raise ValueError(f"shape mismatch: ode_weight {W.shape}, ode_init {x0.shape}, prior {Q.shape} / {R.shape}")
        theta, Bt = ode_fun.pack_params(params)
        sizes = [a.shape[0] for a, nd in ((W, 4), (x0, 3), (Q, 4), (R, 4)) if a.ndim == nd]
        if Bt is not None:
            sizes.append(Bt)
        if len(set(sizes)) > 1:
            raise ValueError(f"inconsistent batch sizes {sizes}")
        self.batched = bool(sizes)
        B = sizes[0] if sizes else 1
        if (ode_fun.n_block, ode_fun.n_bmeas) != (d, m):
            raise ValueError(f"ODE '{ode_fun.name}' has (n_block, n_bmeas) = ({ode_fun.n_block}, {ode_fun.n_bmeas}) "
                             f"but ode_weight has ({d}, {m})")
        self.B, self.N, self.d, self.p, self.m = B, int(n_steps), d, p, m
        dev = self.dev
        self._W = dev.to_device(_bm(W, W.ndim == 4))
        self._x0 = dev.to_device(_bm(x0, x0.ndim == 3))
        self._Q = dev.to_device(_bm(Q, Q.ndim == 4))
        self._R = dev.to_device(_bm(R, R.ndim == 4))
        self._theta = dev.to_device(_bm(theta, Bt is not None)) if theta.size else None
        self.cfg = _lib.SolveCfg(n_traj=B, n_steps=self.N, n_block=d, n_bstate=p, n_bmeas=m, rhs_id=ode_fun.rhs_id,
                                 interrogate=itg_id, kalman_type=_KALMAN[kalman_type], n_theta=ode_fun.n_theta,
                                 flags=(_lib.FLAG_STORE_PRED if store_pred else 0) |
                                 (_lib.FLAG_BATCH_MINOR if batch_minor else 0), t_min=float(t_min),
                                 t_max=float(t_max), seed=0, traj_offset=int(traj_offset))
        self.inp = _lib.SolveIn(
            ode_weight=self._W.ptr, ode_weight_batched=int(W.ndim == 4),
            ode_init=self._x0.ptr, ode_init_batched=int(x0.ndim == 3),
            prior_weight=self._Q.ptr, prior_weight_batched=int(Q.ndim == 4),
            prior_var=self._R.ptr, prior_var_batched=int(R.ndim == 4),
            theta=self._theta.ptr if self._theta is not None else None, theta_batched=int(Bt is not None))
        self._store_pred = store_pred
        self.layout = None                 # layout of the last launch (RK_LAYOUT_*)
        self.last_mode = None              # RK_MODE_* of the last launch
        self._bufs = {}                    # layout -> (mean_state, var_state)
        self._ws = None                    # device scratch for the dense path
        self.mean_state = self.var_state = self.mean_pred = self.var_pred = self.x_state = None
        self._out = _lib.SolveOut()
        self.generation = 0                # bumped by every launch and every update(): lazily read results check it

    def update(self, ode_init=None, prior_pars=None, **params):
        """
        Replace inputs of the same shapes in place (device buffers and outputs are reused) -- what a sampler does between
        two log-density evaluations: new initial values, prior scales and ODE parameters for every trajectory.
        """
        self.generation += 1
        if ode_init is not None:
            x0 = np.asarray(ode_init, dtype=np.float64)
            self._x0.upload(_bm(x0, x0.ndim == 3))
        if prior_pars is not None:
            Q, R = (np.asarray(a, dtype=np.float64) for a in prior_pars)
            self._Q.upload(_bm(Q, Q.ndim == 4))
            self._R.upload(_bm(R, R.ndim == 4))
        if params:
            theta, Bt = self._ode_fun.pack_params(params)
            if self._theta is None:
                raise TypeError("this ODE has no parameters")
            self._theta.upload(_bm(theta, Bt is not None))

    def _prepare_out(self, mode):
        """Ask the library which layout this call uses and (once) allocate the outputs for it."""
        lay = C.c_int32(0)
        _lib.check(self.dev.lib.rk_solve_layout(C.byref(self.cfg), mode, C.byref(lay)))
        lay = lay.value
        dev, N1, d, p, B = self.dev, self.N + 1, self.d, self.p, self.B
        if lay not in self._bufs:
            if lay == _lib.LAYOUT_TILE3:
                mb, vb = C.c_size_t(0), C.c_size_t(0)
                _lib.check(dev.lib.rk_solve_sizes(C.byref(self.cfg), lay, C.byref(mb), C.byref(vb)))
                self._bufs[lay] = (None, dev.empty((N1, B, d, 3, 4), pad_bytes=vb.value - N1 * B * d * 96))  # + scratch tail
            elif lay == _lib.LAYOUT_TILE4:
                mb, vb = C.c_size_t(0), C.c_size_t(0)
                _lib.check(dev.lib.rk_solve_sizes(C.byref(self.cfg), lay, C.byref(mb), C.byref(vb)))
                self._bufs[lay] = (None, dev.empty((N1, B, d, 20), pad_bytes=vb.value - N1 * B * d * 160))
            elif lay == _lib.LAYOUT_TILEP:                                   # blocked tiles, n_bstate = 5 .. 8: [Sigma | mu]
                self._bufs[lay] = (None, dev.empty((N1, B, d, p * p + p)))
            elif lay == _lib.LAYOUT_TRAJ_MAJOR:
                self._bufs[lay] = (dev.empty((B, N1, d, p)), dev.empty((B, N1, d, p, p)))
            else:
                self._bufs[lay] = (dev.empty((N1, d, p, B)), dev.empty((N1, d, p, p, B)))
        self.layout = lay
        self.mean_state, self.var_state = self._bufs[lay]
        if self._store_pred and self.mean_pred is None:
            if lay == _lib.LAYOUT_TRAJ_MAJOR:                                 # the dense path keeps the reference's own layout
                self.mean_pred, self.var_pred = dev.empty((B, N1, d, p)), dev.empty((B, N1, d, p, p))
            else:
                self.mean_pred, self.var_pred = dev.empty((N1, d, p, B)), dev.empty((N1, d, p, p, B))
        if mode == _lib.MODE_SIM and self.x_state is None and not getattr(self, "_no_path", False):
            self.x_state = dev.empty((N1, d, p, B))
        wsb = C.c_size_t(0)
        _lib.check(self.dev.lib.rk_solve_workspace_bytes(C.byref(self.cfg), mode, C.byref(wsb)))
        if wsb.value and (self._ws is None or self._ws.nbytes < wsb.value):
            try:
                self._ws = dev.empty((wsb.value // 8,))
            except Exception:
                # The records of the square-root small-block path's two-kernel backward pass are OPTIONAL (include/rodeo_kalman.h:
                # without them the one-kernel form runs, same results, about twice the time) and large -- N d (3 p^2 + p) B
                # doubles, 13 GB at the headline shape with p = 8.  Every other workspace is required: re-raise.
                optional = (self.cfg.kalman_type == _lib.KALMAN_SQRT and self.layout == _lib.LAYOUT_BATCH_MINOR and
                            mode != _lib.MODE_FILTER)
                if not optional:
                    raise
                self._ws = None
        self._out = _lib.SolveOut(
            workspace=self._ws.ptr if self._ws is not None else None,
            workspace_bytes=self._ws.nbytes if self._ws is not None else 0,
            mean_state=self.mean_state.ptr if self.mean_state is not None else None, var_state=self.var_state.ptr,
            mean_pred=self.mean_pred.ptr if self.mean_pred is not None else None,
            var_pred=self.var_pred.ptr if self.var_pred is not None else None,
            x_state=self.x_state.ptr if self.x_state is not None else None)

    # ---- launches (asynchronous) ----
    def _call(self, fn, key, mode):
        self.generation += 1               # whatever is in the output buffers now belongs to an earlier call
        self._prepare_out(mode)
        self.last_mode = mode
        self.cfg.seed = _seed(key)
        _lib.check(fn(self.dev.h, C.byref(self.cfg), C.byref(self.inp), C.byref(self._out)))

    def filter(self, key=None):
        self._call(self.dev.lib.rk_solve_filter, key, _lib.MODE_FILTER)

    def mv(self, key=None):
        self._call(self.dev.lib.rk_solve_mv, key, _lib.MODE_MV)

    def sim(self, key=None):
        self._call(self.dev.lib.rk_solve_sim, key, _lib.MODE_SIM)

    def sync(self):
        self.dev.sync()

    # ---- results in the reference's layouts ----
    def _host(self, arr):
        a = arr.batch_first()                       # (B, N+1, ...)
        return a if self.batched else a[0]

    def state_host(self):
        """(mean, var) of the last filter() / mv() in the reference layout (views of one download, no transposes)."""
        if self.layout == _lib.LAYOUT_TILE3:
            t = np.moveaxis(self.var_state.to_host(), 1, 0)         # (B, N+1, d, 3, 4): rows [Sigma | mu]
            mean, var = t[..., 3], t[..., :3]
            return (mean, var) if self.batched else (mean[0], var[0])
        if self.layout == _lib.LAYOUT_TILE4:
            t = np.moveaxis(self.var_state.to_host(), 1, 0)         # (B, N+1, d, 20): [Sigma (16) | mu (4)]
            mean, var = t[..., 16:], t[..., :16].reshape(t.shape[:-1] + (4, 4))
            return (mean, var) if self.batched else (mean[0], var[0])
        if self.layout == _lib.LAYOUT_TILEP:
            p = self.p
            t = np.moveaxis(self.var_state.to_host(), 1, 0)         # (B, N+1, d, p*p + p): [Sigma row-major | mu]
            mean, var = t[..., p * p:], t[..., :p * p].reshape(t.shape[:-1] + (p, p))
            return (mean, var) if self.batched else (mean[0], var[0])
        if self.layout == _lib.LAYOUT_TRAJ_MAJOR:                    # already the reference layout
            mean, var = self.mean_state.to_host(), self.var_state.to_host()
            return (mean, var) if self.batched else (mean[0], var[0])
        return self._host(self.mean_state), self._host(self.var_state)

    def pred_host(self):
        if self.layout == _lib.LAYOUT_TRAJ_MAJOR:
            mean, var = self.mean_pred.to_host(), self.var_pred.to_host()
            return (mean, var) if self.batched else (mean[0], var[0])
        return self._host(self.mean_pred), self._host(self.var_pred)

    def x_host(self):
        return self._host(self.x_state)

    # algorithmic HBM bytes per trajectory-step
    def bytes_per_traj_step(self, kind="mv"):
        d, p = self.d, self.p
        return 3 * d * p * (p + 1) * 8 if kind == "mv" else (2 * d * p * (p + 1) + d * p) * 8


_plan_cache = collections.OrderedDict()     # a few device-resident plans, reused when only the numbers change


def cached_plan(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type="standard",
                store_pred=False, batch_minor=False, **params):
    """
    The device-resident ``SolvePlan`` of an earlier call with the same static configuration (ODE, grid, interrogation,
    weight matrix, array shapes, flags) with its inputs replaced in place (``SolvePlan.update``), or a new one.  For the
    callers that bring back a few doubles per trajectory and are called in a loop (``inference.basic``, ``fenrir``):
    rebuilding the plan would cost more than the kernels.  At most four plans are kept.
    """
    W = np.asarray(ode_weight, dtype=np.float64)
    shapes = tuple((k, np.shape(v)) for k, v in sorted(params.items()))
    itg = getattr(interrogate, "func", interrogate), tuple(sorted(getattr(interrogate, "keywords", {}).items()))
    key = (id(ode_fun), W.shape, W.tobytes(), np.shape(ode_init), tuple(np.shape(a) for a in prior_pars), float(t_min),
           float(t_max), int(n_steps), itg, kalman_type, bool(store_pred), bool(batch_minor), shapes)
    plan = _plan_cache.get(key)
    if plan is not None:
        _plan_cache.move_to_end(key)
        plan.update(ode_init=ode_init, prior_pars=prior_pars, **params)
        return plan
    plan = SolvePlan(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type,
                     store_pred=store_pred, batch_minor=batch_minor, **params)
    plan._keep_ode = ode_fun                 # the key holds id(ode_fun): keep the object alive with the plan
    _plan_cache[key] = plan
    while len(_plan_cache) > 4:
        _plan_cache.popitem(last=False)
    return plan


def _solve_filter(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate,
                  prior_weight, prior_var, kalman_funs=None, **params):
    """
    Forward pass (src/rodeo/solve.py:31-122).  Returns the reference's dict:
    ``{"state_pred": (mean, var), "state_filt": (mean, var)}``, each of time length ``n_steps + 1`` with index 0 equal
    to ``(ode_init, 0)``.  ``kalman_funs`` may be the module ``rodeo_amd.kalmantv.standard`` (default).
    """
    kalman_type = "standard"
    if kalman_funs is not None and getattr(kalman_funs, "KALMAN_TYPE", "standard") != "standard":
        kalman_type = kalman_funs.KALMAN_TYPE
    plan = SolvePlan(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, (prior_weight, prior_var),
                     kalman_type, store_pred=True, **params)
    plan.filter(key)
    return {"state_pred": plan.pred_host(), "state_filt": plan.state_host()}


def _pad_two_to_three(ode_weight, ode_init, prior_pars):
    """
    n_bstate = 2 has no MFMA-tile kernels of its own; it runs on the n_bstate = 3 ones with a third state component that
    is decoupled from the other two: Q = diag(Q, 1), R = diag(R, 1), zero weight, zero initial value.  Its cross
    covariances start as exact zeros and stay exact zeros (products with and sums of zeros), the pivoted LU never picks
    its row for the other two columns, and the draws of the first two components use the same normals -- so the leading
    2 x 2 part is the n_bstate = 2 computation, term for term.
    """
    W, x0 = np.asarray(ode_weight, dtype=np.float64), np.asarray(ode_init, dtype=np.float64)
    Q, R = (np.asarray(a, dtype=np.float64) for a in prior_pars)

    def grow(M):                                        # (..., 2, 2) -> (..., 3, 3) with a one in the corner
        out = np.zeros(M.shape[:-2] + (3, 3))
        out[..., :2, :2] = M
        out[..., 2, 2] = 1.0
        return out
    W3 = np.concatenate([W, np.zeros(W.shape[:-1] + (1,))], axis=-1)
    x3 = np.concatenate([x0, np.zeros(x0.shape[:-1] + (1,))], axis=-1)
    return W3, x3, (grow(Q), grow(R))


def _two_state_on_tiles(ode_weight, kalman_type):
    W = np.shape(ode_weight)
    return kalman_type == "standard" and len(W) >= 3 and W[-1] == 2 and W[-2] == 1


def solve_mv(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
             kalman_type="standard", **params):
    """Mean and variance of the stochastic ODE solver (src/rodeo/solve.py:208-302)."""
    if _two_state_on_tiles(ode_weight, kalman_type):
        W3, x3, pp3 = _pad_two_to_three(ode_weight, ode_init, prior_pars)
        m, v = solve_mv(key, ode_fun, W3, x3, t_min, t_max, n_steps, interrogate, pp3, kalman_type, **params)
        return np.ascontiguousarray(m[..., :2]), np.ascontiguousarray(v[..., :2, :2])
    plan = SolvePlan(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type,
                     **params)
    plan.mv(key)
    return plan.state_host()


def solve_sim(key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars,
              kalman_type="standard", **params):
    """Draw one sample solution per trajectory (src/rodeo/solve.py:125-205)."""
    if _two_state_on_tiles(ode_weight, kalman_type):
        W3, x3, pp3 = _pad_two_to_three(ode_weight, ode_init, prior_pars)
        return np.ascontiguousarray(solve_sim(key, ode_fun, W3, x3, t_min, t_max, n_steps, interrogate, pp3, kalman_type,
                                              **params)[..., :2])
    plan = SolvePlan(ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate, prior_pars, kalman_type,
                     **params)
    plan.sim(key)
    return plan.x_host()
