"""
Python right-hand sides on the device: ``ode.from_python`` TRACES an ordinary ``ode_fun(X, t, **params)`` -- written
with NumPy exactly like the JAX functions of the reference (README.md:92-99, docs/examples/lorenz.md:85-92) -- into the
scalar-generic HIP ``rhs`` that ``ode.from_source`` compiles with hiprtc.  This is what JAX does with the callable
inside ``lax.scan`` (src/rodeo/solve.py:70-78): the function is run once on symbolic inputs; arithmetic, indexing,
``np.array([...])``, powers and the elementary functions (sin, cos, tan, exp, log, sqrt, tanh, sinh, cosh, arctan,
arcsin, arccos, log1p, expm1) are recorded; data-dependent Python control flow cannot be traced
(neither can it under ``jax.jit``).  The Jacobian that ``interrogate_kramer`` needs comes from forward-mode duals on
the generated code, the counterpart of ``jax.jacfwd`` (src/rodeo/interrogate.py:76).

The generated struct is passed through the same ``rk_register_rhs_source`` entry point as hand-written source; the
Python function itself stays the host twin used for input preparation (``first_order_pad``).
"""
import hashlib
import numbers
import numpy as np

# NumPy ufunc (method looked up on object arrays) -> device function (rodeo_amd/csrc/dual.hpp)
_FUNCS = {"sin": "sin", "cos": "cos", "tan": "tan", "exp": "exp", "log": "log", "sqrt": "sqrt", "tanh": "tanh",
          "sinh": "sinh", "cosh": "cosh", "arctan": "atan", "arcsin": "asin", "arccos": "acos", "log1p": "log1p",
          "expm1": "expm1"}


def _lit(v):
    """C++ double literal that round-trips."""
    v = float(v)
    if not np.isfinite(v):
        raise ValueError("non-finite constant in the traced right-hand side")
    s = repr(v)
    if "e" not in s and "." not in s and "n" not in s:
        s += ".0"
    return s if v >= 0 else f"({s})"


class Sym:
    """A traced scalar: a C++ expression string."""
    __array_priority__ = 1000.0
    __slots__ = ("code",)

    def __init__(self, code):
        self.code = code

    @staticmethod
    def _c(x):
        if isinstance(x, Sym):
            return x.code
        if isinstance(x, (numbers.Real, np.floating, np.integer)) and not isinstance(x, bool):
            return _lit(x)
        raise TypeError(f"cannot trace an operation with {type(x).__name__}")

    def _bin(self, other, op, swap=False):
        a, b = (Sym._c(other), self.code) if swap else (self.code, Sym._c(other))
        return Sym(f"({a} {op} {b})")

    def __add__(self, o): return self._bin(o, "+")
    def __radd__(self, o): return self._bin(o, "+", True)
    def __sub__(self, o): return self._bin(o, "-")
    def __rsub__(self, o): return self._bin(o, "-", True)
    def __mul__(self, o): return self._bin(o, "*")
    def __rmul__(self, o): return self._bin(o, "*", True)
    def __truediv__(self, o): return self._bin(o, "/")
    def __rtruediv__(self, o): return self._bin(o, "/", True)
    def __neg__(self): return Sym(f"(-{self.code})")
    def __pos__(self): return self

    def __pow__(self, n):
        if isinstance(n, (numbers.Integral, np.integer)) or (isinstance(n, float) and n == int(n)):
            n = int(n)
            if n == 0:
                return Sym("1.0")
            base, k = self, abs(n)
            out = None
            while k:                                   # square-and-multiply: x**5 = x * (x*x) * (x*x)
                if k & 1:
                    out = base if out is None else out * base
                k >>= 1
                if k:
                    base = base * base
            return out if n > 0 else 1.0 / out
        if isinstance(n, float) and n == 0.5:
            return self.sqrt()
        if isinstance(n, (float, np.floating)):
            return (float(n) * self.log()).exp()
        raise TypeError("traced power: the exponent must be a number")

    def __rpow__(self, a):                               # a ** x = exp(x log a)
        return (self * float(np.log(float(a)))).exp()

    def _cmp(self, *_):
        raise TypeError("data-dependent control flow cannot be traced into device code (the same restriction as "
                        "under jax.jit); write the right-hand side with arithmetic only")
    __lt__ = __le__ = __gt__ = __ge__ = __bool__ = _cmp
    __float__ = __int__ = _cmp


def _make_fun(cname):
    def method(self):
        return Sym(f"{cname}({self.code})")
    return method


for _n, _c in _FUNCS.items():                            # np.sin(object array) calls element.sin()
    setattr(Sym, _n, _make_fun(_c))


def _symbols(shape, fmt):
    a = np.empty(shape, dtype=object)
    for idx in np.ndindex(*shape):
        a[idx] = Sym(fmt(*idx))
    return a


def trace_source(fun, n_vars, n_deriv_used, param_spec, struct_name):
    """Run ``fun`` on symbols; returns (source, ndep, n_bmeas).  ``param_spec`` = ((name, size), ...).  A function that
    returns (n_vars, M) with M > 1 (several measurements per block, src/rodeo/solve.py:48-51 -- e.g. the "non-block" form
    of a small system) yields a struct with ``out[D][M]`` for ``rk::AutoJacM`` (csrc/solve_small_m_kernels.hpp)."""
    X = _symbols((n_vars, n_deriv_used), lambda b, j: f"X[{b}][{j}]")
    params, off = {}, 0
    for name, size in param_spec:
        params[name] = _symbols((size,), lambda k, off=off: f"th[{off + k}]")
        off += size
    out = fun(X, Sym("t"), **params)
    out = np.asarray(out, dtype=object)
    if out.shape == (n_vars,):
        out = out[:, None]
    m_max = 256 if n_vars == 1 else 4          # one block holding all variables (non-block form): the dense path beyond 4
    if out.ndim != 2 or out.shape[0] != n_vars or not 1 <= out.shape[1] <= m_max:
        raise ValueError(f"the traced ode_fun must return shape ({n_vars}, n_bmeas) with n_bmeas in 1..{m_max}, got {out.shape}")
    n_bmeas = out.shape[1]
    lines = []
    for b in range(n_vars):
        for i in range(n_bmeas):
            lines.append(f"        out[{b}]{'[%d]' % i if n_bmeas > 1 else ''} = {Sym._c(out[b, i])};")
    body = "\n".join(lines)
    # the same expressions one block at a time: on the tile kernels a lane needs the output of ITS block only, and evaluating
    # all n_vars of them in every lane made a 32-variable ring's forward step 20 x slower than its arithmetic (round 4)
    one = "\n".join(f"            case {b}: out = {Sym._c(out[b, 0])}; break;" for b in range(n_vars)) if n_bmeas == 1 else ""
    used = [j for b in range(n_vars) for j in range(n_deriv_used) if f"X[{b}][{j}]" in body]
    ndep = max(used) + 1 if used else 1
    n_theta = max(off, 0)
    out_t = "T (&out)[D][M]" if n_bmeas > 1 else "T (&out)[D]"
    m_line = f"\n    static constexpr int M = {n_bmeas};" if n_bmeas > 1 else ""
    one_fn = f"""
    static constexpr bool HAS_RHS_ONE = true;
    // output b alone (b may differ from lane to lane: the wave runs the cases its lanes ask for)
    template <class T, int P>
    __device__ __forceinline__ static void rhs_one(int b, const T (&X)[D][P], double t, const double (&th)[NTHETA], T& out) {{
        switch (b) {{
{one}
            default: break;
        }}
    }}""" if one else ""
    src = f"""
// generated by rodeo_amd.trace from the Python function {getattr(fun, '__name__', 'ode_fun')!r}
struct {struct_name} {{
    static constexpr int D = {n_vars};{m_line}
    static constexpr int NTHETA = {max(n_theta, 1)};
    static constexpr int NDEP = {ndep};
    template <class T, int P>
    __device__ __forceinline__ static void rhs(const T (&X)[D][P], double t, const double (&th)[NTHETA], {out_t}) {{
        static_assert(P >= NDEP, "the right-hand side reads more derivatives than the prior carries");
{body}
    }}{one_fn}
}};
"""
    return src, ndep, n_bmeas


def _host_twin(fun, n_vars, n_bmeas=1):
    """The Python function itself, vectorised over leading batch axes of X and of the parameters."""
    def host(X, t, **params):
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 2 and all(np.ndim(v) <= 1 for v in params.values()):
            return np.asarray(fun(X, t, **params), dtype=np.float64).reshape(n_vars, n_bmeas)
        lead = np.broadcast_shapes(X.shape[:-2], *[np.shape(v)[:-1] for v in params.values() if np.ndim(v) >= 2])
        Xb = np.broadcast_to(X, lead + X.shape[-2:])
        pb = {k: (np.broadcast_to(v, lead + np.shape(v)[-1:]) if np.ndim(v) >= 2 else v) for k, v in params.items()}
        out = np.empty(lead + (n_vars, n_bmeas))
        for idx in np.ndindex(*lead):
            out[idx] = np.asarray(fun(Xb[idx], t, **{k: (v[idx] if np.ndim(v) >= 2 else v) for k, v in pb.items()}),
                                  dtype=np.float64).reshape(n_vars, n_bmeas)
        return out
    return host


_cache = {}


def from_python(fun, n_vars, n_deriv_used=2, name=None, **param_sizes):
    """
    ``DeviceODE`` from an ordinary Python right-hand side ``fun(X, t, **params)`` (X of shape (n_block, n_bstate),
    return shape (n_block, n_bmeas); here ``n_vars`` = n_block).  ``param_sizes``: keyword -> length of that parameter vector, e.g. ``theta=3``.
    ``n_deriv_used``: how many leading derivatives the function may read (first-order ODEs read X[:, 0] only).
    """
    from . import ode
    spec = tuple((k, int(v)) for k, v in param_sizes.items())
    key = (fun, int(n_vars), int(n_deriv_used), spec)
    if key in _cache:
        return _cache[key]
    probe_src, _, _ = trace_source(fun, n_vars, n_deriv_used, spec, "TracedOde")
    tag = hashlib.sha1(probe_src.encode()).hexdigest()[:10]
    struct = f"Traced_{tag}"
    src, ndep, n_bmeas = trace_source(fun, n_vars, n_deriv_used, spec, struct)
    wrapper = "AutoJacM" if n_bmeas > 1 else "AutoJac"
    dev = ode.from_source(f"{wrapper}<{struct}>", src, n_vars, spec, _host_twin(fun, n_vars, n_bmeas),
                          name=name or getattr(fun, "__name__", struct), n_bmeas=n_bmeas)
    dev.source, dev.ndep = src, ndep
    _cache[key] = dev
    return dev
