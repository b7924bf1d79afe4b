// Forward-mode dual numbers for user-supplied right-hand sides: the device counterpart of the
// jax.jacfwd(ode_fun) call of src/rodeo/interrogate.py:76, restricted to what interrogate_kramer keeps -- the
// block-diagonal d f_b / d X[b][:] (interrogate.py:78-79).
//
// A user ODE written once on a generic scalar type,
//     struct MyOde {
//         static constexpr int D = ..., NTHETA = ..., NDEP = 1;
//         template <class T, int P> __device__ static void rhs(const T (&X)[D][P], double t,
//                                                              const double (&th)[NTHETA], T (&out)[D]);
//     };
// becomes a complete right-hand side (f and fjac) through rk::AutoJac<MyOde>.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace rk {

template <int P>
struct Dual {
    double v;
    double d[P];
    __device__ __forceinline__ Dual() : v(0.0) {
#pragma unroll
        for (int i = 0; i < P; ++i) d[i] = 0.0;
    }
    __device__ __forceinline__ Dual(double x) : v(x) {
#pragma unroll
        for (int i = 0; i < P; ++i) d[i] = 0.0;
    }
};

#define RK_DUAL_BIN(OP, VEXPR, DEXPR)                                                                    \
    template <int P>                                                                                      \
    __device__ __forceinline__ Dual<P> operator OP(const Dual<P>& a, const Dual<P>& b) {                  \
        Dual<P> r;                                                                                        \
        r.v = VEXPR;                                                                                      \
        _Pragma("unroll") for (int i = 0; i < P; ++i) r.d[i] = DEXPR;                                     \
        return r;                                                                                         \
    }

RK_DUAL_BIN(+, a.v + b.v, a.d[i] + b.d[i])
RK_DUAL_BIN(-, a.v - b.v, a.d[i] - b.d[i])
RK_DUAL_BIN(*, a.v * b.v, fma(a.d[i], b.v, a.v * b.d[i]))
RK_DUAL_BIN(/, a.v / b.v, (a.d[i] - (a.v / b.v) * b.d[i]) / b.v)
#undef RK_DUAL_BIN

// a / b with a reciprocal and one correction (Markstein): q = a y, q + (a - b q) y with y = 1 / b.  Within an ulp of the
// IEEE quotient (exact for most a, for all a when b = 3); y is a constant when b is, and loop-invariant when b is a
// parameter -- three operations on the step's dependent chain instead of the eleven of a full division.
__device__ __forceinline__ double div_by_scalar(double a, double b, double y) {
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);
}

// Mixed operations with a plain double keep the structural zeros of the constant's derivative out of the arithmetic
// (0 * x cannot be folded by the compiler under IEEE rules).
template <int P>
__device__ __forceinline__ Dual<P> operator+(const Dual<P>& a, double b) { Dual<P> r = a; r.v = a.v + b; return r; }
template <int P>
__device__ __forceinline__ Dual<P> operator+(double a, const Dual<P>& b) { Dual<P> r = b; r.v = a + b.v; return r; }
template <int P>
__device__ __forceinline__ Dual<P> operator-(const Dual<P>& a, double b) { Dual<P> r = a; r.v = a.v - b; return r; }
template <int P>
__device__ __forceinline__ Dual<P> operator-(double a, const Dual<P>& b) {
    Dual<P> r;
    r.v = a - b.v;
#pragma unroll
    for (int i = 0; i < P; ++i) r.d[i] = -b.d[i];
    return r;
}
template <int P>
__device__ __forceinline__ Dual<P> operator*(const Dual<P>& a, double b) {
    Dual<P> r;
    r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < P; ++i) r.d[i] = a.d[i] * b;
    return r;
}
template <int P>
__device__ __forceinline__ Dual<P> operator*(double a, const Dual<P>& b) { return b * a; }
template <int P>
__device__ __forceinline__ Dual<P> operator/(const Dual<P>& a, double b) {
    const double y = 1.0 / b;
    Dual<P> r;
    r.v = div_by_scalar(a.v, b, y);
#pragma unroll
    for (int i = 0; i < P; ++i) r.d[i] = div_by_scalar(a.d[i], b, y);
    return r;
}
template <int P>
__device__ __forceinline__ Dual<P> operator/(double a, const Dual<P>& b) {
    Dual<P> r;
    r.v = a / b.v;
    const double f = -r.v / b.v;
#pragma unroll
    for (int i = 0; i < P; ++i) r.d[i] = f * b.d[i];
    return r;
}

template <int P>
__device__ __forceinline__ Dual<P> operator-(const Dual<P>& a) {
    Dual<P> r;
    r.v = -a.v;
#pragma unroll
    for (int i = 0; i < P; ++i) r.d[i] = -a.d[i];
    return r;
}

// plain-double overloads so that generic code inside namespace rk can call sin(x), exp(x), ... on either scalar type
__device__ __forceinline__ double sin(double x) { return ::sin(x); }
__device__ __forceinline__ double cos(double x) { return ::cos(x); }
__device__ __forceinline__ double exp(double x) { return ::exp(x); }
__device__ __forceinline__ double log(double x) { return ::log(x); }
__device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
__device__ __forceinline__ double tanh(double x) { return ::tanh(x); }
__device__ __forceinline__ double tan(double x) { return ::tan(x); }
__device__ __forceinline__ double sinh(double x) { return ::sinh(x); }
__device__ __forceinline__ double cosh(double x) { return ::cosh(x); }
__device__ __forceinline__ double atan(double x) { return ::atan(x); }
__device__ __forceinline__ double asin(double x) { return ::asin(x); }
__device__ __forceinline__ double acos(double x) { return ::acos(x); }
__device__ __forceinline__ double log1p(double x) { return ::log1p(x); }
__device__ __forceinline__ double expm1(double x) { return ::expm1(x); }

#define RK_DUAL_FUN(NAME, VEXPR, DFAC)                                      \
    template <int P>                                                         \
    __device__ __forceinline__ Dual<P> NAME(const Dual<P>& a) {              \
        Dual<P> r;                                                           \
        r.v = VEXPR;                                                         \
        const double fac = DFAC;                                             \
        _Pragma("unroll") for (int i = 0; i < P; ++i) r.d[i] = fac * a.d[i]; \
        return r;                                                            \
    }
RK_DUAL_FUN(sin, ::sin(a.v), ::cos(a.v))
RK_DUAL_FUN(cos, ::cos(a.v), -::sin(a.v))
RK_DUAL_FUN(exp, ::exp(a.v), r.v)
RK_DUAL_FUN(log, ::log(a.v), 1.0 / a.v)
RK_DUAL_FUN(sqrt, ::sqrt(a.v), 0.5 / r.v)
RK_DUAL_FUN(tanh, ::tanh(a.v), 1.0 - r.v * r.v)
RK_DUAL_FUN(tan, ::tan(a.v), 1.0 + r.v * r.v)
RK_DUAL_FUN(sinh, ::sinh(a.v), ::cosh(a.v))
RK_DUAL_FUN(cosh, ::cosh(a.v), ::sinh(a.v))
RK_DUAL_FUN(atan, ::atan(a.v), 1.0 / (1.0 + a.v * a.v))
RK_DUAL_FUN(asin, ::asin(a.v), 1.0 / ::sqrt(1.0 - a.v * a.v))
RK_DUAL_FUN(acos, ::acos(a.v), -1.0 / ::sqrt(1.0 - a.v * a.v))
RK_DUAL_FUN(log1p, ::log1p(a.v), 1.0 / (1.0 + a.v))
RK_DUAL_FUN(expm1, ::expm1(a.v), r.v + 1.0)
#undef RK_DUAL_FUN

// Adapter: a scalar-generic `rhs` -> the (f, fjac) interface of csrc/rhs.hpp
template <class U>
struct AutoJac {
    static constexpr int D = U::D;
    static constexpr int NTHETA = U::NTHETA;
    static constexpr int NDEP = U::NDEP;
    static constexpr bool HAS_TILE_FORM = false;
    static constexpr bool HAS_TILE3_FORM = false;
    template <int P>
    __device__ __forceinline__ static void f(const double (&X)[D][P], double t, const double (&th)[NTHETA],
                                             double (&out)[D]) {
        U::template rhs<double, P>(X, t, th, out);
    }
    template <int P>
    __device__ __forceinline__ static void fjac(const double (&X)[D][P], double t, const double (&th)[NTHETA],
                                                double (&out)[D], double (&J)[D][P]) {
#pragma unroll
        for (int b = 0; b < D; ++b) {
            // seed the P directions of block b; the other blocks are constants (their partials are dropped)
            Dual<P> Xd[D][P], od[D];
#pragma unroll
            for (int bb = 0; bb < D; ++bb)
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    Xd[bb][j] = Dual<P>(X[bb][j]);
                    if (bb == b) Xd[bb][j].d[j] = 1.0;
                }
            U::template rhs<Dual<P>, P>(Xd, t, th, od);
            out[b] = od[b].v;
#pragma unroll
            for (int j = 0; j < P; ++j) J[b][j] = od[b].d[j];
        }
    }
    // What the MFMA-tile kernels need (NDEP == 1, every lane works for ONE block): f_blk and d f_blk / d X[blk][0] from a
    // single evaluation with one dual direction, instead of D evaluations with P directions each.
    static constexpr bool HAS_FJAC0 = true;
    template <class V, class = void> struct has_one_ { static constexpr bool value = false; };
    template <class V> struct has_one_<V, decltype((void)V::HAS_RHS_ONE)> { static constexpr bool value = V::HAS_RHS_ONE; };
    // a right-hand side with `rhs_one` (rodeo_amd.trace writes it): the lane's OWN block's output alone -- the full evaluation
    // computes all D outputs in every lane and throws D - 1 of them away (a traced ring of 32 variables: 176 us per forward step)
    static constexpr bool HAS_F_BLOCK = has_one_<U>::value;
    template <int P>
    __device__ __forceinline__ static void fjac0_block(const double (&X)[D][P], double t, const double (&th)[NTHETA],
                                                       int blk, double& fb, double& J0) {
        Dual<1> Xd[D][P];
#pragma unroll
        for (int bb = 0; bb < D; ++bb) {
#pragma unroll
            for (int j = 0; j < P; ++j) Xd[bb][j] = Dual<1>(X[bb][j]);
            Xd[bb][0].d[0] = bb == blk ? 1.0 : 0.0;
        }
        if constexpr (has_one_<U>::value) {
            Dual<1> o1;
            U::template rhs_one<Dual<1>, P>(blk, Xd, t, th, o1);
            fb = o1.v; J0 = o1.d[0];
        } else {
            Dual<1> od[D];
            U::template rhs<Dual<1>, P>(Xd, t, th, od);
            fb = od[0].v; J0 = od[0].d[0];
#pragma unroll
            for (int bb = 1; bb < D; ++bb) {
                fb = bb == blk ? od[bb].v : fb;
                J0 = bb == blk ? od[bb].d[0] : J0;
            }
        }
    }
    template <int P>
    __device__ __forceinline__ static double f_block(const double (&X)[D][P], double t, const double (&th)[NTHETA], int blk) {
        double o1 = 0.0;
        if constexpr (has_one_<U>::value) U::template rhs_one<double, P>(blk, X, t, th, o1);
        return o1;
    }
};

}  // namespace rk
