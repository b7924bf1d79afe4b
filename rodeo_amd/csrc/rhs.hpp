// Device code for the `ode_fun(X, t, **params)` callable of src/rodeo/solve.py:218 and for the block-diagonal of
// its Jacobian that interrogate_kramer extracts from jax.jacfwd (src/rodeo/interrogate.py:76-79).
//
// A right-hand side is a struct with
//   D       n_block,  NTHETA  number of packed parameters per trajectory  (n_bmeas = 1 for all of these)
//   f   (X[D][P], t, th[NTHETA], out[D])              out[b]    = f_b(X, t)
//   fjac(X[D][P], t, th[NTHETA], out[D], J[D][P])     J[b][j]   = d f_b / d X[b][j]   (off-block terms dropped,
//                                                                  interrogate.py:70)
// NumPy mirrors used by the tests: oracle/odes.py.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace rk {

// x / 3, correctly rounded, without the ~12-instruction IEEE division sequence: q = x * RN(1/3) followed by one
// fused Newton correction with the exact residual (Markstein's division-by-constant; 3 has no all-ones mantissa).
__device__ __forceinline__ double div3(double x) {
    const double c = 0.33333333333333331;
    const double q = x * c;
    return fma(fma(-3.0, q, x), c, q);
}

// FitzHugh-Nagumo, README.md:92-99 / docs/examples/parameter.md:60-68.  theta = (a, b, c).
struct FitzHughNagumo {
    static constexpr int D = 2;
    static constexpr int NTHETA = 3;
    static constexpr int NDEP = 1;     // f and J depend only on the first NDEP entries X[b][0..NDEP) of every block
    template <int P>
    __device__ __forceinline__ static void f(const double (&X)[D][P], double, const double (&th)[NTHETA],
                                             double (&out)[D]) {
        const double a = th[0], b = th[1], c = th[2];
        const double V = X[0][0], R = X[1][0];
        out[0] = c * (V - div3(V * V * V) + R);
        out[1] = -1 / c * (V - a + b * R);
    }
    template <int P>
    __device__ __forceinline__ static void fjac(const double (&X)[D][P], double t, const double (&th)[NTHETA],
                                                double (&out)[D], double (&J)[D][P]) {
        f<P>(X, t, th, out);
#pragma unroll
        for (int b = 0; b < D; ++b)
#pragma unroll
            for (int j = 0; j < P; ++j) J[b][j] = 0.0;
        const double V = X[0][0];
        J[0][0] = th[2] * (1.0 - V * V);
        J[1][0] = -th[1] / th[2];
    }
    // Optional "tile form" used by the MFMA-tile forward kernel (one (trajectory, block) per 16 lanes): every lane
    // evaluates only ITS block's f_b and J_b0 from own = X[b][0] and oth = X[1-b][0] through per-lane coefficients,
    //     f_b = k0 own + k1 own^3 + k2 oth + k3 ,   d f_b / d X[b][0] = k4 + k5 own^2 ,
    // instead of both blocks plus selects (11 fewer VALU instructions per step).  Same function as f / fjac; the
    // association differs from the reference expression at rounding level (c V - (c/3) V^3 + c R).
    static constexpr bool HAS_TILE_FORM = true;
    static constexpr bool HAS_TILE3_FORM = false;
    static constexpr int NTILEK = 6;
    __device__ __forceinline__ static void tile_consts(int blk, const double (&th)[NTHETA], double (&k)[NTILEK]) {
        const double a = th[0], b = th[1], c = th[2];
        if (blk == 0) { k[0] = c; k[1] = -c / 3; k[2] = c; k[3] = 0.0; k[4] = c; k[5] = -c; }
        else { k[0] = -b / c; k[1] = 0.0; k[2] = -1 / c; k[3] = a / c; k[4] = -b / c; k[5] = 0.0; }
    }
    __device__ __forceinline__ static void tile_eval(const double (&k)[NTILEK], double own, double oth, double,
                                                     double& f, double& J0) {
        const double o2 = own * own;
        f = fma(k[1], o2 * own, fma(k[0], own, fma(k[2], oth, k[3])));
        J0 = fma(k[5], o2, k[4]);
    }
};

// Lorenz63, docs/examples/lorenz.md:85-92.  theta = (rho, sigma, beta).
struct Lorenz63 {
    static constexpr int D = 3;
    static constexpr int NTHETA = 3;
    static constexpr int NDEP = 1;     // f and J depend only on the first NDEP entries X[b][0..NDEP) of every block
    template <int P>
    __device__ __forceinline__ static void f(const double (&X)[D][P], double, const double (&th)[NTHETA],
                                             double (&out)[D]) {
        const double rho = th[0], sigma = th[1], beta = th[2];
        const double x = X[0][0], y = X[1][0], z = X[2][0];
        out[0] = -sigma * x + sigma * y;
        out[1] = rho * x - y - x * z;
        out[2] = -beta * z + x * y;
    }
    template <int P>
    __device__ __forceinline__ static void fjac(const double (&X)[D][P], double t, const double (&th)[NTHETA],
                                                double (&out)[D], double (&J)[D][P]) {
        f<P>(X, t, th, out);
#pragma unroll
        for (int b = 0; b < D; ++b)
#pragma unroll
            for (int j = 0; j < P; ++j) J[b][j] = 0.0;
        J[0][0] = -th[1];
        J[1][0] = -1.0;
        J[2][0] = -th[2];
    }
    static constexpr bool HAS_TILE_FORM = false;
    // Optional three-block "tile form" used by the p = 4 MFMA-tile forward kernel: every lane evaluates only ITS block's
    // f_b from its own first state and those of the next / previous / second-previous block of the trajectory
    // (n1, p1, p2; cyclic in the DPP row, unused neighbours have zero coefficients),
    //     f_b = k0 own + k1 n1 + k2 p1 + k3 (p1 n1) + k4 (p2 p1),      d f_b / d X[b][0] = k0  (a constant),
    // instead of all three blocks plus selects.  Same function as f / fjac.
    static constexpr bool HAS_TILE3_FORM = true;
    __device__ __forceinline__ static void tile3_consts(int blk, const double (&th)[NTHETA], double (&k)[5]) {
        const double rho = th[0], sigma = th[1], beta = th[2];
        // (plain selects: an if / else-if chain over a zero-initialised array came out of hipcc with block 2's k0 = -1)
        k[0] = blk == 0 ? -sigma : (blk == 1 ? -1.0 : -beta);      // x' = -sigma x + sigma y
        k[1] = blk == 0 ? sigma : 0.0;                              // y' = -y + rho x - x z
        k[2] = blk == 1 ? rho : 0.0;                                // z' = -beta z + x y
        k[3] = blk == 1 ? -1.0 : 0.0;
        k[4] = blk == 2 ? 1.0 : 0.0;
    }
};

// Second-order ODE of Chkrebtii et al, docs/examples/higher_order.md:47-59:  x'' = sin(2t) - x.
struct HigherOrder {
    static constexpr int D = 1;
    static constexpr int NTHETA = 1;   // unused (one dummy slot keeps the array types non-empty)
    static constexpr int NDEP = 1;
    template <int P>
    __device__ __forceinline__ static void f(const double (&X)[D][P], double t, const double (&)[NTHETA],
                                             double (&out)[D]) {
        out[0] = sin(2 * t) - X[0][0];
    }
    template <int P>
    __device__ __forceinline__ static void fjac(const double (&X)[D][P], double t, const double (&th)[NTHETA],
                                                double (&out)[D], double (&J)[D][P]) {
        f<P>(X, t, th, out);
#pragma unroll
        for (int j = 0; j < P; ++j) J[0][j] = 0.0;
        J[0][0] = -1.0;
    }
    static constexpr bool HAS_TILE_FORM = false;
    static constexpr bool HAS_TILE3_FORM = false;
};

}  // namespace rk
