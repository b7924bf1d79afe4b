// Register-resident square-root Kalman step maps (n_bmeas = 1), one (trajectory, block) per lane: the device form of
// src/rodeo/kalmantv/square_root.py and src/rodeo/utils.py:10-24 (add_sqrt).  NumPy mirror: oracle/sqrt_ops.py.
// Every `L*` argument is a (lower) square-root factor; factors are unique only up to column signs (like the R of
// jnp.linalg.qr), so parity is asserted on L L^T.
#pragma once
#include "linalg_small.hpp"

namespace rk {

// F (P x P, lower) with F F^T = A A^T + B B^T for A (P x KA), B (P x KB): Householder QR of the stacked [A^T; B^T],
// F = R^T (utils.py:22-24).
template <int P, int KA, int KB>
__device__ __forceinline__ void add_sqrt(const double (&A)[P][KA], const double (&B)[P][KB], double (&F)[P][P]) {
    constexpr int R = KA + KB;
    double S[R][P];
#pragma unroll
    for (int i = 0; i < KA; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) S[i][j] = A[j][i];
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) S[KA + i][j] = B[j][i];
#pragma unroll
    for (int c = 0; c < P; ++c) {
        if (c < R) {
            double tail2 = 0.0;
#pragma unroll
            for (int i = c + 1; i < R; ++i) tail2 = fma(S[i][c], S[i][c], tail2);
            const double nrm = fast_sqrt_pos(fma(S[c][c], S[c][c], tail2));      // (<= 1 ulp, linalg_small.hpp: the Householder
                                                                                  //  columns are one dependent chain, sqrt() + a
                                                                                  //  division were 45 of its ~60 instructions)
            const double alpha = S[c][c] >= 0.0 ? -nrm : nrm;
            const double v0 = S[c][c] - alpha;
            const double vn2 = fma(v0, v0, tail2);
            const double tau = vn2 != 0.0 ? 2.0 * fast_rcp(vn2) : 0.0;     // all-zero column: no reflection
#pragma unroll
            for (int j = c + 1; j < P; ++j) {
                double d = v0 * S[c][j];
#pragma unroll
                for (int i = c + 1; i < R; ++i) d = fma(S[i][c], S[i][j], d);
                d *= tau;
                S[c][j] = fma(-d, v0, S[c][j]);
#pragma unroll
                for (int i = c + 1; i < R; ++i) S[i][j] = fma(-d, S[i][c], S[i][j]);
            }
            S[c][c] = vn2 != 0.0 ? alpha : S[c][c];
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) F[i][j] = (j <= i && j < R) ? S[j][i] : 0.0;
}

// L X = B (L lower P x P), B (P x NR) overwritten.  The P reciprocals of the diagonal (fast_rcp, <= 1 ulp) are formed once
// instead of P NR divisions.
template <int P, int NR>
__device__ __forceinline__ void solve_lower(const double (&L)[P][P], double (&B)[P][NR]) {
    double rd[P];
#pragma unroll
    for (int i = 0; i < P; ++i) rd[i] = fast_rcp(L[i][i]);
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int i = 0; i < P; ++i) {
            double s = B[i][j];
#pragma unroll
            for (int l = 0; l < i; ++l) s = fma(-L[i][l], B[l][j], s);
            B[i][j] = s * rd[i];
        }
}
// L^T X = B
template <int P, int NR>
__device__ __forceinline__ void solve_upper_t(const double (&L)[P][P], double (&B)[P][NR]) {
    double rd[P];
#pragma unroll
    for (int i = 0; i < P; ++i) rd[i] = fast_rcp(L[i][i]);
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int i = P - 1; i >= 0; --i) {
            double s = B[i][j];
#pragma unroll
            for (int l = i + 1; l < P; ++l) s = fma(-L[l][i], B[l][j], s);
            B[i][j] = s * rd[i];
        }
}

// square_root.py:56-57
template <int P>
__device__ __forceinline__ void sqrt_predict(const double (&Q)[P][P], const double (&LR)[P][P], const double (&mu)[P],
                                             const double (&L)[P][P], double (&mup)[P], double (&Lp)[P][P]) {
    mv<P, P>(Q, mu, mup);
    double QL[P][P];
    mm<P, P, P>(Q, L, QL);
    add_sqrt<P, P, P>(QL, LR, Lp);
}

// square_root.py:88-99 for n_bmeas = 1, x_meas = 0; var_meas is the 1 x KV "factor" vm
template <int P, int KV>
__device__ __forceinline__ void sqrt_update_m1(const double (&W)[P], double a, const double (&vm)[KV],
                                               const double (&mup)[P], const double (&Lp)[P][P], double (&mu)[P],
                                               double (&L)[P][P]) {
    const double yhat = dot<P>(W, mup) + a;
    double wl[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double s = W[0] * Lp[0][j];
#pragma unroll
        for (int i = 1; i < P; ++i) s = fma(W[i], Lp[i][j], s);
        wl[j] = s;
    }
    double s2 = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) s2 = fma(wl[j], wl[j], s2);
#pragma unroll
    for (int k = 0; k < KV; ++k) s2 = fma(vm[k], vm[k], s2);
    const double sfac = fast_sqrt_pos(s2);              // |add_sqrt(W L, vm)| (1 x 1)
    const double rsf = fast_rcp(sfac);                  // (one reciprocal for the 2 P divisions by it)
    double t1[P], t2[P], t3[P], K[P];
#pragma unroll
    for (int j = 0; j < P; ++j) t1[j] = W[j] * rsf;     // solve_triangular(s, W)
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double s = t1[0] * Lp[0][j];
#pragma unroll
        for (int i = 1; i < P; ++i) s = fma(t1[i], Lp[i][j], s);
        t2[j] = s;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) t3[i] = dot<P>(t2, Lp[i]);
#pragma unroll
    for (int i = 0; i < P; ++i) K[i] = t3[i] * rsf;     // solve_triangular(s^T, .)^T
    const double innov = 0.0 - yhat;
    double KW[P][P], KWL[P][P], A1[P][P], B1[P][KV];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        mu[i] = fma(K[i], innov, mup[i]);
#pragma unroll
        for (int j = 0; j < P; ++j) KW[i][j] = K[i] * W[j];
#pragma unroll
        for (int k = 0; k < KV; ++k) B1[i][k] = K[i] * vm[k];
    }
    mm<P, P, P>(KW, Lp, KWL);
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) A1[i][j] = Lp[i][j] - KWL[i][j];
    add_sqrt<P, P, KV>(A1, B1, L);
}

// square_root.py:170-175: G = (L_p^{-T} (L_p^{-1} Q) L_f L_f^T)^T ;  also J L_f = (I - G Q) L_f (square_root.py:214-215)
template <int P>
__device__ __forceinline__ void sqrt_gain(const double (&Q)[P][P], const double (&Lf)[P][P], const double (&Lp)[P][P],
                                          double (&G)[P][P], double (&JL)[P][P]) {
    double Sf[P][P], I1[P][P], I2[P][P];
    mm_nt<P, P, P>(Lf, Lf, Sf);
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) I1[i][j] = Q[i][j];
    solve_lower<P, P>(Lp, I1);
    mm<P, P, P>(I1, Sf, I2);
    solve_upper_t<P, P>(Lp, I2);
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) G[i][j] = I2[j][i];
    double GQ[P][P];
    mm<P, P, P>(G, Q, GQ);
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) GQ[i][j] = (i == j ? 1.0 : 0.0) - GQ[i][j];
    mm<P, P, P>(GQ, Lf, JL);
}

}  // namespace rk
