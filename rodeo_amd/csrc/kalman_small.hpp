// Per-block Kalman step maps on register-resident blocks (n_bmeas = 1), one (trajectory, block) per lane.
// Each function cites the reference lines it computes; NumPy mirrors: oracle/kalman_ops.py.
#pragma once
#include "linalg_small.hpp"

namespace rk {

// standard.py:57-59 with mean_state = 0 (solve.py:52):  mu- = Q mu ;  Sigma- = (Q Sigma) Q^T + R.
// Used by the forward kernel AND re-evaluated by the backward kernels (the predicted moments are not stored);
// it is written with explicit fma chains so both evaluations round identically.
template <int P>
__device__ __forceinline__ void predict_block(const double (&Q)[P][P], const double (&R)[P][P],
                                              const double (&mu)[P], const double (&S)[P][P],
                                              double (&mup)[P], double (&Sp)[P][P]) {
    mv<P, P>(Q, mu, mup);
    double A[P][P];
    mm<P, P, P>(Q, S, A);
    mm_nt<P, P, P>(A, Q, Sp);
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) Sp[i][j] = Sp[i][j] + R[i][j];
}

// standard.py:93-102 for n_bmeas = 1, x_meas = 0 (solve.py:51):
//   yhat = W mu- + a ; S = (W Sigma-) W^T + V ; K = Sigma- W^T / S ; mu = mu- + K (0 - yhat) ; Sigma = Sigma- - K (W Sigma-)
template <int P>
__device__ __forceinline__ void update_block_m1(const double (&W)[P], double a, double V,
                                                const double (&mup)[P], const double (&Sp)[P][P],
                                                double (&mu)[P], double (&S)[P][P]) {
    const double yhat = dot<P>(W, mup) + a;
    double WS[P], SW[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double s = W[0] * Sp[0][j];
#pragma unroll
        for (int i = 1; i < P; ++i) s = fma(W[i], Sp[i][j], s);
        WS[j] = s;
    }
    const double Smm = dot<P>(WS, W) + V;
#pragma unroll
    for (int i = 0; i < P; ++i) SW[i] = dot<P>(Sp[i], W);
    const double rS = fast_rcp(Smm);
    const double innov = 0.0 - yhat;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const double K = SW[i] * rS;
        mu[i] = fma(K, innov, mup[i]);
#pragma unroll
        for (int j = 0; j < P; ++j) S[i][j] = fma(-K, WS[j], Sp[i][j]);
    }
}

// standard.py:175-176:  T = Sigma_f Q^T ;  G = solve(Sigma-, T^T)^T  (LU with partial pivoting, utils.py:119)
template <int P>
__device__ __forceinline__ void smooth_gain(const double (&Q)[P][P], const double (&Sf)[P][P],
                                            const double (&Sp)[P][P], double (&T)[P][P], double (&G)[P][P]) {
    mm_nt<P, P, P>(Sf, Q, T);
    double A[P][P], X[P][P];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) {
            A[i][j] = Sp[i][j];
            X[i][j] = T[j][i];
        }
    lu_solve<P, P>(A, X);
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) G[i][j] = X[j][i];
}

// standard.py:213-216:  mu_s = mu_f + G (mu_next - mu-) ;  Sigma_s = Sigma_f + (G (Sigma_next - Sigma-)) G^T
template <int P>
__device__ __forceinline__ void smooth_mv_block(const double (&G)[P][P], const double (&mf)[P],
                                                const double (&Sf)[P][P], const double (&mp)[P],
                                                const double (&Sp)[P][P], double (&ms)[P], double (&Ss)[P][P]) {
    double dm[P], D[P][P], GD[P][P], GDG[P][P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        dm[i] = ms[i] - mp[i];
#pragma unroll
        for (int j = 0; j < P; ++j) D[i][j] = Ss[i][j] - Sp[i][j];
    }
    double gm[P];
    mv<P, P>(G, dm, gm);
    mm<P, P, P>(G, D, GD);
    mm_nt<P, P, P>(GD, G, GDG);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        ms[i] = mf[i] + gm[i];
#pragma unroll
        for (int j = 0; j < P; ++j) Ss[i][j] = Sf[i][j] + GDG[i][j];
    }
}

// standard.py:251-254:  mean_sim = mu_f + G (x_next - mu-) ;  var_sim = Sigma_f - G T^T
template <int P>
__device__ __forceinline__ void smooth_sim_block(const double (&G)[P][P], const double (&T)[P][P],
                                                 const double (&mf)[P], const double (&Sf)[P][P],
                                                 const double (&mp)[P], const double (&xn)[P],
                                                 double (&msim)[P], double (&Ssim)[P][P]) {
    double dm[P], gm[P], GT[P][P];
#pragma unroll
    for (int i = 0; i < P; ++i) dm[i] = xn[i] - mp[i];
    mv<P, P>(G, dm, gm);
    mm_nt<P, P, P>(G, T, GT);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        msim[i] = mf[i] + gm[i];
#pragma unroll
        for (int j = 0; j < P; ++j) Ssim[i][j] = Sf[i][j] - GT[i][j];
    }
}

// x = mean + psd_factor(var) z
template <int P>
__device__ __forceinline__ void mvn_draw(const double (&mean)[P], const double (&var)[P][P],
                                         const double (&z)[P], double (&x)[P]) {
    double L[P][P];
    psd_factor<P>(var, L);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        double s = mean[i];
#pragma unroll
        for (int k = 0; k <= i; ++k) s = fma(L[i][k], z[k], s);
        x[i] = s;
    }
}

}  // namespace rk
