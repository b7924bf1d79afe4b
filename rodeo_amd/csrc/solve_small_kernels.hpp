// Device templates of the lane-per-trajectory forward pass and of the standalone interrogation, shared by the
// ahead-of-time build (solve_small.hip, built-in right-hand sides) and by the hiprtc build for user-supplied
// right-hand sides (rhs_jit.hip).  RTC-safe: no host code, no <hip/hip_runtime.h> under __HIPCC_RTC__.
#pragma once
#include "rk_enums.hpp"
#include "kalman_small.hpp"
#include "philox.hpp"
#include "solve_args.hpp"

namespace rk {

// ---- interrogation of one trajectory (all blocks): src/rodeo/interrogate.py -----------------------------------
// Produces W_meas = ode_weight + wgt_meas (solve.py:79), mean_meas and var_meas (n_bmeas = 1 -> scalars per block).
template <class RHS, int P, int ITG>
__device__ __forceinline__ void interrogate_traj(const double (&W)[RHS::D][P], const double (&th)[RHS::NTHETA],
                                                 double t, const double (&mup)[RHS::D][P],
                                                 const double (&Sp)[RHS::D][P][P], uint64_t seed, uint32_t traj,
                                                 uint32_t step, double (&wgt)[RHS::D][P], double (&a)[RHS::D],
                                                 double (&V)[RHS::D]) {
    constexpr int D = RHS::D;
    double f[D];
    if constexpr (ITG == RK_INTERROGATE_KRAMER) {
        // interrogate.py:75-84: wgt_meas = -J ; mean_meas = -f + J mu- ; var_meas = 0
        double J[D][P];
        RHS::template fjac<P>(mup, t, th, f, J);
#pragma unroll
        for (int blk = 0; blk < D; ++blk) {
            a[blk] = -f[blk] + dot<P>(J[blk], mup[blk]);
            V[blk] = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) wgt[blk][j] = -J[blk][j];
        }
    } else {
        if constexpr (ITG == RK_INTERROGATE_CHKREBTII) {
            // interrogate.py:22-34,46: x_b ~ N(mu-_b, Sigma-_b) ; mean_meas = -f(x) ; var_meas = W Sigma- W^T
            double xs[D][P];
#pragma unroll
            for (int blk = 0; blk < D; ++blk) {
                double z[P];
                normals<P>(seed, traj, step, (uint32_t)blk, PURPOSE_INTERROGATE, z);
                mvn_draw<P>(mup[blk], Sp[blk], z, xs[blk]);
            }
            RHS::template f<P>(xs, t, th, f);
        } else {
            // interrogate.py:61 / :114: mean_meas = -f(mu-)
            RHS::template f<P>(mup, t, th, f);
        }
#pragma unroll
        for (int blk = 0; blk < D; ++blk) {
            a[blk] = -f[blk];
#pragma unroll
            for (int j = 0; j < P; ++j) wgt[blk][j] = 0.0;
            if constexpr (ITG == RK_INTERROGATE_SCHOBER) {
                V[blk] = 0.0;                                     // interrogate.py:60
            } else {
                double WS[P];                                     // interrogate.py:110-113 / :26-29
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    double s = W[blk][0] * Sp[blk][0][j];
#pragma unroll
                    for (int i = 1; i < P; ++i) s = fma(W[blk][i], Sp[blk][i][j], s);
                    WS[j] = s;
                }
                V[blk] = dot<P>(WS, W[blk]);
            }
        }
    }
}

// ---- forward pass: one lane per trajectory -----------------------------------------------------------------------
template <class RHS, int P, int ITG, bool STORE_PRED>
__global__ void __launch_bounds__(64) fwd_kernel(SolveArgs a) {
    constexpr int D = RHS::D;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const size_t B = (size_t)a.B;

    double Q[D][P][P], R[D][P][P], W[D][P], th[RHS::NTHETA];
#pragma unroll
    for (int blk = 0; blk < D; ++blk) {
        load_block_consts<P>(a, blk, b, Q[blk], R[blk]);
#pragma unroll
        for (int j = 0; j < P; ++j) W[blk][j] = ld(a.W, (size_t)blk * P + j, a.W_b, a.B, b);
    }
#pragma unroll
    for (int k = 0; k < RHS::NTHETA; ++k) th[k] = a.theta ? ld(a.theta, k, a.theta_b, a.B, b) : 0.0;

    double mu[D][P], S[D][P][P];
#pragma unroll
    for (int blk = 0; blk < D; ++blk)
#pragma unroll
        for (int i = 0; i < P; ++i) {
            mu[blk][i] = ld(a.x0, (size_t)blk * P + i, a.x0_b, a.B, b);
#pragma unroll
            for (int j = 0; j < P; ++j) S[blk][i][j] = 0.0;
        }

    // time index 0: (ode_init, 0) for filt and pred (solve.py:114-121)
#pragma unroll
    for (int blk = 0; blk < D; ++blk)
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const size_t em = (size_t)blk * P + i;
            a.mean[em * B + b] = mu[blk][i];
            if (STORE_PRED) a.mean_pred[em * B + b] = mu[blk][i];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                a.var[(em * P + j) * B + b] = 0.0;
                if (STORE_PRED) a.var_pred[(em * P + j) * B + b] = 0.0;
            }
        }

    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
    const size_t mstride = (size_t)D * P * B, vstride = (size_t)D * P * P * B;
    for (int n = 0; n < a.N; ++n) {
        double mup[D][P], Sp[D][P][P];
#pragma unroll
        for (int blk = 0; blk < D; ++blk) predict_block<P>(Q[blk], R[blk], mu[blk], S[blk], mup[blk], Sp[blk]);

        const double t = a.t_min + (a.t_max - a.t_min) * (double)(n + 1) / (double)a.N;     // solve.py:74
        double wgt[D][P], am[D], V[D];
        interrogate_traj<RHS, P, ITG>(W, th, t, mup, Sp, a.seed, traj, (uint32_t)n, wgt, am, V);

        double* mo = a.mean + (size_t)(n + 1) * mstride + b;
        double* vo = a.var + (size_t)(n + 1) * vstride + b;
#pragma unroll
        for (int blk = 0; blk < D; ++blk) {
            double Wm[P];
#pragma unroll
            for (int j = 0; j < P; ++j) Wm[j] = W[blk][j] + wgt[blk][j];                      // solve.py:79
            update_block_m1<P>(Wm, am[blk], V[blk], mup[blk], Sp[blk], mu[blk], S[blk]);
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const size_t em = (size_t)blk * P + i;
                mo[em * B] = mu[blk][i];
#pragma unroll
                for (int j = 0; j < P; ++j) vo[(em * P + j) * B] = S[blk][i][j];
            }
        }
        if (STORE_PRED) {
            double* mpo = a.mean_pred + (size_t)(n + 1) * mstride + b;
            double* vpo = a.var_pred + (size_t)(n + 1) * vstride + b;
#pragma unroll
            for (int blk = 0; blk < D; ++blk)
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const size_t em = (size_t)blk * P + i;
                    mpo[em * B] = mup[blk][i];
#pragma unroll
                    for (int j = 0; j < P; ++j) vpo[(em * P + j) * B] = Sp[blk][i][j];
                }
        }
    }
}

// ---- one interrogation for a batch (per-step boundary) ------------------------------------------------------------
template <class RHS, int P, int ITG>
__global__ void __launch_bounds__(64) interrogate_kernel(SolveArgs a, double t, int step, const double* mean_pred,
                                                         const double* var_pred, double* wgt_meas,
                                                         double* mean_meas, double* var_meas, int sqrt_mode) {
    constexpr int D = RHS::D;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const size_t B = (size_t)a.B;
    double W[D][P], th[RHS::NTHETA], mup[D][P], Sp[D][P][P];
#pragma unroll
    for (int blk = 0; blk < D; ++blk)
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const size_t em = (size_t)blk * P + i;
            W[blk][i] = ld(a.W, em, a.W_b, a.B, b);
            mup[blk][i] = mean_pred[em * B + b];
#pragma unroll
            for (int j = 0; j < P; ++j) Sp[blk][i][j] = var_pred[(em * P + j) * B + b];
        }
#pragma unroll
    for (int k = 0; k < RHS::NTHETA; ++k) th[k] = a.theta ? ld(a.theta, k, a.theta_b, a.B, b) : 0.0;
    double wgt[D][P], am[D], V[D];
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
    if constexpr (ITG == RK_INTERROGATE_CHKREBTII) {
        if (sqrt_mode) {
            // interrogate.py:35-42 (kalman_type = "square-root"): var_state_pred is the FACTOR L-, var_meas = W L- has shape
            // (1, p) and the draw is mu- + (W L-) . z, one scalar added to every entry; the factor's column signs are
            // normalised (diag >= 0) like in the fused kernels and the oracle.  var_meas here: (d, 1, p, B).
            double xs[D][P], f[D];
#pragma unroll
            for (int blk = 0; blk < D; ++blk) {
                double z[P], WL[P];
                normals<P>(a.seed, traj, (uint32_t)step, (uint32_t)blk, PURPOSE_INTERROGATE, z);
                double shift = 0.0;
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    double sj = W[blk][0] * Sp[blk][0][j];
#pragma unroll
                    for (int i = 1; i < P; ++i) sj = fma(W[blk][i], Sp[blk][i][j], sj);
                    WL[j] = sj;
                    shift = fma(sj, Sp[blk][j][j] < 0.0 ? -z[j] : z[j], shift);
                    var_meas[((size_t)blk * P + j) * B + b] = sj;
                    wgt_meas[((size_t)blk * P + j) * B + b] = 0.0;
                }
#pragma unroll
                for (int j = 0; j < P; ++j) xs[blk][j] = mup[blk][j] + shift;
            }
            RHS::template f<P>(xs, t, th, f);
#pragma unroll
            for (int blk = 0; blk < D; ++blk) mean_meas[(size_t)blk * B + b] = -f[blk];
            return;
        }
    }
    interrogate_traj<RHS, P, ITG>(W, th, t, mup, Sp, a.seed, traj, (uint32_t)step, wgt, am, V);
#pragma unroll
    for (int blk = 0; blk < D; ++blk) {
        mean_meas[(size_t)blk * B + b] = am[blk];
        var_meas[(size_t)blk * B + b] = V[blk];
#pragma unroll
        for (int j = 0; j < P; ++j) wgt_meas[((size_t)blk * P + j) * B + b] = wgt[blk][j];
    }
}

}  // namespace rk
