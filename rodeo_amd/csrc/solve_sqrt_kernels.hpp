// Forward kernel of the fused square-root solver (kalman_type = "square-root", see solve_sqrt.hip), as a template over the
// right-hand side: built ahead of time for the built-in ODEs (solve_sqrt.hip) and by hiprtc for user-supplied / traced
// ones (rhs_jit.hip).  RTC-safe: no host code, no <hip/hip_runtime.h> under __HIPCC_RTC__.
#pragma once
#include "rk_enums.hpp"
#include "kalman_small.hpp"
#include "philox.hpp"
#include "solve_args.hpp"
#include "sqrt_small.hpp"

namespace rk {

template <class RHS, int P, int ITG>
__global__ void __launch_bounds__(64) fwd_sqrt_kernel(SolveArgs a) {
    constexpr int D = RHS::D;
    constexpr int KV = ITG == RK_INTERROGATE_CHKREBTII ? P : 1;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const size_t B = (size_t)a.B;
    double W[D][P], th[RHS::NTHETA], mu[D][P], L[D][P][P];
#pragma unroll
    for (int k = 0; k < RHS::NTHETA; ++k) th[k] = a.theta ? ld(a.theta, k, a.theta_b, a.B, b) : 0.0;
#pragma unroll
    for (int blk = 0; blk < D; ++blk)
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const size_t em = (size_t)blk * P + i;
            W[blk][i] = ld(a.W, em, a.W_b, a.B, b);
            mu[blk][i] = ld(a.x0, em, a.x0_b, a.B, b);
            a.mean[em * B + b] = mu[blk][i];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                L[blk][i][j] = 0.0;
                a.var[(em * P + j) * B + b] = 0.0;
            }
        }
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
    const size_t mstride = (size_t)D * P * B, vstride = (size_t)D * P * P * B;
    for (int n = 0; n < a.N; ++n) {
        double mup[D][P], Lp[D][P][P];
#pragma unroll
        for (int blk = 0; blk < D; ++blk) {
            double Q[P][P], LR[P][P];
            load_block_consts<P>(a, blk, b, Q, LR);
            sqrt_predict<P>(Q, LR, mu[blk], L[blk], mup[blk], Lp[blk]);                  // square_root.py:56-57
        }
        const double t = a.t_min + (a.t_max - a.t_min) * (double)(n + 1) / (double)a.N;  // solve.py:74
        // ---- interrogation (interrogate.py) with the factor standing where the reference puts it ----
        double f[D], wgt[D][P], am[D], vm[D][KV];
        if constexpr (ITG == RK_INTERROGATE_KRAMER) {
            double J[D][P];
            RHS::template fjac<P>(mup, t, th, f, J);
#pragma unroll
            for (int blk = 0; blk < D; ++blk) {
                am[blk] = -f[blk] + dot<P>(J[blk], mup[blk]);
                vm[blk][0] = 0.0;
#pragma unroll
                for (int j = 0; j < P; ++j) wgt[blk][j] = -J[blk][j];
            }
        } else {
            double WL[D][P];
#pragma unroll
            for (int blk = 0; blk < D; ++blk)
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    double s = W[blk][0] * Lp[blk][0][j];
#pragma unroll
                    for (int i = 1; i < P; ++i) s = fma(W[blk][i], Lp[blk][i][j], s);
                    WL[blk][j] = s;
                }
            if constexpr (ITG == RK_INTERROGATE_CHKREBTII) {
                // interrogate.py:36-42: var_meas = W L- (1 x p) ; x = mu- + (W L-) . z  (one scalar added to every entry)
                double xs[D][P];
#pragma unroll
                for (int blk = 0; blk < D; ++blk) {
                    double z[P];
                    normals<P>(a.seed, traj, (uint32_t)n, (uint32_t)blk, PURPOSE_INTERROGATE, z);
#pragma unroll
                    for (int j = 0; j < P; ++j) z[j] = Lp[blk][j][j] < 0.0 ? -z[j] : z[j];   // sign-normalised factor
                    const double shift = dot<P>(WL[blk], z);
#pragma unroll
                    for (int j = 0; j < P; ++j) { xs[blk][j] = mup[blk][j] + shift; vm[blk][j] = WL[blk][j]; }
                }
                RHS::template f<P>(xs, t, th, f);
            } else {
                RHS::template f<P>(mup, t, th, f);
#pragma unroll
                for (int blk = 0; blk < D; ++blk)
                    vm[blk][0] = ITG == RK_INTERROGATE_RODEO ? dot<P>(WL[blk], W[blk]) : 0.0;   // interrogate.py:110-113 / :60
            }
#pragma unroll
            for (int blk = 0; blk < D; ++blk) {
                am[blk] = -f[blk];
#pragma unroll
                for (int j = 0; j < P; ++j) wgt[blk][j] = 0.0;
            }
        }
        double* mo = a.mean + (size_t)(n + 1) * mstride + b;
        double* vo = a.var + (size_t)(n + 1) * vstride + b;
#pragma unroll
        for (int blk = 0; blk < D; ++blk) {
            double Wm[P];
#pragma unroll
            for (int j = 0; j < P; ++j) Wm[j] = W[blk][j] + wgt[blk][j];                   // solve.py:79
            sqrt_update_m1<P, KV>(Wm, am[blk], vm[blk], mup[blk], Lp[blk], mu[blk], L[blk]);
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const size_t em = (size_t)blk * P + i;
                mo[em * B] = mu[blk][i];
#pragma unroll
                for (int j = 0; j < P; ++j) vo[(em * P + j) * B] = L[blk][i][j];
            }
        }
    }
}

}  // namespace rk
