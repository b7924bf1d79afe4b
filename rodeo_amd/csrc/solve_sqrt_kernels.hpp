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

// One lane per (trajectory, block): the D blocks of a trajectory sit in D neighbouring lanes (64 / D trajectories per wave)
// and exchange their interrogation points through ds_bpermute once per step; every lane evaluates the right-hand side
// for its whole trajectory and keeps its own block's row.  (One lane per trajectory walked the D blocks one after the
// other: with 1024 trajectories that is 16 waves on the whole chip, each issuing D times the instructions per step --
// 9.4 ms against 4.7 for the headline shape, scripts/nderiv_times.py.)
__device__ __forceinline__ double sq_from_lane(double x, int src_lane) {
    const int lo_ = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x));
    const int hi_ = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x));
    return __hiloint2double(hi_, lo_);
}
template <int D>
__device__ __forceinline__ double sq_pick(const double (&v)[D], int blk) {
    double x = v[0];
#pragma unroll
    for (int k = 1; k < D; ++k) x = blk == k ? v[k] : x;
    return x;
}

template <class RHS, int P, int ITG>
__global__ void __launch_bounds__(64) fwd_sqrt_kernel(SolveArgs a) {
    constexpr int D = RHS::D, TPW_ = 64 / D;
    static_assert(D >= 1 && D <= 64, "square-root filter: at most 64 blocks");
    constexpr int KV = ITG == RK_INTERROGATE_CHKREBTII ? P : 1;
    constexpr bool HOIST = P <= 5;                        // Q and R^{1/2} of the lane's block stay in registers
    const int lane = threadIdx.x, tl = lane / D, blk_raw = lane - tl * D, base = lane - blk_raw;
    const int b_raw = blockIdx.x * TPW_ + tl;
    const bool valid = tl < TPW_ && b_raw < a.B;          // (other lanes repeat a real unit and store nothing)
    const int b = b_raw < a.B ? b_raw : a.B - 1, blk = blk_raw;
    const size_t B = (size_t)a.B;
    double W[P], th[RHS::NTHETA], mu[P], L[P][P], Qh[HOIST ? P : 1][HOIST ? P : 1], LRh[HOIST ? P : 1][HOIST ? P : 1];
#pragma unroll
    for (int k = 0; k < RHS::NTHETA; ++k) th[k] = a.theta ? ld(a.theta, k, a.theta_b, a.B, b) : 0.0;
    if constexpr (HOIST) load_block_consts<P>(a, blk, b, Qh, LRh);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const size_t em = (size_t)blk * P + i;
        W[i] = ld(a.W, em, a.W_b, a.B, b);
        mu[i] = ld(a.x0, em, a.x0_b, a.B, b);
        if (valid) a.mean[em * B + b] = mu[i];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            L[i][j] = 0.0;
            if (valid) a.var[(em * P + j) * B + b] = 0.0;
        }
    }
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
    const size_t mstride = (size_t)D * P * B, vstride = (size_t)D * P * P * B;
    for (int n = 0; n < a.N; ++n) {
        double mup[P], Lp[P][P];
        if constexpr (HOIST) {
            sqrt_predict<P>(Qh, LRh, mu, L, mup, Lp);                                    // square_root.py:56-57
        } else {
            double Q[P][P], LR[P][P];
            load_block_consts<P>(a, blk, b, Q, LR);
            sqrt_predict<P>(Q, LR, mu, L, mup, Lp);
        }
        const double t = a.t_min + (a.t_max - a.t_min) * (double)(n + 1) / (double)a.N;  // solve.py:74
        // ---- interrogation (interrogate.py) with the factor standing where the reference puts it ----
        double am, vm[KV], wgt[P], WL[P];
        if constexpr (ITG != RK_INTERROGATE_KRAMER) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                double s = W[0] * Lp[0][j];
#pragma unroll
                for (int i = 1; i < P; ++i) s = fma(W[i], Lp[i][j], s);
                WL[j] = s;
            }
        }
        // the point of this lane's block: mu-, or for interrogate_chkrebtii the draw x = mu- + (W L-) . z with one scalar added
        // to every entry (interrogate.py:36-42: var_meas = W L-, 1 x p)
        double xo[P];
        if constexpr (ITG == RK_INTERROGATE_CHKREBTII) {
            double z[P];
            normals<P>(a.seed, traj, (uint32_t)n, (uint32_t)blk, PURPOSE_INTERROGATE, z);
#pragma unroll
            for (int j = 0; j < P; ++j) z[j] = Lp[j][j] < 0.0 ? -z[j] : z[j];          // sign-normalised factor
            const double shift = dot<P>(WL, z);
#pragma unroll
            for (int j = 0; j < P; ++j) { xo[j] = mup[j] + shift; vm[j] = WL[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) xo[j] = mup[j];
        }
        double X[D][P];                                   // the whole trajectory's points, in every one of its lanes
#pragma unroll
        for (int bb = 0; bb < D; ++bb)
#pragma unroll
            for (int j = 0; j < P; ++j) X[bb][j] = D == 1 ? xo[j] : sq_from_lane(xo[j], base + bb);
        double f[D];
        if constexpr (ITG == RK_INTERROGATE_KRAMER) {
            double J[D][P];
            RHS::template fjac<P>(X, t, th, f, J);
            double Jo[P];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                double col[D];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) col[bb] = J[bb][j];
                Jo[j] = sq_pick<D>(col, blk);
            }
            am = -sq_pick<D>(f, blk) + dot<P>(Jo, mup);
            vm[0] = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) wgt[j] = -Jo[j];
        } else {
            RHS::template f<P>(X, t, th, f);
            am = -sq_pick<D>(f, blk);
            if constexpr (ITG != RK_INTERROGATE_CHKREBTII)
                vm[0] = ITG == RK_INTERROGATE_RODEO ? dot<P>(WL, W) : 0.0;              // interrogate.py:110-113 / :60
#pragma unroll
            for (int j = 0; j < P; ++j) wgt[j] = 0.0;
        }
        double Wm[P];
#pragma unroll
        for (int j = 0; j < P; ++j) Wm[j] = W[j] + wgt[j];                              // solve.py:79
        sqrt_update_m1<P, KV>(Wm, am, vm, mup, Lp, mu, L);
        if (valid) {
            double* mo = a.mean + (size_t)(n + 1) * mstride + b;
            double* vo = a.var + (size_t)(n + 1) * vstride + b;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const size_t em = (size_t)blk * P + i;
                mo[em * B] = mu[i];
#pragma unroll
                for (int j = 0; j < P; ++j) vo[(em * P + j) * B] = L[i][j];
            }
        }
    }
}

}  // namespace rk
