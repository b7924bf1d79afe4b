// Kernel argument block shared by the fused solver kernels (batch-minor inputs of include/rodeo_kalman.h).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace rk {

struct SolveArgs {
    int B, N, D;
    double t_min, t_max;
    uint64_t seed, traj_offset;
    const double *W, *x0, *Q, *R, *theta;
    int W_b, x0_b, Q_b, R_b, theta_b;
    double *mean, *var, *mean_pred, *var_pred, *x;
};

__device__ __forceinline__ double ld(const double* p, size_t e, int batched, int B, int b) {
    return batched ? p[e * (size_t)B + b] : p[e];
}

template <int P>
__device__ __forceinline__ void load_block_consts(const SolveArgs& a, int blk, int b, double (&Q)[P][P],
                                                  double (&R)[P][P]) {
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const size_t e = ((size_t)blk * P + i) * P + j;
            Q[i][j] = ld(a.Q, e, a.Q_b, a.B, b);
            R[i][j] = ld(a.R, e, a.R_b, a.B, b);
        }
}

}  // namespace rk
