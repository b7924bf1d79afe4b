// probe16: the LU panel's column step (rodeo_amd/csrc/solve_dense_panel.hpp, rl_panel_col) in isolation: one wave, 160 rows x 16
// columns in registers (three rows per lane), all 16 columns unrolled -- cycles per column, and variants of its parts.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -Irodeo_amd/csrc -Iinclude scripts/probe/probe16.hip -o scripts/probe/bin/probe16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include "linalg_small.hpp"
namespace rk { constexpr int LU_NB = 16; }
#include "solve_dense_panel.hpp"
using namespace rk;

// the shipped column step with parts switched off: VAR bit 0 = no pivot search, bit 1 = no pivot-row broadcast, bit 2 = no update
template <int RS, int J, int VAR>
__device__ __forceinline__ int colv(double (&a)[RS][LU_NB], int (&pos)[RS], int gpos, int n) {
    typedef unsigned long long u64;
    const unsigned span = (unsigned)(n - gpos);
    int pj, ol, os;
    if constexpr ((VAR & 1) != 0) { pj = gpos; ol = gpos & 63; os = gpos >> 6; }
    else {
    // ---- pivot: largest |a|, smallest position, over the rows whose position is >= gpos ----
    u64 bk = (unsigned)(pos[0] - gpos) < span ? (u64)__double_as_longlong(fabs(a[0][J])) + 1ull : 0ull;
    int bs = 0;
    unsigned tie = 0u;                                          // != 0: two candidates of this lane have the same |a| (exact path)
#pragma unroll
    for (int s = 1; s < RS; ++s) {
        const u64 ks = (unsigned)(pos[s] - gpos) < span ? (u64)__double_as_longlong(fabs(a[s][J])) + 1ull : 0ull;
        tie |= ks == bk ? (unsigned)(ks | (ks >> 32)) : 0u;
        const bool better = ks > bk;
        bk = better ? ks : bk;
        bs = better ? s : bs;
    }
    int bp = pos[0];
#pragma unroll
    for (int s = 1; s < RS; ++s) bp = bs == s ? pos[s] : bp;
    // The wave maximum is found on the high words; one matching lane is the rule, and then its candidate is the pivot.
    // Ties (on the high word across lanes, or exact ones inside a lane) take the exact path.
    const unsigned key = bk != 0ull ? (unsigned)(bk >> 32) + 1u : 0u;
    const unsigned mkey = wave_max_u32_fused(key);
    const u64 mm = __builtin_amdgcn_ballot_w64(key == mkey);
    const u64 tt = RS > 1 ? __builtin_amdgcn_ballot_w64(tie != 0u) : 0ull;
    if (mkey != 0u && __builtin_popcountll(mm) == 1 && tt == 0ull) {
        ol = (int)__builtin_ctzll(mm);
        pj = __builtin_amdgcn_readlane(bp, ol);
        os = RS > 1 ? __builtin_amdgcn_readlane(bs, ol) : 0;
    } else {
        if (mkey == 0u) {
            pj = gpos;                                          // nothing in play (n reached): keep the row
        } else {
            double best = -1.0;
            int bq = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < RS; ++s) {
                const double v = fabs(a[s][J]);
                if ((unsigned)(pos[s] - gpos) < span && (v > best || (v == best && pos[s] < bq))) { best = v; bq = pos[s]; }
            }
            const double m2 = wave_max_f64(best);
            pj = wave_min_i32(best == m2 ? bq : 0x7fffffff);
            if (pj == 0x7fffffff) pj = gpos;
        }
        int ms = -1;
#pragma unroll
        for (int s = RS - 1; s >= 0; --s) ms = pos[s] == pj ? s : ms;
        ol = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(ms >= 0));
        os = __builtin_amdgcn_readlane(ms, ol);
    }
    }
    // ---- the pivot row to every lane (scalars) ----
    double prow[LU_NB];
    if constexpr ((VAR & 2) != 0) {
#pragma unroll
        for (int c = J; c < LU_NB; ++c) prow[c] = 1.0 + 0.001 * c;
    } else if constexpr ((VAR & 8) != 0) {                  // ds_bpermute: every lane fetches lane ol's registers
        const int sel = ol << 2;
        auto bp = [&](double x) {
            return __hiloint2double(__builtin_amdgcn_ds_bpermute(sel, __double2hiint(x)), __builtin_amdgcn_ds_bpermute(sel, __double2loint(x)));
        };
        if (RS == 1 || os == 0) {
#pragma unroll
            for (int c = J; c < LU_NB; ++c) prow[c] = bp(a[0][c]);
        } else if (RS == 2 || os == 1) {
#pragma unroll
            for (int c = J; c < LU_NB; ++c) prow[c] = bp(a[RS > 1 ? 1 : 0][c]);
        } else {
#pragma unroll
            for (int c = J; c < LU_NB; ++c) prow[c] = bp(a[RS > 2 ? 2 : 0][c]);
        }
    } else if constexpr ((VAR & 16) != 0) {                 // through LDS: the owner writes, every lane reads the same addresses
        __shared__ __attribute__((aligned(16))) double rowbuf[LU_NB];
        const bool own = (int)(threadIdx.x & 63) == ol;
        if (RS == 1 || os == 0) {
            if (own) {
#pragma unroll
                for (int c = J; c < LU_NB; ++c) rowbuf[c] = a[0][c];
            }
        } else if (RS == 2 || os == 1) {
            if (own) {
#pragma unroll
                for (int c = J; c < LU_NB; ++c) rowbuf[c] = a[RS > 1 ? 1 : 0][c];
            }
        } else {
            if (own) {
#pragma unroll
                for (int c = J; c < LU_NB; ++c) rowbuf[c] = a[RS > 2 ? 2 : 0][c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = J; c < LU_NB; ++c) prow[c] = rowbuf[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
    if (RS == 1 || os == 0) {
#pragma unroll
        for (int c = J; c < LU_NB; ++c) prow[c] = rl_readlane_f64(a[0][c], ol);
    } else if (RS == 2 || os == 1) {
#pragma unroll
        for (int c = J; c < LU_NB; ++c) prow[c] = rl_readlane_f64(a[RS > 1 ? 1 : 0][c], ol);
    } else {
#pragma unroll
        for (int c = J; c < LU_NB; ++c) prow[c] = rl_readlane_f64(a[RS > 2 ? 2 : 0][c], ol);
    }
    }
    const double rinv = fast_rcp(prow[J]);
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        int t = pos[s] == pj ? gpos : pos[s];                    // interchange gpos <-> pj
        t = pos[s] == gpos ? pj : t;
        pos[s] = t;
        const bool below = (unsigned)(t - gpos - 1) < span - 1u;             // still below the diagonal
        const double lm = a[s][J] * rinv;
        const double lmz = below ? lm : 0.0;                    // (rows out of play: a - 0 * u = a exactly)
        a[s][J] = below ? lm : a[s][J];
        if constexpr ((VAR & 4) == 0)
#pragma unroll
        for (int c = J + 1; c < LU_NB; ++c) a[s][c] = fma(-lmz, prow[c], a[s][c]);
    }
    return pj;
}



template <int RS, int J, int VAR>
struct Cols {
    static __device__ __forceinline__ void run(double (&a)[RS][LU_NB], int (&pos)[RS], int n, int& acc) {
        if constexpr (J < LU_NB) {
            if constexpr (VAR == 0) acc += rl_panel_col<RS, J>(a, pos, J, n);
            else acc += colv<RS, J, VAR>(a, pos, J, n);
            Cols<RS, J + 1, VAR>::run(a, pos, n, acc);
        }
    }
};

template <int RS, int VAR>
__global__ void __launch_bounds__(64) k(const double* in, double* out, long long* cyc, int n, int reps) {
    const int lane = threadIdx.x;
    double a0[RS][LU_NB];
#pragma unroll
    for (int s = 0; s < RS; ++s)
#pragma unroll
        for (int c = 0; c < LU_NB; ++c) a0[s][c] = in[((size_t)blockIdx.x * 192 + lane + 64 * s) * LU_NB + c];
    long long total = 0;
    int acc = 0;
    double sum = 0.0;
    for (int r = 0; r < reps; ++r) {
        double a[RS][LU_NB];
        int pos[RS];
#pragma unroll
        for (int s = 0; s < RS; ++s) {
            pos[s] = lane + 64 * s < n ? lane + 64 * s : 0x7fffffff;
#pragma unroll
            for (int c = 0; c < LU_NB; ++c) a[s][c] = a0[s][c] + 1e-9 * r;
        }
        const long long t0 = __builtin_amdgcn_s_memtime();
        Cols<RS, 0, VAR>::run(a, pos, n, acc);
        const long long t1 = __builtin_amdgcn_s_memtime();
        total += t1 - t0;
#pragma unroll
        for (int s = 0; s < RS; ++s)
#pragma unroll
            for (int c = 0; c < LU_NB; ++c) sum += a[s][c];
    }
    out[blockIdx.x * 64 + lane] = sum + acc;
    if (lane == 0) cyc[blockIdx.x] = total;
}

template <int RS, int VAR>
void run(const char* name, int n) {
    const int reps = 50, nb = 64;
    std::vector<double> h((size_t)nb * 192 * 16);
    srand(1);
    for (auto& v : h) v = (rand() / (double)RAND_MAX) - 0.5;
    double *in, *out; long long* cyc;
    hipMalloc(&in, h.size() * 8); hipMalloc(&out, nb * 64 * 8); hipMalloc(&cyc, nb * 8);
    hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<RS, VAR>), dim3(nb), dim3(64), 0, 0, in, out, cyc, n, reps);
    hipDeviceSynchronize();
    std::vector<long long> c(nb);
    hipMemcpy(c.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : c) m += v; m /= nb;
    printf("%-60s rows %3d: %7.0f cycles per column\n", name, n, m / (reps * 16.0));
    hipFree(in); hipFree(out); hipFree(cyc);
}
int main() {
    run<3, 0>("shipped column step, three rows per lane", 160);
    run<3, 1>("  without the pivot search", 160);
    run<3, 2>("  without the pivot row's broadcast", 160);
    run<3, 4>("  without the update", 160);
    run<3, 3>("  update only", 160);
    run<3, 6>("  search only", 160);
    run<3, 5>("  broadcast only", 160);
    run<3, 7>("  nothing (reciprocal, position bookkeeping)", 160);
    run<3, 8>("  broadcast by ds_bpermute", 160);
    run<3, 16>("  broadcast through LDS (owner writes, all read)", 160);
    run<2, 0>("shipped column step, two rows per lane", 128);
    run<1, 0>("shipped column step, one row per lane", 64);
    return 0;
}
