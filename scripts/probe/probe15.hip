// probe15: the backward chain step of bwd_mv_tile4_kernel (solve_tile4.hip, consumer) in isolation -- what do its 245 cycles
// consist of?  One wave per workgroup, one workgroup per CU; per step
//     V1 = MF(Ss - Sp, Gt, 0);  ms = MF(Gt, ms - mp, mf);  Ss = MF(V1, Gt, Sf)
// with MF = v_mfma_f64_4x4x4_4b; variants add the kernel's surroundings one at a time.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form scripts/probe/probe15.hip -o scripts/probe/bin/probe15
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double MF(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
constexpr int CH = 16, TICKS = 256;
__shared__ __attribute__((aligned(16))) double g_lds[8192];

// VAR 0: operands in registers, no LDS
//     1: + the two image writes per step (unconditional ds_write_b64)
//     2: + exec-masked image writes as in the kernel (valid lanes / one lane in four)
//     3: 2 + five LDS reads per step, two steps ahead
//     4: 3 + one workgroup barrier per 16 steps
//     5: only the Ss chain (no mean MFMA)
//     6: 0 with the subtraction folded away (Ss chain: two MFMAs back to back)
//     7: one MFMA per step (dependent chain of single MFMAs): the latency of one v_mfma_f64_4x4x4_4b
//     8: one v_add_f64 per step (dependent): VALU fp64 latency
__shared__ volatile int g_stop;
__shared__ __attribute__((aligned(16))) double g_noise[3 * 3840];
// NOISE: waves 1..3 of the workgroup imitate the producers' LDS traffic while the chain runs: per "tick" 56 ds_write_b64 and 20
// ds_read_b64 per lane (the hand-off and the landing-zone reads of solve_tile4.hip's producers), then PAUSE cycles of sleep
template <int VAR, int NOISE>
__global__ void __launch_bounds__(NOISE ? 256 : 64) k(double* out, long long* cyc, double seed) {
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) g_stop = 0;
    for (int e = threadIdx.x; e < 8192; e += blockDim.x) g_lds[e] = 1e-3 * (e % 97) * seed;
    __syncthreads();
    if (NOISE && threadIdx.x >= 64) {
        double* mine = g_noise + (threadIdx.x >> 6) * 3840 - 3840 + lane;
        double acc = seed;
        while (g_stop == 0) {
#pragma unroll
            for (int i = 0; i < 56; ++i) mine[(i % 60) * 64] = acc + i;
#pragma unroll
            for (int i = 0; i < 20; ++i) acc += mine[(i % 60) * 64];
            if (NOISE > 1) __builtin_amdgcn_s_sleep(NOISE);
        }
        out[blockIdx.x * 256 + threadIdx.x] = acc;
        return;
    }
    double Ss = 1.0 + 1e-3 * lane, ms = 0.5;
    const bool valid = (lane >> 4) < 3, st_m = valid && (lane & 3) == 0;
    const double* rd = g_lds + lane;
    double* img = g_lds + 4096 + lane;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < TICKS; ++t) {
        double Sp[CH], Gt[CH], Sf[CH], mp[CH], mf[CH];
        auto load = [&](int s) {
            if (VAR >= 3 && VAR <= 4) {
                const double* q = rd + s * 192;
                Sp[s] = q[0]; Gt[s] = q[64]; Sf[s] = q[128]; mp[s] = q[3072 + 0]; mf[s] = q[3072 + 64];
            } else { Sp[s] = 1e-3 * seed; Gt[s] = 0.5 * seed; Sf[s] = 1.0 * seed; mp[s] = 0.1 * seed; mf[s] = 0.2 * seed; }
        };
#pragma unroll
        for (int s = 0; s < 2; ++s) load(s);
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (s + 2 < CH) load(s + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (VAR == 7) { Ss = MF(Ss, Gt[s], Sf[s]); }
            else if (VAR == 8) { Ss = Ss - Sp[s]; }
            else {
                const double V1 = MF(VAR == 6 ? Ss : Ss - Sp[s], Gt[s], 0.0);
                if (VAR != 5 && VAR != 6) ms = MF(Gt[s], ms - mp[s], mf[s]);
                Ss = MF(V1, Gt[s], Sf[s]);
            }
            if (VAR == 1) { img[s * 128] = Ss; img[s * 128 + 64] = ms; }
            if (VAR >= 2 && VAR <= 4) { if (valid) img[s * 128] = Ss; if (st_m) img[s * 128 + 64] = ms; }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (VAR == 4) __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) g_stop = 1;
    out[blockIdx.x * 256 + lane] = Ss + ms;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VAR, int NOISE = 0>
void run(const char* name) {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 8);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<VAR, NOISE>), dim3(256), dim3(NOISE ? 256 : 64), 0, 0, out, cyc, 1.0);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= 256;
    printf("%-72s %7.1f cycles per step\n", name, m / (TICKS * CH));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<8>("one dependent v_add_f64 per step");
    run<7>("one dependent v_mfma_f64_4x4x4_4b per step");
    run<6>("Ss chain, two MFMAs back to back (no subtraction, no mean)");
    run<5>("Ss chain: sub -> MFMA -> MFMA");
    run<0>("the step: sub, sub, MFMA V1, MFMA ms, MFMA Ss (registers only)");
    run<1>("+ two unconditional image writes");
    run<2>("+ exec-masked image writes (as in the kernel)");
    run<3>("+ five LDS reads per step, two steps ahead");
    run<4>("+ a workgroup barrier per 16 steps");
    run<3, 1>("the same + three waves writing / reading LDS back to back");
    run<3, 8>("the same + three waves at ~ the producers' LDS rate (bursts, sleep 8)");
    run<3, 32>("the same + three waves, bursts with sleep 32");
    run<0, 1>("registers-only step + three waves writing / reading LDS back to back");
    return 0;
}
