// Probe 6: the backward consumer's instruction stream in isolation, adding one ingredient at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ double MF(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void store_after(const void* row, unsigned voff, double v, unsigned long long mask, double after) {
    asm volatile("s_nop 0\n\ts_mov_b64 exec, %3\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_mov_b64 exec, -1" :: "v"(voff), "v"(v), "s"(row), "s"(mask), "v"(after) : "memory");
}
// MODE bit 0: asm masked store per step; bit 1: 48 LDS reads per 16 steps (fresh each tick); bit 2: barrier per tick;
// bit 3: the other three waves of the workgroup run dependent-free f64 FMA loops; bit 4: s_setprio 3
template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, long long* cyc, int ticks, double seed, char* sink, size_t row_bytes) {
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    __shared__ double lds[2][16 * 64 * 3];
    for (int i = threadIdx.x; i < 2 * 16 * 64 * 3; i += 256) (&lds[0][0])[i] = 1e-3 * (i % 7);
    __syncthreads();
    if (wave != 0) {
        double x0 = seed + l, x1 = seed * 2, x2 = seed * 3, x3 = 0.5, x4 = 0.25, x5 = 0.125;
        for (int t = 0; t < ticks; ++t) {
            if (MODE & 8) {
#pragma unroll 1
                for (int i = 0; i < 30; ++i) { x0 = fma(x0, 0.999, 1e-3); x1 = fma(x1, 0.999, 1e-3); x2 = fma(x2, 0.999, 1e-3);
                                                x3 = fma(x3, 0.999, 1e-3); x4 = fma(x4, 0.999, 1e-3); x5 = fma(x5, 0.999, 1e-3); }
            }
            if (MODE & 4) __syncthreads();
        }
        out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5;
        return;
    }
    if (MODE & 16) __builtin_amdgcn_s_setprio(3);
    double ms = seed + l * 1e-3;
    const unsigned long long mask = __ballot(l < 48);
    const unsigned voff = l * 8;
    const char* row = sink + (size_t)blockIdx.x * 512;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ticks; ++t) {
        double mp[16], g[16], mf[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE & 2) { const double* q = &lds[t & 1][(u * 64 + l) * 3]; mp[u] = q[0]; g[u] = q[1]; mf[u] = q[2]; }
            else { mp[u] = 0.25; g[u] = 1e-2 * (l & 3); mf[u] = 0.5; }
        }
        __builtin_amdgcn_sched_barrier(0);
        double d = ms - mp[0];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double v = MF(d, g[u], 0.0);
            ms = MF(v, g[u], mf[u]);
            d = ms - mp[(u + 1) & 15];
            if (MODE & 1) store_after(row + (size_t)(t * 16 + u) * row_bytes, voff, ms, mask, d);
        }
        ms = d + mp[0];
        if (MODE & 4) __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = ms;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int wgs) {
    const int ticks = 100;
    const size_t row_bytes = (size_t)wgs * 512;
    double* out; char* sink; long long* cyc;
    CK(hipMalloc(&out, 256 * 8)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&sink, (size_t)ticks * 16 * row_bytes + 4096));
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, out, cyc, ticks, 0.5, sink, row_bytes); CK(hipDeviceSynchronize()); }
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-78s wgs %3d: %7.2f cycles per step\n", name, wgs, (double)h / ticks / 16);
    CK(hipFree(out)); CK(hipFree(sink)); CK(hipFree(cyc));
}
int main() {
    for (int wgs : {1, 256, 512}) {
        run<0>("chain (add, MFMA, MFMA), operands in registers", wgs);
        run<1>("+ masked store with scalar row pointer", wgs);
        run<3>("+ 48 LDS reads per 16 steps, issued first", wgs);
        run<7>("+ workgroup barrier per 16 steps (3 idle waves)", wgs);
        run<15>("+ the 3 other waves run f64 FMA loops (180 FMAs per tick each)", wgs);
        run<31>("+ s_setprio 3 on the chain wave", wgs);
    }
    return 0;
}
