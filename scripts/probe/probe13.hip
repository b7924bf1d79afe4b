// Probe 13: what one output per step costs a single-wave dependent chain (the forward tile kernels), and whether a
// second "writer" wave per workgroup (results through LDS, whole rows to memory every 16 steps) is cheaper.
//   mode 0: chain only (add, MFMA, MFMA, the forward kernels' kind of step)      mode 1: + raw buffer store b64, 48 lanes
//   mode 2: + ds_write_b64, all lanes                                            mode 3: + ds_write_b64, 48 lanes (exec mask)
//   mode 5: mode 1 with the store one step late (previous step's value, issued behind the first MFMA)       mode 6: same for mode 3
//   mode 4: mode 3 + a writer wave: barrier every 16 steps, writer reads the 16 rows back (b128) and stores them (b128)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double MF(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
constexpr int ROW = 384;                       // bytes per (wave, step): 48 lanes x 8
template <int MODE>
__global__ void __launch_bounds__(MODE == 4 ? 128 : 64) k(double* out, long long* cyc, int ticks, double seed, char* sink, size_t row_bytes) {
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    __shared__ __attribute__((aligned(16))) char lds[2][16 * ROW];
    const char* base = sink + (size_t)blockIdx.x * ROW;
    if (MODE == 4 && wave == 1) {
        // writer: 16 rows x 384 B = 384 pieces of 16 B = 6 per lane
        __syncthreads();
        for (int t = 0; t < ticks; ++t) {
            __syncthreads();                                       // chunk t is in lds[t & 1]
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)t * 16 * row_bytes), 0, (int)(15 * row_bytes + ROW), 0x00020000);
            u32x4 v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { const int j = 64 * i + l; v[i] = *(const u32x4*)(&lds[t & 1][(j / 24) * ROW + (j % 24) * 16]); }
#pragma unroll
            for (int i = 0; i < 6; ++i) { const int j = 64 * i + l; __builtin_amdgcn_raw_buffer_store_b128(v[i], rs, (int)((j / 24) * row_bytes) + (j % 24) * 16, 0, 0); }
        }
        return;
    }
    double ms = seed + l * 1e-3;
    const double g = 1e-2 * (l & 3), mf = 0.5, mp = 0.25;
    const int vo = l < 48 ? l * 8 : (int)0x80000000;
    if (MODE == 4) __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ticks; ++t) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)t * 16 * row_bytes), 0, (int)(15 * row_bytes + ROW), 0x00020000);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double d = ms - mp;
            const double v = MF(d, g, 0.0);
            if (MODE == 5 && (u > 0 || t > 0)) { u32x2 b; __builtin_memcpy(&b, &ms, 8); __builtin_amdgcn_raw_buffer_store_b64(b, rs, vo, (int)(((u + 15) & 15) * row_bytes), 0); }
            if (MODE == 6 && l < 48) *(double*)(&lds[t & 1][((u + 15) & 15) * ROW + l * 8]) = ms;
            const double w = MF(g, d, mf);                         // (an independent MFMA, like the mean's)
            ms = MF(v, g, w);
            if (MODE == 1) { u32x2 b; __builtin_memcpy(&b, &ms, 8); __builtin_amdgcn_raw_buffer_store_b64(b, rs, vo, (int)(u * row_bytes), 0); }
            if (MODE == 2) *(double*)(&lds[t & 1][u * ROW + (l % 48) * 8]) = ms;
            if (MODE == 3 || MODE == 4) { if (l < 48) *(double*)(&lds[t & 1][u * ROW + l * 8]) = ms; }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 4) __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 4) __syncthreads();
    out[(size_t)blockIdx.x * 64 + l] = ms;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int wgs) {
    const int ticks = 200;
    const size_t row_bytes = (size_t)wgs * ROW;
    double* out; char* sink; long long* cyc;
    CK(hipMalloc(&out, (size_t)wgs * 64 * 8)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&sink, (size_t)ticks * 16 * row_bytes + 4096));
    float ms = 0; hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(MODE == 4 ? 128 : 64), 0, 0, out, cyc, ticks, 0.5, sink, row_bytes);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-62s wgs %4d: %7.2f cycles per step (workgroup 0), kernel %7.1f us = %6.1f ns per step\n", name, wgs, (double)h / ticks / 16, ms * 1e3, ms * 1e6 / ticks / 16);
    CK(hipFree(out)); CK(hipFree(sink)); CK(hipFree(cyc));
}
int main() {
    for (int wgs : {256, 512, 1024}) {
        run<0>("chain only (add, 3 MFMA of which 2 dependent)", wgs);
        run<1>("+ raw buffer store b64, 48 lanes, per step", wgs);
        run<2>("+ ds_write_b64, all lanes, per step", wgs);
        run<3>("+ ds_write_b64, 48 lanes (exec mask), per step", wgs);
        run<5>("+ raw buffer store b64 of the previous step's value, behind MFMA 1", wgs);
        run<6>("+ ds_write_b64 48 lanes of the previous step's value, behind MFMA 1", wgs);
        run<4>("+ ds_write 48 lanes, writer wave stores whole rows per 16 steps", wgs);
    }
    return 0;
}
