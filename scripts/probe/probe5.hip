// Probe 5: issue cost of six 1-KiB LDS-DMA pieces (global_load_lds_dwordx4) per wave, four ways, next to plain loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void dma_a(const void* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int MODE>
__global__ void __launch_bounds__(256) k(const char* __restrict__ g, size_t row_stride, long long* cyc, double* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char zone[4][6144];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned z = __builtin_amdgcn_readfirstlane(lds_addr(zone[wave]));
    const char* base = g + (size_t)(blockIdx.x * 4 + wave) * 384;
    long long t_issue = 0, t_all = 0;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        const char* rows = base + (size_t)it * 16 * row_stride;
        const long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) {           // scattered: lane (s,g) reads its 96 B in six 16-B pieces; per-piece M0 save/restore
            const char* in = rows + (size_t)(lane >> 2) * row_stride + (lane & 3) * 96;
#pragma unroll
            for (int i = 0; i < 6; ++i) dma_a(in + 16 * i, z + 1024 * i);
        } else if (MODE == 1) {    // row-contiguous pieces; per-piece M0 save/restore
#pragma unroll
            for (int i = 0; i < 6; ++i) { const int j = 64 * i + lane; dma_a(rows + (size_t)(j / 24) * row_stride + (j % 24) * 16, z + 1024 * i); }
        } else if (MODE == 2) {    // row-contiguous, one statement, M0 written twice, instruction offsets
            const char* s[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { const int j = 64 * i + lane; s[i] = rows + (size_t)(j / 24) * row_stride + (j % 24) * 16 - 1024 * (i % 3); }
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\tglobal_load_lds_dwordx4 %3, off offset:2048\n\t"
                         "s_mov_b32 m0, %8\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %4, off\n\tglobal_load_lds_dwordx4 %5, off offset:1024\n\tglobal_load_lds_dwordx4 %6, off offset:2048\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "s"(z), "s"(z + 3072) : "memory");
        } else if (MODE == 3) {    // the compiler builtin
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int j = 64 * i + lane;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rows + (size_t)(j / 24) * row_stride + (j % 24) * 16),
                                                 (__attribute__((address_space(3))) void*)(zone[wave] + 1024 * i), 16, 0, 0);
            }
        }
        double2 v[6];
        if (MODE == 4) {           // plain loads to registers (scattered, as the producers did before)
            const char* in = rows + (size_t)(lane >> 2) * row_stride + (lane & 3) * 96;
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = *(const double2*)(in + 16 * i);
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_amdgcn_s_memtime();
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 6; ++i) acc += v[i].x + v[i].y;
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) { const double2 w = *(const double2*)(zone[wave] + 96 * lane + 16 * i); acc += w.x + w.y; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        t_issue += t1 - t0; t_all += t2 - t0;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t_issue; cyc[1] = t_all; }
}
template <int MODE> void run(const char* name, int wgs) {
    const size_t row_stride = (size_t)2048 * 96;          // C2: 2048 tiles per time row
    const int iters = 200;
    char* g; double* sink; long long* cyc;
    CK(hipMalloc(&g, (size_t)iters * 16 * row_stride + 4096)); CK(hipMemset(g, 0, (size_t)iters * 16 * row_stride + 4096));
    CK(hipMalloc(&sink, (size_t)wgs * 256 * 8)); CK(hipMalloc(&cyc, 16));
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, g, row_stride, cyc, sink, iters); CK(hipDeviceSynchronize()); }
    long long h[2]; CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
    printf("%-72s wgs %4d: issue %7.1f cycles, issue+land %7.1f cycles per 6 KiB\n", name, wgs, (double)h[0] / iters, (double)h[1] / iters);
    CK(hipFree(g)); CK(hipFree(sink)); CK(hipFree(cyc));
}
int main() {
    for (int wgs : {1, 128}) {
        run<0>("LDS-DMA, 64 scattered 16-B pieces per instruction, M0 per piece", wgs);
        run<1>("LDS-DMA, row-contiguous pieces, M0 per piece", wgs);
        run<2>("LDS-DMA, row-contiguous, one statement, M0 twice + offsets", wgs);
        run<3>("LDS-DMA, row-contiguous, __builtin_amdgcn_global_load_lds", wgs);
        run<4>("plain global_load_dwordx4 to VGPRs (scattered)", wgs);
    }
    return 0;
}
