// Probe 4: the backward consumer's dependent chain  d = ms - mp; v = MF(d, g, 0); ms = MF(v, g, mf)  in isolation
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ double MF(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
template <int MODE>
__global__ void k(double* out, long long* cyc, int iters, double seed, double* sink) {
    const int l = threadIdx.x;
    __shared__ double lds[16 * 64 * 3];
    for (int i = l; i < 16 * 64 * 3; i += 64) lds[i] = 1e-3 * (i % 7);
    __syncthreads();
    double ms = seed + l * 1e-3, mp = 0.25, g = 1e-2 * (l & 3), mf = 0.5;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE >= 1) { mp = lds[(u * 64 + l) * 3]; g = lds[(u * 64 + l) * 3 + 1]; mf = lds[(u * 64 + l) * 3 + 2]; }
            const double d = ms - mp;
            const double v = MF(d, g, 0.0);
            ms = MF(v, g, mf);
            if (MODE >= 2) sink[(size_t)(i * 16 + u) * 64 + l] = ms;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = ms;
    if (l == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    double *out, *sink; long long* cyc; CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&sink, (size_t)200 * 16 * 64 * 8));
    const int iters = 200;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.5, sink); CK(hipDeviceSynchronize()); }
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-60s cycles per step = %8.2f\n", name, (double)h / iters / 16);
}
int main() {
    run<0>("chain only (registers)");
    run<1>("chain + 3 LDS reads per step");
    run<2>("chain + LDS reads + global store per step");
    return 0;
}
