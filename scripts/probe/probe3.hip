// Probe 3: does a v_mfma_f64_4x4x4 block the issuing wave's (independent) fp64 VALU ops?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int MODE>
__global__ void k(double* out, long long* cyc, int iters, double seed) {
    const int l = threadIdx.x;
    double a = seed + l * 1e-3, b = 1.0 + l * 1e-6;
    double m0 = 0.5, m1 = 0.25, f0 = 0.1, f1 = 0.2, f2 = 0.3, f3 = 0.4, f4 = .5, f5 = .6, f6 = .7, f7 = .8;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { m0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m1, 0, 0, 0); }
            if (MODE == 1) { m0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m1, 0, 0, 0);
                             f0 = fma(a, b, f0); f1 = fma(a, b, f1); f2 = fma(a, b, f2); f3 = fma(a, b, f3); }
            if (MODE == 2) { m0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m1, 0, 0, 0);
                             f0 = fma(a, b, f0); f1 = fma(a, b, f1); f2 = fma(a, b, f2); f3 = fma(a, b, f3);
                             f4 = fma(a, b, f4); f5 = fma(a, b, f5); f6 = fma(a, b, f6); f7 = fma(a, b, f7); }
            if (MODE == 3) { f0 = fma(a, b, f0); f1 = fma(a, b, f1); f2 = fma(a, b, f2); f3 = fma(a, b, f3);
                             f4 = fma(a, b, f4); f5 = fma(a, b, f5); f6 = fma(a, b, f6); f7 = fma(a, b, f7); }
            if (MODE == 4) { // interleaved mfma, 4 fma, mfma, 4 fma
                m0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m0, 0, 0, 0); f0 = fma(a, b, f0); f1 = fma(a, b, f1); f2 = fma(a, b, f2); f3 = fma(a, b, f3);
                m1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m1, 0, 0, 0); f4 = fma(a, b, f4); f5 = fma(a, b, f5); f6 = fma(a, b, f6); f7 = fma(a, b, f7); }
            if (MODE == 5) { // 2 mfma + 8 32-bit ops (int adds)
                m0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m1, 0, 0, 0);
                int x0 = __double2loint(f0), x1 = __double2loint(f1);
                x0 = x0 * 3 + u; x1 = x1 * 5 + u; x0 ^= x1; x1 += x0; x0 = x0 * 7 + 1; x1 = x1 * 9 + 2; x0 ^= x1; x1 += x0;
                f0 = __hiloint2double(__double2hiint(f0), x0); f1 = __hiloint2double(__double2hiint(f1), x1); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + l] = m0 + m1 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    double* out; long long* cyc; CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.5); CK(hipDeviceSynchronize()); }
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-50s cycles per group = %8.2f\n", name, (double)h / iters / 8);
}
int main() {
    run<0>("2 mfma");
    run<1>("2 mfma + 4 fma");
    run<2>("2 mfma + 8 fma");
    run<3>("8 fma");
    run<4>("mfma,4fma,mfma,4fma interleaved");
    run<5>("2 mfma + ~10 int ops");
    return 0;
}
