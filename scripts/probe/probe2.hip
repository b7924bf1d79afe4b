// Hardware probe 2 (not part of the product): fp64 VALU dependent-latency by operand position, DPP, rcp, sqrt,
// 16x16x4 f64 MFMA, two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using d4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ double dpp_ror4(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xf, 0xf, false);   // row_ror:4
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void timing_kernel(double* out, long long* cyc, int iters, double seed) {
    const int l = threadIdx.x;
    double a = seed + l * 1e-3, b = 1.0 + l * 1e-6, c = 0.5 + l * 1e-9, c2 = 0.25, c3 = 0.125, c4 = 0.0625;
    d4 acc = {0.1, 0.2, 0.3, 0.4};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (MODE == 0) c = fma(a, b, c);                       // dep via addend
        else if (MODE == 1) c = fma(c, a, b);                  // dep via multiplicand
        else if (MODE == 2) c = c + a;                         // add dep
        else if (MODE == 3) c = c * b;                         // mul dep
        else if (MODE == 4) c = __builtin_amdgcn_rcp(c);       // rcp dep (rcp(rcp(x)))
        else if (MODE == 5) c = __builtin_amdgcn_sqrt(c) + 0.0;  // sqrt
        else if (MODE == 6) c = dpp_ror4(c);                   // 2 dpp movs dep
        else if (MODE == 7) c = dpp_ror4(c) + a;               // dpp + add
        else if (MODE == 8) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);   // dep via C
        else if (MODE == 9) {                                  // 8 independent fma
            c = fma(a, b, c); c2 = fma(a, b, c2); c3 = fma(a, b, c3); c4 = fma(a, b, c4);
            acc[0] = fma(a, b, acc[0]); acc[1] = fma(a, b, acc[1]); acc[2] = fma(a, b, acc[2]); acc[3] = fma(a, b, acc[3]);
        } else if (MODE == 10) {                               // full-precision reciprocal: rcp + 2 Newton steps
            double r = __builtin_amdgcn_rcp(c);
            r = fma(fma(-c, r, 1.0), r, r);
            r = fma(fma(-c, r, 1.0), r, r);
            c = r + 1.5;
        } else if (MODE == 11) {                               // mfma 4x4x4 -> fma(via addend) -> mfma
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c, 0.0, 0, 0, 0);
            c = fma(a, b, c);
        } else if (MODE == 12) {                               // mfma -> mul -> mfma
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c, 0.0, 0, 0, 0);
            c = c * b;
        } else if (MODE == 13) {                               // fma -> fma(mult) -> fma(addend) mixture, 3 ops
            c = fma(c, a, b); c = fma(a, b, c); c = c * b;
        } else if (MODE == 14) {                               // readlane broadcast + add
            c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(c), 5), __builtin_amdgcn_readlane(__double2loint(c), 5)) + a;
        } else if (MODE == 15) {                               // ds_swizzle? use __shfl_xor 4
            c = __shfl_xor(c, 4) + a;
        } else if (MODE == 16) {
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        } else if (MODE == 17) {
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c, 0.0, 0, 0, 0);
        } else if (MODE == 18) {
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(c, b, 0.0, 0, 0, 0);
        } else if (MODE == 19) {
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0); c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
        } else if (MODE == 20) {
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c, 0.0, 0, 0, 0); c = __builtin_amdgcn_mfma_f64_4x4x4f64(c, b, 0.0, 0, 0, 0);
        }
      }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + l] = c + c2 + c3 + c4 + acc[0] + acc[1] + acc[2] + acc[3];
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run_timing(const char* name, int per_iter, int nblocks = 1, int threads = 64) {
    double* out; long long* cyc;
    CK(hipMalloc(&out, nblocks * threads * sizeof(double)));
    CK(hipMalloc(&cyc, nblocks * sizeof(long long)));
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((timing_kernel<MODE>), dim3(nblocks), dim3(threads), 0, 0, out, cyc, iters, 0.5);
        CK(hipDeviceSynchronize());
    }
    std::vector<long long> h(nblocks);
    CK(hipMemcpy(h.data(), cyc, nblocks * sizeof(long long), hipMemcpyDeviceToHost));
    printf("%-52s blocks=%4d thr=%3d cycles/op = %8.3f\n", name, nblocks, threads, (double)h[0] / iters / per_iter / 16);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    run_timing<0>("fma dep via addend", 1);
    run_timing<1>("fma dep via multiplicand", 1);
    run_timing<2>("add dep", 1);
    run_timing<3>("mul dep", 1);
    run_timing<4>("rcp dep", 1);
    run_timing<5>("sqrt(+0) dep", 1);
    run_timing<6>("2x dpp row_ror:4 dep", 1);
    run_timing<7>("2x dpp + add dep", 1);
    run_timing<8>("mfma 16x16x4 f64 dep via C", 1);
    run_timing<9>("8 independent fma", 8);
    run_timing<10>("rcp + 2 Newton + add", 1);
    run_timing<11>("mfma4x4x4 -> fma(addend)", 1);
    run_timing<12>("mfma4x4x4 -> mul", 1);
    run_timing<13>("fma(mult) fma(addend) mul chain (3 ops)", 3);
    run_timing<14>("readlane x2 + add", 1);
    run_timing<15>("shfl_xor(double) + add", 1);
    run_timing<16>("mfma4x4x4 dep via C", 1);
    run_timing<17>("mfma4x4x4 dep via B", 1);
    run_timing<18>("mfma4x4x4 dep via A", 1);
    run_timing<19>("mfma4x4x4 4 independent", 4);
    run_timing<20>("mfma4x4x4 B then A chain (2 ops)", 2);
    run_timing<17>("mfma4x4x4 dep via B, 8 waves/CU", 1, 1, 512);
    run_timing<19>("mfma4x4x4 4 independent, 8 waves/CU", 4, 1, 512);
    // two / four waves per SIMD: 256 threads = 4 waves per block on one CU -> one per SIMD; 512 -> two per SIMD
    run_timing<0>("fma dep via addend, 4 waves/CU", 1, 1, 256);
    run_timing<0>("fma dep via addend, 8 waves/CU", 1, 1, 512);
    run_timing<9>("8 independent fma, 8 waves/CU", 8, 1, 512);
    run_timing<9>("8 independent fma, 16 waves/CU", 8, 1, 1024);
    return 0;
}
