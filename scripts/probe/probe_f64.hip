// Hardware probe (not part of the product): lane layout, issue rate and dependent latency of
// v_mfma_f64_4x4x4_4b_f64 / v_mfma_f64_16x16x4_f64, and fp64 VALU / rcp / DPP / bpermute costs on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---- layout probe: D = A*B (+0) with A,B filled so that each product identifies its (i,k),(k,j) ----
__global__ void layout_kernel(const double* a_in, const double* b_in, double* d_out) {
    const int l = threadIdx.x;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a_in[l], b_in[l], 0.0, 0, 0, 0);
    d_out[l] = d;
}

template <int MODE>
__global__ void timing_kernel(double* out, long long* cyc, int iters, double seed) {
    const int l = threadIdx.x;
    double a = seed + l * 1e-3, b = 1.0 + l * 1e-6, c = 0.5, c2 = 0.25, c3 = 0.125, c4 = 0.0625;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // dependent mfma 4x4x4 chain (D -> C)
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        } else if (MODE == 1) {     // dependent mfma chain through the B operand (D -> B)
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c, 0.0, 0, 0, 0);
        } else if (MODE == 2) {     // 4 independent mfma
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
        } else if (MODE == 3) {     // dependent fma chain
            c = fma(a, c, b);
        } else if (MODE == 4) {     // 4 independent fma
            c = fma(a, c, b); c2 = fma(a, c2, b); c3 = fma(a, c3, b); c4 = fma(a, c4, b);
        } else if (MODE == 5) {     // dependent rcp (hardware approx) chain
            c = __builtin_amdgcn_rcp(c + 1.5);
        } else if (MODE == 6) {     // dependent IEEE division chain
            c = 1.0 / (c + 1.5);
        } else if (MODE == 7) {     // dependent shfl (ds_bpermute) chain on a double
            c = __shfl(c, (l + 5) & 63) + 1.0;
        } else if (MODE == 8) {     // dependent mfma D -> A operand
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(c, b, 0.0, 0, 0, 0);
        } else if (MODE == 9) {     // mfma then dependent fma then mfma (mixed chain)
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c, 0.0, 0, 0, 0);
            c = fma(c, 0.5, b);
        } else if (MODE == 10) {    // dependent 16x16x4 chain
            using d4 = __attribute__((ext_vector_type(4))) double;
            static_assert(sizeof(d4) == 32, "");
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + l] = c + c2 + c3 + c4;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run_timing(const char* name, int per_iter, int nblocks = 1) {
    double* out; long long* cyc;
    CK(hipMalloc(&out, nblocks * 64 * sizeof(double)));
    CK(hipMalloc(&cyc, nblocks * sizeof(long long)));
    const int iters = 20000;
    hipLaunchKernelGGL((timing_kernel<MODE>), dim3(nblocks), dim3(64), 0, 0, out, cyc, iters, 0.5);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((timing_kernel<MODE>), dim3(nblocks), dim3(64), 0, 0, out, cyc, iters, 0.5);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(nblocks);
    CK(hipMemcpy(h.data(), cyc, nblocks * sizeof(long long), hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz constant?  report both ticks and wall-derived ns
    printf("%-44s blocks=%4d  memtime ticks/op = %8.3f   wall ns/op = %8.3f\n", name, nblocks,
           (double)h[0] / iters / per_iter, ms * 1e6 / iters / per_iter);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    // ---- layout ----
    std::vector<double> A(64), B(64), D(64);
    double *da, *db, *dd;
    CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dd, 512));
    // experiment 1: A lane l = 1 only for one lane (la), B = 1 for one lane (lb): find which D lanes light up
    printf("LAYOUT PROBE v_mfma_f64_4x4x4_4b_f64: for each (lane_a, lane_b) with D != 0 print lanes\n");
    // find for each A lane the set (block, i, k): do it by setting A[l]=1 for single lane, B = lane-coded values
    for (int la = 0; la < 64; ++la) {
        for (int l = 0; l < 64; ++l) { A[l] = (l == la) ? 1.0 : 0.0; B[l] = 100.0 + l; }
        CK(hipMemcpy(da, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, B.data(), 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
        CK(hipMemcpy(D.data(), dd, 512, hipMemcpyDeviceToHost));
        printf("A lane %2d ->", la);
        for (int l = 0; l < 64; ++l) if (D[l] != 0.0) printf("  D[%2d]=B[%2d]", l, (int)(D[l] - 100.0));
        printf("\n");
    }
    // ---- timing ----
    printf("\nTIMING (one wave unless blocks>1; s_memtime ticks; wall ns)\n");
    run_timing<0>("mfma4x4x4 dependent via C", 1);
    run_timing<1>("mfma4x4x4 dependent via B", 1);
    run_timing<8>("mfma4x4x4 dependent via A", 1);
    run_timing<2>("mfma4x4x4 4 independent", 4);
    run_timing<3>("v_fma_f64 dependent", 1);
    run_timing<4>("v_fma_f64 4 independent", 4);
    run_timing<5>("v_rcp_f64+add dependent", 1);
    run_timing<6>("IEEE div+add dependent", 1);
    run_timing<7>("shfl(double)+add dependent", 1);
    run_timing<9>("mfma(viaB)+fma dependent pair", 1);
    run_timing<2>("mfma4x4x4 4 independent, 1024 blocks", 4, 1024);
    run_timing<4>("v_fma_f64 4 independent, 1024 blocks", 4, 1024);
    run_timing<4>("v_fma_f64 4 independent, 2048 blocks", 4, 2048);
    return 0;
}
