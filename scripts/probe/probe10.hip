// Probe 10: lane layout of v_mfma_f64_16x16x4_f64 (A, B one double per lane; C/D four doubles per lane) and its cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void layout(double* out) {
    const int l = threadIdx.x;
    for (int l0 = 0; l0 < 64; ++l0) {
        d4 c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(l + 1), l == l0 ? 1.0 : 0.0, c, 0, 0, 0);
        for (int v = 0; v < 4; ++v) out[(l0 * 64 + l) * 4 + v] = c[v];
    }
}
template <int NACC>
__global__ void cost(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int n = 0; n < iters; ++n) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[l] = s;
    if (l == 0) cyc[0] = t1 - t0;
}
template <int NACC> void run_cost() {
    double* out; long long* cyc; CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((cost<NACC>), dim3(1), dim3(64), 0, 0, out, cyc, iters); CK(hipDeviceSynchronize()); }
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%d independent accumulators: %.2f cycles per MFMA (2048 flop each)\n", NACC, (double)h / iters / NACC);
}
int main() {
    double* d; CK(hipMalloc(&d, 64 * 64 * 4 * 8));
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d); CK(hipDeviceSynchronize());
    static double h[64 * 64 * 4]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    // with B = delta at lane l0 and A(lane) = lane + 1: nonzero outputs (lane, v) carry the A lane that met B's (k, j)
    for (int l0 : {0, 1, 15, 16, 17, 33, 63}) {
        printf("B one-hot at lane %2d -> nonzero D entries (lane,vgpr:A-lane):", l0);
        int cnt = 0;
        for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) { double x = h[(l0 * 64 + l) * 4 + v]; if (x != 0 && cnt++ < 20) printf(" (%d,%d:%d)", l, v, (int)x - 1); }
        printf("  [%d]\n", cnt);
    }
    run_cost<1>(); run_cost<2>(); run_cost<4>(); run_cost<8>();
    return 0;
}
