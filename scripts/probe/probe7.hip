// Probe 7: accuracy of v_rcp_f64 and of its refinements (max relative error in units of 2^-53 over random inputs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k(const double* x, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = x[i];
    const double y0 = __builtin_amdgcn_rcp(s);
    double y1 = fma(fma(-s, y0, 1.0), y0, y0);
    double y2 = fma(fma(-s, y1, 1.0), y1, y1);
    const double e = fma(-s, y0, 1.0);
    const double yc = fma(y0, fma(e, e, e), y0);
    out[i] = y0; out[n + i] = y1; out[2 * n + i] = y2; out[3 * n + i] = yc;
}
int main() {
    const int n = 1 << 20;
    double* hx = (double*)malloc(n * 8); double* ho = (double*)malloc(4 * n * 8);
    srand(1);
    for (int i = 0; i < n; ++i) { double m = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX / RAND_MAX; hx[i] = ldexp(m, (rand() % 200) - 100) * ((i & 1) ? -1 : 1); }
    double *dx, *dout; CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dout, 4 * n * 8));
    CK(hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ho, dout, 4 * n * 8, hipMemcpyDeviceToHost));
    const char* names[4] = {"v_rcp_f64", "+ 1 Newton step", "+ 2 Newton steps (fast_rcp)", "+ 1 cubic step y0 (1 + e + e^2)"};
    for (int v = 0; v < 4; ++v) {
        double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double ex = 1.0L / (long double)hx[i];
            const double rel = (double)fabsl(((long double)ho[v * n + i] - ex) / ex);
            if (rel > worst) worst = rel;
        }
        printf("%-36s max relative error %.3e = %.2f x 2^-53\n", names[v], worst, worst / ldexp(1.0, -53));
    }
    return 0;
}
