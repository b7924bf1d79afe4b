// Probe 11: achievable HBM bandwidth on the box ("nominal peaks to be replaced by on-box measurements").
// Four streaming kernels over buffers far larger than the 256 MB Infinity Cache: read-only (sum), write-only (fill),
// copy, triad a = b + s*c; 16-byte accesses per lane, grid-stride, HIP-event timing, best of 5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_read(const double2* __restrict__ a, size_t n, double* sink) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = a[i]; acc += v.x + v.y;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}
__global__ void k_fill(double2* __restrict__ a, size_t n, double s) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = make_double2(s, s);
}
__global__ void k_copy(double2* __restrict__ a, const double2* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = b[i];
}
__global__ void k_triad(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c, size_t n, double s) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 x = b[i], y = c[i];
        a[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
    }
}

template <class F> static double best_ms(F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t bytes = (size_t)4 << 30, n = bytes / sizeof(double2);
    double2 *a, *b, *c; double* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
    for (int wg_per_cu : {4, 8, 16, 32}) {
        const dim3 grid(256 * wg_per_cu), block(256);
        const double r = best_ms([&] { hipLaunchKernelGGL(k_read, grid, block, 0, 0, a, n, sink); });
        const double f = best_ms([&] { hipLaunchKernelGGL(k_fill, grid, block, 0, 0, a, n, 1.0); });
        const double cp = best_ms([&] { hipLaunchKernelGGL(k_copy, grid, block, 0, 0, a, b, n); });
        const double t = best_ms([&] { hipLaunchKernelGGL(k_triad, grid, block, 0, 0, a, b, c, n, 0.5); });
        printf("grid %5d x 256: read %7.1f GB/s  fill %7.1f GB/s  copy %7.1f GB/s  triad %7.1f GB/s\n", grid.x,
               bytes / r * 1e-6, bytes / f * 1e-6, 2.0 * bytes / cp * 1e-6, 3.0 * bytes / t * 1e-6);
    }
    return 0;
}
