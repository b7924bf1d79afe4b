// Probe 12: where do the single-wave workgroups of a forward-like kernel land after (a) a kernel of the same kind and
// (b) a backward-like kernel (256-thread workgroups with 66 KB of LDS)?  Each forward wave runs a fixed dependent fp64
// chain and records its HW_ID / XCC_ID and its own start / end time; the host prints, per predecessor, the kernel time and
// the histogram of "waves that shared their SIMD with k - 1 others".  (fwd_tile3_kernel takes 0.46 ms
// in a filter-only stream and 0.76 ms behind bwd_mv_tile3_kernel at B = 2048 = one wave per SIMD.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(64) fwd_like(double* out, unsigned* ids, long long* tt, int iters, double seed) {
    __shared__ double pad[64];                                   // (the real kernel has a small static LDS array too)
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
    for (int i = 0; i < iters; ++i) {                            // dependent chain: mfma + fma, like a filter step
        x = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, 1e-9, 0, 0, 0);
        x = fma(x, 0.25, 1e-3);
    }
    pad[threadIdx.x] = x;
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        ids[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        ids[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        tt[2 * blockIdx.x] = t0; tt[2 * blockIdx.x + 1] = t1;
    }
    out[blockIdx.x * 64 + threadIdx.x] = pad[63 - threadIdx.x];
}

__global__ void __launch_bounds__(256) bwd_like(double* buf, size_t n, int iters) {
    __shared__ double lds[66 * 128];                             // 66 KB: two or three workgroups per CU
    double acc = 0.0;
    for (int i = 0; i < iters; ++i) {
        const size_t j = ((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)i * 256 * gridDim.x) % n;
        acc += buf[j];
        lds[(threadIdx.x + i) % (66 * 128)] = acc;
    }
    __syncthreads();
    buf[((size_t)blockIdx.x * 256 + threadIdx.x) % n] = acc + lds[threadIdx.x];
}

int main(int argc, char** argv) {
    const int n_wg = argc > 1 ? atoi(argv[1]) : 1024;
    const int iters = 12000;
    double* out; unsigned* ids; long long* tt; double* big;
    const size_t nbig = (size_t)1 << 28;                         // 2 GiB
    CK(hipMalloc(&out, (size_t)n_wg * 64 * 8)); CK(hipMalloc(&ids, (size_t)n_wg * 8)); CK(hipMalloc(&tt, (size_t)n_wg * 16));
    CK(hipMalloc(&big, nbig * 8)); CK(hipMemset(big, 0, nbig * 8));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned> h(2 * n_wg); std::vector<long long> ht(2 * n_wg);
    auto report = [&](const char* what, float ms) {
        CK(hipMemcpy(h.data(), ids, (size_t)n_wg * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ht.data(), tt, (size_t)n_wg * 16, hipMemcpyDeviceToHost));
        std::map<unsigned long long, int> per_simd;
        for (int w = 0; w < n_wg; ++w) {
            const unsigned id = h[2 * w], xcc = h[2 * w + 1] & 0xf;
            const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            per_simd[((unsigned long long)xcc << 32) | (se << 16) | (sh << 12) | (cu << 4) | simd]++;
        }
        int hist[9] = {0}; long long tmin = ht[0], tmax = ht[1]; double dur = 0;
        for (auto& kv : per_simd) hist[kv.second > 8 ? 8 : kv.second] += kv.second;
        for (int w = 0; w < n_wg; ++w) { tmin = ht[2*w] < tmin ? ht[2*w] : tmin; tmax = ht[2*w+1] > tmax ? ht[2*w+1] : tmax; dur += (double)(ht[2*w+1] - ht[2*w]); }
        printf("%-34s kernel %.3f ms | distinct SIMDs used %zu | waves by SIMD occupancy 1:%d 2:%d 3:%d 4+:%d | mean wave time %.1f us, span %.1f us (100 MHz clock)\n",
               what, ms, per_simd.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7] + hist[8],
               dur / n_wg / 100.0, (double)(tmax - tmin) / 100.0);
    };
    auto fwd = [&]() { hipLaunchKernelGGL(fwd_like, dim3(n_wg), dim3(64), 0, st, out, ids, tt, iters, 0.5); };
    auto bwd = [&]() { hipLaunchKernelGGL(bwd_like, dim3(n_wg / 2), dim3(256), 0, st, big, nbig, 400); };
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        fwd(); fwd();
        CK(hipEventRecord(e0, st)); fwd(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
        report("forward after forward", ms);
        bwd();
        CK(hipEventRecord(e0, st)); fwd(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
        report("forward after backward-like", ms);
        bwd(); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st)); fwd(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
        report("... with a sync in between", ms);
    }
    return 0;
}
