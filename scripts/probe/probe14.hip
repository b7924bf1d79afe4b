// probe14: what bounds the inner loop of the dense GEMM (solve_dense.hip, wg_gemm_dma_body) -- fp64 MFMA 16x16x4 fed from LDS
// by 8 waves per CU.  Per "chunk" every wave multiplies NT tiles x 4 k-steps; variants of the read / MFMA order.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form scripts/probe/probe14.hip -o scripts/probe/bin/probe14
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int SK = 176, SR = 18, BUF = 2944, NT = 13, REP = 200;
__shared__ __attribute__((aligned(16))) double g_lds[4 * BUF];

template <int VAR>
__global__ void __launch_bounds__(512) k(double* out, long long* cyc, int nt_run, const double* src) {
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < 4 * BUF; e += blockDim.x) g_lds[e] = 1e-3 * (e % 97);
    __syncthreads();
    const double* As = g_lds + hi * SK + lo;             // k-major A
    const double* Bs = g_lds + 2 * BUF + lo * SR + hi;   // row-major B
    d4 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = d4{0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        const double* A0 = As + (r & 1) * BUF;
        const double* B0 = Bs + (r & 1) * BUF;
        if (VAR == 0 || VAR == 5 || VAR == 6) {           // tile by tile: 8 reads, 4 MFMAs
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int e = wave + 8 * q, ti = e / 10, tj = e - 10 * ti;
                double fa[4], fb[4];
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) { fa[kq] = A0[16 * ti + 4 * SK * kq]; fb[kq] = B0[16 * SR * tj + 4 * kq]; }
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kq], fb[kq], acc[q], 0, 0, 0);
            }
        } else if (VAR == 1) {                            // software-pipelined: reads of tile q + 1 before the MFMAs of tile q
            double fa[2][4], fb[2][4];
            { const int e = wave, ti = e / 10, tj = e - 10 * ti;
#pragma unroll
              for (int kq = 0; kq < 4; ++kq) { fa[0][kq] = A0[16 * ti + 4 * SK * kq]; fb[0][kq] = B0[16 * SR * tj + 4 * kq]; } }
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                if (q + 1 < NT) {
                    const int e = wave + 8 * (q + 1), ti = e / 10, tj = e - 10 * ti;
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) { fa[(q + 1) & 1][kq] = A0[16 * ti + 4 * SK * kq]; fb[(q + 1) & 1][kq] = B0[16 * SR * tj + 4 * kq]; }
                }
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[q & 1][kq], fb[q & 1][kq], acc[q], 0, 0, 0);
            }
        } else if (VAR == 2) {                            // MFMAs only (operands from registers)
            double fa = A0[0], fb = B0[0];
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc[q], 0, 0, 0);
        } else if (VAR == 3) {                            // reads only
            double s = 0;
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int e = wave + 8 * q, ti = e / 10, tj = e - 10 * ti;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) s += A0[16 * ti + 4 * SK * kq] + B0[16 * SR * tj + 4 * kq];
            }
            acc[0][0] += s;
        } else if (VAR == 4) {                            // rectangle: this wave's 5 row tiles x up to 3 column tiles, A fragments reused
            // wave = (rh, cg): rh = wave >> 2 row half, cg = wave & 3: column tiles cg, cg + 4, (cg + 8 if < 10)
            const int rh = wave >> 2, cg = wave & 3;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                double fa[5], fb[3];
#pragma unroll
                for (int i = 0; i < 5; ++i) fa[i] = A0[16 * (5 * rh + i) + 4 * SK * kq];
#pragma unroll
                for (int j = 0; j < 3; ++j) fb[j] = B0[16 * SR * (cg + 4 * j < 10 ? cg + 4 * j : cg) + 4 * kq];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[0], acc[i], 0, 0, 0);
                    acc[5 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[1], acc[5 + i], 0, 0, 0);
                }
                if (cg < 2) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[10 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[2], acc[10 + i], 0, 0, 0);
                }
            }
        }
        if (VAR == 5 || VAR == 6 || VAR == 7) {
            if (VAR >= 6) {                               // the chunk's DMA loads: 6 per wave, 16 bytes per lane, into the other buffer pair
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int q = wave + 8 * (j % 3);
                    if (q < 22) {
                        const int gidx = q * 64 + lane;
                        const size_t off = (size_t)((r * 16 + gidx / 88) % 160) * 160 + 2 * ((gidx % 88) < 80 ? (gidx % 88) : 79) + (size_t)blockIdx.x * 25600;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                            (__attribute__((address_space(3))) void*)(g_lds + ((r + 1) & 1) * BUF + (j / 3) * 2 * BUF + q * 128), 16, 0, 0);
                    }
                }
            }
            // (VAR 7: the loads were issued BEFORE the MFMAs -- see above; here only the barrier)
            __syncthreads();
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int q = 0; q < NT; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VAR>
void run(const char* name, int threads) {
    double* out; long long* cyc; double* src; hipMalloc(&src, 256 * 25600 * 8); hipMemset(src, 0, 256 * 25600 * 8);
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(threads), 0, 0, out, cyc, 1, src);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= 256;
    const double per_chunk = m / REP;
    // MFMA pipe time per chunk per SIMD: (threads / 256) waves x 13 x 4 x 64
    printf("%-44s threads %3d: %8.0f cycles per chunk   (MFMA issue per SIMD: %d)\n", name, threads, per_chunk, (threads / 256) * 13 * 4 * 64);
    hipFree(out); hipFree(cyc); hipFree(src);
}
int main() {
    run<0>("tile by tile (8 reads -> 4 MFMAs)", 512);
    run<1>("pipelined (next tile's reads first)", 512);
    run<2>("MFMAs only", 512);
    run<3>("LDS reads only", 512);
    run<4>("rectangle 5 x 2(3): 8 reads -> 10-13 MFMAs", 512);
    run<5>("tile by tile + barrier per chunk", 512);
    run<6>("tile by tile + 6 DMA loads + barrier per chunk", 512);
    run<0>("tile by tile", 256);
    run<1>("pipelined", 256);
    run<2>("MFMAs only", 256);
    run<4>("rectangle", 256);
    return 0;
}
