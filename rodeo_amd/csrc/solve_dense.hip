// Dense large-block solver (BASELINE config 5: the "non-block" form of src/rodeo/prior/indep_init.py:8-23 -- one
// dense block of n_vars * n_deriv states with n_bmeas = n_vars measurements; examples/solve_nb.py:55-260 is the
// reference's own statement of this path).  One 512-thread workgroup (8 waves, one per CU) per trajectory runs the
// whole time loop of
//   src/rodeo/solve.py:31-122 (forward),  src/rodeo/solve.py:257-301 (backward mean/variance smoother)  and
//   src/rodeo/solve.py:137-205 (backward sampler of solve_sim)
// with every p x p operand in global memory and fp64 MFMA products; all LU solves use partial pivoting like
// src/rodeo/utils.py:119.  Right-hand sides: the linear ODE of config 5 inside the kernels; any other one through an
// interrogation kernel that hiprtc builds around it, launched between the two halves of the forward step
// (solve_dense_itg_kernels.hpp).  All four interrogations (interrogate.py:13-115).
//
// Layout: trajectory-major = the reference's own layout with a leading batch axis:
//   mean (B, N+1, p), var (B, N+1, p, p)  (n_block = 1), so a workgroup streams contiguous p x p matrices.
//
// Structure (what made it 2.1x faster than the first version, which inlined a templated GEMM at every call site and
// ran one wave per SIMD):
//   * ONE instance of every building block.  The kernels are descriptor loops over a step's phases; wg_gemm,
//     wg_lu_solve, the rank-16 update, the panel factorisation and the block-diagonal products are real (non-inlined)
//     functions.  Their arguments arrive in VGPRs, so every uniform one is made scalar again (uni()), and the LDS is a
//     file-scope array (through a pointer argument it would be addressed with flat loads).  Kernel code: 8 KB.
//   * wg_gemm: both operands staged through LDS in k-chunks of 32 (a transposed operand only changes its staging),
//     16 x 16 tiles of v_mfma_f64_16x16x4_f64, accumulators in registers across the chunks.
//   * wg_lu_solve: right-looking, 16-column panels.  Panel: wave 0 alone, rows in registers, each row carries its
//     LAPACK position (no physical interchanges), pivot search by DPP reductions, pivot row broadcast through LDS.
//     The panel's net row permutation is applied to the trailing columns and right-hand sides in one pass (one
//     thread per column).  Triangular solves thread-per-column from the LDS panel; trailing matrix AND right-hand
//     sides updated by rank-16 MFMA tiles whose A fragments come from the LDS panel and B fragments from an LDS strip.
//   * products with the block-diagonal Q of prior.indep_init skip their exact-zero terms (checked on the device).
// hipcc pitfalls met here: `if (s == runtime_slot)` selections over a register array become a dynamic index and
// move the array to scratch (use selects / per-row state); all 120 LDS reads of a 16 x 16 triangular solve are issued
// up front and spill unless each row's reads are tied to the previous result; a spill reload in a loop costs an
// s_waitcnt vmcnt(0), i.e. it serialises every outstanding global load.
// Measured on C5 (B = 256, p = 160, m = 32): 101 -> 46 ms per 50 steps = 19.6 % of the fp64 peak on the reference
// algorithm's flop count.  The phases now run at the chip's memory limit: a step moves ~8 MB per trajectory (the LU's
// ten trailing updates alone 3 MB), 2 GB over the 256 workgroups, far beyond the L2s; the rank-16 updates reach
// 7 TB/s aggregate.  Next step would be an LU whose [Sigma^- | T^T] stays in registers (200 tiles over 8 waves).
// Phase timing: build with -DRK_DENSE_STAMPS, run scripts/bench_configs.py c5 with RK_DENSE_STAMPS=1.
#include <type_traits>
#include <cstdlib>
#include "common.hpp"
#include "linalg_small.hpp"
#include "solve_args.hpp"
#include "solve_dense_itg_kernels.hpp"
#include "philox.hpp"

namespace rk {

struct DenseArgs {
    int B, N, p, m, itg;
    double t_min, t_max;
    const double *W, *x0, *Q, *R, *theta;      // W (m,p), Q,R (p,p) shared; x0 (p[,B]); theta = A (m,m[,B])
    int x0_b, theta_b;
    double *mean, *var;                        // (B, N+1, p), (B, N+1, p, p)
    double* ws;                                // workspace, ws_stride doubles per trajectory
    size_t ws_stride;
    int n0;                                    // dense_fwd_kernel<2>: the step whose update it runs
    unsigned long long seed, traj_offset;      // Philox stream of the draws (interrogate_chkrebtii, solve_sim)
    double* x;                                 // solve_sim: draws, batch-minor like the lane kernels: x[(n p + i) B + b]
    double *mean_pred, *var_pred;              // RK_FLAG_STORE_PRED (_solve_filter): (B, N+1, p), (B, N+1, p, p), or null
    double* fac_pred;                          // square-root form: the predicted factors L^-_n (B, N+1, p, p) for the backward pass, or null
    int lu_regs;                               // wg_lu_solve: 1 = register-resident forward elimination (default), 0 = panel loop over memory
};

constexpr int DT = 512, NWAVE = DT / 64;

// Optional phase timing (compile with -DRK_DENSE_STAMPS): cycles per phase, summed over the steps of workgroup 0,
// in the spare doubles before the block-diagonal flag at the end of trajectory 0's workspace.
#ifdef RK_DENSE_STAMPS
#define RK_STAMP_DECL(ws_end) double* const stamps_ = (double*)(ws_end); long long stamp_t_ = __builtin_amdgcn_s_memtime()
#define RK_STAMP(k) do { __syncthreads(); if (stamps_ && blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); stamps_[-2 - (k)] += (double)(t_ - stamp_t_); stamp_t_ = t_; } } while (0)
#define RK_STAMP_RESET() stamp_t_ = __builtin_amdgcn_s_memtime()
#define RK_STAMP_ZERO() do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < 14; ++k_) stamps_[-2 - k_] = 0.0; __syncthreads(); } while (0)
#define RK_STAMP_ZERO_N(n_) do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < (n_); ++k_) stamps_[-2 - k_] = 0.0; __syncthreads(); } while (0)
#else
#define RK_STAMP_ZERO_N(n_)
#define RK_STAMP_DECL(ws_end)
#define RK_STAMP(k)
#define RK_STAMP_RESET()
#define RK_STAMP_ZERO()
#endif

typedef double d4 __attribute__((ext_vector_type(4)));

// Arguments of a real (non-inlined) device function arrive in VGPRs and everything derived from them counts as
// divergent (vector compares, exec-mask branches, per-lane addresses).  These make the workgroup-uniform ones scalar again.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <class T>
__device__ __forceinline__ T* uni(T* ptr) {                 // generic pointer (e.g. a descriptor on the caller's stack)
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi_ << 32) | lo_);
}
// ... of a pointer into GLOBAL memory: rebuilt from scalars a generic pointer loses its address space and every access
// through it becomes a flat load / store (64-bit per-lane address, both wait counters).  The result is therefore TYPED
// as a global pointer (gd / cgd / gi below) and stays so through the helpers: scalar base + 32-bit lane offset.
typedef __attribute__((address_space(1))) double gd;
typedef __attribute__((address_space(1))) const double cgd;
typedef __attribute__((address_space(1))) int gi;
__device__ __forceinline__ unsigned long long uni_bits(const void* ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi_ << 32) | lo_;
}
__device__ __forceinline__ gd* uni_g(double* ptr) { return (gd*)uni_bits(ptr); }
__device__ __forceinline__ cgd* uni_g(const double* ptr) { return (cgd*)uni_bits(ptr); }
__device__ __forceinline__ gi* uni_g(int* ptr) { return (gi*)uni_bits(ptr); }

constexpr int LDS_DOUBLES = 17488;               // 136.6 KiB: GEMM staging / LU panel + U strip / register-resident LU (one workgroup per CU)
// The workgroup's LDS, at file scope so that the real (non-inlined) device functions below address it as LDS: through
// a pointer argument they would see a generic pointer and emit flat loads.
__shared__ __attribute__((aligned(16))) double g_lds[LDS_DOUBLES];
constexpr int LU_NB = 16, LU_MAXN = 320, LU_LD = LU_NB + 1;
constexpr int BD_MAX = 8, BD_MAXP = 320;
__shared__ int g_ppiv[LU_NB], g_cur[LU_MAXN], g_mlist[2 * LU_NB], g_mcount;
__shared__ double g_rdiag[LU_NB], g_rowbuf[2 * LU_NB];
__shared__ double g_qd[BD_MAXP * BD_MAX];         // diagonal blocks of a block-diagonal Q

// ---------------------------------------------------------------------------------------------------------------
// C (M x N, ldc) = ce * E + cab * op(A) op(B); row-major operands in global memory, the whole workgroup cooperates.
// (A rectangle-per-wave tile ownership -- 3 A + 5 B fragments from LDS for 15 MFMAs instead of two reads per MFMA -- was built
// and measured 10-35 % SLOWER in round 3: the tiles' per-MFMA guards cost more than the fragment reads save.)
// (So was a software-pipelined staging -- the next chunk's operands loaded into 24 registers per thread behind the current chunk's
// MFMAs, stored to LDS after the barrier: with 13 accumulator tiles per wave the function spills (253 scratch instructions, most
// of them in the chunk loop) and the products ran 30-50 % slower.)
// (And the same staging restricted to the skinny m x p / p x m products of the update -- 3 accumulator tiles, no spills -- also
// measured slower: 89 -> 115 k, 37 -> 53 k, 59 -> 69 k cycles.  The staging round trips are not what these products wait for.)
// v_mfma_f64_16x16x4_f64 with BOTH operands staged through LDS in k-chunks of 32 (As[k][i], Bs[k][j], row stride 177
// doubles: the fragment reads and the transposing stores are both at most 2-way on the banks), so a transposed operand
// only changes how its chunk is staged and there is ONE copy of the inner loop: the kernels issue their products
// from a descriptor loop (GemmOp), not from inlined call sites -- the first version of this file inlined a templated
// GEMM at every call site (22 000 instructions per kernel, several times the instruction cache) and ran one wave per
// SIMD, so nothing overlapped its MFMAs.  The output is processed in 160 x 160 macro-blocks of 16 x 16 tiles; wave w
// owns tiles w, w + 16, ... (at most 7) and keeps their accumulators in registers across the k-chunks.
// E may alias C (each element is read and written by the same lane).
// Fragment layout (profiles/r01_probe10_mfma_f64_16x16x4.log): A lane = 16 k + i, B lane = 16 k + j,
// D[i][j] in lane 16 (i % 4) + j, register i / 4; one MFMA = 64 cycles.
// ---------------------------------------------------------------------------------------------------------------
struct GemmOp {
    double* C; int ldc;
    const double* A; int lda;
    const double* B; int ldb;
    int M, N, K;
    const double* E; int lde;
    double ce, cab;
    bool ta, tb;
};
constexpr int G_KC = 32, G_SP = 177, G_MB = 160, G_TPW = ((G_MB / 16) * (G_MB / 16) + NWAVE - 1) / NWAVE;
static_assert(2 * G_KC * G_SP <= LDS_DOUBLES, "GEMM staging exceeds the LDS buffer");
static_assert((G_MB / 16) * (G_MB / 16) <= G_TPW * NWAVE, "tiles per wave");

// dst[k][j] = src[k * ld + j] for k < kn, j < jn; zero elsewhere in the 32 x 160 chunk  (source rows -> LDS rows)
__device__ __forceinline__ void stage_rows(double* dst, cgd* src, int ld, int kn, int jn) {
    constexpr int IT = G_KC * G_MB / DT;
    static_assert(IT * DT == G_KC * G_MB && IT % 5 == 0, "chunk size / workgroup size");
    for (int q0 = 0; q0 < IT; q0 += 5) {
        double v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int e = threadIdx.x + (q0 + q) * DT, k = e / G_MB, j = e - k * G_MB;
            v[q] = (k < kn && j < jn) ? src[k * ld + j] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int e = threadIdx.x + (q0 + q) * DT, k = e / G_MB, j = e - k * G_MB;
            dst[k * G_SP + j] = v[q];
        }
    }
}
// dst[k][i] = src[i * ld + k]  (source rows -> LDS columns): 8 lanes cover 32 consecutive k of one source row
__device__ __forceinline__ void stage_cols(double* dst, cgd* src, int ld, int kn, int in) {
    for (int e = threadIdx.x; e < G_MB * 8; e += DT) {
        const int kg = e & 7, i = e >> 3;
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (4 * kg + u < kn && i < in) ? src[i * ld + 4 * kg + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) dst[(4 * kg + u) * G_SP + i] = v[u];
    }
}

__device__ __noinline__ void wg_gemm_staged(const GemmOp& g_) {
    double* const lds = g_lds;
    const GemmOp& gr = *uni(&g_);
    struct { gd* C; cgd *A, *B, *E; int ldc, lda, ldb, lde, M, N, K; double ce, cab; bool ta, tb; } g;
    g.C = uni_g(gr.C); g.ldc = uni(gr.ldc); g.A = uni_g(gr.A); g.lda = uni(gr.lda); g.B = uni_g(gr.B); g.ldb = uni(gr.ldb);
    g.M = uni(gr.M); g.N = uni(gr.N); g.K = uni(gr.K); g.E = uni_g(gr.E); g.lde = uni(gr.lde);
    g.ce = uni(gr.ce); g.cab = uni(gr.cab); g.ta = uni((int)gr.ta) != 0; g.tb = uni((int)gr.tb) != 0;
    double* const As = lds;
    double* const Bs = lds + G_KC * G_SP;
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
    const int wave = uni((int)(threadIdx.x >> 6));                  // tile bookkeeping stays scalar
    const int frag = hi * G_SP + lo;
    for (int i0 = 0; i0 < g.M; i0 += G_MB)
        for (int j0 = 0; j0 < g.N; j0 += G_MB) {
            const int mb = min(G_MB, g.M - i0), nbk = min(G_MB, g.N - j0);
            const int nt = (nbk + 15) >> 4, T = ((mb + 15) >> 4) * nt;
            d4 acc[G_TPW];
#pragma unroll
            for (int q = 0; q < G_TPW; ++q) acc[q] = d4{0, 0, 0, 0};
            for (int k0 = 0; k0 < g.K; k0 += G_KC) {
                const int kn = min(G_KC, g.K - k0);
                __syncthreads();                                    // the previous chunk has been consumed
                if (g.ta) stage_rows(As, g.A + (size_t)k0 * g.lda + i0, g.lda, kn, mb);      // A is K x M
                else      stage_cols(As, g.A + (size_t)i0 * g.lda + k0, g.lda, kn, mb);      // A is M x K
                if (g.tb) stage_cols(Bs, g.B + (size_t)j0 * g.ldb + k0, g.ldb, kn, nbk);     // B is N x K
                else      stage_rows(Bs, g.B + (size_t)k0 * g.ldb + j0, g.ldb, kn, nbk);     // B is K x N
                __syncthreads();
#pragma unroll
                for (int q = 0; q < G_TPW; ++q) {
                    const int e = wave + NWAVE * q;
                    if (e < T) {
                        const int ti = e / nt, tj = e - ti * nt;
                        const double* ap = As + frag + 16 * ti;
                        const double* bp = Bs + frag + 16 * tj;
#pragma unroll
                        for (int kq = 0; kq < G_KC / 4; ++kq)
                            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * kq * G_SP], bp[4 * kq * G_SP], acc[q], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < G_TPW; ++q) {
                const int e = wave + NWAVE * q;
                if (e < T) {
                    const int ti = e / nt, tj = e - ti * nt;
                    const int j = j0 + 16 * tj + lo;
                    double ev[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int ii = i0 + 16 * ti + 4 * v + hi;
                        ev[v] = (g.E && ii < g.M && j < g.N) ? g.E[ii * g.lde + j] : 0.0;
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int ii = i0 + 16 * ti + 4 * v + hi;
                        if (ii < g.M && j < g.N) g.C[ii * g.ldc + j] = fma(g.ce, ev[v], g.cab * acc[q][v]);
                    }
                }
            }
        }
    __syncthreads();
}


// ---------------------------------------------------------------------------------------------------------------
// Round 4: the same product with the operands brought into LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no
// ds_write pass) into TWO buffers per operand, k-chunks of 16: the loads of chunk c + 1 are in flight while the MFMAs of
// chunk c run, one barrier per chunk.  The staged version above did load -> LDS write -> barrier -> MFMA one after the
// other and spent as long waiting for its 80 KB per chunk as multiplying (stamps of round 4: 157 k / 189 k cycles per
// 160^3 product against 66 k of MFMA issue).  An LDS-DMA writes 64 x 16 bytes CONTIGUOUSLY (lane-linear), the source
// address is per lane: so the LDS images are plain linear arrays and the padding that keeps the fragment reads off each
// other's banks is filled from clamped source addresses --
//   * an operand whose memory rows run along k ("k-major": A given transposed, B given as is): image X[k][c], row stride
//     176 doubles (16 mod 32: the four k rows of a fragment read lie on disjoint banks), 22 DMA instructions per chunk;
//   * an operand whose memory rows run along i / j ("row-major": A as is, B given transposed): image Y[r][k], row stride 18
//     doubles (9 granules of 16 bytes: 8 of data, one of padding; 36 r + 2 k dwords: conflict-free), 23 instructions.
// Same MFMAs in the same k order as the staged version: the same bits.  Needs K a multiple of 16, even leading dimensions
// and 16-byte aligned operands (the DMA moves 16-byte granules); anything else takes the staged version.
// ---------------------------------------------------------------------------------------------------------------
constexpr int GD_KC = 16, GD_SK = 176, GD_SR = 18, GD_BUF = 2944;         // 23 x 128 doubles >= 16 x 176, 160 x 18
static_assert(4 * GD_BUF <= LDS_DOUBLES, "DMA GEMM buffers exceed the LDS buffer");

// one chunk of one operand: KM: rows k0 .. k0 + 15 of src (row stride ld), columns c0 .. (image X[k][c]);
// else rows r0 .. r0 + nr - 1, columns k0 .. k0 + 15 (image Y[r][k])
template <bool KM>
__device__ __forceinline__ void gd_issue(double* dst, cgd* src, int ld, int k0, int c0, int nvalid, int wave, int lane) {
    constexpr int NI = KM ? 22 : 23;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int q = wave + NWAVE * r;                             // DMA instruction: granules 64 q .. 64 q + 63 of the image
        if (q < NI) {
            const int gidx = q * 64 + lane;
            size_t off;
            if (KM) {
                const int k = gidx / 88, c2 = gidx - k * 88;
                const int col = min(2 * c2, max(ld - c0 - 2, 0));       // (padding columns: any readable pair of this row)
                off = (size_t)(k0 + k) * ld + c0 + col;
            } else {
                const int rr = gidx / 9, part = gidx - rr * 9;
                off = (size_t)(c0 + min(rr, nvalid - 1)) * ld + k0 + 2 * min(part, 7);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                             (__attribute__((address_space(3))) void*)(dst + q * 128), 16, 0, 0);
        }
    }
}

template <bool AKM, bool BKM>
__device__ __forceinline__ void wg_gemm_dma_body(gd* C, int ldc, cgd* A, int lda, cgd* B, int ldb, int M, int N, int K,
                                                 cgd* E, int lde, double ce, double cab) {
    double* const lds = g_lds;
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
    const int wave = uni((int)(threadIdx.x >> 6));
    const int fragA = AKM ? hi * GD_SK + lo : lo * GD_SR + hi;
    const int fragB = BKM ? hi * GD_SK + lo : lo * GD_SR + hi;
    constexpr int stepA = AKM ? 4 * GD_SK : 4, stepB = BKM ? 4 * GD_SK : 4;      // per kq
    constexpr int tileA = AKM ? 16 : 16 * GD_SR, tileB = BKM ? 16 : 16 * GD_SR;  // per tile index
    for (int i0 = 0; i0 < M; i0 += G_MB)
        for (int j0 = 0; j0 < N; j0 += G_MB) {
            const int mb = min(G_MB, M - i0), nbk = min(G_MB, N - j0);
            const int nt = (nbk + 15) >> 4, T = ((mb + 15) >> 4) * nt;
            d4 acc[G_TPW];
#pragma unroll
            for (int q = 0; q < G_TPW; ++q) acc[q] = d4{0, 0, 0, 0};
            // C = E +- A B (ce = 1, |cab| = 1: the smoother's Sigma_f + (G D) G^T, the filter's Sigma- - K (W~ Sigma-)): E goes into the
            // accumulators up front, its loads under the first chunk's DMA, instead of 13 load -> store round trips per wave at the
            // end (C may be E: no load of the next tile can pass a store of this one -- 43 k cycles of (GD)G^T against G D).  The sum
            // then runs E + a1 b1 + a2 b2 ... instead of E + (a1 b1 + ...): another rounding order, same terms.
            const bool pre = E != nullptr && ce == 1.0 && (cab == 1.0 || cab == -1.0);
            if (pre) {
#pragma unroll
                for (int q = 0; q < G_TPW; ++q) {
                    const int e = wave + NWAVE * q;
                    if (e < T) {
                        const int ti = e / nt, tj = e - ti * nt;
                        const int j = min(j0 + 16 * tj + lo, N - 1);
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[q][v] = cab * E[min(i0 + 16 * ti + 4 * v + hi, M - 1) * lde + j];     // (clamped, unmasked)
                    }
                }
            }
            auto issue = [&](int k0, int buf) {
                gd_issue<AKM>(lds + buf * GD_BUF, A, lda, k0, i0, mb, wave, lane);
                gd_issue<BKM>(lds + (2 + buf) * GD_BUF, B, ldb, k0, j0, nbk, wave, lane);
            };
            // (Round 4: THREE buffers per operand with counted waits -- s_waitcnt vmcnt(this wave's DMAs per chunk) + an LDS-only barrier, so
            //  that chunk c + 2 is in flight while chunk c is multiplied -- changed nothing: G D 141 k, (GD)G^T 168 k cycles as before.  A
            //  chunk of the 160 x 160 products is 52 MFMAs per wave = 6.7 k cycles of the SIMD's matrix pipe for its two waves (probe14:
            //  7.5 k with the LDS reads) against a ~5 k round trip that two buffers already hide; what is left over the 64 k of the
            //  MFMAs alone is the barrier per chunk, LDS reads that do not overlap MFMAs, and the first and last chunk.)
            __syncthreads();                                        // whoever used the LDS before is done with it
            issue(0, 0);
            for (int k0 = 0, buf = 0; k0 < K; k0 += GD_KC, buf ^= 1) {
                __syncthreads();                                    // chunk k0 has landed (hipcc drains the DMAs in front of the barrier); buffer buf ^ 1 is free
                if (k0 + GD_KC < K) issue(k0 + GD_KC, buf ^ 1);
                const double* const As = lds + buf * GD_BUF + fragA;
                const double* const Bs = lds + (2 + buf) * GD_BUF + fragB;
                // The tiles of this wave one after the other, reads and MFMAs alternating in pairs of k-steps.  Measured
                // (scripts/probe/probe14.hip, profiles/r04_probe14_gemm_inner_loop.log): one wave issues an fp64 16x16x4 MFMA per
                // 64 cycles and nothing else meanwhile; two waves of a SIMD reach one per 32 only with operands in registers --
                // fed from LDS their read phases and MFMA phases take turns and the SIMD settles at one MFMA per 64 cycles
                // whatever the order.  Tried in this loop and measured slower: the fragments of tile q + 1 requested ahead of
                // tile q's MFMAs (+19 %), all of a k-step's fragments first and then its 13 MFMAs (+18 %), four reads and four
                // MFMAs per tile in straight-line code (+11 %).
#pragma unroll
                for (int q = 0; q < G_TPW; ++q) {
                    const int e = wave + NWAVE * q;
                    if (e < T) {
                        const int ti = e / nt, tj = e - ti * nt;
                        const double* const ap = As + tileA * ti;
                        const double* const bp = Bs + tileB * tj;
#pragma unroll
                        for (int kq = 0; kq < GD_KC / 4; ++kq)
                            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[stepA * kq], bp[stepB * kq], acc[q], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < G_TPW; ++q) {
                const int e = wave + NWAVE * q;
                if (e < T) {
                    const int ti = e / nt, tj = e - ti * nt;
                    const int j = j0 + 16 * tj + lo;
                    double ev[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int ii = i0 + 16 * ti + 4 * v + hi;
                        ev[v] = (E && !pre && ii < M && j < N) ? E[ii * lde + j] : 0.0;
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int ii = i0 + 16 * ti + 4 * v + hi;
                        if (ii < M && j < N) C[ii * ldc + j] = pre ? cab * acc[q][v] : fma(ce, ev[v], cab * acc[q][v]);
                    }
                }
            }
        }
    __syncthreads();
}

// (Round 4: a barrier-free path for the SKINNY products of the filter step -- M = n_bmeas = 32 rows, a wave owning whole column tiles and
//  feeding its MFMAs straight from memory, 60 fragment loads in flight per lane -- was measured slower than the chunked path it was
//  meant to replace: W~ Sigma- 75 k -> 107 k cycles, (Sigma- W~^T)^T 47 k -> 97 k: 8-byte fragment loads spread over 16 rows per
//  instruction cost the address path more than the ten barriers and round trips they avoid.)
__device__ __noinline__ void wg_gemm(const GemmOp& g_) {
    const GemmOp& gr = *uni(&g_);
    const int K = uni(gr.K), lda = uni(gr.lda), ldb = uni(gr.ldb);
    const unsigned long long pa = uni_bits(gr.A), pb = uni_bits(gr.B);
    if ((K & (GD_KC - 1)) != 0 || ((lda | ldb) & 1) != 0 || ((pa | pb) & 15) != 0) { wg_gemm_staged(g_); return; }
    auto* const C = uni_g(gr.C);
    auto* const A = uni_g(gr.A);
    auto* const B = uni_g(gr.B);
    auto* const E = uni_g(gr.E);
    const int ldc = uni(gr.ldc), M = uni(gr.M), N = uni(gr.N), lde = uni(gr.lde);
    const double ce = uni(gr.ce), cab = uni(gr.cab);
    const bool ta = uni((int)gr.ta) != 0, tb = uni((int)gr.tb) != 0;
    // A given transposed (K x M): its rows run along k; B given as is (K x N): likewise
    if (ta) { if (!tb) wg_gemm_dma_body<true, true>(C, ldc, A, lda, B, ldb, M, N, K, E, lde, ce, cab);
              else wg_gemm_dma_body<true, false>(C, ldc, A, lda, B, ldb, M, N, K, E, lde, ce, cab); }
    else    { if (!tb) wg_gemm_dma_body<false, true>(C, ldc, A, lda, B, ldb, M, N, K, E, lde, ce, cab);
              else wg_gemm_dma_body<false, false>(C, ldc, A, lda, B, ldb, M, N, K, E, lde, ce, cab); }
}

// y (M) = ce * e + cab * op(A) x ; A (M x K) or, if TA, A is (K x M) and op(A) = A^T
template <bool TA>
__device__ __forceinline__ void wg_gemv(double* y, const double* A, int lda, const double* x, int M, int K, const double* e,
                        double ce, double cab) {
    for (int i = threadIdx.x; i < M; i += DT) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s = fma(TA ? A[(size_t)k * lda + i] : A[(size_t)i * lda + k], x[k], s);
        s *= cab;
        if (e) s = fma(ce, e[i], s);
        y[i] = s;
    }
    __syncthreads();
}

// In-place LU with partial pivoting of A (n x n, lda) -- first maximum of |a_ik| like LAPACK getrf -- then
// X = A^{-1} Bm for the nr right-hand-side columns of Bm (n x nr, ldb), overwritten.  piv: n ints in global memory.
__device__ __noinline__ void wg_lu_solve_unblocked(double* lds, double* A, int lda, double* Bm, int ldb, int n, int nr, int* piv) {
    double* const red_v = lds;                             // DT doubles
    int* const red_i = (int*)(lds + DT);                   // DT ints
    for (int k = 0; k < n; ++k) {
        // pivot search
        double best = -1.0;
        int bi = k;
        for (int i = k + threadIdx.x; i < n; i += DT) {
            const double v = fabs(A[(size_t)i * lda + k]);
            if (v > best) { best = v; bi = i; }             // ascending i per thread: first maximum
        }
        red_v[threadIdx.x] = best;
        red_i[threadIdx.x] = bi;
        __syncthreads();
        for (int s = DT / 2; s > 0; s >>= 1) {
            if (threadIdx.x < s) {
                const double v2 = red_v[threadIdx.x + s];
                const int i2 = red_i[threadIdx.x + s];
                if (v2 > red_v[threadIdx.x] || (v2 == red_v[threadIdx.x] && i2 < red_i[threadIdx.x])) {
                    red_v[threadIdx.x] = v2;
                    red_i[threadIdx.x] = i2;
                }
            }
            __syncthreads();
        }
        const int pk = red_i[0];
        if (threadIdx.x == 0) piv[k] = pk;
        if (pk != k)
            for (int j = threadIdx.x; j < n; j += DT) {
                const double t = A[(size_t)k * lda + j];
                A[(size_t)k * lda + j] = A[(size_t)pk * lda + j];
                A[(size_t)pk * lda + j] = t;
            }
        __syncthreads();
        const double r = 1.0 / A[(size_t)k * lda + k];
        for (int i = k + 1 + threadIdx.x; i < n; i += DT) A[(size_t)i * lda + k] *= r;
        __syncthreads();
        // trailing update: rows k+1.., columns k+1..
        const int rem = n - k - 1;
        for (int e = threadIdx.x; e < rem * rem; e += DT) {
            const int i = k + 1 + e / rem, j = k + 1 + e % rem;
            A[(size_t)i * lda + j] = fma(-A[(size_t)i * lda + k], A[(size_t)k * lda + j], A[(size_t)i * lda + j]);
        }
        __syncthreads();
    }
    // each thread solves whole right-hand-side columns: row swaps, forward (unit lower), backward (upper)
    for (int c = threadIdx.x; c < nr; c += DT) {
        for (int k = 0; k < n; ++k) {                       // P b: all row interchanges first (LAPACK laswp)
            const int pk = piv[k];
            if (pk != k) {
                const double t = Bm[(size_t)k * ldb + c];
                Bm[(size_t)k * ldb + c] = Bm[(size_t)pk * ldb + c];
                Bm[(size_t)pk * ldb + c] = t;
            }
        }
        for (int k = 0; k < n; ++k) {
            const double bk = Bm[(size_t)k * ldb + c];
            for (int i = k + 1; i < n; ++i)
                Bm[(size_t)i * ldb + c] = fma(-A[(size_t)i * lda + k], bk, Bm[(size_t)i * ldb + c]);
        }
        for (int k = n - 1; k >= 0; --k) {
            double s = Bm[(size_t)k * ldb + c];
            for (int i = k + 1; i < n; ++i) s = fma(-A[(size_t)k * lda + i], Bm[(size_t)i * ldb + c], s);
            Bm[(size_t)k * ldb + c] = s / A[(size_t)k * lda + k];
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// Blocked LU solve (right-looking, panels of 16 columns): same pivots (first maximum of |a_ik|, LAPACK getrf) and the
// same L, U up to the order of summation; divisions by the pivots are multiplications with their reciprocals (as
// getf2 scales its columns).  Per panel:
//   1. the panel (rows k0.., 16 columns) is copied to LDS and factored by WAVE 0 ALONE, wave-synchronously (no
//      workgroup barrier inside the 16 columns; maximum and first index through DPP row reductions + readlane);
//   2. the net row permutation of the panel's 16 interchanges is applied to the columns right of the panel and to the
//      right-hand sides in one parallel gather / scatter (the columns left of it hold L, which is never read again
//      because the right-hand sides are eliminated on the fly);
//   3. U12 = L11^{-1} A12 and B1 = L11^{-1} B1, one thread per column, written to memory and to an LDS strip;
//   4. the rank-16 update of the trailing matrix AND of the right-hand sides: 16 x 16 tiles, A fragments from the LDS
//      panel, B fragments from the LDS strip, four MFMAs per tile straight onto the loaded C tile.
// Back substitution: per 16-row block from the bottom, a thread-per-column solve with U11 and the same rank-16 update
// of the rows above.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {        // LDS traffic between the lanes of ONE wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace rk
#include "solve_dense_panel.hpp"
namespace rk {

// wave 0: factor the LDS panel (rows x nb, row stride LU_LD) in place; g_ppiv[j] = pivot row of column j
// (panel-relative); g_cur[r] = the original (panel-relative) row that ends up in position r.
// Lane l keeps rows l, l + 64, ... (RS of them) in registers for the whole panel and the rows never move: each row
// carries its current POSITION (what LAPACK's interchanges would have made of it), the pivot search is "largest |.|,
// smallest position", an interchange swaps two positions, the pivot row reaches the lanes through readlane, the rank-1
// update multiplies the rows that are out of play by an exact zero, and the rows are written back to the panel AT their
// positions.  The 16 columns are unrolled through templates, so every register index is a constant (with
// `if (s == ps)` selections hipcc makes the index dynamic and moves the rows to scratch).  The column step is
// rl_panel_col (round 4: written for the register-resident elimination, then taken over here -- the first column step,
// DPP fp64 maximum + minimum of positions, the pivot row through a 16-double LDS buffer between two wave syncs and an
// exec-masked update, took ~2.4 k cycles per column).
template <int RS, int J>
__device__ __forceinline__ void lu_panel_col(double (&a)[RS][LU_NB], int (&pos)[RS], int l, int rows) {
    const int pj = rl_panel_col<RS, J>(a, pos, J, rows);
    if (l == 0) g_ppiv[J] = pj;
}
template <int RS, int J>
__device__ __forceinline__ void lu_panel_cols(double (&a)[RS][LU_NB], int (&pos)[RS], int nb, int l, int rows) {
    if constexpr (J < LU_NB) {
        if (J < nb) lu_panel_col<RS, J>(a, pos, l, rows);
        lu_panel_cols<RS, J + 1>(a, pos, nb, l, rows);
    }
}
template <int RS>
__device__ __forceinline__ void lu_panel_regs(int rows, int nb, int poff) {
    double* const panel = g_lds + poff;
    const int l = threadIdx.x;                             // lane of wave 0
    double a[RS][LU_NB];
    int pos[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        pos[s] = (l + 64 * s < rows) ? l + 64 * s : 0x7fffffff;
#pragma unroll
        for (int c = 0; c < LU_NB; ++c) a[s][c] = (l + 64 * s < rows) ? panel[(l + 64 * s) * LU_LD + c] : 0.0;
    }
    wave_lds_sync();                                        // every row is in registers before any is written back
    lu_panel_cols<RS, 0>(a, pos, nb, l, rows);
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        if (pos[s] != 0x7fffffff) {
            g_cur[pos[s]] = l + 64 * s;
#pragma unroll
            for (int c = 0; c < LU_NB; ++c) panel[pos[s] * LU_LD + c] = a[s][c];
        }
    }
}

__device__ __noinline__ void lu_panel_wave1(int rows_, int nb_, int poff_ = 0) { lu_panel_regs<1>(uni(rows_), uni(nb_), uni(poff_)); }   // rows <= 64
__device__ __noinline__ void lu_panel_wave2(int rows_, int nb_, int poff_ = 0) { lu_panel_regs<2>(uni(rows_), uni(nb_), uni(poff_)); }   // rows <= 128
__device__ __noinline__ void lu_panel_wave3(int rows_, int nb_, int poff_ = 0) { lu_panel_regs<3>(uni(rows_), uni(nb_), uni(poff_)); }   // rows <= 192
__device__ __noinline__ void lu_panel_wave5(int rows_, int nb_, int poff_ = 0) { lu_panel_regs<5>(uni(rows_), uni(nb_), uni(poff_)); }   // rows <= 320
// wave 0: the panel in the LDS buffer at poff, with as few register rows per lane as its row count needs (the column steps
// are a dependent chain on this one wave -- the critical path of the whole solve)
__device__ __forceinline__ void lu_panel_wave(int rows, int nb, int poff) {
    if (rows <= 64) lu_panel_wave1(rows, nb, poff);
    else if (rows <= 128) lu_panel_wave2(rows, nb, poff);
    else if (rows <= 192) lu_panel_wave3(rows, nb, poff);
    else lu_panel_wave5(rows, nb, poff);
}

// rank-16 update  C <- C - L U  for the tiles (rb, ct): rows r0 + 16 rb.., a strip of column tiles; L rows from the LDS
// panel `lp` (row stride LU_LD, row index rb * 16 + i, columns < nb), U from the LDS strip `us` (row stride usp).
// Column tiles ct < ctA address C1 (ld1, n1 valid columns), the others C2 (ld2, n2 valid columns) at strip offset offB.
__device__ __noinline__ void lu_rank_update(int lp_off, int lrows_, int nb_, int us_off, int usp_, int rt_,
                                            double* C1_, int ld1_, int n1_, int ctA_, double* C2_, int ld2_, int n2_, int ctB_,
                                            int offB_, int mrows_, int wave0_ = 0, int nwaves_ = NWAVE) {
    const double* const lp = g_lds + uni(lp_off);
    const double* const us = g_lds + uni(us_off);
    const int lrows = uni(lrows_), nb = uni(nb_), usp = uni(usp_), rt = uni(rt_), ld1 = uni(ld1_), n1 = uni(n1_), ctA = uni(ctA_);
    const int ld2 = uni(ld2_), n2 = uni(n2_), ctB = uni(ctB_), offB = uni(offB_), mrows = uni(mrows_);
    auto* const C1 = uni_g(C1_);
    auto* const C2 = uni_g(C2_);
    // the tiles are dealt to the waves wave0 .. wave0 + nwaves - 1 (look-ahead: wave 0 factors the next panel meanwhile)
    const int wave0 = uni(wave0_), nwaves = uni(nwaves_);
    const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6)) - wave0, lo = lane & 15, hi = lane >> 4;
    if (wave < 0 || wave >= nwaves) return;
    const int nct = ctA + ctB, tiles = rt * nct;
    // Tiles wave, wave + 8, ... in groups of RU_G: the C values of the NEXT group are loaded before the current group is
    // multiplied and stored (a tile is 4 loads, 4 MFMAs, 4 stores; without this every tile waits a full memory round trip).
    constexpr int RU_G = 4;
    auto tile_of = [&](int e, gd*& Cm, int& ld, int& cj, int& sj, int& rb, bool& cok) {
        rb = e / nct;
        const int ct = e - rb * nct;
        const bool inA = ct < ctA;
        Cm = inA ? C1 : C2;
        ld = inA ? ld1 : ld2;
        cj = (inA ? ct : ct - ctA) * 16 + lo;
        sj = (inA ? ct * 16 : offB + (ct - ctA) * 16) + lo;
        cok = cj < (inA ? n1 : n2);
    };
    auto load_group = [&](int e0, d4 (&c)[RU_G]) {
#pragma unroll
        for (int u = 0; u < RU_G; ++u) {
            const int e = e0 + u * nwaves;
            if (e < tiles) {
                gd* Cm; int ld, cj, sj, rb; bool cok;
                tile_of(e, Cm, ld, cj, sj, rb, cok);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int ii = rb * 16 + 4 * v + hi;
                    c[u][v] = (cok && ii < mrows) ? Cm[ii * ld + cj] : 0.0;
                }
            }
        }
    };
    d4 cur[RU_G], nxt[RU_G];
    load_group(wave, cur);
    for (int e0 = wave; e0 < tiles; e0 += RU_G * nwaves) {
        load_group(e0 + RU_G * nwaves, nxt);
#pragma unroll
        for (int u = 0; u < RU_G; ++u) {
            const int e = e0 + u * nwaves;
            if (e < tiles) {
                gd* Cm; int ld, cj, sj, rb; bool cok;
                tile_of(e, Cm, ld, cj, sj, rb, cok);
                const int li = min(rb * 16 + lo, lrows - 1);
                d4 acc = cur[u];
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const int k = 4 * kq + hi;
                    const double a = k < nb ? -lp[li * LU_LD + k] : 0.0;
                    const double b = us[k * usp + sj];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int ii = rb * 16 + 4 * v + hi;
                    if (cok && ii < mrows) Cm[ii * ld + cj] = acc[v];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RU_G; ++u) cur[u] = nxt[u];
    }
}

// x <- E11^{-1} x for ONE right-hand-side column per lane and a 16 x 16 triangular block E11 whose row j lies in LDS at
// prow0[j * LU_LD + 0 .. 15].  LOWER: forward substitution from row 0, else back substitution from row 15; UNIT: unit diagonal
// (the L of the LU), else x_j is scaled by rdiag[j] = 1 / e_jj.  Entries of x from nb on are zero on entry and on exit; columns
// >= nb of the staged block are zero or meet those zeros.
// Round 4: the 16 lanes of a DPP row share the block -- lane i keeps COLUMN i of it (16 registers, 16 LDS reads), and
// E[j][i] reaches the whole row inside the FMA (v_fmac_f64_dpp ... row_newbcast:i, the one 64-bit DPP form gfx90a+ has).  The
// first version read every entry as an LDS broadcast, one row ahead of the chain: 120 reads per column and ~5 k cycles per
// block (stamps of round 4); terms and their order are unchanged (even i into s0, odd i into s1, s0 + s1), so are the bits.
// Every 16-lane row that takes part must be entirely active (all call sites cut their column ranges at multiples of 16).
template <int I>
__device__ __forceinline__ void dense_fnmac_bc(double& acc, double e, double y) {       // acc -= e(lane I of the row) * y
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(e), "v"(y), "n"(I));
}
template <bool LOWER, int J, int I>
__device__ __forceinline__ void trsm16_row(double& s0, double& s1, const double (&x)[LU_NB], double ej) {
    if constexpr (I < LU_NB) {
        if constexpr (LOWER ? I < J : I > J) {
            if constexpr (I & 1) dense_fnmac_bc<I>(s1, ej, x[I]);
            else dense_fnmac_bc<I>(s0, ej, x[I]);
        }
        trsm16_row<LOWER, J, I + 1>(s0, s1, x, ej);
    }
}
template <bool LOWER, bool UNIT, int STEP>
__device__ __forceinline__ void trsm16_steps(double (&x)[LU_NB], double (&er)[LU_NB], int nb, const double* rdiag) {
    if constexpr (STEP < LU_NB) {
        constexpr int J = LOWER ? STEP : LU_NB - 1 - STEP;
        double s0 = x[J], s1 = 0.0;
        trsm16_row<LOWER, J, 0>(s0, s1, x, er[J]);
        const double sacc = s0 + s1;
        x[J] = J < nb ? (UNIT ? sacc : sacc * rdiag[J]) : 0.0;
        trsm16_steps<LOWER, UNIT, STEP + 1>(x, er, nb, rdiag);
    }
}
template <bool LOWER, bool UNIT>
__device__ __forceinline__ void trsm16(double (&x)[LU_NB], const double* prow0, int nb, const double* rdiag) {
    double er[LU_NB];
    const int li = threadIdx.x & 15;
#pragma unroll
    for (int j = 0; j < LU_NB; ++j) er[j] = prow0[j * LU_LD + li];
    // (a VALU write of a VGPR needs two wait states before a DPP operand reads it; the block entries come from LDS, this
    //  keeps their loads and whatever the compiler does to them in front of the chain)
#pragma unroll
    for (int j = 0; j < LU_NB; ++j) asm volatile("" : "+v"(er[j]));
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int j = 0; j < LU_NB; ++j) asm volatile("" : "+v"(er[j]));
    trsm16_steps<LOWER, UNIT, 0>(x, er, nb, rdiag);
}

// Net row permutation of one panel applied to `ncols` columns of M (row stride ld, rows relative to k0): one thread
// per column loads the (at most 32) moved rows of its column and stores them to their new places -- no barrier, no
// thread touches another thread's column.  Row bases are uniform (scalar), the column is the per-thread offset.
__device__ __forceinline__ void lu_swap_cols(gd* M, int ld, int ncols, int nm, int my_dst, int my_src, int t0 = threadIdx.x, int nt = DT) {
    // my_dst / my_src: the moved rows' destinations and sources, entry li in lane li of every wave (lu_swap_lists), handed out
    // by readlane
    for (int cc = t0; cc < ncols; cc += nt) {
        double tmp[2 * LU_NB];
#pragma unroll
        for (int li = 0; li < 2 * LU_NB; ++li)
            if (li < nm) tmp[li] = (M + __builtin_amdgcn_readlane(my_src, li) * ld)[cc];
#pragma unroll
        for (int li = 0; li < 2 * LU_NB; ++li)
            if (li < nm) (M + __builtin_amdgcn_readlane(my_dst, li) * ld)[cc] = tmp[li];
    }
}
__device__ __forceinline__ void lu_swap_lists(int nm, int& my_dst, int& my_src) {
    const int l = threadIdx.x & 63;
    my_dst = l < nm ? g_mlist[l] : 0;
    my_src = l < nm ? g_cur[my_dst] : 0;
}

// X = L11^{-1} M for the 16-row block M (row stride ld, `ncols` columns; unit lower triangle from the LDS panel), one
// thread per column; results to memory and to the LDS strip at column offset `soff` (zero in the padding columns up
// to `npad` and in rows >= nb)
__device__ __forceinline__ void lu_trsm_lower(gd* M, int ld, int ncols, int npad, int nb, double* us, int usp, int soff, int poff = 0,
                                              int t0 = threadIdx.x, int nt = DT) {
    const double* const panel = g_lds + poff;
    for (int c = t0; c < npad; c += nt) {
        asm volatile("" ::: "memory");
        const bool ok = c < ncols;
        double x[LU_NB];
#pragma unroll
        for (int j = 0; j < LU_NB; ++j) x[j] = (ok && j < nb) ? (M + j * ld)[c] : 0.0;
        trsm16<true, true>(x, panel, nb, nullptr);
#pragma unroll
        for (int j = 0; j < LU_NB; ++j) {
            if (ok && j < nb) (M + j * ld)[c] = x[j];
            us[j * usp + soff + c] = x[j];
        }
    }
}

// X1 = U11^{-1} B1 for one 16-row block of the back substitution (U11 in the LDS panel buffer at rows k0..), one
// thread per right-hand-side column; results to memory and to the LDS strip
__device__ __forceinline__ void lu_trsm_upper(gd* M, int ld, int k0, int nb, int nr, int nbp, double* us, int usp) {
    const double* const panel = g_lds;
    for (int c = threadIdx.x; c < nbp; c += DT) {
        asm volatile("" ::: "memory");
        double x[LU_NB];
#pragma unroll
        for (int j = 0; j < LU_NB; ++j) x[j] = (c < nr && j < nb) ? (M + j * ld)[c] : 0.0;
        trsm16<false, false>(x, panel + k0 * LU_LD, nb, g_rdiag);
#pragma unroll
        for (int j = 0; j < LU_NB; ++j) {
            if (c < nr && j < nb) (M + j * ld)[c] = x[j];
            us[j * usp + c] = x[j];
        }
    }
}

// Back substitution  X = U^{-1} Y  with the whole right-hand side resident in registers (n, nr <= 160: at most 10 x 10
// tiles of 16 x 16, wave w owns the tiles w, w + 8, ... in row-major order, 13 accumulators): Y is read once and X
// written once, where the block-row loop over global memory moved 1.8 MB per solve at p = 160.  Per block row k from the
// bottom: the block column of U is staged in LDS; the owners of row k put their tiles into per-column-tile LDS
// scratches; 16 threads per column tile solve with U11 (the substitution of lu_trsm_upper, all ten tiles at once); the
// owners take the solved tiles back and every wave updates its tiles above with rank-16 MFMAs (A fragments from the
// staged block column, B fragments from the scratches).  Same operations in the same order as the block-row loop.
constexpr int BS_T = 10, BS_Q = (BS_T * BS_T + NWAVE - 1) / NWAVE;

// X = E^{-1} B with the right-hand side resident in registers, E triangular (solve_dense_sqrt.hpp: the back substitution of
// the LU below is its upper / non-transposed case)
template <bool LOWER>
__device__ __noinline__ void wg_tri_solve_regs(const double* T_, int ldt_, int trans_, double* Bm_, int ldb_, int n_, int nr_,
                                               double* ws_end_ = nullptr, const double* dm_ = nullptr, double* mu_ = nullptr);
// the sizes at which wg_lu_solve ends in wg_tri_solve_regs (and can take the smoother's mean update along)
__device__ __forceinline__ bool lu_backsub_in_regs(int n, int nr) { return n > 64 && n <= 16 * BS_T && nr <= 16 * BS_T && DT == 512; }

}  // namespace rk
#include "solve_dense_lu_regs.hpp"
namespace rk {

// regs_: the forward elimination with [A | Bm] resident in registers (solve_dense_lu_regs.hpp) where the sizes allow it;
// 0 = always the panel loop over global memory below (RK_DENSE_LU=global: the bit-equality test's other leg)
__device__ __noinline__ void wg_lu_solve(double* A_, int lda_, double* Bm_, int ldb_, int n_, int nr_, int* piv_,
                                         double* ws_end_ = nullptr, int regs_ = 1, const double* dm_ = nullptr, double* mu_ = nullptr) {
    double* const lds = g_lds;
    auto* const A = uni_g(A_);
    auto* const Bm = uni_g(Bm_);
    auto* const piv = uni_g(piv_);
    auto* const ws_end = uni_g(ws_end_);
    (void)ws_end;
    const int lda = uni(lda_), ldb = uni(ldb_), n = uni(n_), nr = uni(nr_);
    RK_STAMP_DECL(ws_end);
    if (uni(regs_) != 0 && n > 64 && n <= RL_N && nr <= RL_N && DT == 512) {
        RK_STAMP_RESET();
        wg_lu_fwd_regs((double*)A, lda, (double*)Bm, ldb, n, nr, (double*)ws_end);
        RK_STAMP_RESET();
        wg_tri_solve_regs<false>((const double*)A, lda, 0, (double*)Bm, ldb, n, nr, nullptr, dm_, mu_);
        RK_STAMP(5);
        return;
    }
    const int nrp = (n + 15) & ~15, nbp = (nr + 15) & ~15;      // strip: [0, nrt) trailing columns, [nrt, nrt + nbp) right-hand sides
    const int usp = (nrp + nbp + 16) | 1;                       // odd row stride
    if (n > LU_MAXN || n * LU_LD + LU_NB * usp > LDS_DOUBLES) { wg_lu_solve_unblocked(lds, (double*)A, lda, (double*)Bm, ldb, n, nr, (int*)piv); return; }
    const int tid = threadIdx.x;
    // Look-ahead (when two panel buffers fit): after the swaps and the L11 solves of panel k, the rank-16 update is applied
    // FIRST to the 16 columns of the next panel by all waves; then wave 0 copies and factors panel k + 1 (wave-synchronous,
    // no workgroup barrier inside) while waves 1-7 apply the update to the other columns and to the right-hand sides.
    // (two panel buffers of n + 16 rows: the 16 rows in front of a panel hold the U12 block of ITS columns while wave 0 prepares it)
    const bool la = 2 * (n + LU_NB) * LU_LD + LU_NB * usp <= LDS_DOUBLES;
    const int pbs = (n + LU_NB) * LU_LD;                        // doubles per panel buffer (look-ahead form)
    const int us_off = la ? 2 * pbs : n * LU_LD;
    auto copy_panel = [&](int k0, int poff, int t0, int nt) {              // rows k0.., columns k0 .. k0 + nb - 1 -> LDS
        const int nb = min(LU_NB, n - k0), rows = n - k0;
        double* const pn = lds + poff;
        for (int e = t0; e < rows * LU_NB; e += nt) {
            const int r = e >> 4, c = e & 15;
            pn[r * LU_LD + c] = c < nb ? A[(k0 + r) * lda + k0 + c] : 0.0;
        }
        for (int r = t0; r < rows; r += nt) g_cur[r] = r;
        if (t0 == 0) g_mcount = 0;
    };
    copy_panel(0, la ? LU_NB * LU_LD : 0, tid, DT);
    __syncthreads();
    if (tid < 64) lu_panel_wave(n, min(LU_NB, n), la ? LU_NB * LU_LD : 0);
    __syncthreads();
    RK_STAMP(1);
    for (int k0 = 0; k0 < n; k0 += LU_NB) {
        const int nb = min(LU_NB, n - k0), rows = n - k0, n_right = n - k0 - nb;
        const int poff = la ? ((k0 / LU_NB) & 1) * pbs + LU_NB * LU_LD : 0;
        double* const panel = lds + poff;
        // the diagonal block back to A (its upper triangle is U11, needed by the back substitution); pivots
        if (tid < nb * LU_NB) {
            const int r = tid >> 4, c = tid & 15;
            if (c < nb) A[(k0 + r) * lda + k0 + c] = panel[r * LU_LD + c];
        }
        if (tid < nb) piv[k0 + tid] = k0 + g_ppiv[tid];
        for (int r = tid; r < rows; r += DT)                          // rows whose content changed
            if (g_cur[r] != r) g_mlist[atomicAdd(&g_mcount, 1)] = r;
        __syncthreads();
        const int nm = g_mcount;
        int my_dst, my_src;
        lu_swap_lists(nm, my_dst, my_src);                             // (in registers before the next panel resets g_cur)
        const int nrt = (n_right + 15) & ~15;                         // trailing columns padded to whole tiles
        const int rt = (n_right + 15) >> 4;
        gd* const A12 = A + (size_t)k0 * lda + k0 + nb;               // rows k0 .., the columns right of the panel
        gd* const B1 = Bm + (size_t)k0 * ldb;
        double* const A22 = (double*)(A + (size_t)(k0 + nb) * lda + k0 + nb);
        double* const B2 = (double*)(Bm + (size_t)(k0 + nb) * ldb);
        if (la && n_right > 0) {
            // Every wave owns a strip of columns and takes it through the whole step by itself -- row interchanges, L11 solve
            // (one lane per column, its U12 entries into the LDS strip), rank-16 update of the tiles below -- with no
            // workgroup barrier in between: wave 0 owns the next panel's 16 columns and goes straight on to copy and factor
            // that panel (the serial chain that bounds the solve: 16 column steps of ~2.4 k cycles), waves 1-7 share the other
            // columns and the right-hand sides.  (The first version ran interchanges, solves and the next panel's update as
            // workgroup phases in FRONT of the panel: 735 k cycles per 160 x 160 solve with 160 right-hand sides, of which
            // 250 k were those phases.)
            __syncthreads();                                           // every wave has its copy of the interchange lists
            const int wave = uni((int)(tid >> 6)), lane = tid & 63;
#ifdef RK_DENSE_STAMPS
            const long long tw0_ = __builtin_amdgcn_s_memtime();
#define RK_WSTAMP(k, t_from) do { if (stamps_ && blockIdx.x == 0 && lane == 0) stamps_[-2 - (k)] += (double)(__builtin_amdgcn_s_memtime() - (t_from)); } while (0)
#else
#define RK_WSTAMP(k, t_from)
#endif
            if (wave == 0) {
                // The next panel's block column never goes back to memory: its rows k0 .. n - 1 come into the other panel
                // buffer WITH the interchanges applied on the way (position r <- row g_cur[r]), the L11 solve of its top
                // 16 rows and the rank-16 update of the rows below run on LDS, and only the U12 block (back substitution) is
                // stored.  One memory round trip on the chain per panel where interchange, solve, update, store and
                // copy were five (42 k of the 71 k cycles per panel; a round trip to this data is ~5 k cycles).
                const int nxt = min(LU_NB, n_right), k1 = k0 + nb;
                double* const nb1 = lds + (((k1 / LU_NB) & 1)) * pbs;    // rows 0 .. 15: rows k0 .. of the block column; then the panel
                {
                    // (source rows first, then all loads of a pass, then the LDS writes: a rolled loop waited for every load in turn;
                    //  loads are clamped, not masked -- a select behind a load makes hipcc wait for that load at once)
                    constexpr int CH = 20;                             // 160 rows x 16 columns = 40 elements per lane: two passes (one pass of 40: slower)
                    const int total = rows * LU_NB;
                    for (int e0 = lane; e0 < total; e0 += 64 * CH) {
                        int srow[CH];
                        double val[CH];
#pragma unroll
                        for (int u = 0; u < CH; ++u) srow[u] = g_cur[min(e0 + 64 * u, total - 1) >> 4];
#pragma unroll
                        for (int u = 0; u < CH; ++u) val[u] = A12[(size_t)srow[u] * lda + min((e0 + 64 * u) & 15, nxt - 1)];
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int e = e0 + 64 * u;
                            if (e < total) nb1[(e >> 4) * LU_LD + (e & 15)] = (e & 15) < nxt ? val[u] : 0.0;
                        }
                    }
                }
                wave_lds_sync();
                if (lane < LU_NB) {
                    double x[LU_NB];
#pragma unroll
                    for (int j = 0; j < LU_NB; ++j) x[j] = j < nb ? nb1[j * LU_LD + lane] : 0.0;
                    trsm16<true, true>(x, panel, nb, nullptr);
#pragma unroll
                    for (int j = 0; j < LU_NB; ++j) {
                        if (lane < nxt && j < nb) A12[(size_t)j * lda + lane] = x[j];
                        lds[us_off + j * usp + lane] = x[j];
                    }
                }
                wave_lds_sync();
                {
                    const int lo = lane & 15, hi = lane >> 4;
                    const double* const lp = lds + poff + nb * LU_LD;
                    const double* const us = lds + us_off;
                    double* const pn = nb1 + LU_NB * LU_LD;                // the panel proper: rows k1 ..
                    for (int rb = 0; rb < rt; ++rb) {
                        const int li = min(rb * 16 + lo, n_right - 1);
                        d4 acc;
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[v] = pn[(rb * 16 + 4 * v + hi) * LU_LD + lo];
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq) {
                            const int k = 4 * kq + hi;
                            const double a_ = k < nb ? -lp[li * LU_LD + k] : 0.0;
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, us[k * usp + lo], acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (rb * 16 + 4 * v + hi < n_right) pn[(rb * 16 + 4 * v + hi) * LU_LD + lo] = acc[v];
                    }
                }
                for (int r = lane; r < n - k1; r += 64) g_cur[r] = r;
                if (lane == 0) g_mcount = 0;
                wave_lds_sync();
                RK_WSTAMP(2, tw0_);                                    // (stamps build: wave 0 up to the panel; then with the panel)
                lu_panel_wave(n - k1, min(LU_NB, n - k1), (((k1 / LU_NB) & 1)) * pbs + LU_NB * LU_LD);
                RK_WSTAMP(3, tw0_);
            } else {
                // column tiles in the order: trailing tiles 1 .. ta - 1, right-hand-side tiles 0 .. tb - 1
                // (round 4: leaving wave 4 idle -- it shares SIMD 0 with the panel wave -- did not shorten the panel chain, 369 k ->
                //  382 k cycles, and stretched the strips, 399 k -> 421 k: the chain is not held up by its SIMD partner's MFMAs)
                const int ta = nrt >> 4, tb = nbp >> 4, T = ta - 1 + tb, per = (T + NWAVE - 2) / (NWAVE - 1);
                const int g0 = (wave - 1) * per, g1 = min(T, g0 + per);
                if (g0 < g1) {
                    const int a0 = min(g0, ta - 1), a1 = min(g1, ta - 1);
                    const int b0 = max(g0 - (ta - 1), 0), b1 = max(g1 - (ta - 1), 0);
                    const int ca = 16 * (1 + a0), cb = 16 * b0;         // first column of the strip in A12 / in B1
                    const int nAc = a1 > a0 ? min(n_right - ca, 16 * (a1 - a0)) : 0;
                    const int nBc = b1 > b0 ? min(nr - cb, 16 * (b1 - b0)) : 0;
                    if (a1 > a0) {
                        if (nm > 0) lu_swap_cols(A12 + ca, lda, nAc, nm, my_dst, my_src, lane, 64);
                        lu_trsm_lower(A12 + ca, lda, nAc, 16 * (a1 - a0), nb, lds + us_off, usp, ca, poff, lane, 64);
                    }
                    if (b1 > b0) {
                        if (nm > 0) lu_swap_cols(B1 + cb, ldb, nBc, nm, my_dst, my_src, lane, 64);
                        lu_trsm_lower(B1 + cb, ldb, nBc, 16 * (b1 - b0), nb, lds + us_off, usp, nrt + cb, poff, lane, 64);
                    }
                    wave_lds_sync();
                    lu_rank_update(poff + nb * LU_LD, n_right, nb, us_off + ca, usp, rt, A22 + ca, lda, nAc, a1 - a0,
                                   B2 + cb, ldb, nBc, b1 - b0, nrt + cb - ca, n_right, wave, 1);
                }
                if (wave == 1) RK_WSTAMP(5, tw0_);                     // (stamps build: wave 1's whole strip)
            }
#undef RK_WSTAMP
        } else {
            if (nm > 0) {
                lu_swap_cols(A12, lda, n_right, nm, my_dst, my_src);
                lu_swap_cols(B1, ldb, nr, nm, my_dst, my_src);
            }
            __syncthreads();
            RK_STAMP(2);
            lu_trsm_lower(A12, lda, n_right, nrt, nb, lds + us_off, usp, 0, poff);
            lu_trsm_lower(B1, ldb, nr, nbp, nb, lds + us_off, usp, nrt, poff);
            __syncthreads();
            RK_STAMP(3);
            if (n_right > 0) {
                lu_rank_update(poff + nb * LU_LD, n_right, nb, us_off, usp, rt, A22, lda, n_right, nrt >> 4, B2, ldb, nr, nbp >> 4, nrt, n_right);
                __syncthreads();
                copy_panel(k0 + nb, 0, tid, DT);
                __syncthreads();
                if (tid < 64) lu_panel_wave(n_right, min(LU_NB, n_right), 0);
            }
        }
        __syncthreads();
        RK_STAMP(4);
    }
    if (lu_backsub_in_regs(n, nr)) {                                    // right-hand side resident in registers
        wg_tri_solve_regs<false>((const double*)A, lda, 0, (double*)Bm, ldb, n, nr, (double*)ws_end, dm_, mu_);
        RK_STAMP_RESET();
        return;
    }
    // back substitution with U, block rows from the bottom
    double* const panel = lds;
    for (int k0 = ((n - 1) / LU_NB) * LU_NB; k0 >= 0; k0 -= LU_NB) {
        const int nb = min(LU_NB, n - k0);
        // block column k0 of U, rows 0 .. k0 + nb - 1, into the panel buffer (U11 = its last nb rows)
        for (int e = tid; e < (k0 + nb) * LU_NB; e += DT) {
            const int r = e >> 4, c = e & 15;
            panel[r * LU_LD + c] = c < nb ? A[r * lda + k0 + c] : 0.0;
        }
        __syncthreads();
        if (tid < nb) g_rdiag[tid] = 1.0 / panel[(k0 + tid) * LU_LD + tid];
        __syncthreads();
        lu_trsm_upper(Bm + (size_t)k0 * ldb, ldb, k0, nb, nr, nbp, lds + us_off, usp);
        __syncthreads();
        RK_STAMP(5);
        if (k0 > 0) lu_rank_update(0, k0, nb, us_off, usp, k0 >> 4, (double*)Bm, ldb, 0, 0, (double*)Bm, ldb, nr, nbp >> 4, 0, k0);
        __syncthreads();
        RK_STAMP(6);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Draws x = mean + F z with F F^T = A, A symmetric positive SEMI-definite (interrogate_chkrebtii, interrogate.py:30-34;
// solve_sim, solve.py:182-194): the reference's multivariate_normal takes the Cholesky factor; a non-positive pivot zeroes
// its column here instead of producing NaN (the rule of the small-block kernels' psd_factor, linalg_small.hpp, and of
// the oracle).  The factor is built TRANSPOSED, U[j][i] = F[i][j], from the lower triangle of A, so that in the
// left-looking sweep the thread of row i reads U[k][i] -- consecutive threads, consecutive addresses -- and the pivot
// row entry U[k][j] is one address for the whole workgroup.  A is not modified.  z: p standard normals in g_lds.
__device__ __noinline__ void wg_psd_draw(const double* A_, double* U_, const double* mean_, double* x_, int n_, int zoff_) {
    auto* const A = uni_g(A_);
    auto* const U = uni_g(U_);
    auto* const mean = uni_g(mean_);
    auto* const x = uni_g(x_);
    const int n = uni(n_), zoff = uni(zoff_);
    double* const piv = g_rowbuf;                          // [0]: the column's pivot
    for (int e = threadIdx.x; e < n * n; e += DT) {        // U[j][i] = A[i][j] for i >= j (the lower triangle, transposed)
        const int j = e / n, i = e - j * n;
        if (i >= j) U[(size_t)j * n + i] = A[(size_t)i * n + j];
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        double sv[(LU_MAXN * 3 + DT - 1) / DT];            // this thread's rows i = j + threadIdx.x + r DT (n <= 768)
        int r = 0;
        for (int i = j + threadIdx.x; i < n; i += DT, ++r) {
            double acc = U[(size_t)j * n + i];
            for (int k = 0; k < j; ++k) acc = fma(-U[(size_t)k * n + i], U[(size_t)k * n + j], acc);
            sv[r] = acc;
            if (i == j) piv[0] = acc;
        }
        __syncthreads();
        const double d = piv[0];
        const bool ok = d > 0.0;
        const double dj = sqrt(ok ? d : 1.0);
        r = 0;
        for (int i = j + threadIdx.x; i < n; i += DT, ++r)
            U[(size_t)j * n + i] = ok ? (i == j ? dj : sv[r] / dj) : 0.0;
        __syncthreads();
    }
    const double* const z = g_lds + zoff;
    for (int i = threadIdx.x; i < n; i += DT) {
        double acc = 0.0;
        for (int j = 0; j <= i; ++j) acc = fma(U[(size_t)j * n + i], z[j], acc);
        x[i] = mean[i] + acc;
    }
    __syncthreads();
}

// p standard normals of (trajectory, step, block 0, purpose) into g_lds[zoff ..] -- the stream of the small-block kernels
// and of oracle/counter_rng.py: element 2c, 2c + 1 = the pair of counter c
__device__ __forceinline__ void wg_normals(unsigned long long seed, unsigned traj, unsigned step, unsigned purpose, int p, int zoff) {
    for (int c = threadIdx.x; c < (p + 1) / 2; c += DT) {
        double z0, z1;
        normal_pair(seed, traj, step, 0u, purpose, (unsigned)c, z0, z1);
        g_lds[zoff + 2 * c] = z0;
        if (2 * c + 1 < p) g_lds[zoff + 2 * c + 1] = z1;
    }
    __syncthreads();
}

struct DenseWs {
    double *A1, *A2, *A3, *A4, *Wt, *WS, *X, *S, *mup, *f, *yhat, *dm;
    int* piv;
};

__device__ __forceinline__ DenseWs carve(double* w, int p, int m) {
    DenseWs d;
    const size_t pp = (size_t)p * p, mp = (size_t)m * p;
    d.A1 = w; d.A2 = d.A1 + pp; d.A3 = d.A2 + pp; d.A4 = d.A3 + pp;
    d.Wt = d.A4 + pp; d.WS = d.Wt + mp; d.X = d.WS + mp; d.S = d.X + mp;
    d.mup = d.S + (size_t)m * m; d.f = d.mup + p; d.yhat = d.f + m; d.dm = d.yhat + m;
    d.piv = (int*)(d.dm + p);
    return d;
}

static size_t dense_ws_doubles(int p, int m) {
    return 4 * (size_t)p * p + 3 * (size_t)m * p + (size_t)m * m + 2 * (size_t)p + 2 * (size_t)m + (size_t)(p + 1) / 2 + 16;
}

// Products with a BLOCK-DIAGONAL Q (what prior.indep_init builds: n_vars blocks of nd x nd, nd = n_deriv): the terms
// a dense product would add are exact zeros times finite numbers, so leaving them out changes nothing in the result
// (same order of the remaining terms) and saves 6 p^3 of the 12.67 p^3 flops of a step.  Whether Q has that structure
// is checked on the device once per solve (dense_qcheck_kernel); otherwise the dense GEMMs run.  The diagonal blocks
// are kept in LDS (qd[i][k] = Q[i][k0(i) + k], row stride BD_MAX); a thread produces the nd outputs of one block
// row / block column from nd loads, lanes along the contiguous direction.
// C1 = Q X (rows of X), and, if X2T: C2 = Q X2T^T (X2T read transposed: the T^T = Q Sigma_f^T of standard.py:175)
__device__ __noinline__ void wg_bd_left(double* C1_, const double* X_, double* C2_, const double* X2T_, int p_, int nd_) {
    const double* const qd = g_qd;
    auto* const C1 = uni_g(C1_);
    auto* const X = uni_g(X_);
    auto* const C2 = uni_g(C2_);
    auto* const X2T = uni_g(X2T_);
    const int p = uni(p_), nd = uni(nd_);
    const int nblk = p / nd, total = nblk * p;
    constexpr int U = 2;                                   // items per pass: all their loads are issued before the arithmetic
    for (int e0 = threadIdx.x; e0 < total; e0 += U * DT) {
        double x[U][BD_MAX], xt[U][BD_MAX];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u * DT, total - 1), ib = e / p, j = e - ib * p, k0 = ib * nd;
#pragma unroll
            for (int k = 0; k < BD_MAX; ++k) {
                x[u][k] = k < nd ? X[(k0 + k) * p + j] : 0.0;
                xt[u][k] = (X2T && k < nd) ? X2T[j * p + k0 + k] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * DT;
            if (e < total) {
                const int ib = e / p, j = e - ib * p, k0 = ib * nd;
#pragma unroll
                for (int r = 0; r < BD_MAX; ++r)
                    if (r < nd) {
                        double s = 0.0, st = 0.0;
#pragma unroll
                        for (int k = 0; k < BD_MAX; ++k)
                            if (k < nd) { s = fma(qd[(k0 + r) * BD_MAX + k], x[u][k], s); st = fma(qd[(k0 + r) * BD_MAX + k], xt[u][k], st); }
                        C1[(k0 + r) * p + j] = s;
                        if (X2T) C2[(k0 + r) * p + j] = st;
                    }
            }
        }
    }
    __syncthreads();
}
// C = X Q^T + E ; and, if D: D = F - C   (the Sigma_next - Sigma^- of standard.py:215 in the same pass)
__device__ __noinline__ void wg_bd_right(double* C_, const double* X_, const double* E_, double* D_, const double* F_, int p_, int nd_) {
    const double* const qd = g_qd;
    auto* const C = uni_g(C_);
    auto* const X = uni_g(X_);
    auto* const E = uni_g(E_);
    auto* const D = uni_g(D_);
    auto* const F = uni_g(F_);
    const int p = uni(p_), nd = uni(nd_);
    const int nblk = p / nd, total = nblk * p;
    constexpr int U = 2;
    for (int e0 = threadIdx.x; e0 < total; e0 += U * DT) {
        double x[U][BD_MAX], ev[U][BD_MAX], fv[U][BD_MAX];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u * DT, total - 1), i = e / nblk, jb = e - i * nblk, k0 = jb * nd;
#pragma unroll
            for (int k = 0; k < BD_MAX; ++k) {
                x[u][k] = k < nd ? X[i * p + k0 + k] : 0.0;
                ev[u][k] = k < nd ? E[i * p + k0 + k] : 0.0;
                fv[u][k] = (D && k < nd) ? F[i * p + k0 + k] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * DT;
            if (e < total) {
                const int i = e / nblk, jb = e - i * nblk, k0 = jb * nd;
#pragma unroll
                for (int r = 0; r < BD_MAX; ++r)
                    if (r < nd) {
                        double s = 0.0;
#pragma unroll
                        for (int k = 0; k < BD_MAX; ++k)
                            if (k < nd) s = fma(x[u][k], qd[(k0 + r) * BD_MAX + k], s);
                        const double c = ev[u][r] + s;
                        C[i * p + k0 + r] = c;
                        if (D) D[i * p + k0 + r] = fv[u][r] - c;
                    }
            }
        }
    }
    __syncthreads();
}

// Round 4: the whole block-diagonal predict in ONE pass.  With Q = blockdiag(Q_I) the nd x nd block (I, J) of
//   Sigma^- = (Q Sigma) Q^T + R          is   (Q_I Sigma_IJ) Q_J^T + R_IJ                      (standard.py:58-59)
//   T^T = Q Sigma_f^T   (standard.py:175) is   Q_I (Sigma_f,JI)^T
// -- a function of ONE block of Sigma (and of its mirror image for T^T): a thread loads block (I, J) [and (J, I)], forms both
// products in registers with the terms and the order of wg_bd_left / wg_bd_right (k ascending into s = 0, then e + s), and
// stores Sigma^-, T^T and D = F - Sigma^- (the Sigma_next - Sigma^- of standard.py:215).  No intermediate Q Sigma in memory
// (the two-pass form wrote and re-read it: 0.4 MB per step) and no transposed read: the mirror block's rows are as
// contiguous as the block's own.  Lanes run along J: a wave's five loads of a block row cover 64 x 40 contiguous bytes.
// C2 = null: only Sigma^- (the forward pass); D = null: no difference.
template <int ND>
__device__ __forceinline__ void wg_bd_predict_nd(gd* C, gd* C2, gd* D, cgd* X, cgd* E, cgd* F, int p) {
    const double* const qd = g_qd;
    double* const stg = g_lds;
    const int nblk = p / ND;
    // Whole block rows per sweep, so that what a sweep produces is ONE contiguous range of rows of each output: the results go
    // to LDS at their place in that range (one region per output) and leave as flat, fully coalesced copies, 16 bytes per lane.
    // Stored straight from the block's thread -- 8-byte pieces 40 bytes apart, five instructions per cache line, every piece
    // a 32-byte write request -- the three outputs cost 80 k of the pass's 212 k cycles at (160, 5) (stamps of round 4, DESIGN.md
    // section 4).  Two kinds of thread per block when T^T is wanted: kind 0 takes block (I, J) to Sigma^- and D, kind 1 takes the
    // mirror block (J, I) to T^T (one thread doing both holds 100 doubles of results and spills).  The loads stay per block:
    // staging the bands of Sigma and F through LDS as well (flat copies in, the column band as [p][W]) was measured slower
    // (181 k against 166 k cycles for the pass) -- a CU has ~48 KB of loads in flight either way.
    const int nreg = 1 + (C2 ? 1 : 0) + (D ? 1 : 0), kinds = C2 ? 2 : 1;
    const int rb = max(1, min((DT / kinds) / nblk, LDS_DOUBLES / (nreg * ND * p)));          // block rows per sweep
    const int per = rb * nblk, reg = rb * ND * p;                          // blocks per sweep; doubles per LDS region
    const int regD = reg, regT = D ? 2 * reg : reg;
    const bool vst = (p & 1) == 0 && (reg & 1) == 0 && ((((size_t)C) | ((size_t)C2) | ((size_t)D)) & 15) == 0;
    typedef double d2v __attribute__((ext_vector_type(2)));
    auto flush = [&](gd* O, int roff, int row0, int nrows) {               // LDS region -> rows row0 .. of O
        const int n_el = nrows * p;
        gd* const o = O + (size_t)row0 * p;
        if (vst) {
            for (int e = 2 * threadIdx.x; e < n_el; e += 2 * DT) {
                d2v v; v.x = stg[roff + e]; v.y = stg[roff + e + 1];
                *reinterpret_cast<__attribute__((address_space(1))) d2v*>(o + e) = v;
            }
        } else {
            for (int e = threadIdx.x; e < n_el; e += DT) o[e] = stg[roff + e];
        }
    };
    __syncthreads();                                                        // (the LDS buffer may still be read by the phase in front)
    for (int ib0 = 0; ib0 < nblk; ib0 += rb) {
        const int nbr = min(rb, nblk - ib0), W = nbr * ND;                  // block rows, rows of this sweep
        const int t = threadIdx.x;
        const int kind = t >= per ? 1 : 0, idx = t - kind * per;
        const bool on = idx < nbr * nblk && kind < kinds;
        const int ibl = on ? idx / nblk : 0, jb = on ? idx - ibl * nblk : 0, i0 = (ib0 + ibl) * ND, j0 = jb * ND;
        const int lo_ = (ibl * ND) * p + j0;                               // the block's place in the sweep's row range
        double x[ND][ND], fv[ND][ND];                                       // kind 0: block (I, J) of Sigma, of F; kind 1: the mirror block
        if (on) {
#pragma unroll
            for (int r = 0; r < ND; ++r)
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    x[r][c] = kind == 0 ? X[(i0 + r) * p + j0 + c] : X[(j0 + r) * p + i0 + c];
                    if (D) fv[r][c] = kind == 0 ? F[(i0 + r) * p + j0 + c] : 0.0;
                }
        }
        if (on && kind == 0) {
            double y[ND][ND];
            // y = Q_I x (k ascending, like wg_bd_left)
#pragma unroll
            for (int r = 0; r < ND; ++r)
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    double s_ = 0.0;
#pragma unroll
                    for (int k = 0; k < ND; ++k) s_ = fma(qd[(i0 + r) * BD_MAX + k], x[k][c], s_);
                    y[r][c] = s_;
                }
#pragma unroll
            for (int r = 0; r < ND; ++r) {
                double ev[ND];
#pragma unroll
                for (int c = 0; c < ND; ++c) ev[c] = E[(i0 + r) * p + j0 + c];
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    double s_ = 0.0;
#pragma unroll
                    for (int k = 0; k < ND; ++k) s_ = fma(y[r][k], qd[(j0 + c) * BD_MAX + k], s_);     // (like wg_bd_right)
                    const double cv = ev[c] + s_;
                    stg[lo_ + r * p + c] = cv;
                    if (D) stg[regD + lo_ + r * p + c] = fv[r][c] - cv;
                }
            }
        } else if (on) {
            // T^T block (I, J) = Q_I (Sigma_f block (J, I))^T: x[r][c] = Sigma[j0 + r][i0 + c]
#pragma unroll
            for (int r = 0; r < ND; ++r)
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    double st = 0.0;
#pragma unroll
                    for (int k = 0; k < ND; ++k) st = fma(qd[(i0 + r) * BD_MAX + k], x[c][k], st);
                    stg[regT + lo_ + r * p + c] = st;
                }
        }
        __syncthreads();
        flush(C, 0, ib0 * ND, W);
        if (D) flush(D, regD, ib0 * ND, W);
        if (C2) flush(C2, regT, ib0 * ND, W);
        __syncthreads();
    }
}
// returns false if nd has no instance (the caller then runs the two passes)
__device__ __noinline__ bool wg_bd_predict(double* C_, double* C2_, double* D_, const double* X_, const double* E_, const double* F_,
                                           int p_, int nd_) {
    auto* const C = uni_g(C_);
    auto* const C2 = uni_g(C2_);
    auto* const D = uni_g(D_);
    auto* const X = uni_g(X_);
    auto* const E = uni_g(E_);
    auto* const F = uni_g(F_);
    const int p = uni(p_), nd = uni(nd_);
    switch (nd) {
        case 2: wg_bd_predict_nd<2>(C, C2, D, X, E, F, p); return true;
        case 3: wg_bd_predict_nd<3>(C, C2, D, X, E, F, p); return true;
        case 4: wg_bd_predict_nd<4>(C, C2, D, X, E, F, p); return true;
        case 5: wg_bd_predict_nd<5>(C, C2, D, X, E, F, p); return true;
        default: return false;
    }
}

// 1.0 if every entry of Q (p x p) outside the nd x nd diagonal blocks is exactly zero, else 0.0
__global__ void dense_qcheck_kernel(const double* Q, int p, int nd, double* flag) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    for (int e = threadIdx.x; e < p * p; e += blockDim.x) {
        const int i = e / p, j = e - i * p;
        if (i / nd != j / nd && Q[e] != 0.0) mine = 1;
    }
    if (mine) atomicOr(&bad, 1);
    __syncthreads();
    if (threadIdx.x == 0) *flag = (bad || nd > BD_MAX || p > BD_MAXP) ? 0.0 : 1.0;
}

__device__ __forceinline__ void load_qd(double* qd, const double* Q, int p, int nd) {
    for (int e = threadIdx.x; e < p * BD_MAX; e += DT) {
        const int i = e / BD_MAX, k = e - i * BD_MAX;
        qd[e] = k < nd ? Q[(size_t)i * p + (i / nd) * nd + k] : 0.0;
    }
    __syncthreads();
}

// y = Q x with block-diagonal Q
__device__ __forceinline__ void wg_bd_matvec(const double* qd, double* y, const double* x, int p, int nd) {
    for (int i = threadIdx.x; i < p; i += DT) {
        const int k0 = (i / nd) * nd;
        double s = 0.0;
        for (int k = 0; k < nd; ++k) s = fma(qd[i * BD_MAX + k], x[k0 + k], s);
        y[i] = s;
    }
}

__device__ __forceinline__ GemmOp gemm_op(double* C, int ldc, const double* A, int lda, bool ta, const double* B, int ldb, bool tb,
                                          int M, int N, int K, const double* E, int lde, double ce, double cab) {
    GemmOp g;
    g.C = C; g.ldc = ldc; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
    g.E = E; g.lde = lde; g.ce = ce; g.cab = cab; g.ta = ta; g.tb = tb;
    return g;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward pass.  The step's dense products are issued from ONE wg_gemm instance in a descriptor loop.
// ---------------------------------------------------------------------------------------------------------------
// MODE 0: the whole forward pass with the built-in linear right-hand side; MODE 1, 2: the pieces around the interrogation
// kernel of a hiprtc right-hand side (dense_solve): 1 = time 0 and step 0 up to mu-, 2 = the update of step n0 and step
// n0 + 1 up to mu-.  (A template parameter: with run-time phase bounds the fused pass lost 9 %.)
// EXTRA: interrogate_chkrebtii's draw and RK_FLAG_STORE_PRED compiled in (kept out of the instance that config 5 times:
// their mere presence cost it 9 % through register allocation around the calls).
template <int MODE, bool EXTRA>
__global__ void __launch_bounds__(DT) dense_fwd_kernel(DenseArgs a) {
    double* const qd = g_qd;
    const int b = blockIdx.x, p = a.p, m = a.m;
    const int nd = p / m;                              // derivatives per variable: x_v = X[v * nd]
    const DenseWs w = carve(a.ws + (size_t)b * a.ws_stride, p, m);
    const bool q_bd = a.ws[a.ws_stride - 1] != 0.0;     // set by dense_qcheck_kernel: Q is block diagonal (indep_init)
    double* mean = a.mean + (size_t)b * (a.N + 1) * p;
    double* var = a.var + (size_t)b * (a.N + 1) * p * p;
    const double* Aode = a.theta;
    if (q_bd) load_qd(qd, a.Q, p, nd);
    RK_STAMP_DECL(a.ws + a.ws_stride);
    if (MODE != 2) {
        RK_STAMP_ZERO();
        // time 0: (ode_init, 0)   (solve.py:53-54, 114-121)
        for (int i = threadIdx.x; i < p; i += DT) mean[i] = a.x0_b ? a.x0[(size_t)i * a.B + b] : a.x0[i];
        for (int e = threadIdx.x; e < p * p; e += DT) var[e] = 0.0;
        if (EXTRA && a.mean_pred) {                      // index 0 of the predictions = (ode_init, 0) too (solve.py:114-121)
            double* mp0 = a.mean_pred + (size_t)b * (a.N + 1) * p;
            double* vp0 = a.var_pred + (size_t)b * (a.N + 1) * p * p;
            for (int i = threadIdx.x; i < p; i += DT) mp0[i] = a.x0_b ? a.x0[(size_t)i * a.B + b] : a.x0[i];
            for (int e = threadIdx.x; e < p * p; e += DT) vp0[e] = 0.0;
        }
    }
    __syncthreads();
    // The step's dense products are issued from ONE wg_gemm call in a descriptor loop over its phases:
    // 0-1 predict (standard.py:57-59), 2 mu-, 3 interrogation and W~ Sigma-, 4 S, 5 (Sigma- W~^T)^T (standard.py:93-97),
    // 6 the LU solve, the mean and Sigma- - K (W~ Sigma-) (standard.py:98-102).
    auto phases = [&](int n, auto ph_lo, auto ph_hi) {
        const double* mu = mean + (size_t)n * p;
        const double* Sig = var + (size_t)n * p * p;
        double* mu_o = mean + (size_t)(n + 1) * p;
        double* Sig_o = var + (size_t)(n + 1) * p * p;
        bool fused = false;
#pragma unroll 1
        for (int ph = decltype(ph_lo)::value; ph < decltype(ph_hi)::value; ++ph) {       // (ONE wg_gemm call site: not unrolled)
            GemmOp g;
            bool run = true;
            switch (ph) {
                case 0:
                    if (q_bd) {
                        fused = wg_bd_predict(w.A2, nullptr, nullptr, Sig, a.R, nullptr, p, nd);      // (Q Sigma) Q^T + R in one pass
                        if (!fused) wg_bd_left(w.A1, Sig, nullptr, nullptr, p, nd);
                        run = false;
                    }
                    else g = gemm_op(w.A1, p, a.Q, p, false, Sig, p, false, p, p, p, nullptr, 0, 0.0, 1.0);
                    break;
                case 1:
                    if (q_bd) { if (!fused) wg_bd_right(w.A2, w.A1, a.R, nullptr, nullptr, p, nd); run = false; }
                    else g = gemm_op(w.A2, p, w.A1, p, false, a.Q, p, true, p, p, p, a.R, p, 1.0, 1.0);
                    break;
                case 2:
                    if (q_bd) wg_bd_matvec(qd, w.mup, mu, p, nd);
                    else wg_gemv<false>(w.mup, a.Q, p, mu, p, p, nullptr, 0.0, 1.0);
                    if (EXTRA && a.mean_pred) {          // state_pred of the reference's _solve_filter (solve.py:99-104)
                        __syncthreads();
                        double* mpo = a.mean_pred + ((size_t)b * (a.N + 1) + n + 1) * p;
                        double* vpo = a.var_pred + ((size_t)b * (a.N + 1) + n + 1) * p * p;
                        for (int i = threadIdx.x; i < p; i += DT) mpo[i] = w.mup[i];
                        for (int e = threadIdx.x; e < p * p; e += DT) vpo[e] = w.A2[e];
                    }
                    if (EXTRA && a.itg == RK_INTERROGATE_CHKREBTII) {
                        // interrogate.py:22-34: the point the ODE is evaluated at, x ~ N(mu-, Sigma-), into w.dm
                        __syncthreads();
                        wg_normals(a.seed, (unsigned)(a.traj_offset + (unsigned long long)b), (unsigned)n, PURPOSE_INTERROGATE, p, 0);
                        wg_psd_draw(w.A2, w.A3, w.mup, w.dm, p, 0);
                    }
                    run = false;
                    break;
                case 3: {
                    // ---- interrogation (interrogate.py): W~ = W - J in w.Wt, the offset a = -f (+ J mu- for kramer,
                    // interrogate.py:81-82) in w.f.  Built in: the linear ODE f = A x, x_v = X[v * nd]; a right-hand
                    // side that arrives through hiprtc has filled both from its own kernel between two launches of this
                    // one (MODE != 0, solve_dense_itg_kernels.hpp) ----
                    // (MODE 0: W~ of the built-in linear ODE does not change with the step -- built once in front of the time loop)
                    if (MODE == 0) __syncthreads();                                       // mu- of phase 2 is complete
                    // yhat = W~ mu- + a (standard.py:93); for the linear ODE f_i = sum_v A_iv x_v and J mu- = (W - W~) mu-:
                    // one wave per measurement row, lanes along the sums, wave reduction
                    for (int i = threadIdx.x >> 6; i < m; i += DT / 64) {
                        const int lane = threadIdx.x & 63;
                        double s = 0.0, jm = 0.0, wm = 0.0;
                        if (MODE == 0) {
                            const double* const xe = EXTRA && a.itg == RK_INTERROGATE_CHKREBTII ? w.dm : w.mup;     // where f is evaluated
                            for (int v = lane; v < m; v += 64) {
                                const double Aiv = a.theta_b ? Aode[((size_t)i * m + v) * a.B + b] : Aode[(size_t)i * m + v];
                                s = fma(Aiv, xe[(size_t)v * nd], s);
                            }
                        }
                        for (int j = lane; j < p; j += 64) {
                            const double wt = w.Wt[(size_t)i * p + j], mj = w.mup[j];
                            if (MODE == 0) jm = fma(a.W[(size_t)i * p + j] - wt, mj, jm);
                            wm = fma(wt, mj, wm);
                        }
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) {
                            s += __shfl_xor(s, off); jm += __shfl_xor(jm, off); wm += __shfl_xor(wm, off);
                        }
                        const double am = MODE != 0 ? w.f[i] : (a.itg == RK_INTERROGATE_KRAMER ? -s + jm : -s);
                        if (lane == 0) w.yhat[i] = wm + am;
                    }
                    g = gemm_op(w.WS, p, w.Wt, p, false, w.A2, p, false, m, p, p, nullptr, 0, 0.0, 1.0);          // W~ Sigma-
                    break;
                }
                case 4: g = gemm_op(w.S, m, w.WS, p, false, w.Wt, p, true, m, m, p, nullptr, 0, 0.0, 1.0); break;  // (W~ Sigma-) W~^T
                case 5: g = gemm_op(w.X, p, w.Wt, p, false, w.A2, p, true, m, p, p, nullptr, 0, 0.0, 1.0); break;  // (Sigma- W~^T)^T
                default:
                    if (a.itg == RK_INTERROGATE_RODEO || (EXTRA && a.itg == RK_INTERROGATE_CHKREBTII)) {   // + var_meas = W Sigma- W^T (W~ = W; interrogate.py:110-113, 26-29)
                        for (int e = threadIdx.x; e < m * m; e += DT) w.S[e] = w.S[e] + w.S[e];
                        __syncthreads();
                    }
                    wg_lu_solve(w.S, m, w.X, p, m, p, w.piv);                                                  // X = K^T (utils.py:119)
                    for (int i = threadIdx.x; i < p; i += DT) {
                        double s = 0.0;
                        for (int j = 0; j < m; ++j) s = fma(w.X[(size_t)j * p + i], 0.0 - w.yhat[j], s);
                        mu_o[i] = w.mup[i] + s;                                      // standard.py:99-100 with x_meas = 0
                    }
                    g = gemm_op(Sig_o, p, w.X, p, true, w.WS, p, false, p, p, m, w.A2, p, 1.0, -1.0);              // Sigma- - K (W~ Sigma-)
                    break;
            }
            if (run) wg_gemm(g);
            RK_STAMP(ph);
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P3 = std::integral_constant<int, 3>;
    using P7 = std::integral_constant<int, 7>;
    if (MODE == 0) {
        // W~ = W - J (solve.py:79) for the built-in linear ODE f = A x: J = A (kramer) or 0, the same at every step and every state
        for (int e = threadIdx.x; e < m * p; e += DT) {
            const int i = e / p, j = e % p;
            double Jij = 0.0;
            if (a.itg == RK_INTERROGATE_KRAMER && j % nd == 0) {
                const int v = j / nd;
                Jij = a.theta_b ? Aode[((size_t)i * m + v) * a.B + b] : Aode[(size_t)i * m + v];
            }
            w.Wt[e] = a.W[e] + (-Jij);                                               // W + wgt_meas   (solve.py:79)
        }
        __syncthreads();
        for (int n = 0; n < a.N; ++n) phases(n, P0{}, P7{});
    } else if (MODE == 1) {
        phases(0, P0{}, P3{});
    } else {
        phases(a.n0, P3{}, P7{});
        if (a.n0 + 1 < a.N) phases(a.n0 + 1, P0{}, P3{});
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward pass (solve.py:257-301, standard.py:160-217), in place on the filtered moments.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(DT) dense_bwd_mv_kernel(DenseArgs a) {
    double* const qd = g_qd;
    double* const lds = g_lds;
    const int b = blockIdx.x, p = a.p, m = a.m;
    const int nd = p / m;
    const DenseWs w = carve(a.ws + (size_t)b * a.ws_stride, p, m);
    const bool q_bd = a.ws[a.ws_stride - 1] != 0.0;     // set by dense_qcheck_kernel: Q is block diagonal (indep_init)
    double* mean = a.mean + (size_t)b * (a.N + 1) * p;
    double* var = a.var + (size_t)b * (a.N + 1) * p * p;
    if (q_bd) load_qd(qd, a.Q, p, nd);
    RK_STAMP_DECL(a.ws + a.ws_stride);
    RK_STAMP_ZERO();
    for (int n = a.N - 1; n >= 1; --n) {
        double* mu_f = mean + (size_t)n * p;
        double* Sig_f = var + (size_t)n * p * p;
        const double* mu_s = mean + (size_t)(n + 1) * p;           // already smoothed (in place)
        const double* Sig_s = var + (size_t)(n + 1) * p * p;
        // phases 0-2: pred[n+1] re-evaluated from filt[n], T^T = Q Sigma_f^T (standard.py:175), the differences;
        // 3: G^T = solve(Sigma-, T^T) (standard.py:176), mean (213-214), G D; 4: Sigma_f + (G D) G^T (215-216)
        bool fused = false;
        for (int ph = 0; ph < 5; ++ph) {
            GemmOp g;
            bool run = true;
            switch (ph) {
                case 0:
                    if (q_bd) {
                        fused = wg_bd_predict(w.A2, w.A3, w.A4, Sig_f, a.R, Sig_s, p, nd);       // Sigma^-, T^T, Sigma_next - Sigma^- in one pass
                        if (!fused) wg_bd_left(w.A1, Sig_f, w.A3, Sig_f, p, nd);
                        run = false;
                        RK_STAMP(6);
                    }
                    else g = gemm_op(w.A1, p, a.Q, p, false, Sig_f, p, false, p, p, p, nullptr, 0, 0.0, 1.0);
                    break;
                case 1:
                    if (q_bd) { if (!fused) wg_bd_right(w.A2, w.A1, a.R, w.A4, Sig_s, p, nd); run = false; }
                    else g = gemm_op(w.A2, p, w.A1, p, false, a.Q, p, true, p, p, p, a.R, p, 1.0, 1.0);
                    break;
                case 2:
                    if (q_bd) { run = false; break; }
                    for (int e = threadIdx.x; e < p * p; e += DT) w.A4[e] = Sig_s[e] - w.A2[e];        // Sigma_next - Sigma-
                    g = gemm_op(w.A3, p, a.Q, p, false, Sig_f, p, true, p, p, p, nullptr, 0, 0.0, 1.0);
                    break;
                case 3: {
                    if (q_bd) wg_bd_matvec(qd, w.mup, mu_f, p, nd);
                    else wg_gemv<false>(w.mup, a.Q, p, mu_f, p, p, nullptr, 0.0, 1.0);
                    __syncthreads();
                    for (int i = threadIdx.x; i < p; i += DT) w.dm[i] = mu_s[i] - w.mup[i];
                    __syncthreads();
                    RK_STAMP(0);
                    const bool mean_fused = lu_backsub_in_regs(p, p);       // (the back substitution takes mu_f + G dm along)
                    wg_lu_solve(w.A2, p, w.A3, p, p, p, w.piv, a.ws + a.ws_stride, a.lu_regs, mean_fused ? w.dm : nullptr,
                                mean_fused ? mu_f : nullptr);                    // A3 <- G^T
                    RK_STAMP_RESET();
                    // mean: mu_f + G dm, the column sums of G^T split over the workgroup (partial sums through LDS)
                    if (!mean_fused) {
                        const int ng = DT / 64;                                     // row groups of G^T
                        for (int i = threadIdx.x & 63; i < p; i += 64) {
                            const int gq = threadIdx.x >> 6;
                            double s = 0.0;
                            #pragma unroll 8
                            for (int j = gq; j < p; j += ng) s = fma(w.A3[(size_t)j * p + i], w.dm[j], s);
                            lds[gq * p + i] = s;
                        }
                        __syncthreads();
                        for (int i = threadIdx.x; i < p; i += DT) {
                            double s = 0.0;
                            for (int gq = 0; gq < ng; ++gq) s += lds[gq * p + i];
                            mu_f[i] = mu_f[i] + s;                                   // standard.py:213-214
                        }
                        __syncthreads();
                    }
                    RK_STAMP(7);
                    g = gemm_op(w.A1, p, w.A3, p, true, w.A4, p, false, p, p, p, nullptr, 0, 0.0, 1.0);    // G D
                    break;
                }
                default:
                    RK_STAMP(8);
                    g = gemm_op(Sig_f, p, w.A1, p, false, w.A3, p, false, p, p, p, Sig_f, p, 1.0, 1.0);    // Sigma_f + (G D) G^T
                    break;
            }
            if (run) wg_gemm(g);
        }
        RK_STAMP(9);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward sampler of solve_sim (solve.py:137-205, standard.py:248-254) on the filtered moments, which stay untouched:
//   x_N ~ N(mu_N, Sigma_N);   x_n ~ N(mu_f + G (x_{n+1} - mu-),  Sigma_f - G T^T),   T = Sigma_f Q^T, G = T (Sigma-)^{-1},
// with the building blocks of the smoothing pass above and the factor / draw of wg_psd_draw.  x[0] = ode_init.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(DT) dense_bwd_sim_kernel(DenseArgs a) {
    double* const qd = g_qd;
    double* const lds = g_lds;
    const int b = blockIdx.x, p = a.p, m = a.m;
    const int nd = p / m;
    const DenseWs w = carve(a.ws + (size_t)b * a.ws_stride, p, m);
    const bool q_bd = a.ws[a.ws_stride - 1] != 0.0;
    const double* mean = a.mean + (size_t)b * (a.N + 1) * p;
    const double* var = a.var + (size_t)b * (a.N + 1) * p * p;
    const unsigned traj = (unsigned)(a.traj_offset + (unsigned long long)b);
    if (q_bd) load_qd(qd, a.Q, p, nd);
    auto put_x = [&](int n, const double* xv) {
        for (int i = threadIdx.x; i < p; i += DT) a.x[((size_t)n * p + i) * a.B + b] = xv[i];
    };
    // terminal draw (solve.py:182-186)
    wg_normals(a.seed, traj, (unsigned)a.N, PURPOSE_SMOOTH, p, 0);
    wg_psd_draw(var + (size_t)a.N * p * p, w.A3, mean + (size_t)a.N * p, w.dm, p, 0);
    put_x(a.N, w.dm);                                           // w.dm = x_{n+1} from here on
    __syncthreads();
    for (int n = a.N - 1; n >= 1; --n) {
        const double* mu_f = mean + (size_t)n * p;
        const double* Sig_f = var + (size_t)n * p * p;
        // pred[n+1] from filt[n]; T^T = Q Sigma_f^T (standard.py:175)
        if (q_bd) {
            if (!wg_bd_predict(w.A2, w.A3, nullptr, Sig_f, a.R, nullptr, p, nd)) {
                wg_bd_left(w.A1, Sig_f, w.A3, Sig_f, p, nd);
                wg_bd_right(w.A2, w.A1, a.R, nullptr, nullptr, p, nd);
            }
        } else {
            wg_gemm(gemm_op(w.A1, p, a.Q, p, false, Sig_f, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
            wg_gemm(gemm_op(w.A2, p, w.A1, p, false, a.Q, p, true, p, p, p, a.R, p, 1.0, 1.0));
            wg_gemm(gemm_op(w.A3, p, a.Q, p, false, Sig_f, p, true, p, p, p, nullptr, 0, 0.0, 1.0));
        }
        if (q_bd) wg_bd_matvec(qd, w.mup, mu_f, p, nd);
        else wg_gemv<false>(w.mup, a.Q, p, mu_f, p, p, nullptr, 0.0, 1.0);
        __syncthreads();
        for (int e = threadIdx.x; e < p * p; e += DT) w.A4[e] = w.A3[e];          // keep T^T: the solve overwrites A3
        for (int i = threadIdx.x; i < p; i += DT) w.dm[i] = w.dm[i] - w.mup[i];   // x_{n+1} - mu-
        __syncthreads();
        wg_lu_solve(w.A2, p, w.A3, p, p, p, w.piv, a.ws + a.ws_stride, a.lu_regs);           // A3 <- G^T (standard.py:176)
        // mean_sim = mu_f + G (x_{n+1} - mu-) (standard.py:250-251): column sums of G^T through LDS, as in the smoother
        {
            const int ng = DT / 64;
            for (int i = threadIdx.x & 63; i < p; i += 64) {
                const int gq = threadIdx.x >> 6;
                double sacc = 0.0;
                #pragma unroll 8
                for (int j = gq; j < p; j += ng) sacc = fma(w.A3[(size_t)j * p + i], w.dm[j], sacc);
                lds[gq * p + i] = sacc;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < p; i += DT) {
                double sacc = 0.0;
                for (int gq = 0; gq < ng; ++gq) sacc += lds[gq * p + i];
                w.mup[i] = mu_f[i] + sacc;                                         // mean_sim (mu- is no longer needed)
            }
            __syncthreads();
        }
        wg_gemm(gemm_op(w.A1, p, w.A3, p, true, w.A4, p, false, p, p, p, Sig_f, p, 1.0, -1.0));   // Sigma_f - G T^T (standard.py:252-253)
        wg_normals(a.seed, traj, (unsigned)n, PURPOSE_SMOOTH, p, 0);
        wg_psd_draw(w.A1, w.A2, w.mup, w.dm, p, 0);                                // x_n -> w.dm
        put_x(n, w.dm);
        __syncthreads();
    }
    put_x(0, mean);                                             // x[0] = ode_init exactly (solve.py:196-204)
    (void)m;
}

}  // namespace rk

#include "solve_dense_sqrt.hpp"
#include "solve_dense_ops.hpp"

namespace rk {

bool is_user_rhs(int rhs_id);
bool user_dense_wanted(const rk_solve_cfg* c);
int user_dense_interrogate(rk_handle h, const rk_solve_cfg* c, const DenseItgArgs& a);

// The built-in linear ODE of config 5, and any hiprtc right-hand side in the non-block form (one block, several
// measurements) beyond what the lane-per-trajectory kernels take (rhs_jit.hip: n_bstate > 9 or n_bmeas > 4).
bool dense_supported(const rk_solve_cfg* c, int mode) {
    return c->rhs_id == RK_RHS_LINEAR_DENSE || (is_user_rhs(c->rhs_id) && user_dense_wanted(c));
}

int dense_check(const rk_solve_cfg* c, const rk_solve_in* in, int mode) {
    RK_REQUIRE(c->n_block == 1, RK_ERR_UNSUPPORTED, "dense path: n_block must be 1 (use prior.indep_init), got %d", c->n_block);
    RK_REQUIRE(c->n_bmeas >= 1 && c->n_bstate % c->n_bmeas == 0 && c->n_bstate / c->n_bmeas >= 2, RK_ERR_INVALID,
               "dense linear ODE: n_bstate (%d) must be n_vars * n_deriv with n_vars = n_bmeas (%d), n_deriv >= 2",
               c->n_bstate, c->n_bmeas);
    RK_REQUIRE(c->n_bstate <= 768, RK_ERR_UNSUPPORTED, "dense path: n_bstate = %d exceeds 768 (LDS staging of the GEMMs)", c->n_bstate);
    RK_REQUIRE(c->kalman_type == RK_KALMAN_STANDARD || c->kalman_type == RK_KALMAN_SQRT, RK_ERR_UNSUPPORTED,
               "dense path: unknown kalman_type %d", c->kalman_type);
    // interrogate.py:36-42: in square-root mode the reference draws mu- + (W L-) z, an (n_bmeas,) vector added to an
    // (n_bstate,) one -- it only broadcasts for n_bmeas = 1 or n_bstate, neither of which is a dense-path shape
    RK_REQUIRE(!(c->kalman_type == RK_KALMAN_SQRT && c->interrogate == RK_INTERROGATE_CHKREBTII), RK_ERR_UNSUPPORTED,
               "dense path: interrogate_chkrebtii with kalman_type=square-root is undefined for 1 < n_bmeas < n_bstate "
               "(the reference's draw mu- + (W L-) z does not broadcast, src/rodeo/interrogate.py:36-42)");
    RK_REQUIRE(!(c->flags & RK_FLAG_BATCH_MINOR), RK_ERR_UNSUPPORTED, "dense path: its layout is trajectory-major (RK_FLAG_BATCH_MINOR)");
    RK_REQUIRE(!in->ode_weight_batched && !in->prior_weight_batched && !in->prior_var_batched, RK_ERR_UNSUPPORTED,
               "dense path: ode_weight and prior_pars must be shared by all trajectories");
    if (c->rhs_id == RK_RHS_LINEAR_DENSE)
        RK_REQUIRE(in->theta && c->n_theta == c->n_bmeas * c->n_bmeas, RK_ERR_INVALID,
                   "dense linear ODE needs theta = A (n_vars x n_vars row-major), n_theta = %d", c->n_bmeas * c->n_bmeas);
    else
        RK_REQUIRE(c->n_theta == 0 || in->theta, RK_ERR_INVALID, "dense path: theta is NULL but n_theta = %d", c->n_theta);
    return RK_OK;
}

// Bytes of out->workspace a dense solve needs.  Square-root form: per-trajectory scratch, then -- for the backward passes,
// unless the caller keeps the predictions anyway (RK_FLAG_STORE_PRED) -- the predicted factors of every step
// (B, N+1, p, p): square_root.py:170-175 solves with L^-_{n+1}, and recomputing it would repeat the forward pass's
// most expensive operation (a 2p x p QR) where storing it costs p^2 doubles per step of 288 GB.
static bool dense_sqrt_keeps_pred_in_ws(const rk_solve_cfg* c, int mode) {
    return c->kalman_type == RK_KALMAN_SQRT && mode != RK_MODE_FILTER && !(c->flags & RK_FLAG_STORE_PRED);
}
size_t dense_ws_bytes(const rk_solve_cfg* c, int mode) {
    const size_t p = c->n_bstate, B = c->n_traj;
    if (c->kalman_type == RK_KALMAN_SQRT)
        return (dense_sq_ws_doubles(c->n_bstate, c->n_bmeas) * B +
                (dense_sqrt_keeps_pred_in_ws(c, mode) ? B * (size_t)(c->n_steps + 1) * p * p : 0)) * sizeof(double);
    return dense_ws_doubles(c->n_bstate, c->n_bmeas) * B * sizeof(double);
}

int dense_solve(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out, int mode) {
    RK_REQUIRE(out->workspace, RK_ERR_INVALID, "dense path needs out->workspace (rk_solve_workspace_bytes)");
    {
        // every kernel below indexes the workspace by trajectory with this stride: check the caller's size on the host
        const size_t need = dense_ws_bytes(c, mode);
        RK_REQUIRE(out->workspace_bytes >= need, RK_ERR_INVALID,
                   "dense path: out->workspace_bytes = %zu, this configuration needs %zu (rk_solve_workspace_bytes)",
                   out->workspace_bytes, need);
        RK_REQUIRE(out->mean_state && out->var_state, RK_ERR_INVALID, "dense path: null output pointer");
    }
    DenseArgs a;
    a.B = c->n_traj; a.N = c->n_steps; a.p = c->n_bstate; a.m = c->n_bmeas; a.itg = c->interrogate;
    a.t_min = c->t_min; a.t_max = c->t_max;
    a.W = in->ode_weight; a.x0 = in->ode_init; a.Q = in->prior_weight; a.R = in->prior_var; a.theta = in->theta;
    a.x0_b = in->ode_init_batched; a.theta_b = in->theta_batched;
    a.mean = out->mean_state; a.var = out->var_state;
    a.ws = (double*)out->workspace; a.ws_stride = dense_ws_doubles(a.p, a.m);
    a.n0 = 0;
    a.seed = c->seed; a.traj_offset = c->traj_offset; a.x = out->x_state;
    a.mean_pred = (c->flags & RK_FLAG_STORE_PRED) ? out->mean_pred : nullptr;         // trajectory-major like mean / var
    a.var_pred = (c->flags & RK_FLAG_STORE_PRED) ? out->var_pred : nullptr;
    RK_REQUIRE(!(c->flags & RK_FLAG_STORE_PRED) || (out->mean_pred && out->var_pred), RK_ERR_INVALID,
               "RK_FLAG_STORE_PRED needs out->mean_pred / var_pred");
    RK_REQUIRE(mode != RK_MODE_SIM || out->x_state, RK_ERR_INVALID, "rk_solve_sim needs out->x_state");
    a.fac_pred = nullptr;
    {
        const char* const lm = getenv("RK_DENSE_LU");          // "regs": the register-resident forward elimination (solve_dense_lu_regs.hpp)
        a.lu_regs = lm && lm[0] == 'r';
    }
    if (c->kalman_type == RK_KALMAN_SQRT) {
        // ---- square-root form (solve_dense_sqrt.hpp) ----
        a.ws_stride = dense_sq_ws_doubles(a.p, a.m);
        if (c->flags & RK_FLAG_STORE_PRED) a.fac_pred = out->var_pred;
        else if (dense_sqrt_keeps_pred_in_ws(c, mode)) a.fac_pred = a.ws + a.ws_stride * (size_t)a.B;
        {   // structure of the predict stack (RK_DENSE_STRUCTURED=0: always the generic QR -- the parity test's other leg)
            const char* const st = getenv("RK_DENSE_STRUCTURED");
            if (st && st[0] == '0') RK_HIP(hipMemsetAsync(a.ws + a.ws_stride - 1, 0, sizeof(double), h->stream));
            else hipLaunchKernelGGL(dense_sqcheck_kernel, dim3(1), dim3(256), 0, h->stream, a.Q, a.R, a.p, a.p / a.m, a.ws + a.ws_stride - 1);
        }
        if (c->rhs_id == RK_RHS_LINEAR_DENSE) {
            LaunchTimer t(h, "dense_sqrt_fwd_kernel");
            hipLaunchKernelGGL((dense_sqrt_fwd_kernel<0>), dim3(a.B), dim3(DT), 0, h->stream, a);
            t.stop();
        } else {
            DenseItgArgs g;
            g.B = a.B; g.N = a.N; g.t_min = a.t_min; g.t_max = a.t_max; g.W = a.W; g.theta = a.theta; g.theta_b = a.theta_b;
            g.ws = a.ws; g.ws_stride = a.ws_stride;
            g.off_Wt = dense_sq_off_Wt(a.p); g.off_mup = dense_sq_off_mup(a.p, a.m); g.off_am = g.off_mup + a.p; g.off_x = g.off_mup;
            LaunchTimer t(h, "dense_sqrt_fwd_kernel<user, stepwise>");
            hipLaunchKernelGGL((dense_sqrt_fwd_kernel<1>), dim3(a.B), dim3(DT), 0, h->stream, a);
            for (int n = 0; n < a.N; ++n) {
                g.n = n;
                const int rc = user_dense_interrogate(h, c, g);
                if (rc) { t.stop(); return rc; }
                a.n0 = n;
                hipLaunchKernelGGL((dense_sqrt_fwd_kernel<2>), dim3(a.B), dim3(DT), 0, h->stream, a);
            }
            t.stop();
        }
        RK_HIP(hipGetLastError());
        if (mode == RK_MODE_SIM) {
            LaunchTimer t(h, "dense_sqrt_bwd_sim_kernel");
            hipLaunchKernelGGL(dense_sqrt_bwd_sim_kernel, dim3(a.B), dim3(DT), 0, h->stream, a);
            t.stop();
            RK_HIP(hipGetLastError());
        }
        if (mode == RK_MODE_MV && a.N >= 2) {
            LaunchTimer t(h, "dense_sqrt_bwd_mv_kernel");
            hipLaunchKernelGGL(dense_sqrt_bwd_mv_kernel, dim3(a.B), dim3(DT), 0, h->stream, a);
            t.stop();
            RK_HIP(hipGetLastError());
        }
        return RK_OK;
    }
    hipLaunchKernelGGL(dense_qcheck_kernel, dim3(1), dim3(256), 0, h->stream, a.Q, a.p, a.p / a.m, a.ws + a.ws_stride - 1);
    const bool extra = c->interrogate == RK_INTERROGATE_CHKREBTII || (c->flags & RK_FLAG_STORE_PRED);
    if (c->rhs_id == RK_RHS_LINEAR_DENSE) {
        LaunchTimer t(h, "dense_fwd_kernel");
        if (extra) hipLaunchKernelGGL((dense_fwd_kernel<0, true>), dim3(a.B), dim3(DT), 0, h->stream, a);
        else hipLaunchKernelGGL((dense_fwd_kernel<0, false>), dim3(a.B), dim3(DT), 0, h->stream, a);
        t.stop();
    } else {
        // a right-hand side built by hiprtc: the step is cut at the interrogation (solve_dense_itg_kernels.hpp) -- the
        // precompiled phases on either side, the user's code in between; N + 1 + N launches queued back to back
        // (a step is ~0.1-1 ms of device work at these sizes)
        DenseItgArgs g;
        g.B = a.B; g.N = a.N; g.t_min = a.t_min; g.t_max = a.t_max; g.W = a.W; g.theta = a.theta; g.theta_b = a.theta_b;
        g.ws = a.ws; g.ws_stride = a.ws_stride;
        {
            const size_t pp = (size_t)a.p * a.p, mp = (size_t)a.m * a.p;      // (carve)
            g.off_Wt = 4 * pp; g.off_mup = 4 * pp + 3 * mp + (size_t)a.m * a.m; g.off_am = g.off_mup + a.p;
            g.off_x = c->interrogate == RK_INTERROGATE_CHKREBTII ? g.off_mup + a.p + 2 * (size_t)a.m : g.off_mup;    // w.dm / w.mup
        }
        LaunchTimer t(h, "dense_fwd_kernel<user, stepwise>");
        hipLaunchKernelGGL((dense_fwd_kernel<1, true>), dim3(a.B), dim3(DT), 0, h->stream, a);
        for (int n = 0; n < a.N; ++n) {
            g.n = n;
            const int rc = user_dense_interrogate(h, c, g);
            if (rc) { t.stop(); return rc; }             // (the bracket's stop event must exist for rk_profile_last)
            a.n0 = n;
            hipLaunchKernelGGL((dense_fwd_kernel<2, true>), dim3(a.B), dim3(DT), 0, h->stream, a);
        }
        t.stop();
    }
    RK_HIP(hipGetLastError());
    if (mode == RK_MODE_SIM) {
        LaunchTimer t(h, "dense_bwd_sim_kernel");
        hipLaunchKernelGGL(dense_bwd_sim_kernel, dim3(a.B), dim3(DT), 0, h->stream, a);
        t.stop();
        RK_HIP(hipGetLastError());
    }
    if (mode == RK_MODE_MV && a.N >= 2) {
        LaunchTimer t(h, "dense_bwd_mv_kernel");
        hipLaunchKernelGGL(dense_bwd_mv_kernel, dim3(a.B), dim3(DT), 0, h->stream, a);
        t.stop();
        RK_HIP(hipGetLastError());
    }
    return RK_OK;
}

}  // namespace rk
