// Blocked MFMA-tile solver for n_bstate = 4 .. 8 (n_bmeas = 1): the general-n_deriv form of the tile path.
//
//   src/rodeo/solve.py:31-122   _solve_filter -> fwd_tilen_kernel        (solve_tilen_kernels.hpp) NB x NB tiles per block
//   src/rodeo/solve.py:257-301  solve_mv      -> tilen_gain_cols_kernel  time-parallel: one 16-lane row per (step, unit) item
//                                                + bwd_mv_tilen_kernel   the carry recursion on blocked MFMA tiles
//   src/rodeo/solve.py:162-204  solve_sim     -> tilen_gain_kernel<SIM>  + bwd_sim_tilen_kernel
//
// Why two kernels for the backward pass: the smoothing gain G_n = Sigma_f Q^T (Sigma-_{n+1})^{-1} (standard.py:175-176)
// and, for solve_sim, the conditional factor and the normals do not depend on the carry, so they are evaluated for ALL
// time steps at once -- one lane per (step, unit), register LU with partial pivoting like the reference's utils.py:119
// -- into a workspace record per item; the sequential part that remains is, per step,
//     solve_mv :  D = S_s - S- ; V = (G D)^T ; S_s = V^T G^T + Sigma_f ; m_s = G (m_s - m-) + mu_f     (standard.py:213-216)
//     solve_sim:  x = G (x - m-) + (mu_f + L z)                                                         (standard.py:248-254)
// on the same blocked tiles as the forward pass, with the records of the next steps loaded LOOKAHEAD steps ahead.
// (The p = 3 / p = 4 solve_mv kernels fuse the two through LDS; at p up to 8 a chunk of records no longer fits there.)
//
// Workspace record of item (n, unit), RS doubles (padded to even):
//     solve_mv :  [ G^T row-major (p*p) | Sigma-_{n+1} (p*p) | mu-_{n+1} (p) ]
//     solve_sim:  [ G^T row-major (p*p) | mu-_{n+1} (p)      | mu_f + L z (p) ]      (n = N: G = 0, mu- = 0: the terminal draw)
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "kalman_small.hpp"
#include "mfma_tile.hpp"
#include "philox.hpp"
#include "rhs.hpp"
#include "solve_args.hpp"
#include "solve_tilen_kernels.hpp"

namespace rk {

__host__ __device__ inline int tilen_rs(int p, bool sim) {
    const int rs = sim ? p * p + 2 * p : 2 * p * p + p;
    return rs + (rs & 1);
}

// ---- phase 1: one lane per (step, unit) ------------------------------------------------------------------------------
// grid = (number of steps) x ceil(n_units / 64) workgroups; block = 64 lanes = 64 consecutive units of one time step.
// The 64 input tiles and the 64 output records of a wave are contiguous in HBM, so both pass through LDS: coalesced
// 512-byte rows on the memory side, one record per lane (odd stride: two-way bank conflicts at most) on the register
// side -- lane-strided global accesses touch 64 cache lines per instruction.
template <int P, bool SIM>
__global__ void __launch_bounds__(64) tilen_gain_kernel(SolveArgs a, const double* __restrict__ tiles, double* __restrict__ ws,
                                                        int blocks_per_step, int n_first) {
    constexpr int PP = P * P + P;
    constexpr int RS0 = SIM ? P * P + 2 * P : 2 * P * P + P, RS = RS0 + (RS0 & 1);
    constexpr bool STAGED = P <= 6;                                  // beyond that the staging registers only add spills (measured)
    constexpr int LP = PP | 1, LR = RS | 1, LMAX = STAGED ? (LP > LR ? LP : LR) : 1;
    __shared__ double sh[64 * LMAX];
    const int D = a.D, n_units = a.B * D, lane = threadIdx.x;
    const int step_idx = blockIdx.x / blocks_per_step;
    const int tau0 = (blockIdx.x - step_idx * blocks_per_step) * 64;
    const int n_here = n_units - tau0 < 64 ? n_units - tau0 : 64;
    const int n = step_idx + n_first;                                // mv: 1 .. N-1 ; sim: 1 .. N (this launch: n_first ..)
    const bool live = lane < n_here;
    const int tau = live ? tau0 + lane : n_units - 1;
    const int b = tau / D, blk = tau - b * D;
    if constexpr (STAGED) {
        // (all PP loads are issued before the first LDS store: a rolled loop would wait for every load in turn)
        const double* src = tiles + ((size_t)n * n_units + tau0) * PP;
        const int cnt = n_here * PP;
        double tmp[PP];
#pragma unroll
        for (int k = 0; k < PP; ++k) { const int i = lane + 64 * k; tmp[k] = src[i < cnt ? i : cnt - 1]; }
#pragma unroll
        for (int k = 0; k < PP; ++k) {
            const int i = lane + 64 * k;
            if (i < cnt) sh[(i / PP) * LP + (i % PP)] = tmp[k];
        }
        __syncthreads();
    }
    double Q[P][P], R[P][P];
    load_block_consts<P>(a, blk, b, Q, R);
    double mf[P], Sf[P][P];
    if constexpr (STAGED) {
        const double* my = sh + (live ? lane : 0) * LP;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            mf[i] = my[P * P + i];
#pragma unroll
            for (int j = 0; j < P; ++j) Sf[i][j] = my[i * P + j];
        }
        __syncthreads();                                             // the staging area is reused for the records
    } else {
        const double2* src = (const double2*)(tiles + ((size_t)n * n_units + tau) * PP);     // PP is even: 16-byte aligned
        double buf[PP];
#pragma unroll
        for (int k = 0; k < PP / 2; ++k) { const double2 v = src[k]; buf[2 * k] = v.x; buf[2 * k + 1] = v.y; }
#pragma unroll
        for (int i = 0; i < P; ++i) {
            mf[i] = buf[P * P + i];
#pragma unroll
            for (int j = 0; j < P; ++j) Sf[i][j] = buf[i * P + j];
        }
    }
    double* rec = STAGED ? sh + lane * LR : ws + ((size_t)n * n_units + tau) * RS;
    if (!STAGED && !live) return;
    const bool term = SIM && n == a.N;                               // terminal draw x_N ~ N(filt[N]) (solve.py:182-186)
    double mp[P], Sp[P][P], T[P][P], A[P][P], X[P][P];
    predict_block<P>(Q, R, mf, Sf, mp, Sp);                          // pred[n+1] from filt[n]   (standard.py:57-59)
    mm_nt<P, P, P>(Sf, Q, T);                                        // T = Sigma_f Q^T          (standard.py:175)
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) { A[i][j] = Sp[i][j]; X[i][j] = T[j][i]; }
    lu_solve<P, P>(A, X);                                            // X = solve(Sigma-, T^T) = G^T (standard.py:176)
    if constexpr (!SIM) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            rec[2 * P * P + i] = mp[i];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                rec[i * P + j] = X[i][j];
                rec[P * P + i * P + j] = Sp[i][j];
            }
        }
    } else {
        const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
        double z[P], G[P][P], GT[P][P], Ssim[P][P], L[P][P];
        normals<P>(a.seed, traj, (uint32_t)n, (uint32_t)blk, PURPOSE_SMOOTH, z);
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int j = 0; j < P; ++j) G[i][j] = term ? 0.0 : X[j][i];
        mm_nt<P, P, P>(G, T, GT);
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int j = 0; j < P; ++j) Ssim[i][j] = Sf[i][j] - GT[i][j];          // standard.py:253-254
        psd_factor<P>(Ssim, L);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            double w = mf[i];
#pragma unroll
            for (int k = 0; k <= i; ++k) w = fma(L[i][k], z[k], w);
            rec[P * P + i] = term ? 0.0 : mp[i];
            rec[P * P + P + i] = w;                                                 // mu_f + L z
#pragma unroll
            for (int j = 0; j < P; ++j) rec[i * P + j] = G[j][i];                   // G^T
        }
    }
    if constexpr (RS != RS0) rec[RS0] = 0.0;                         // the padding double
    if constexpr (STAGED) {
        __syncthreads();
        double* dst = ws + ((size_t)n * n_units + tau0) * RS;
        const int cnt = n_here * RS;
#pragma unroll 8
        for (int k = 0; k < RS; ++k) {
            const int i = lane + 64 * k;
            if (i < cnt) dst[i] = sh[(i / RS) * LR + (i % RS)];
        }
    }
}

// ---- raw buffer accesses on a window (scalar base, per-lane byte offset; out-of-range lanes load 0.0 / store nothing) ----
constexpr int TN_OOR = (int)0x80000000;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t rs, int off) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
    double d;
    __builtin_memcpy(&d, &v, 8);
    return d;
}
__device__ __forceinline__ void buf_st(double d, __amdgpu_buffer_rsrc_t rs, int off) {
    u32x2 v;
    __builtin_memcpy(&v, &d, 8);
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_window(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

// ---- phase 1 for solve_mv: one 16-lane DPP row per item, one matrix column per lane ----------------------------------------
// The lane-per-item kernel above holds five p x p matrices per lane: 390 / 474 / 512 (+ 168 spilled) VGPRs at p = 6 / 7 / 8, one
// wave per SIMD, and 2.2 / 4.3 / 9.0 ms on the headline shape against 0.3 ms of fp64 VALU work (solve_sim still uses it).  Here an item is a 16-lane
// row of the wave (four items per wave: the four units of the chain wave, looped over a chunk of time steps), lane j < 8
// holds COLUMN j of the p x p matrices and lane 8 + j column j of the right-hand sides, and whatever a lane needs from
// another column comes through the DPP operand of a 64-bit FMA (v_fmac_f64_dpp row_newbcast:k = lane k of the own row):
//     lanes 0..7 :  v = Sigma_f[:, j]   y = Q v = (Q Sigma_f)[:, j]      A[:, j] = sum_k y_k[:] Q[j][k] + R[:, j] = Sigma-[:, j]
//     lanes 8..15:  v = Sigma_f[j, :]   y = Q v = (Sigma_f Q^T)[j, :]^T  = column j of T^T, the right-hand sides of utils.py:119
// and the LU with partial pivoting of [Sigma- | T^T] runs with every lane on its own column: the pivot search in column k
// is lane k's, the row index and the multipliers l_i reach the other lanes by row_newbcast:k, the row swaps are selects
// on the lane's own 8 values.  Back substitution: the right-hand-side lanes read U[k][i] from lane i.  Every sum has the
// terms and the order of the lane-per-item kernel (lu_factor_fwd / lu_back, mm / mm_nt of linalg_small.hpp): the records
// are the same to the bit (tests/test_gpu_tilen.py compares the two kernels).  About 40 VGPRs per p instead of 60 per p.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int J>
__device__ __forceinline__ void fmac_bc(double& acc, double x, double y) {          // acc += x(lane J of the row) * y
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
}
template <int J>
__device__ __forceinline__ void fnmac_bc(double& acc, double x, double y) {         // acc -= x(lane J of the row) * y
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
}
// A VALU write of a VGPR needs two wait states before a DPP operand reads it, and nothing checks that for inline
// assembly: this ties the values to an s_nop, so that their producers stay in front of it and the DPP reads behind.
template <int P>
__device__ __forceinline__ void dpp_fence(double (&x)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) asm volatile("" : "+v"(x[i]));
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int i = 0; i < P; ++i) asm volatile("" : "+v"(x[i]));
}
template <int J>
__device__ __forceinline__ double row_bc(double x) {                                // lane J of the row, in every lane
    int lo_ = __double2loint(x), hi_ = __double2hiint(x);
    lo_ = __builtin_amdgcn_mov_dpp(lo_, 0x150 + J, 0xF, 0xF, false);
    hi_ = __builtin_amdgcn_mov_dpp(hi_, 0x150 + J, 0xF, 0xF, false);
    return __hiloint2double(hi_, lo_);
}

// One item (one 16-lane row): from this lane's column v of Sigma_f (lanes 0..7: column j, lanes 8..15: row j) and mu_f to
// Sigma-[:, j] (Sp, lanes 0..7), mu-_j (mp, lanes 0..7) and column j of G^T (A, lanes 8..15).
template <int P>
__device__ __forceinline__ void cols_gain_item(const double (&Qc)[P], const double (&Qr)[P], const double (&Rc)[P], int h,
                                               const double (&v)[P], double mu, double (&A)[P], double (&Sp)[P], double& mp) {
    // ---- y = Q v  (mm(Q, Sigma_f) column j | mm_nt(Sigma_f, Q) row j), mu- = Q mu_f  (standard.py:57-59, 175) ----
    double y[P];
    mp = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) y[i] = 0.0;
    static_for<0, P>([&](auto K) {
        constexpr int k = decltype(K)::value;
#pragma unroll
        for (int i = 0; i < P; ++i) fmac_bc<k>(y[i], Qc[i], v[k]);
        fmac_bc<k>(mp, mu, Qr[k]);
    });
    // ---- Sigma- = (Q Sigma_f) Q^T + R: column j in the lanes 0..7; the others keep their right-hand-side column ----
    double sp[P];
#pragma unroll
    for (int i = 0; i < P; ++i) sp[i] = 0.0;
    dpp_fence<P>(y);
    static_for<0, P>([&](auto K) {
        constexpr int k = decltype(K)::value;
#pragma unroll
        for (int i = 0; i < P; ++i) fmac_bc<k>(sp[i], y[i], Qr[k]);
    });
#pragma unroll
    for (int i = 0; i < P; ++i) {
        Sp[i] = sp[i] + Rc[i];
        A[i] = h ? y[i] : Sp[i];
    }
    // ---- LU with partial pivoting of [Sigma- | T^T], one column per lane (lu_factor_fwd, utils.py:119) ----
    double rpv[P];
    static_for<0, P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        int piv = k;
        double best = fabs(A[k]);
#pragma unroll
        for (int i = k + 1; i < P; ++i) {
            const double w = fabs(A[i]);
            const bool gt = w > best;
            best = gt ? w : best;
            piv = gt ? i : piv;
        }
        piv = __builtin_amdgcn_mov_dpp(piv, 0x150 + k, 0xF, 0xF, false);        // column k's choice
#pragma unroll
        for (int i = k + 1; i < P; ++i) {
            const bool sw = piv == i;
            const double t = A[k];
            A[k] = sw ? A[i] : t;
            A[i] = sw ? t : A[i];
        }
        rpv[k] = fast_rcp(A[k]);                      // (lane k's is the pivot's)
        if constexpr (k + 1 < P) {
            double l[P - k - 1];
#pragma unroll
            for (int i = k + 1; i < P; ++i) l[i - k - 1] = A[i] * rpv[k];
            dpp_fence<P - k - 1>(l);
#pragma unroll
            for (int i = k + 1; i < P; ++i) fnmac_bc<k>(A[i], l[i - k - 1], A[k]);
        }
    });
    // ---- back substitution (lu_back): the right-hand-side lanes read U[k][i] from lane i's A[k] ----
    dpp_fence<P>(A);
    static_for<0, P>([&](auto KK) {
        constexpr int k = P - 1 - decltype(KK)::value;
        double s = A[k];
        static_for<k + 1, P>([&](auto I) {
            constexpr int i = decltype(I)::value;
            fnmac_bc<i>(s, A[k], A[i]);
        });
        const double x = s * row_bc<k>(rpv[k]);
        A[k] = h ? x : A[k];
    });
}

template <int P>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
tilen_gain_cols_kernel(SolveArgs a, const double* __restrict__ tiles, double* __restrict__ ws, int n_first, int n_last, int chunk) {
    static_assert(P >= 2 && P <= 8, "one column per lane: p <= 8");
    constexpr int PP = P * P + P, RS0 = 2 * P * P + P, RS = RS0 + (RS0 & 1);
    const int n_units = a.B * a.D, lane = threadIdx.x, g = lane >> 4, h = (lane >> 3) & 1, j = lane & 7;
    const int n_valid = n_units - (int)blockIdx.x * 4 < 4 ? n_units - (int)blockIdx.x * 4 : 4;
    const bool live = g < n_valid && j < P;
    const int tau_raw = blockIdx.x * 4 + g, tau = tau_raw < n_units ? tau_raw : n_units - 1, jj = j < P ? j : P - 1;
    const int b = tau / a.D, blk = tau - b * a.D;
    double Qc[P], Qr[P], Rc[P];                           // Q[:, j], Q[j, :], R[:, j]
#pragma unroll
    for (int i = 0; i < P; ++i) {
        Qc[i] = ld(a.Q, ((size_t)blk * P + i) * P + jj, a.Q_b, a.B, b);
        Qr[i] = ld(a.Q, ((size_t)blk * P + jj) * P + i, a.Q_b, a.B, b);
        Rc[i] = ld(a.R, ((size_t)blk * P + i) * P + jj, a.R_b, a.B, b);
    }
    // The wave's four tiles of a time row (4 PP doubles) and its four records (4 RS doubles) are contiguous in HBM: whole rows
    // move with 16 bytes per lane (3 + 5 instructions per item at p = 8 where one access per lane and matrix row took 19 --
    // the CU's one address path, ~40 cycles for 64 separate addresses, was what bounded the first version: 3.0 ms at p = 8,
    // 1.9 ms at p = 5), and the lanes pick their columns out of LDS.  One wave per workgroup: LDS accesses of a wave
    // execute in order, no barrier.  Rows past the last unit load as zeros and are not stored (buffer range).
    constexpr int NLT = (4 * PP * 8 + 1023) / 1024, NLW = (4 * RS * 8 + 1023) / 1024;
    constexpr int IN_D = NLT * 128, OUT_D = NLW * 128, DUMP = IN_D + OUT_D;
    __shared__ __attribute__((aligned(16))) double sh[IN_D + OUT_D + 2];
    double* const in = sh;
    double* const out = sh + IN_D;
    int iV[P], iOut[P];                                   // LDS indices (doubles)
#pragma unroll
    for (int k = 0; k < P; ++k) {
        iV[k] = g * PP + (h ? jj * P + k : k * P + jj);                             // Sigma_f[k][j] | Sigma_f[j][k]
        iOut[k] = live ? IN_D + g * RS + (h ? 0 : P * P) + k * P + j : DUMP;        // G^T[k][j]     | Sigma-[k][j]
    }
    const int iMu = g * PP + P * P + jj;
    const int iMp = (live && !h) ? IN_D + g * RS + 2 * P * P + j : DUMP;
    const int iPad = (RS != RS0 && lane % 16 == 0) ? IN_D + g * RS + RS0 : DUMP;
    const size_t tstride = (size_t)n_units * PP, wstride = (size_t)n_units * RS;
    const double* const tw = tiles + (size_t)blockIdx.x * 4 * PP;
    double* const ww = ws + (size_t)blockIdx.x * 4 * RS;
    const int tbytes = n_valid * PP * 8, wbytes = n_valid * RS * 8;
    const int n0 = n_first + (int)blockIdx.y * chunk, n1 = n0 + chunk - 1 < n_last ? n0 + chunk - 1 : n_last;
    u32x4 row[NLT];
    auto gload = [&](int n) {
        const __amdgpu_buffer_rsrc_t t = buf_window(tw + (size_t)n * tstride, tbytes);
#pragma unroll
        for (int i = 0; i < NLT; ++i) row[i] = __builtin_amdgcn_raw_buffer_load_b128(t, 1024 * i + 16 * lane, 0, 0);
    };
    if (n0 <= n1) gload(n0);
    for (int n = n0; n <= n1; ++n) {
        double v[P], mu;
        {
            u32x4* const dst = (u32x4*)in;
#pragma unroll
            for (int i = 0; i < NLT; ++i) dst[64 * i + lane] = row[i];
#pragma unroll
            for (int k = 0; k < P; ++k) v[k] = sh[iV[k]];
            mu = sh[iMu];
        }
        if (n < n1) gload(n + 1);                         // (the next item's rows fly behind this item's arithmetic)
        double A[P], Sp[P], mp;
        cols_gain_item<P>(Qc, Qr, Rc, h, v, mu, A, Sp, mp);
        // ---- the record [G^T | Sigma- | mu-] ----
#pragma unroll
        for (int i = 0; i < P; ++i) sh[iOut[i]] = h ? A[i] : Sp[i];
        sh[iMp] = mp;
        if constexpr (RS != RS0) sh[iPad] = 0.0;
        const __amdgpu_buffer_rsrc_t w = buf_window(ww + (size_t)n * wstride, wbytes);
        const u32x4* const src = (const u32x4*)out;
#pragma unroll
        for (int i = 0; i < NLW; ++i) __builtin_amdgcn_raw_buffer_store_b128(src[64 * i + lane], w, 1024 * i + 16 * lane, 0, 0);
    }
}

// ---- phase 2: the sequential chains on blocked tiles ---------------------------------------------------------------------
// One wave = 4 units (lane = 16 r + 4 g + c).  The records of the next TN_RING - 1 steps are always in flight (static
// register names in a fully unrolled ring, so hipcc counts vmcnt per load and the loop-carried slots need no copies).
// All accesses are raw buffer loads / stores on a window over the wave's four units of one time row: lanes of the zero
// padding (and of units past the end) carry an out-of-range offset, so their loads return 0.0 and their stores are
// dropped -- no selects and no exec-mask branches in the step.
constexpr int TN_RING = 4;        // (22 memory operations per step at NB = 2: the 6-bit vmcnt covers about three steps anyway)

// At most 256 VGPRs (waves_per_eu): two chain waves per SIMD at larger batches.
template <int NB>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
bwd_mv_tilen_kernel(SolveArgs a, double* __restrict__ tiles, const double* __restrict__ ws, int P,
                                                          int n_top, int n_bot) {
    const int n_units = a.B * a.D, PP = P * P + P, RS = tilen_rs(P, false);
    const int lane = threadIdx.x, r = lane >> 4, g = (lane >> 2) & 3, c = lane & 3;
    const bool valid = blockIdx.x * 4 + g < n_units;
    // per-lane byte offsets inside the wave's window of a tile row (4 PP doubles) / a workspace row (4 RS doubles)
    int oS[NB][NB], oM[NB], oMst[NB], oG[NB][NB], oP[NB][NB], oMp[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int i = 4 * k + r;
        const bool iv = valid && i < P;
        oM[k] = iv ? (g * PP + P * P + i) * 8 : TN_OOR;
        oMst[k] = (iv && c == 0) ? oM[k] : TN_OOR;
        oMp[k] = iv ? (g * RS + 2 * P * P + i) * 8 : TN_OOR;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int j = 4 * bb + c;
            const bool in = iv && j < P;
            oS[k][bb] = in ? (g * PP + i * P + j) * 8 : TN_OOR;
            oG[k][bb] = in ? (g * RS + i * P + j) * 8 : TN_OOR;
            oP[k][bb] = in ? (g * RS + P * P + i * P + j) * 8 : TN_OOR;
        }
    }
    const size_t tstride = (size_t)n_units * PP, wstride = (size_t)n_units * RS;
    const double* const tw = tiles + (size_t)blockIdx.x * 4 * PP;
    const double* const ww = ws + (size_t)blockIdx.x * 4 * RS;
    const int tbytes = 4 * PP * 8, wbytes = 4 * RS * 8;
    // this launch smooths the steps n_top .. n_bot (descending); its carry is the tile of step n_top + 1: filt[N] for the
    // first chunk (solve.py:279-282), the smoothed state the previous chunk's launch left there otherwise
    double Ms[NB][NB], ms[NB];
    {
        const __amdgpu_buffer_rsrc_t t = buf_window(tw + (size_t)(n_top + 1) * tstride, tbytes);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            ms[k] = buf_ld(t, oM[k]);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) Ms[k][bb] = buf_ld(t, oS[k][bb]);
        }
    }
    struct Rec { double Gt[NB][NB], Sp[NB][NB], Sf[NB][NB], mp[NB], mf[NB]; };
    auto load = [&](int n, Rec& q) {
        const __amdgpu_buffer_rsrc_t t = buf_window(tw + (size_t)n * tstride, tbytes);
        const __amdgpu_buffer_rsrc_t w = buf_window(ww + (size_t)n * wstride, wbytes);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            q.mp[k] = buf_ld(w, oMp[k]);
            q.mf[k] = buf_ld(t, oM[k]);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                q.Gt[k][bb] = buf_ld(w, oG[k][bb]);
                q.Sp[k][bb] = buf_ld(w, oP[k][bb]);
                q.Sf[k][bb] = buf_ld(t, oS[k][bb]);
            }
        }
    };
    auto step = [&](int n, const Rec& q) {
        double Dm[NB][NB], dm[NB], V1[NB][NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            dm[k] = ms[k] - q.mp[k];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) Dm[k][bb] = Ms[k][bb] - q.Sp[k][bb];
        }
        bmm_tn0<NB>(Dm, q.Gt, V1);                       // (G D)^T
        bmm_tn<NB>(V1, q.Gt, q.Sf, Ms);                  // G D G^T + Sigma_f      (standard.py:215-216)
        bmv_t<NB>(q.Gt, dm, q.mf, ms);                   // G (m_s - m-) + mu_f    (standard.py:213-214)
        const __amdgpu_buffer_rsrc_t t = buf_window(tw + (size_t)n * tstride, tbytes);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            buf_st(ms[k], t, oMst[k]);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) buf_st(Ms[k][bb], t, oS[k][bb]);
        }
    };
    // steps n_top .. n_bot through a ring of TN_RING records: slot s is consumed and at once refilled with the record
    // TN_RING steps further down, so TN_RING - 1 steps of loads are in flight all the time, also across loop iterations
    // (indices below n_bot are clamped: a harmless reload that is never consumed)
    int n = n_top;
    Rec q[TN_RING];
#pragma unroll
    for (int s = 0; s < TN_RING; ++s) load(n - s >= n_bot ? n - s : n_bot, q[s]);
    while (n - n_bot + 1 >= TN_RING) {
#pragma unroll
        for (int s = 0; s < TN_RING; ++s) {
            step(n - s, q[s]);
            const int nn = n - s - TN_RING;
            load(nn >= n_bot ? nn : n_bot, q[s]);
        }
        n -= TN_RING;
    }
#pragma unroll
    for (int s = 0; s < TN_RING; ++s)
        if (s < n - n_bot + 1) step(n - s, q[s]);         // (uniform condition)
}

// ---- solve_mv chain with coalesced rows (template over n_bstate) --------------------------------------------------------
// The chain above issues 22 scattered 8-byte memory instructions per step (16 loads, 6 stores: every lane its own tile
// element); each costs the address path ~40 cycles whatever the data volume, and together they -- not the 20 MFMAs --
// set its step time (measured: without the stores 1.36 instead of 1.80 ms at n_bstate = 5).  Used at n_bstate = 7, 8 (see
// tilen_solve for the measured crossover).  The four units of a wave are
// contiguous in the workspace row (4 RS doubles) and in the tile row (4 (p^2 + p) doubles), so this version moves whole
// rows: 16-byte-per-lane raw buffer loads into registers TS_LA steps ahead, ds_write_b128 into a per-wave LDS slot, the tile
// elements out of LDS in the D layout (padding lanes read a zero word), and the smoothed tiles back through LDS as one
// coalesced 16-byte-per-lane store.  Same arithmetic in the same order as bwd_mv_tilen_kernel.
constexpr int TS_LA = 4;                                   // steps of row loads in flight

template <int P>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
bwd_mv_tilen_rows_kernel(SolveArgs a, double* __restrict__ tiles, const double* __restrict__ ws, int n_top, int n_bot) {
    constexpr int NB = P <= 4 ? 1 : 2, PP = P * P + P, RS0 = 2 * P * P + P, RS = RS0 + (RS0 & 1);
    constexpr int WB = 4 * RS * 8, TB = 4 * PP * 8;                         // bytes of the wave's workspace / tile row part
    constexpr int NLW = (WB + 1023) / 1024, NLT = (TB + 1023) / 1024;       // 16-byte-per-lane instructions per part
    constexpr int SLOT = (NLW + NLT) * 128;                                 // doubles per staging slot (whole instructions)
    constexpr int TOFF = NLW * 128;                                         // tile part inside a slot
    __shared__ __attribute__((aligned(16))) double lds[2 * SLOT + NLT * 128 + 2];
    double* const outz = lds + 2 * SLOT;                                    // output staging (NLT * 128 doubles)
    constexpr int ZERO = 2 * SLOT + NLT * 128, DUMP = ZERO + 1;             // a zero word for padding lanes; a dump word
    const int n_units = a.B * a.D;
    const int lane = threadIdx.x, r = lane >> 4, g = (lane >> 2) & 3, c = lane & 3;
    const int n_valid = n_units - (int)blockIdx.x * 4 < 4 ? n_units - (int)blockIdx.x * 4 : 4;
    const bool valid = g < n_valid;
    if (lane == 0) { lds[ZERO] = 0.0; lds[DUMP] = 0.0; }
    // LDS offsets (doubles, relative to a slot / to the output staging) of this lane's tile elements
    int lG[NB][NB], lP[NB][NB], lS[NB][NB], lMp[NB], lMf[NB], oS[NB][NB], oM[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int i = 4 * k + r;
        const bool iv = valid && i < P;
        lMp[k] = iv ? g * RS + 2 * P * P + i : -1;
        lMf[k] = iv ? TOFF + g * PP + P * P + i : -1;
        oM[k] = (iv && c == 0) ? g * PP + P * P + i : -1;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int j = 4 * bb + c;
            const bool in = iv && j < P;
            lG[k][bb] = in ? g * RS + i * P + j : -1;
            lP[k][bb] = in ? g * RS + P * P + i * P + j : -1;
            lS[k][bb] = in ? TOFF + g * PP + i * P + j : -1;
            oS[k][bb] = in ? g * PP + i * P + j : -1;
        }
    }
    const size_t tstride = (size_t)n_units * PP, wstride = (size_t)n_units * RS;
    const double* const tw = tiles + (size_t)blockIdx.x * 4 * PP;
    const double* const ww = ws + (size_t)blockIdx.x * 4 * RS;
    const int tbytes = n_valid * PP * 8, wbytes = n_valid * RS * 8;         // rows past the last unit read as zeros
    struct Row { u32x4 w[NLW], t[NLT]; };
    auto gload = [&](int n, Row& q) {
        const __amdgpu_buffer_rsrc_t rw = buf_window(ww + (size_t)n * wstride, wbytes);
        const __amdgpu_buffer_rsrc_t rt = buf_window(tw + (size_t)n * tstride, tbytes);
#pragma unroll
        for (int i = 0; i < NLW; ++i) q.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, 1024 * i + 16 * lane, 0, 0);
#pragma unroll
        for (int i = 0; i < NLT; ++i) q.t[i] = __builtin_amdgcn_raw_buffer_load_b128(rt, 1024 * i + 16 * lane, 0, 0);
    };
    auto to_lds = [&](const Row& q, int slot) {
        u32x4* dst = (u32x4*)(lds + slot * SLOT);
#pragma unroll
        for (int i = 0; i < NLW; ++i) dst[64 * i + lane] = q.w[i];
#pragma unroll
        for (int i = 0; i < NLT; ++i) dst[64 * (NLW + i) + lane] = q.t[i];
    };
    struct Rec { double Gt[NB][NB], Sp[NB][NB], Sf[NB][NB], mp[NB], mf[NB]; };
    auto from_lds = [&](int slot, Rec& q) {
        const double* s0 = lds + slot * SLOT;
        auto rd = [&](int off) { return off >= 0 ? s0[off] : lds[ZERO]; };
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            q.mp[k] = rd(lMp[k]);
            q.mf[k] = rd(lMf[k]);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                q.Gt[k][bb] = rd(lG[k][bb]);
                q.Sp[k][bb] = rd(lP[k][bb]);
                q.Sf[k][bb] = rd(lS[k][bb]);
            }
        }
    };
    // carry: the tile of step n_top + 1 (filt[N] for the first launch, solve.py:279-282)
    double Ms[NB][NB], ms[NB];
    {
        Row q;
        const __amdgpu_buffer_rsrc_t rt = buf_window(tw + (size_t)(n_top + 1) * tstride, tbytes);
#pragma unroll
        for (int i = 0; i < NLW; ++i) q.w[i] = u32x4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NLT; ++i) q.t[i] = __builtin_amdgcn_raw_buffer_load_b128(rt, 1024 * i + 16 * lane, 0, 0);
        to_lds(q, 0);
        Rec c0;
        from_lds(0, c0);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            ms[k] = c0.mf[k];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) Ms[k][bb] = c0.Sf[k][bb];
        }
    }
    auto step = [&](int n, const Rec& q) {
        double Dm[NB][NB], dm[NB], V1[NB][NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            dm[k] = ms[k] - q.mp[k];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) Dm[k][bb] = Ms[k][bb] - q.Sp[k][bb];
        }
        bmm_tn0<NB>(Dm, q.Gt, V1);                       // (G D)^T
        bmm_tn<NB>(V1, q.Gt, q.Sf, Ms);                  // G D G^T + Sigma_f      (standard.py:215-216)
        bmv_t<NB>(q.Gt, dm, q.mf, ms);                   // G (m_s - m-) + mu_f    (standard.py:213-214)
        // the smoothed tiles back as whole rows: D layout -> LDS -> 16 bytes per lane
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            outz[oM[k] >= 0 ? oM[k] : DUMP - 2 * SLOT] = ms[k];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) outz[oS[k][bb] >= 0 ? oS[k][bb] : DUMP - 2 * SLOT] = Ms[k][bb];
        }
        const __amdgpu_buffer_rsrc_t rt = buf_window(tw + (size_t)n * tstride, tbytes);
        const u32x4* src = (const u32x4*)outz;
#pragma unroll
        for (int i = 0; i < NLT; ++i) __builtin_amdgcn_raw_buffer_store_b128(src[64 * i + lane], rt, 1024 * i + 16 * lane, 0, 0);
    };
    // rows of the next TS_LA steps in registers; the record of the next step already in LDS / in `nxt` while the current
    // one is consumed (indices below n_bot are clamped: a harmless reload that is never consumed)
    Row rows[TS_LA];
    int n = n_top;
#pragma unroll
    for (int s = 0; s < TS_LA; ++s) gload(n - s >= n_bot ? n - s : n_bot, rows[s]);
    Rec cur, nxt;
    to_lds(rows[0], 1);
    from_lds(1, cur);
    { const int nn = n - TS_LA; gload(nn >= n_bot ? nn : n_bot, rows[0]); }
    int parity = 0;                                       // the slot the NEXT record goes to
    while (n >= n_bot) {
#pragma unroll
        for (int s = 0; s < TS_LA; ++s) {
            if (n - s >= n_bot) {                         // (uniform)
                // stage the record of step n - s - 1 (rows[(s + 1) % TS_LA]) and refill that register slot
                constexpr int dummy = 0; (void)dummy;
                const int s1 = (s + 1) % TS_LA;
                to_lds(rows[s1], parity);
                from_lds(parity, nxt);
                { const int nn = n - s - 1 - TS_LA; gload(nn >= n_bot ? nn : n_bot, rows[s1]); }
                parity ^= 1;
                step(n - s, cur);
                cur = nxt;
            }
        }
        n -= TS_LA;
    }
}

// ---- solve_mv backward pass in ONE kernel: the chain wave fed through LDS by gain waves -----------------------------------------
// The two-kernel form writes a record of 2 p^2 + p doubles per (step, unit) to HBM and reads it back (3.4 x the backward pass's
// algorithmic traffic, profiles/r03_nderiv5_pmc_traffic.json), and its chain spends most of its step on the 16 scattered loads
// that fetch the record and on waiting for its own stores.  Here a workgroup is four units: wave 0 runs the chain of
// bwd_mv_tilen_kernel, waves 1 .. NPROD produce the records of the next chunk of FZ_C time steps (cols_gain_item: a 16-lane row
// per unit, every NPROD-th step) straight into
// LDS, double-buffered, one barrier per chunk; the filtered tiles they load as whole rows stay in LDS for the chain too, so
// the chain's only memory instructions are LDS accesses: it writes the smoothed tiles over the filtered rows in their LDS slot
// and the producers store them as whole rows before they refill the slot.  HBM sees the filtered tiles once and the smoothed
// tiles once.  The producers run ahead of the chain (lower n), so the in-place stores never meet a load.
// Where the time goes (s_memtime stamps per wave, RK_TILEN_STAMPS=1, scripts/probe/fz_stamps.py; cycles per step at p = 5 / 8):
// chain 594 / 648 of work + 312 / 781 at the barriers, producers 600-880 / 1110-1400 per step, i.e. 2400-3500 / 4500-5600
// per item (four items per producer and four steps): the workgroup's ten waves per CU are VALU-issue-bound on the gain
// items -- (2 x 1800 + 2 x 594) / 4 SIMDs = 1200 cycles per step at p = 5 is what is measured.  Tried and removed: a ring of
// eight slots with an LDS item counter and ready / consumed flags instead of the chunk barrier (the waves of a workgroup
// share SIMDs unevenly and the slowest producer sets the pace of a chunk): 2.45 / 2.90 / 4.28 / 4.70 ms at p = 5 .. 8
// against 2.32 / 2.77 / 3.35 / 4.80 -- the flags and run-time slot addresses cost the chain what the balance gains.
constexpr int FZ_C = 4;                                   // time steps per chunk

template <int P, int NPROD>
__global__ void __launch_bounds__(64 * (1 + NPROD)) __attribute__((amdgpu_waves_per_eu(2)))
bwd_mv_tilen_fused_kernel(SolveArgs a, double* __restrict__ tiles, int n_top, int n_bot, double* dbg) {
    static_assert(P >= 5 && P <= 8, "blocked tiles with NB = 2");
    // dbg (RK_TILEN_STAMPS=1, workgroup 0 only): per wave [cycles of work, cycles at the chunk barriers] (s_memtime)
    uint64_t t_work = 0, t_wait = 0, t0 = dbg ? __builtin_amdgcn_s_memtime() : 0;
    auto lap = [&](uint64_t& acc) { if (dbg) { const uint64_t t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; } };
    static_assert(FZ_C % NPROD == 0, "every producer wave has the same slots in every chunk");
    constexpr int NT = 64 * (1 + NPROD);
    constexpr int NB = 2, PP = P * P + P, RS0 = 2 * P * P + P, RS = RS0 + (RS0 & 1);
    constexpr int NLT = (4 * PP * 8 + 1023) / 1024, TW = NLT * 128;         // a staged tile row: whole 16-byte-per-lane loads
    static_assert(4 * PP < TW, "the tail of a staged tile row is the zero word of the chain's padding lanes");
    constexpr int RW = 4 * RS + 2;                                          // a record row + [dump word, zero word]
    constexpr int REC = 0, TIL = 2 * FZ_C * RW, TOTAL = TIL + 2 * FZ_C * TW;
    __shared__ __attribute__((aligned(16))) double sh[TOTAL];
    const int n_units = a.B * a.D, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_valid = n_units - (int)blockIdx.x * 4 < 4 ? n_units - (int)blockIdx.x * 4 : 4;
    const size_t tstride = (size_t)n_units * PP;
    double* const tw = tiles + (size_t)blockIdx.x * 4 * PP;
    const int tbytes = n_valid * PP * 8;
    const int nch = (n_top - n_bot + FZ_C) / FZ_C;                           // chunks of this launch
    for (int e = threadIdx.x; e < 2 * FZ_C; e += NT) sh[REC + e * RW + 4 * RS + 1] = 0.0;      // the records' zero words
    __syncthreads();
    if (wave != 0) {
        // ================= producers: wave 1 + pw has the slots pw, pw + NPROD, ... of every chunk =================
        const int pw = wave - 1, g = lane >> 4, h = (lane >> 3) & 1, j = lane & 7;
        const int tau_raw = blockIdx.x * 4 + g, tau = tau_raw < n_units ? tau_raw : n_units - 1, jj = j < P ? j : P - 1;
        const int b = tau / a.D, blk = tau - b * a.D;
        double Qc[P], Qr[P], Rc[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            Qc[i] = ld(a.Q, ((size_t)blk * P + i) * P + jj, a.Q_b, a.B, b);
            Qr[i] = ld(a.Q, ((size_t)blk * P + jj) * P + i, a.Q_b, a.B, b);
            Rc[i] = ld(a.R, ((size_t)blk * P + i) * P + jj, a.R_b, a.B, b);
        }
        int iV[P], iOut[P];                               // indices inside a tile slot / a record slot
#pragma unroll
        for (int k = 0; k < P; ++k) {
            iV[k] = g * PP + (h ? jj * P + k : k * P + jj);
            iOut[k] = j < P ? g * RS + (h ? 0 : P * P) + k * P + j : 4 * RS;         // (lanes without a column: the dump word)
        }
        const int iMu = g * PP + P * P + jj;
        const int iMp = (j < P && !h) ? g * RS + 2 * P * P + j : 4 * RS;
        const int iPad = (RS != RS0 && lane % 16 == 0) ? g * RS + RS0 : 4 * RS;
        u32x4 row[NLT];
        auto gload = [&](int n) {
            const __amdgpu_buffer_rsrc_t t = buf_window(tw + (size_t)n * tstride, tbytes);
#pragma unroll
            for (int i = 0; i < NLT; ++i) row[i] = __builtin_amdgcn_raw_buffer_load_b128(t, 1024 * i + 16 * lane, 0, 0);
        };
        // this wave's steps: n_top - pw, n_top - pw - NPROD, ... (the pattern runs through the chunk boundaries)
        if (n_top - pw >= n_bot) gload(n_top - pw);
        // iteration ch (re)fills the slots of buffer ch & 1 for chunk ch; what the chain left there two chunks ago -- the smoothed
        // tiles of chunk ch - 2, written in place over the filtered rows -- goes to HBM first, as whole rows.  Two more
        // iterations drain the last two chunks.
        for (int ch = 0; ch < nch + 2; ++ch) {
            const int buf = ch & 1, n_hi = n_top - ch * FZ_C;
#pragma unroll
            for (int s0 = 0; s0 < FZ_C; s0 += NPROD) {
                const int sl = s0 + pw, n = n_hi - sl, n_old = n + 2 * FZ_C;
                double* const til = sh + TIL + (buf * FZ_C + sl) * TW;
                double* const rec = sh + REC + (buf * FZ_C + sl) * RW;
                if (ch >= 2 && n_old >= n_bot) {           // (uniform)
                    const __amdgpu_buffer_rsrc_t t = buf_window(tw + (size_t)n_old * tstride, tbytes);
                    const u32x4* const src = (const u32x4*)til;
#pragma unroll
                    for (int i = 0; i < NLT; ++i) __builtin_amdgcn_raw_buffer_store_b128(src[64 * i + lane], t, 1024 * i + 16 * lane, 0, 0);
                }
                if (ch < nch && n >= n_bot) {              // (uniform)
                    u32x4* const dst = (u32x4*)til;
#pragma unroll
                    for (int i = 0; i < NLT; ++i) dst[64 * i + lane] = row[i];
                    double v[P], mu;
#pragma unroll
                    for (int k = 0; k < P; ++k) v[k] = til[iV[k]];
                    mu = til[iMu];
                    if (n - NPROD >= n_bot) gload(n - NPROD);
                    double A[P], Sp[P], mp;
#ifdef RK_FZ_NOPROD       // experiment build: the producers only move data (wrong results; the chain wave's own speed)
#pragma unroll
                    for (int i = 0; i < P; ++i) { A[i] = v[i]; Sp[i] = v[i]; }
                    mp = mu;
#else
                    cols_gain_item<P>(Qc, Qr, Rc, h, v, mu, A, Sp, mp);
#endif
#pragma unroll
                    for (int i = 0; i < P; ++i) rec[iOut[i]] = h ? A[i] : Sp[i];
                    rec[iMp] = mp;
                    if constexpr (RS != RS0) rec[iPad] = 0.0;
                }
            }
            lap(t_work);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            lap(t_wait);
        }
        if (dbg && blockIdx.x == 0 && lane == 0) { dbg[2 * wave] = (double)t_work; dbg[2 * wave + 1] = (double)t_wait; }
        return;
    }
    // ================= wave 0: the chain (bwd_mv_tilen_kernel's step, operands out of LDS) =================
    // (the chain is the serial part: it goes first wherever it shares a SIMD with producer waves)
    __builtin_amdgcn_s_setprio(3);
    const int r = lane >> 4, g = (lane >> 2) & 3, c = lane & 3;
    const bool valid = g < n_valid;
    // l*: where this lane's tile elements sit in a record / tile slot (padding lanes: the slots' zero words).  The smoothed
    // tiles go back INTO the tile slot: w* = index in slot 0 + slot * ws* (padding lanes of real units write their exact
    // zeros onto the zero word; lanes of units past the end write into their own unused part of the row; the copies of a mean
    // entry in the columns c > 0 go to the record slot's dump word).
    int lG[NB][NB], lP[NB][NB], lS[NB][NB], lMp[NB], lMf[NB], oS[NB][NB], oM[NB], wS[NB][NB], wM[NB], wsM[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int i = 4 * k + r;
        const bool iv = i < P;
        lMp[k] = iv ? g * RS + 2 * P * P + i : 4 * RS + 1;
        lMf[k] = iv ? g * PP + P * P + i : TW - 1;
        oM[k] = (valid && iv) ? (g * PP + P * P + i) * 8 : TN_OOR;
        const bool wm = valid && iv && c == 0;
        wM[k] = wm ? TIL + g * PP + P * P + i : REC + 4 * RS;
        wsM[k] = wm ? TW : RW;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int jc = 4 * bb + c;
            const bool in = iv && jc < P;
            lG[k][bb] = in ? g * RS + i * P + jc : 4 * RS + 1;
            lP[k][bb] = in ? g * RS + P * P + i * P + jc : 4 * RS + 1;
            lS[k][bb] = in ? g * PP + i * P + jc : TW - 1;
            oS[k][bb] = (valid && in) ? (g * PP + i * P + jc) * 8 : TN_OOR;
            wS[k][bb] = TIL + (valid ? (in ? g * PP + i * P + jc : TW - 1) : g * PP);
        }
    }
    double Ms[NB][NB], ms[NB];
    {
        const __amdgpu_buffer_rsrc_t t = buf_window(tw + (size_t)(n_top + 1) * tstride, tbytes);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            ms[k] = buf_ld(t, oM[k]);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) Ms[k][bb] = buf_ld(t, oS[k][bb]);
        }
    }
    // What the step costs is latency, not work (s_memtime stamps of the first version: 1074 cycles per step at p = 5 with the
    // producers idle, for 20 MFMAs = 320 cycles of issue): (i) the record of step n - 1 is read out of LDS into registers
    // BEFORE the arithmetic of step n (inside a chunk); (ii) the smoothed tiles go back into the LDS tile slot and leave from
    // there in the producers' hands -- with its own global stores the chain waited (vmcnt) every step until the previous
    // step's stores had read their data, 520 of the 1080 cycles (copies of the stored values in other registers did not help); (iii) the chunk barrier waits for LDS
    // only (fz_barrier): nobody in this kernel reads the stores.
    struct Rec { double Gt[NB][NB], Sp[NB][NB], Sf[NB][NB], mp[NB], mf[NB]; };
    auto lds_load = [&](Rec& q, const double* rec, const double* til) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            q.mp[k] = rec[lMp[k]];
            q.mf[k] = til[lMf[k]];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                q.Gt[k][bb] = rec[lG[k][bb]];
                q.Sp[k][bb] = rec[lP[k][bb]];
                q.Sf[k][bb] = til[lS[k][bb]];
            }
        }
    };
    auto step = [&](const Rec& q, int slot) {
        double Dm[NB][NB], dm[NB], V1[NB][NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            dm[k] = ms[k] - q.mp[k];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) Dm[k][bb] = Ms[k][bb] - q.Sp[k][bb];
        }
        bmm_tn0<NB>(Dm, q.Gt, V1);                        // (G D)^T
        bmm_tn<NB>(V1, q.Gt, q.Sf, Ms);                   // G D G^T + Sigma_f      (standard.py:215-216)
        bmv_t<NB>(q.Gt, dm, q.mf, ms);                    // G (m_s - m-) + mu_f    (standard.py:213-214)
#ifndef RK_FZ_NOSTORE      // (experiment build without the hand-over: what it costs the chain)
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            sh[wM[k] + slot * wsM[k]] = ms[k];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) sh[wS[k][bb] + slot * TW] = Ms[k][bb];
        }
#endif
    };
    auto fz_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    for (int ch0 = 0; ch0 < nch; ch0 += 2) {
#pragma unroll
        for (int buf = 0; buf < 2; ++buf) {
            const int ch = ch0 + buf;
            if (ch < nch) {                                // (uniform)
                lap(t_work);
                fz_barrier();
                lap(t_wait);
                const int n_hi = n_top - ch * FZ_C;
                Rec q[2];
                lds_load(q[0], sh + REC + (buf * FZ_C) * RW, sh + TIL + (buf * FZ_C) * TW);
#pragma unroll
                for (int sl = 0; sl < FZ_C; ++sl) {
                    if (n_hi - sl >= n_bot) {
                        if (sl + 1 < FZ_C)                 // (slots past n_bot hold stale records: loaded, never used)
                            lds_load(q[(sl + 1) & 1], sh + REC + (buf * FZ_C + sl + 1) * RW, sh + TIL + (buf * FZ_C + sl + 1) * TW);
                        step(q[sl & 1], buf * FZ_C + sl);
                    }
                }
            }
        }
    }
    fz_barrier();                                          // (the producers' two draining iterations)
    fz_barrier();
    lap(t_work);
    if (dbg && blockIdx.x == 0 && lane == 0) { dbg[0] = (double)t_work; dbg[1] = (double)t_wait; }
}

template <int NB>
__global__ void __launch_bounds__(64) bwd_sim_tilen_kernel(SolveArgs a, const double* __restrict__ tiles, const double* __restrict__ ws, int P) {
    const int D = a.D, n_units = a.B * D, PP = P * P + P, RS = tilen_rs(P, true);
    const int lane = threadIdx.x, r = lane >> 4, g = (lane >> 2) & 3, c = lane & 3;
    const int tau_raw = blockIdx.x * 4 + g;
    const bool valid = tau_raw < n_units;
    const int tau = valid ? tau_raw : n_units - 1;
    const int b = tau / D, blk = tau - b * D;
    int oG[NB][NB], oMp[NB], oW[NB];
    bool st[NB];
    size_t ox[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int i = 4 * k + r;
        const bool iv = valid && i < P;
        st[k] = iv && c == 0;
        oMp[k] = iv ? (g * RS + P * P + i) * 8 : TN_OOR;
        oW[k] = iv ? (g * RS + P * P + P + i) * 8 : TN_OOR;
        ox[k] = ((size_t)blk * P + (i < P ? i : 0)) * (size_t)a.B + b;          // x_state (N+1, d, p, B)
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int j = 4 * bb + c;
            oG[k][bb] = (iv && j < P) ? (g * RS + i * P + j) * 8 : TN_OOR;
        }
    }
    const size_t wstride = (size_t)n_units * RS, xstride = (size_t)D * P * a.B;
    const double* const ww = ws + (size_t)blockIdx.x * 4 * RS;
    const int wbytes = 4 * RS * 8;
    struct Rec { double Gt[NB][NB], mp[NB], w[NB]; };
    auto load = [&](int n, Rec& q) {
        const __amdgpu_buffer_rsrc_t w = buf_window(ww + (size_t)n * wstride, wbytes);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            q.mp[k] = buf_ld(w, oMp[k]);
            q.w[k] = buf_ld(w, oW[k]);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) q.Gt[k][bb] = buf_ld(w, oG[k][bb]);
        }
    };
    double x[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) x[k] = 0.0;
    auto step = [&](int n, const Rec& q) {
        double dx[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) dx[k] = x[k] - q.mp[k];
        bmv_t<NB>(q.Gt, dx, q.w, x);                     // x_n = G (x_{n+1} - mu-) + mu_f + L z   (standard.py:251-254, solve.py:179)
        double* xo = a.x + (size_t)n * xstride;
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (st[k]) xo[ox[k]] = x[k];
    };
    int n = a.N;                                          // steps n = N .. 1 (n = N: the terminal draw), ring as above
    Rec q[TN_RING];
#pragma unroll
    for (int s = 0; s < TN_RING; ++s) load(n - s >= 1 ? n - s : 1, q[s]);
    while (n >= TN_RING) {
#pragma unroll
        for (int s = 0; s < TN_RING; ++s) {
            step(n - s, q[s]);
            const int nn = n - s - TN_RING;
            load(nn >= 1 ? nn : 1, q[s]);
        }
        n -= TN_RING;
    }
#pragma unroll
    for (int s = 0; s < TN_RING; ++s)
        if (s < n) step(n - s, q[s]);
    // x[0] = ode_init exactly (solve.py:196-204): the mean of tile time 0
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (st[k]) a.x[ox[k]] = tiles[(size_t)tau * PP + P * P + 4 * k + r];
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
static inline int tilen_nb(int p) { return p <= 4 ? 1 : 2; }

template <class RHS, int NB>
static int launch_fwd_tilen_nb(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles) {
    constexpr int NW = TileWaves<RHS::D>::value;
    const dim3 grid(NW == 1 ? div_up(a.B * RHS::D, Tpw<RHS::D>::value) : a.B), block(64 * NW);
    const int P = c->n_bstate;
    launch_placement_primer(h, grid, block);           // (common.hpp: exact one-wave-per-SIMD placement behind any kernel)
    LaunchTimer t(h, "fwd_tilen_kernel");
    switch (c->interrogate) {
        case RK_INTERROGATE_KRAMER:
            hipLaunchKernelGGL((fwd_tilen_kernel<RHS, RK_INTERROGATE_KRAMER, NB>), grid, block, 0, h->stream, a, tiles, P); break;
        case RK_INTERROGATE_SCHOBER:
            hipLaunchKernelGGL((fwd_tilen_kernel<RHS, RK_INTERROGATE_SCHOBER, NB>), grid, block, 0, h->stream, a, tiles, P); break;
        case RK_INTERROGATE_RODEO:
            hipLaunchKernelGGL((fwd_tilen_kernel<RHS, RK_INTERROGATE_RODEO, NB>), grid, block, 0, h->stream, a, tiles, P); break;
        case RK_INTERROGATE_CHKREBTII:
            hipLaunchKernelGGL((fwd_tilen_kernel<RHS, RK_INTERROGATE_CHKREBTII, NB>), grid, block, 0, h->stream, a, tiles, P); break;
        default:
            set_error("tile path: interrogate id %d not supported", c->interrogate);
            return RK_ERR_UNSUPPORTED;
    }
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}
template <class RHS>
static int launch_fwd_tilen(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles) {
    return tilen_nb(c->n_bstate) == 1 ? launch_fwd_tilen_nb<RHS, 1>(h, c, a, tiles) : launch_fwd_tilen_nb<RHS, 2>(h, c, a, tiles);
}

bool is_user_rhs(int rhs_id);
bool user_tile_available(const rk_solve_cfg* c, int which);
int user_forward_tile(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int which);

bool tilen_supported(const rk_solve_cfg* c, int mode) {
    if (c->flags & (RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR)) return false;
    if (c->kalman_type != RK_KALMAN_STANDARD || c->n_bmeas != 1) return false;
    if (c->n_bstate < 4 || c->n_bstate > 8) return false;
    if (c->interrogate < RK_INTERROGATE_RODEO || c->interrogate > RK_INTERROGATE_CHKREBTII) return false;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) return c->n_block == 2;
    if (c->rhs_id == RK_RHS_LORENZ63) return c->n_block == 3;
    if (c->rhs_id == RK_RHS_HIGHER_ORDER) return c->n_block == 1;
    if (is_user_rhs(c->rhs_id)) return user_tile_available(c, 5);       // hiprtc build of fwd_tilen_kernel (rhs_jit.hip)
    return false;
}

size_t tilen_tile_doubles(const rk_solve_cfg* c) {
    const size_t pp = (size_t)c->n_bstate * (c->n_bstate + 1);
    return (size_t)(c->n_steps + 1) * c->n_block * (size_t)c->n_traj * pp;
}

size_t tilen_ws_doubles(const rk_solve_cfg* c, int mode) {
    if (mode == RK_MODE_FILTER) return 0;
    return (size_t)(c->n_steps + 1) * c->n_block * (size_t)c->n_traj * tilen_rs(c->n_bstate, mode == RK_MODE_SIM);
}

int tilen_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, double* ws, size_t ws_bytes, int mode) {
    const int P = c->n_bstate, n_units = a.B * a.D;
    RK_REQUIRE((size_t)n_units * (size_t)P * (P + 1) < 0x7fffffffull && (size_t)n_units * tilen_rs(P, mode == RK_MODE_SIM) < 0x7fffffffull,
               RK_ERR_UNSUPPORTED, "blocked tile path: n_traj * n_block too large for 32-bit record offsets");
    if (mode != RK_MODE_FILTER) {
        const size_t need = tilen_ws_doubles(c, mode) * sizeof(double);
        RK_REQUIRE(need == 0 || (ws && ws_bytes >= need), RK_ERR_INVALID,
                   "blocked tile path: out->workspace_bytes = %zu, this call needs %zu (rk_solve_workspace_bytes)", ws_bytes, need);
    }
    int rc;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) rc = launch_fwd_tilen<FitzHughNagumo>(h, c, a, tiles);
    else if (c->rhs_id == RK_RHS_LORENZ63) rc = launch_fwd_tilen<Lorenz63>(h, c, a, tiles);
    else if (is_user_rhs(c->rhs_id)) rc = user_forward_tile(h, c, a, tiles, 5);
    else rc = launch_fwd_tilen<HigherOrder>(h, c, a, tiles);
    if (rc || mode == RK_MODE_FILTER) return rc;
    const bool sim = mode == RK_MODE_SIM;
    const int bps = div_up(n_units, 64);
    auto launch_gain = [&](hipStream_t st, int n_first, int n_count) -> int {
        RK_REQUIRE((size_t)bps * (size_t)n_count < 0x7fffffffull, RK_ERR_UNSUPPORTED,
                   "blocked tile path: n_steps * n_traj * n_block too large for one launch");
        const dim3 grid((unsigned)(bps * n_count)), block(64);
#define RK_GAIN(P_)                                                                                     \
    case P_:                                                                                            \
        if (sim) hipLaunchKernelGGL((tilen_gain_kernel<P_, true>), grid, block, 0, st, a, tiles, ws, bps, n_first);   \
        else hipLaunchKernelGGL((tilen_gain_kernel<P_, false>), grid, block, 0, st, a, tiles, ws, bps, n_first);      \
        break;
        switch (P) { RK_GAIN(4) RK_GAIN(5) RK_GAIN(6) RK_GAIN(7) RK_GAIN(8) }
#undef RK_GAIN
        RK_HIP(hipGetLastError());
        return RK_OK;
    };
    const dim3 cgrid(div_up(n_units, 4)), cblock(64);
    if (sim) {
        {
            LaunchTimer t(h, "tilen_gain_kernel<sim>");
            rc = launch_gain(h->stream, 1, a.N);
            t.stop();
            if (rc) return rc;
        }
        LaunchTimer t(h, "bwd_sim_tilen_kernel");
        if (tilen_nb(P) == 1) hipLaunchKernelGGL((bwd_sim_tilen_kernel<1>), cgrid, cblock, 0, h->stream, a, tiles, ws, P);
        else hipLaunchKernelGGL((bwd_sim_tilen_kernel<2>), cgrid, cblock, 0, h->stream, a, tiles, ws, P);
        t.stop();
        RK_HIP(hipGetLastError());
        return RK_OK;
    }
    if (a.N < 2) return RK_OK;
    const int steps = a.N - 1;
    // measured (FN headline shape, ms): element-per-lane chain 1.80 / 2.33 / 3.41 / 4.09 at n_bstate = 5 / 6 / 7 / 8, coalesced
    // rows through LDS 2.07 / 2.30 / 2.82 / 3.72 -- the LDS traffic of the staging costs what the address path saves until the
    // tiles are nearly full; RK_TILEN_CHAIN = rows | scattered overrides the choice
    static const char* const force = getenv("RK_TILEN_CHAIN");
    const bool scattered = force ? force[0] == 's' : P <= 6;
    auto launch_chain = [&](int n_top, int n_bot) -> int {
        if (scattered) {
            if (tilen_nb(P) == 1) hipLaunchKernelGGL((bwd_mv_tilen_kernel<1>), cgrid, cblock, 0, h->stream, a, tiles, ws, P, n_top, n_bot);
            else hipLaunchKernelGGL((bwd_mv_tilen_kernel<2>), cgrid, cblock, 0, h->stream, a, tiles, ws, P, n_top, n_bot);
        } else {
#define RK_ROWS(P_) case P_: hipLaunchKernelGGL((bwd_mv_tilen_rows_kernel<P_>), cgrid, cblock, 0, h->stream, a, tiles, ws, n_top, n_bot); break;
            switch (P) { RK_ROWS(4) RK_ROWS(5) RK_ROWS(6) RK_ROWS(7) RK_ROWS(8) }
#undef RK_ROWS
        }
        RK_HIP(hipGetLastError());
        return RK_OK;
    };
    // (Tried: the chain and the gain items of the next time chunk overlapped on two streams -- the chain is latency-bound
    // on one wave per SIMD -- measured 2.84 ms against 3.02 ms sequential at n_deriv = 5 and slower at 6: the gain grid's
    // workgroups take every free slot and the two kernels mostly alternate.  A fused producer / consumer workgroup like the
    // p = 3 kernels: 2.76 ms at n_deriv = 5, 5.1 ms at 6 -- its producers need > 256 VGPRs, so one workgroup per CU and
    // two rounds; capped at 256 VGPRs they spill: 4.9 ms.  Both removed.)
    // One kernel (bwd_mv_tilen_fused_kernel) at n_bstate >= 5; RK_TILEN_BWD = split selects the two-kernel form below (kept as the
    // parity test's other leg and for n_bstate = 4 with interrogate_chkrebtii, whose solve_mv has no hand-trimmed kernel).
    const char* const bforce = getenv("RK_TILEN_BWD");
    if (P >= 5 && !(bforce && bforce[0] == 's')) {
        // producer waves per chain wave: RK_TILEN_PROD = 2 | 4 (measured below)
        const char* const pforce = getenv("RK_TILEN_PROD");
        const int nprod = pforce ? atoi(pforce) : 4;
        double* const dbg = getenv("RK_TILEN_STAMPS") ? ws : nullptr;        // (the two-kernel form's workspace is free here)
        const dim3 fgrid(div_up(n_units, 4)), fblock(64 * (1 + (nprod == 2 ? 2 : 4)));
        LaunchTimer t(h, "bwd_mv_tilen_fused_kernel");
#define RK_FUSED(P_)                                                                                                         \
    case P_:                                                                                                                 \
        if (nprod == 2) hipLaunchKernelGGL((bwd_mv_tilen_fused_kernel<P_, 2>), fgrid, fblock, 0, h->stream, a, tiles, a.N - 1, 1, dbg);   \
        else hipLaunchKernelGGL((bwd_mv_tilen_fused_kernel<P_, 4>), fgrid, fblock, 0, h->stream, a, tiles, a.N - 1, 1, dbg);      \
        break;
        switch (P) { RK_FUSED(5) RK_FUSED(6) RK_FUSED(7) RK_FUSED(8) }
#undef RK_FUSED
        t.stop();
        RK_HIP(hipGetLastError());
        return RK_OK;
    }
    // phase 1 of solve_mv: one 16-lane row per item (tilen_gain_cols_kernel).  Measured, headline shape, ms at n_bstate = 5 .. 8:
    // one lane per item 1.27 / 2.20 / 4.32 / 9.00, rows of lanes 1.18 / 1.71 / 2.33 / 2.64 (profiles/r03_nderiv_times_v2.jsonl);
    // RK_TILEN_GAIN = lanes | cols overrides the choice (both forms give the same bits: tests/test_gpu_tilen.py)
    const char* const gforce = getenv("RK_TILEN_GAIN");      // (read per call: the parity test switches it)
    const bool cols = gforce ? gforce[0] == 'c' : true;
    if (cols) {
        constexpr int CHUNK = 32;                          // time steps per wave: the block constants are loaded once per chunk
        const dim3 ggrid(div_up(n_units, 4), div_up(steps, CHUNK));
        LaunchTimer t(h, "tilen_gain_cols_kernel");
#define RK_COLS(P_) case P_: hipLaunchKernelGGL((tilen_gain_cols_kernel<P_>), ggrid, cblock, 0, h->stream, a, tiles, ws, 1, steps, CHUNK); break;
        switch (P) { RK_COLS(4) RK_COLS(5) RK_COLS(6) RK_COLS(7) RK_COLS(8) }
#undef RK_COLS
        t.stop();
        RK_HIP(hipGetLastError());
    } else {
        LaunchTimer t(h, "tilen_gain_kernel");
        rc = launch_gain(h->stream, 1, steps);
        t.stop();
        if (rc) return rc;
    }
    LaunchTimer t(h, "bwd_mv_tilen_kernel");
    rc = launch_chain(a.N - 1, 1);
    t.stop();
    return rc;
}

}  // namespace rk
