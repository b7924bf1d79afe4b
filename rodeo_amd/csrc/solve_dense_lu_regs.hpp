// Register-resident forward elimination of the dense LU solve  X = A^{-1} Bm  (src/rodeo/utils.py:105-119, called by
// standard._smooth, src/rodeo/kalmantv/standard.py:175-176): n, nr <= 160.  Included by solve_dense.hip.
//
// What it replaces: the right-looking panel loop of wg_lu_solve kept [A | Bm] in global memory and read and wrote the
// whole trailing matrix once per 16-column panel -- 4.5 of the 8 MB a backward step of config 5 moved
// (profiles/r03_c5_pmc_traffic_dense.json), at the per-CU memory rate.  Here the 160 x 320 matrix [A | Bm] is loaded ONCE
// into the registers of the workgroup's eight waves (200 tiles of 16 x 16 in the D layout of v_mfma_f64_16x16x4, 25 per
// wave = 200 VGPRs) and leaves as U (upper triangle, LAPACK row order) and Y = L^{-1} P Bm, written once, panel by panel.
//
// Pivoting WITHOUT row movement.  A register tile cannot be indexed by a run-time row, so rows never move: every row
// carries its LAPACK position (what getrf's interchanges would have made of it -- the pivot search is "largest |a|, smallest
// position", so the pivots are LAPACK's, ties included), a row that has been a pivot is retired, and the rank-16 update
// runs over all ten row tiles with the multipliers of retired rows set to zero (C - 0 * U = C exactly).  Per panel k:
//   P1  the two waves that own column tile k put it into the LDS panel buffer (double-buffered over k);
//   P2  wave 0 factors the panel in registers (three rows per lane, all 16 columns unrolled; rows in place): it leaves -L for the
//       rows still active (zeros elsewhere) in the panel buffer, the 16 x 16 block L11 \ U11 in pivot order, and the table
//       "row -> pivot index of this panel";
//   P3  every wave drops the 16 pivot rows of its live column tiles into the LDS strip (16 x 320, pivot order);
//   P4  one thread per column: U12 = L11^{-1} A12, Y_k = L11^{-1} B_k in the strip (trsm16, the substitution of the old path);
//   P5  the strip goes to memory (rows 16k.. of U and of Y: LAPACK order), and every wave updates its live tiles:
//       four MFMAs per tile, A fragments from the LDS panel, B fragments from the strip.
// Operations and their order are those of the old path (same panel arithmetic, same trsm16, acc = C, then k = 0..15 in four
// MFMAs), so U and Y have the same bits; wg_tri_solve_regs then runs the back substitution unchanged.
// Wave 0 needs 96 registers for its panel rows on top of its 25 tiles: it parks them in LDS around P2 (left to the register
// allocator they went to scratch, with reloads inside the MFMA groups).
#pragma once

namespace rk {

constexpr int RL_T = 10;                                   // tiles per dimension
constexpr int RL_N = 16 * RL_T;                            // 160
constexpr int RL_PB = RL_N * LU_LD;                        // doubles per panel buffer
constexpr int RL_D11 = 2 * RL_PB;                          // 16 x 16 block L11 \ U11 (row stride LU_LD)
constexpr int RL_USP = 336;                                // strip row stride: 16 mod 32 doubles (B fragments of two k rows on disjoint banks)
constexpr int RL_STRIP = RL_D11 + LU_NB * LU_LD;
constexpr int RL_PARK = RL_STRIP + LU_NB * RL_USP;         // wave 0's parked tiles
constexpr int RL_PARK_TILES = 25;
constexpr int RL_END = RL_PARK + RL_PARK_TILES * 256;
static_assert(RL_END <= LDS_DOUBLES, "register-resident LU: LDS layout exceeds the buffer");
__shared__ int g_rl_pos[RL_N];
// (the panel loop over memory and this path never run at the same time: its row list g_cur[LU_MAXN] serves here as
//  "row -> pivot index of the current panel" and as the list of the rows still active)
#define g_rl_pidx (g_cur + RL_N)
#define g_rl_list g_cur
static_assert(LU_MAXN >= 2 * RL_N, "g_cur is reused by the register-resident LU");

// (rl_readlane_f64, wave_max_u32_fused and the column step rl_panel_col live in solve_dense.hip: the panel loop over memory
//  uses the same column step)
template <int RS, int J>
__device__ __forceinline__ void rl_panel_cols(double (&a)[RS][LU_NB], int (&pos)[RS], int k0, int nb, int n) {
    if constexpr (J < LU_NB) {
        if (J < nb) rl_panel_col<RS, J>(a, pos, k0 + J, n);
        rl_panel_cols<RS, J + 1>(a, pos, k0, nb, n);
    }
}


template <int RS>
__device__ __forceinline__ void rl_panel_body(double* pb, int k0, int nb, int n, int nact, int lane) {
    double a[RS][LU_NB];
    int pos[RS], row[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        const int i = lane + 64 * s;
        row[s] = i < nact ? g_rl_list[i] : -1;
        pos[s] = row[s] >= 0 ? g_rl_pos[row[s]] : 0x7fffffff;
#pragma unroll
        for (int c = 0; c < LU_NB; ++c) a[s][c] = row[s] >= 0 ? pb[row[s] * LU_LD + c] : 0.0;
    }
    wave_lds_sync();                                            // every row is in registers before the buffer is rewritten
    rl_panel_cols<RS, 0>(a, pos, k0, nb, n);
    double* const d11 = g_lds + RL_D11;
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        if (row[s] >= 0) {
            const int r = row[s], pp = pos[s];
            const bool live = pp >= k0 + nb;                    // still to be eliminated: its multipliers take part in the update
#pragma unroll
            for (int c = 0; c < LU_NB; ++c) pb[r * LU_LD + c] = live ? -a[s][c] : 0.0;
            if (!live) {
#pragma unroll
                for (int c = 0; c < LU_NB; ++c) d11[(pp - k0) * LU_LD + c] = a[s][c];
            }
            g_rl_pidx[r] = live ? -1 : pp - k0;
            g_rl_pos[r] = pp;
        }
    }
    if (lane < LU_NB && lane >= nb) {                           // (a partial last panel: the rows of the block past nb)
#pragma unroll
        for (int c = 0; c < LU_NB; ++c) d11[lane * LU_LD + c] = 0.0;
    }
}

// wave 0: factor panel k (columns k0 .. k0 + nb - 1 of all n rows, in the LDS buffer at pb_off).  A real call: inlined into
// wg_lu_fwd_regs its 100 registers came on top of that function's 200 tile registers, and although the caller parks its
// tiles in LDS around it the allocator sent them to scratch, with reloads inside the MFMA groups.
__device__ __noinline__ void rl_panel(int pb_off_, int k0_, int nb_, int n_) {
    double* const pb = g_lds + uni(pb_off_);
    const int k0 = uni(k0_), nb = uni(nb_), n = uni(n_), lane = threadIdx.x & 63;
    // the rows retired by earlier panels: zero multipliers, no pivot index; the others in compact order to the lanes
    int base = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int r = lane + 64 * s;
        const bool in = r < RL_N;
        const bool act = in && r < n && g_rl_pos[in ? r : 0] >= k0;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(act);
        const int idx = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (act) g_rl_list[idx] = r;
        else if (in) {
#pragma unroll
            for (int c = 0; c < LU_NB; ++c) pb[r * LU_LD + c] = 0.0;
            g_rl_pidx[r] = -1;
        }
        base += (int)__builtin_popcountll(bal);
    }
    wave_lds_sync();
    const int nact = base;                                      // = n - k0
    if (nact > 128) rl_panel_body<3>(pb, k0, nb, n, nact, lane);
    else if (nact > 64) rl_panel_body<2>(pb, k0, nb, n, nact, lane);
    else rl_panel_body<1>(pb, k0, nb, n, nact, lane);
}

// Forward elimination.  On return A holds U (upper triangle and the diagonal blocks, LAPACK row order) and Bm holds
// Y = L^{-1} P Bm -- what the panel loop of wg_lu_solve leaves for the back substitution.
__device__ __noinline__ void wg_lu_fwd_regs(double* A_, int lda_, double* Bm_, int ldb_, int n_, int nr_, double* ws_end_ = nullptr) {
    auto* const A = uni_g(A_);
    auto* const Bm = uni_g(Bm_);
    auto* const ws_end = uni_g(ws_end_);
    (void)ws_end;
    RK_STAMP_DECL(ws_end);
    const int lda = uni(lda_), ldb = uni(ldb_), n = uni(n_), nr = uni(nr_);
    double* const lds = g_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = uni((int)(tid >> 6)), lo = lane & 15, hi = lane >> 4;
    const int g = wave & 3, h = wave >> 2;                      // column group, row half (waves w, w + 4 share a SIMD)
    // column tiles of group g in [A | Bm] (A: 0..9, Bm: 10..19): A tiles g, g + 4, (g + 8), the rest from Bm -- five each,
    // and the A tiles still live at panel k are spread evenly over the groups
    int ct[5];
    ct[0] = g; ct[1] = g + 4;
    ct[2] = g < 2 ? g + 8 : g + 10;
    ct[3] = g < 2 ? g + 10 : g + 14;
    ct[4] = g < 2 ? g + 14 : g + 16;
    const int rt0 = 5 * h;                                      // first row tile of this wave
    d4 t[5][5];
#pragma unroll
    for (int ci = 0; ci < 5; ++ci) {
        const bool inA = ct[ci] < RL_T;
        cgd* const M = inA ? (cgd*)A : (cgd*)Bm;
        const int ld = inA ? lda : ldb, nc = inA ? n : nr;
        const int col = min(16 * (inA ? ct[ci] : ct[ci] - RL_T) + lo, nc - 1);
#pragma unroll
        for (int ri = 0; ri < 5; ++ri)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                t[ri][ci][v] = M[min(16 * (rt0 + ri) + 4 * v + hi, n - 1) * ld + col];       // clamped, unmasked: all loads in flight
    }
    if ((n & 15) | (nr & 15) | (n < RL_N) | (nr < RL_N)) {      // rows / columns past the end: exact zeros
#pragma unroll
        for (int ci = 0; ci < 5; ++ci) {
            const bool inA = ct[ci] < RL_T;
            const int nc = inA ? n : nr;
            const bool cok = 16 * (inA ? ct[ci] : ct[ci] - RL_T) + lo < nc;
#pragma unroll
            for (int ri = 0; ri < 5; ++ri)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (!cok || 16 * (rt0 + ri) + 4 * v + hi >= n) t[ri][ci][v] = 0.0;
        }
    }
    for (int r = tid; r < RL_N; r += DT) g_rl_pos[r] = r;
    RK_STAMP(11);
    const int npan = (n + 15) >> 4;
    int us_off = RL_STRIP;
    asm volatile("" : "+v"(us_off));                         // (opaque, for the same reason as the park base below)
    double* const us = g_lds + us_off;
    // column tile kk of A (slot co of its group's list) into panel buffer kk & 1
    auto put_panel = [&](int kk) {
        double* const pbn = lds + (kk & 1) * RL_PB;
        const int co = kk >> 2;
#pragma unroll
        for (int ri = 0; ri < 5; ++ri)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const double x = co == 0 ? t[ri][0][v] : (co == 1 ? t[ri][1][v] : t[ri][2][v]);
                pbn[(16 * (rt0 + ri) + 4 * v + hi) * LU_LD + lo] = x;
            }
    };
    // wave 0: factor panel kk; its tiles wait in LDS meanwhile (its panel rows need their registers)
    auto factor_panel = [&](int kk) {
        // (an opaque base: folded into the accesses' constants the offsets pass the 64 KB immediate range, and hipcc then
        //  keeps one address REGISTER per access -- a hundred of them -- live across the loop)
        int park_off = RL_PARK + lane;                          // (an opaque INDEX: an opaque pointer would lose its LDS address space)
        asm volatile("" : "+v"(park_off));
        double* const park = g_lds + park_off;
#pragma unroll
        for (int ri = 0; ri < 5; ++ri)
#pragma unroll
            for (int ci = 0; ci < 5; ++ci)
#pragma unroll
                for (int v = 0; v < 4; ++v) park[((ri * 5 + ci) * 4 + v) * 64] = t[ri][ci][v];
        __builtin_amdgcn_s_setprio(3);                          // (the chain everything waits for: ahead of its SIMD partner's MFMAs)
        rl_panel((kk & 1) * RL_PB, 16 * kk, min(16, n - 16 * kk), n);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int ri = 0; ri < 5; ++ri)
#pragma unroll
            for (int ci = 0; ci < 5; ++ci)
#pragma unroll
                for (int v = 0; v < 4; ++v) t[ri][ci][v] = park[((ri * 5 + ci) * 4 + v) * 64];
    };
    // rank-16 update of this wave's five tiles of column slot CI with panel k (buffer pb): four MFMAs per tile
    auto update_slot = [&](auto CI, const double* pb) {
        constexpr int ci = decltype(CI)::value;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const double b = us[(4 * kq + hi) * RL_USP + 16 * ct[ci] + lo];
#pragma unroll
            for (int ri = 0; ri < 5; ++ri) {
                const double av = pb[(16 * (rt0 + ri) + lo) * LU_LD + 4 * kq + hi];
                t[ri][ci] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, t[ri][ci], 0, 0, 0);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
    if (g == 0) put_panel(0);
    __syncthreads();
    if (wave == 0) factor_panel(0);
    __syncthreads();
    RK_STAMP(1);
    for (int k = 0; k < npan; ++k) {
        const int k0 = 16 * k, nb = min(16, n - k0);
        const double* const pb = lds + (k & 1) * RL_PB;
        // ---- P3: the panel's pivot rows of the live column tiles into the strip (pivot order) ----
        {
            int pi[5][4];
#pragma unroll
            for (int ri = 0; ri < 5; ++ri)
#pragma unroll
                for (int v = 0; v < 4; ++v) pi[ri][v] = g_rl_pidx[16 * (rt0 + ri) + 4 * v + hi];
#pragma unroll
            for (int ri = 0; ri < 5; ++ri)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bool mine = pi[ri][v] >= 0;
                    if (__builtin_amdgcn_ballot_w64(mine) != 0) {
#pragma unroll
                        for (int ci = 0; ci < 5; ++ci)
                            if (ct[ci] > k) {
                                if (mine) us[pi[ri][v] * RL_USP + 16 * ct[ci] + lo] = t[ri][ci][v];
                            }
                    }
                }
        }
        __syncthreads();
        RK_STAMP(2);
        // ---- P4: U12 = L11^{-1} A12, Y_k = L11^{-1} B_k, one thread per column ----
        {
            const int col = 16 * (k + 1) + tid;
            if (col < 2 * RL_N) {
                double x[LU_NB];
#pragma unroll
                for (int j = 0; j < LU_NB; ++j) x[j] = j < nb ? us[j * RL_USP + col] : 0.0;
                trsm16<true, true>(x, lds + RL_D11, nb, nullptr);
#pragma unroll
                for (int j = 0; j < LU_NB; ++j) us[j * RL_USP + col] = x[j];
            }
        }
        __syncthreads();
        RK_STAMP(3);
        // ---- P5: rows k0 .. of U and Y to memory; rank-16 update of the live tiles.  Look-ahead: the owners of column
        // tile k + 1 update it first and hand it to wave 0, which factors panel k + 1 while the other waves update the rest
        // (the panel is a dependent chain of 16 pivot searches on one wave: nothing else can shorten it) ----
        {
            const int j = tid >> 5;                             // strip row
            if (j < nb) {
                for (int cc = (tid & 31) + 16 * (k + 1); cc < 2 * RL_N; cc += 32) {
                    const double x = us[j * RL_USP + cc];
                    if (cc < RL_N) { if (cc < n) A[(k0 + j) * lda + cc] = x; }
                    else if (cc - RL_N < nr) Bm[(k0 + j) * ldb + cc - RL_N] = x;
                }
            }
            if (tid < LU_NB * LU_NB) {                          // the diagonal block (U11 in its upper triangle)
                const int r = tid >> 4, c = tid & 15;
                if (r < nb && c < nb) A[(k0 + r) * lda + k0 + c] = lds[RL_D11 + r * LU_LD + c];
            }
        }
        if (k + 1 < npan) {                                     // (after the last panel no row is left to update)
            const int kn = k + 1, con = kn >> 2;
            const bool own_next = g == (kn & 3);
            if (own_next) {
                if (con == 0) update_slot(I0{}, pb);
                else if (con == 1) update_slot(I1{}, pb);
                else update_slot(I2{}, pb);
                put_panel(kn);
            }
            __syncthreads();                                    // (D11 / pidx of panel k have been consumed: P4, P3)
            if (wave == 0) factor_panel(kn);
            const int skip = own_next ? con : -1;
            if (ct[0] > k && skip != 0) update_slot(I0{}, pb);
            if (ct[1] > k && skip != 1) update_slot(I1{}, pb);
            if (ct[2] > k && skip != 2) update_slot(I2{}, pb);
            if (ct[3] > k) update_slot(I3{}, pb);
            if (ct[4] > k) update_slot(I4{}, pb);
            __syncthreads();
        }
        RK_STAMP(4);
    }
    __syncthreads();
}

}  // namespace rk
