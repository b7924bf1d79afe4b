// Dense square-root filter / smoother / sampler: kalman_type = "square-root" on the non-block path (one dense block of
// n_vars * n_deriv states, prior.indep_init), i.e. src/rodeo/kalmantv/square_root.py:30-261 with src/rodeo/utils.py:10-24
// (add_sqrt = R^T of the QR of the stacked transposed factors) inside the time loops of src/rodeo/solve.py:31-122,
// 137-205, 257-301.  Included by solve_dense.hip (same translation unit: it shares the workgroup's LDS, wg_gemm,
// lu_rank_update and the substitution kernels of the covariance-form path).
//
// Why it exists: in covariance form the reference's recursion is numerically dead on BASELINE config 5 (stiff linear
// ODE, exact measurement: |x - expm(A) x0| = 6e8 at t = 1 on the oracle and on the device alike); the square-root form
// of the same reference reaches 1.8e-8.
//
// Representation: every factor is kept as the reference keeps it -- LOWER triangular L, row-major, in the var arrays
// (solve.py returns the factors in this mode).  A QR only ever produces R (never Q): add_sqrt(A, B) = R^T.
//
// Building blocks (all workgroup-wide, 512 threads, operands in global memory / L2):
//   wg_qr_r        blocked Householder QR of a stacked M x n matrix, R only: 16-column panels in compact WY form
//                  (H_1 ... H_16 = I - V T V^T).  Per panel: (i) the panel is factored with the columns spread over
//                  lanes -- lane (slab, c) of the 32 slabs of 16 lanes holds column c of the panel's 16 triangle rows
//                  (replicated in every slab) and of its slab's share of the rows below, so a dot product of the pivot
//                  column with all 16 columns is ONE value per lane, reduced over the slabs by two shuffles and one LDS
//                  exchange (one barrier per column); LAPACK dlarfg's beta / tau / scaling; (ii) V^T V on
//                  v_mfma_f64_16x16x4 (both fragments are the same LDS read), T by 16 lanes (dlarft's forward
//                  recurrence) while the other waves already run (iii) W = V^T A2 on MFMA with V from the LDS panel and
//                  A2 streamed from memory; (iv) W <- T^T W in LDS; (v) A2 <- A2 - V W = the LU path's rank-16 update.
//   wg_tri_solve   X = E^{-1} B for a triangular E given as a row-major triangle or its transpose, both directions,
//                  the right-hand side resident in registers (the back substitution of the LU path, generalised).
//   wg_transpose   dst = src^T through LDS bands (optionally only the upper triangle of src: L = R^T).
// LAW OF THE SAMPLER (solve_sim in this mode): draws are x = mean + L z, i.e. N(mean, L L^T), L = the conditional factor of
// square_root.smooth_sim.  The reference passes that factor to jax.random.multivariate_normal(method="svd") in the
// COVARIANCE slot (src/rodeo/solve.py:179,182-186 with square_root.py:259), i.e. samples N(mean, L): not reproduced (not a law
// when L is not symmetric PSD; MIGRATION.md).  tests/test_gpu_solver.py::test_square_root_sim_law_is_L_Lt pins it by moments.
#pragma once

namespace rk {

#ifndef RK_STAMP_TRI
#define RK_STAMP_TRI 0
#endif
constexpr int QR_RT = 30;                               // panel rows below the triangle per lane: panels of up to 496 rows
constexpr int QR_MAXM = 16 + 16 * QR_RT;

__device__ __forceinline__ double readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ double bperm_f64(double x, int byte_addr) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(x)),
                            __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(x)));
}

// ---------------------------------------------------------------------------------------------------------------
// dst (rows x cols, ldd) = src^T, src (cols x rows, lds_).  tri != 0: src is upper triangular (square), only src[j][i]
// with j <= i is read and dst gets exact zeros above its diagonal (L = R^T).  Bands of 16 dst rows through LDS
// (buf[j][ti], row stride 17): the loads read 128-byte pieces of src rows, the stores write whole dst rows.
// ---------------------------------------------------------------------------------------------------------------
__device__ __noinline__ void wg_transpose(double* dst_, int ldd_, const double* src_, int lds__, int rows_, int cols_, int tri_) {
    auto* const dst = uni_g(dst_);
    auto* const src = uni_g(src_);
    const int ldd = uni(ldd_), lds_ = uni(lds__), rows = uni(rows_), cols = uni(cols_), tri = uni(tri_);
    double* const buf = g_lds;
    const int cmax = LDS_DOUBLES / 17;                  // columns per pass
    for (int j0 = 0; j0 < cols; j0 += cmax) {
        const int jn = min(cmax, cols - j0);
        for (int i0 = 0; i0 < rows; i0 += 16) {
            const int in = min(16, rows - i0);
            const int jhi = tri ? min(jn, i0 + 16 - j0) : jn;          // src rows that can be non-zero in this band
            __syncthreads();
            for (int e = threadIdx.x; e < jhi * 16; e += DT) {
                const int j = e >> 4, ti = e & 15;
                buf[j * 17 + ti] = ti < in ? src[(size_t)(j0 + j) * lds_ + i0 + ti] : 0.0;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < in * jn; e += DT) {
                const int ti = e / jn, j = e - ti * jn;
                const bool nz = !tri || (j0 + j <= i0 + ti);
                dst[(size_t)(i0 + ti) * ldd + j0 + j] = (nz && j < jhi) ? buf[j * 17 + ti] : 0.0;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// Panel factorisation (see the file header).  Thread t: column c = t & 15 of slab s = t >> 4; tri[i] = panel row i
// (i < 16, replicated over the slabs), tall[i] = panel row 16 + s + 32 i.  Cross-lane traffic stays in the VALU: the
// pivot column reaches the 16 lanes of a slab by DPP row_newbcast (gfx90a+), the triangle rows by v_readlane, the four
// slabs of a wave are summed with v_permlane32_swap / v_permlane16_swap (gfx950); only the sum over the eight waves
// goes through LDS (one barrier per column, buffers alternate).  The pivot lane keeps its column UNSCALED (nothing reads
// it again) and remembers its own scale factor: no per-element selects in the update, V = x * scale at write-out.
// ---------------------------------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ double row_bcast_f64(double x) {           // lane J of every 16-lane row to its whole row
    int lo_ = __double2loint(x), hi_ = __double2hiint(x);
    lo_ = __builtin_amdgcn_mov_dpp(lo_, 0x150 + J, 0xF, 0xF, false);       // (no `old` operand: every lane has a source)
    hi_ = __builtin_amdgcn_mov_dpp(hi_, 0x150 + J, 0xF, 0xF, false);
    return __hiloint2double(hi_, lo_);
}
__device__ __forceinline__ double slab_sum(double x) {                // x(l) + x(l^16) + x(l^32) + x(l^48), in every lane
    unsigned lo_ = __double2loint(x), hi_ = __double2hiint(x);
    auto a = __builtin_amdgcn_permlane32_swap(lo_, lo_, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi_, hi_, false, false);
    x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo_ = __double2loint(x); hi_ = __double2hiint(x);
    a = __builtin_amdgcn_permlane16_swap(lo_, lo_, false, false);
    b = __builtin_amdgcn_permlane16_swap(hi_, hi_, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// The panel is factored by the FIRST FOUR waves, one per SIMD (the column steps are VALU-issue-bound: with two waves on a
// SIMD each step took the sum of both instruction streams -- 8 waves x 15 rows per lane ran 1.6x slower than 4 x 30); the
// other four only take part in the barriers.
constexpr int QR_PW = 4;                                // panel waves
// acc += x(lane J of this lane's 16-lane row) * y: the broadcast rides on the FMA itself (gfx90a+: 64-bit DPP exists for
// row_newbcast only) -- no separate move per entry of the pivot column.  Inline assembly: hipcc has no builtin for it.
template <int J>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double x, double y) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
}

template <int RT, int J>
__device__ __forceinline__ void qr_panel_col(double (&tr)[16], double (&ta)[RT], double& myscale, bool act, int c, int wave,
                                             int lane, int pg_off, int tau_off) {
    double gt = 0.0;
    double* const bb = g_lds + pg_off + (J & 1) * (16 * QR_PW);      // [c][wave]: partial dots of the rows below the triangle
    if (act) {
        // dot products of the pivot column (lane J of every row: its triangle entries are the same in all rows, its tall
        // entries those of this lane's slab) with this lane's column
        double g4[4] = {0.0, 0.0, 0.0, 0.0}, t2[2] = {0.0, 0.0};      // independent accumulators: no back-to-back dependence
        asm volatile("s_nop 1");                                      // (VALU write -> DPP read of the same register: 2 wait states)
#pragma unroll
        for (int i = 0; i < RT; ++i) fmac_rowbcast<J>(g4[i & 3], ta[i], ta[i]);
#pragma unroll
        for (int i = J + 1; i < 16; ++i) fmac_rowbcast<J>(t2[i & 1], tr[i], tr[i]);
        gt = t2[0] + t2[1];
        double gu = slab_sum((g4[0] + g4[1]) + (g4[2] + g4[3]));
        if (lane < 16) bb[c * QR_PW + wave] = gu;
    }
    __syncthreads();
    if (act) {
        double G = 0.0, Nn = 0.0;
#pragma unroll
        for (int w = 0; w < QR_PW; ++w) { G += bb[c * QR_PW + w]; Nn += bb[J * QR_PW + w]; }
        // LAPACK dlarfg: beta = -sign(alpha) ||(alpha, x)||, tau = (beta - alpha) / beta, v = x / (alpha - beta), v_J = 1
        // (||x||^2 = the pivot column's dot product with itself: lane J's triangle part, column J's exchanged sums)
        const double alpha = readlane_f64(tr[J], J), xn2 = readlane_f64(gt, J) + Nn;
        const bool refl = xn2 != 0.0;
        const double beta = refl ? -copysign(sqrt(fma(alpha, alpha, xn2)), alpha) : alpha;
        const double tau = refl ? (beta - alpha) * fast_rcp(beta) : 0.0;
        const double scale = refl ? fast_rcp(alpha - beta) : 0.0;
        if (threadIdx.x == 0) g_lds[tau_off + J] = tau;
        // w_c = tau (a_Jc + v_below . a_c) for the columns right of J;  a_c <- a_c - v w_c with v = x scale.  The pivot lane
        // (and the columns left of it) run with w_c = 0: untouched.
        const bool right = c > J, piv = c == J;
        const double wc = right ? tau * fma(scale, gt + G, tr[J]) : 0.0;
        myscale = piv ? scale : myscale;
        tr[J] = piv ? beta : tr[J] - wc;
        const double z = -(scale * wc);
        asm volatile("s_nop 1");
#pragma unroll
        for (int i = J + 1; i < 16; ++i) fmac_rowbcast<J>(tr[i], tr[i], z);
#pragma unroll
        for (int i = 0; i < RT; ++i) fmac_rowbcast<J>(ta[i], ta[i], z);
    }
}
template <int RT, int J>
__device__ __forceinline__ void qr_panel_cols(double (&tr)[16], double (&ta)[RT], double& myscale, bool act, int c, int wave,
                                              int lane, int nb, int pg_off, int tau_off) {
    if constexpr (J < 16) {
        if (J < nb) qr_panel_col<RT, J>(tr, ta, myscale, act, c, wave, lane, pg_off, tau_off);
        else if (threadIdx.x == 0) g_lds[tau_off + J] = 0.0;
        qr_panel_cols<RT, J + 1>(tr, ta, myscale, act, c, wave, lane, nb, pg_off, tau_off);
    }
}

// Factors the panel P (Mk <= 16 + 16 RT rows, nb <= 16 columns, row stride ld, in global memory): R11's upper triangle back
// to P, V (unit lower trapezoid, explicit ones and zeros; zero rows up to the next multiple of 16) to the LDS panel
// g_lds[r * LU_LD + c], tau to g_lds[tau_off ..].  Thread t < 256: column c = t & 15 of slab s = t >> 4 (16 slabs).
// Pt: the panel's rows from the 17th on (P + 16 ld, or -- stacked triangular block on top, wg_qr_r -- the other block).
template <int RT>
__device__ __forceinline__ void qr_panel_body(double* P_, const double* Pt_, int ld_, int Mk_, int nb_, int pg_off_, int tau_off_) {
    auto* const P = uni_g(P_);
    auto* const Pt = uni_g(Pt_);
    const int ld = uni(ld_), Mk = uni(Mk_), nb = uni(nb_), pg_off = uni(pg_off_), tau_off = uni(tau_off_);
    const int tid = threadIdx.x, c = tid & 15, s = tid >> 4, lane = tid & 63, wave = uni((int)(tid >> 6));
    const bool act = wave < QR_PW;
    double tr[16], ta[RT], myscale = 0.0;
    if (act) {
#pragma unroll
        for (int i = 0; i < 16; ++i) tr[i] = (c < nb && i < Mk) ? P[(size_t)i * ld + c] : 0.0;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int r = 16 + s + 16 * i;
            ta[i] = (c < nb && r < Mk) ? Pt[(size_t)(r - 16) * ld + c] : 0.0;
        }
    }
    qr_panel_cols<RT, 0>(tr, ta, myscale, act, c, wave, lane, nb, pg_off, tau_off);
    double* const panel = g_lds;
    if (act) {
        if (s == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i < Mk) panel[i * LU_LD + c] = i > c ? tr[i] * myscale : ((i == c && c < nb) ? 1.0 : 0.0);
                if (i <= c && c < nb && i < Mk) P[(size_t)i * ld + c] = tr[i];
            }
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int r = 16 + s + 16 * i;
            if (r < Mk) panel[r * LU_LD + c] = ta[i] * myscale;
        }
    }
    for (int e = tid; e < (((Mk + 15) & ~15) - Mk) * 16; e += DT) panel[(Mk + (e >> 4)) * LU_LD + (e & 15)] = 0.0;   // whole tiles
    __syncthreads();
}
__device__ __noinline__ void qr_panel14(double* P, const double* Pt, int ld, int Mk, int nb, int pg_off, int tau_off) { qr_panel_body<14>(P, Pt, ld, Mk, nb, pg_off, tau_off); }
__device__ __noinline__ void qr_panel30(double* P, const double* Pt, int ld, int Mk, int nb, int pg_off, int tau_off) { qr_panel_body<QR_RT>(P, Pt, ld, Mk, nb, pg_off, tau_off); }

// T (16 x 16 upper triangular, g_lds[t_off + i * 16 + j]) of the compact WY form from the LDS panel V and tau:
// T_jj = tau_j, T(0:j, j) = -tau_j T(0:j, 0:j) (V^T V)(0:j, j)   (LAPACK dlarft, forward / columnwise).
// Ends WITHOUT a barrier: T is complete after the caller's next barrier.
__device__ __forceinline__ void qr_build_T(int Mk_, int gp_off_, int t_off_, int tau_off_) {
    const int Mk = uni(Mk_), gp_off = uni(gp_off_), t_off = uni(t_off_), tau_off = uni(tau_off_);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni((int)(tid >> 6)), lo = lane & 15, hi = lane >> 4;
    const double* const panel = g_lds;
    double* const gp = g_lds + gp_off;
    d4 acc = d4{0, 0, 0, 0};
    for (int r0 = 4 * wave; r0 < Mk; r0 += 4 * NWAVE) {
        const int r = r0 + hi;
        const double a = r < Mk ? panel[r * LU_LD + lo] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) gp[wave * 256 + (4 * v + hi) * 16 + lo] = acc[v];
    __syncthreads();
    if (tid < 256) {
        double sacc = 0.0;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) sacc += gp[w * 256 + tid];
        gp[tid] = sacc;                                  // (its own element of slot 0: nobody else reads or writes it)
    }
    __syncthreads();
    if (tid < 16) {
        const int i = tid;
        double Tr[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double tau_j = g_lds[tau_off + j];
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < j; ++k) sacc = fma(k >= i ? Tr[k] : 0.0, gp[k * 16 + j], sacc);
            Tr[j] = i == j ? tau_j : (i < j ? -tau_j * sacc : 0.0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) g_lds[t_off + i * 16 + j] = Tr[j];
    }
}

// A2 <- (I - V T^T V^T) A2 for the Mk x n2 matrix right of the panel, ONE pass over memory: a unit = (column tile, half of
// the row tiles); the two halves of a column tile run in the same round on neighbouring waves.  A wave loads its (up to
// 16) tiles of 16 x 16 into registers -- the loaded tile IS the B fragment of W = V^T A2 (v_mfma_f64_16x16x4: register
// v of a tile holds its rows 4v .. 4v+3) -- accumulates its half of W, exchanges the halves through LDS (one barrier
// per round, buffers alternate), forms W' = T^T W with four MFMAs (W's accumulator is again a B fragment), applies
// tile -= V_rows W' onto the registers it still holds and stores them: every element of A2 is read once and written once.
constexpr int QF_T = 16;                                // row tiles per unit: panels of up to 512 rows
// A2t: A2's rows from the 17th on (A2 + 16 lda, or the other block of a stack with its triangular block on top).
// (inlined into wg_qr_r, its only caller -- like the panel and T builders: as a real function it saved and restored 96 callee-saved
// VGPRs through scratch on every call, 27 calls per QR: 49 KB per wave and call, about HALF of the forward kernel's 14.5 MB of
// memory traffic per trajectory-step; profiles/r03_c5_pmc_traffic_dense.json)
__device__ __forceinline__ void qr_fused_update(double* A2_, double* A2t_, int lda_, int Mk_, int n2_, int t_off_, int px_off_) {
    auto* const A2 = uni_g(A2_);
    auto* const A2t = uni_g(A2t_);
    const int lda = uni(lda_), Mk = uni(Mk_), n2 = uni(n2_), t_off = uni(t_off_), px_off = uni(px_off_);
    const double* const panel = g_lds;                                // rows Mk .. 16 nrt - 1 are zero (qr_panel_body)
    const double* const T = g_lds + t_off;
    const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6)), lo = lane & 15, hi = lane >> 4;
    const int nct = (n2 + 15) >> 4, nrt = (Mk + 15) >> 4, h0 = (nrt + 1) >> 1;
    const int rounds = (2 * nct + NWAVE - 1) / NWAVE;
    for (int rd = 0; rd < rounds; ++rd) {
        const int u = rd * NWAVE + wave;                              // (scalar: wave is)
        const bool live = u < 2 * nct;
        const int ct = u >> 1, half = u & 1;
        const int rb0 = half ? h0 : 0, nq = live ? (half ? nrt - h0 : h0) : 0;
        const int j = ct * 16 + lo;
        const bool cok = live && j < n2;
        double* const px = g_lds + px_off + (rd & 1) * (NWAVE * 256);
        // Addresses: a scalar row-tile base plus four per-lane offsets (rows 4v + hi of a tile, column j).  No load is
        // masked (a select behind a load makes hipcc wait for that load at once): a row past the end reads the last row,
        // a column past the end column 0 -- finite values of the matrix that meet zero rows of V and are never stored.
        const int rows_left = Mk - 16 * rb0;
        const int jc = cok ? j : 0;
        int voff[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) voff[v] = (4 * v + hi) * lda + jc;
        d4 t[QF_T];
#pragma unroll
        for (int q = 0; q < QF_T; ++q) {
            if (q < nq) {
                gd* const tb = rb0 + q == 0 ? A2 : A2t + (size_t)(16 * (rb0 + q - 1)) * lda;     // scalar
                const int rl = rows_left - 16 * q;                            // rows of this tile that exist (>= 1)
                if (rl >= 16) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) t[q][v] = tb[voff[v]];
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) t[q][v] = tb[min(4 * v + hi, rl - 1) * lda + jc];
                }
            }
        }
        d4 acc = d4{0, 0, 0, 0};
        const double* const pb = panel + 16 * rb0 * LU_LD + hi * LU_LD + lo;          // V^T fragments: [16 q + 4 v + hi][lo]
#pragma unroll
        for (int q = 0; q < QF_T; ++q) {
            if (q < nq) {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[(16 * q + 4 * v) * LU_LD], t[q][v], acc, 0, 0, 0);
            }
        }
        if (live) {
#pragma unroll
            for (int v = 0; v < 4; ++v) px[wave * 256 + (4 * v + hi) * 16 + lo] = acc[v];
        }
        __syncthreads();
        if (live) {
            d4 wsum, wp = d4{0, 0, 0, 0};
#pragma unroll
            for (int v = 0; v < 4; ++v)
                wsum[v] = px[(wave & ~1) * 256 + (4 * v + hi) * 16 + lo] + px[(wave | 1) * 256 + (4 * v + hi) * 16 + lo];
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)                                    // W' = T^T W
                wp = __builtin_amdgcn_mfma_f64_16x16x4f64(T[(4 * kq + hi) * 16 + lo], wsum[kq], wp, 0, 0, 0);
            const double* const eb = panel + 16 * rb0 * LU_LD + lo * LU_LD + hi;      // V fragments: [16 q + lo][4 kq + hi]
#pragma unroll
            for (int q = 0; q < QF_T; ++q) {
                if (q < nq) {
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq)
                        t[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-eb[16 * q * LU_LD + 4 * kq], wp[kq], t[q], 0, 0, 0);
                    gd* const tb = rb0 + q == 0 ? A2 : A2t + (size_t)(16 * (rb0 + q - 1)) * lda;
                    const int rl = rows_left - 16 * q;
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (cok && 4 * v + hi < rl) tb[voff[v]] = t[q][v];
                }
            }
        }
    }
    __syncthreads();
}

// (the same as a real function: the 3p x p stacks of the smoothing pass -- up to 15 row tiles per unit -- measured 5 % FASTER through
// the call, save / restore included, than with the update inlined; the forward pass's shorter stacks 6 % faster inlined)
__device__ __noinline__ void qr_fused_update_call(double* A2, double* A2t, int lda, int Mk, int n2, int t_off, int px_off) {
    qr_fused_update(A2, A2t, lda, Mk, n2, t_off, px_off);
}

// Fallback beyond the LDS panel's limits: column-by-column Householder in global memory (one wave per trailing column).
__device__ __noinline__ void wg_qr_r_unblocked(double* S_, int ld_, int M_, int n_) {
    auto* const S = uni_g(S_);
    const int ld = uni(ld_), M = uni(M_), n = uni(n_);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni((int)(tid >> 6));
    double* const red = g_lds;
    for (int j = 0; j < n && j < M; ++j) {
        double part = 0.0;
        for (int r = j + 1 + tid; r < M; r += DT) { const double x = S[(size_t)r * ld + j]; part = fma(x, x, part); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        double xn2 = 0.0;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) xn2 += red[w];
        const double alpha = S[(size_t)j * ld + j];
        const bool refl = xn2 != 0.0;
        const double beta = refl ? -copysign(sqrt(fma(alpha, alpha, xn2)), alpha) : alpha;
        const double tau = refl ? (beta - alpha) / beta : 0.0;
        const double scale = refl ? 1.0 / (alpha - beta) : 0.0;
        for (int c = j + 1 + wave; c < n; c += NWAVE) {
            double dot = 0.0;
            for (int r = j + 1 + lane; r < M; r += 64) dot = fma(S[(size_t)r * ld + j], S[(size_t)r * ld + c], dot);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
            const double wc = tau * fma(scale, dot, S[(size_t)j * ld + c]);
            for (int r = j + 1 + lane; r < M; r += 64) S[(size_t)r * ld + c] = fma(-(S[(size_t)r * ld + j] * scale), wc, S[(size_t)r * ld + c]);
            if (lane == 0) S[(size_t)j * ld + c] -= wc;
        }
        __syncthreads();
        if (tid == 0) S[(size_t)j * ld + j] = beta;
        __syncthreads();
    }
}

// In place: the upper triangle of the top n x n block of S (M x n, row stride ld, M >= 1) becomes the R of S's QR
// factorisation (LAPACK's sign convention: r_jj = -sign(a_jj) ||.||); everything below the diagonal is left undefined.
//
// tp_top = n > 0 (n a multiple of 16): S = [T ; A] with T (n x n) UPPER TRIANGULAR on top -- add_sqrt's stack with the
// transposed lower factor first (the R of a stack does not depend on the order of its rows).  The reflectors of panel k
// then have their support in T's rows 16 k .. 16 k + 15 and in A: T's rows below are zero in the panel's columns and stay
// out of the panel, of T and of the update (LAPACK's dtpqrt: 2 n^3 instead of 3.33 n^3 flops for A of n rows).
// nd > 0 on top of that: A is BLOCK upper triangular with nd x nd blocks (A = (Q L)^T for a lower triangular L and a
// block-diagonal Q, prior.indep_init: A[i][j] = sum_k L[k][i] Q[j][k] needs a k >= i in j's block), so in the columns
// of panel k only A's rows up to the end of the block of column 16 k + 15 are non-zero: the panel's row count grows from
// 16 + nd ... to 16 + n instead of shrinking from 2 n (0.31 of the generic tile count at n = 160, nd = 5).
__device__ __noinline__ void wg_qr_r(double* S_, int ld_, int M_, int n_, double* ws_end_ = nullptr, int tp_top_ = 0, int nd_ = 0) {
    auto* const S = uni_g(S_);
    auto* const ws_end = uni_g(ws_end_);
    (void)ws_end;
    const int ld = uni(ld_), M = uni(M_), n = uni(n_), nd = uni(nd_);
    const int tp = (uni(tp_top_) == n && (n & 15) == 0 && M > n) ? n : 0;       // (anything else: the generic path, correct for any S)
    RK_STAMP_DECL(ws_end);
    const int Mmax = tp ? 16 + (M - tp) : M;
    const int gp_off = (((Mmax + 15) & ~15) * LU_LD + 1) & ~1, t_off = gp_off + NWAVE * 256, tau_off = t_off + 256, pg_off = tau_off + 16;
    const int px_off = pg_off + 2 * 16 * QR_PW;                // partial W tiles of the fused update: 2 x 8 x 256
    if (Mmax > QR_MAXM || px_off + 2 * NWAVE * 256 > LDS_DOUBLES) { wg_qr_r_unblocked((double*)S, ld, M, n); return; }
    __syncthreads();
    for (int k0 = 0; k0 < n && k0 < M; k0 += 16) {
        const int nb = min(16, n - k0), n2 = n - k0 - nb;
        int Mk = M - k0;
        double* Pt = (double*)(S + (size_t)(k0 + 16) * ld + k0);
        if (tp) {
            const int mb = M - tp, rl = nd > 0 ? min(mb, ((k0 + nb - 1) / nd + 1) * nd) : mb;
            Mk = 16 + rl;
            Pt = (double*)(S + (size_t)tp * ld + k0);
        }
        if (Mk <= 16 + 16 * 14) qr_panel14((double*)(S + (size_t)k0 * ld + k0), Pt, ld, Mk, nb, pg_off, tau_off);
        else qr_panel30((double*)(S + (size_t)k0 * ld + k0), Pt, ld, Mk, nb, pg_off, tau_off);
        RK_STAMP(10);
        if (n2 > 0) {
            qr_build_T(Mk, gp_off, t_off, tau_off);
            RK_STAMP(11);
            if (M > 2 * n) qr_fused_update_call((double*)(S + (size_t)k0 * ld + k0 + nb), Pt + nb, ld, Mk, n2, t_off, px_off);
            else qr_fused_update((double*)(S + (size_t)k0 * ld + k0 + nb), Pt + nb, ld, Mk, n2, t_off, px_off);
            RK_STAMP(13);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// X = E^{-1} B in place, E triangular (n x n, non-unit diagonal), E[i][j] = trans ? T[j][i] : T[i][j]; LOWER: E is lower
// triangular (forward substitution from the top), else upper (back substitution from the bottom).  n, nr <= 160: the
// right-hand side stays in registers (lu_backsub_regs' scheme: 100 tiles over 8 waves; per 16-row block the block
// column of E is staged in LDS, the owners of the block's tiles pass them through LDS scratches to 16 threads per
// column tile for the 16 x 16 substitution, and every wave updates its remaining tiles with rank-16 MFMAs).
// ---------------------------------------------------------------------------------------------------------------
constexpr int TS_SCR = BS_T * 16 * LU_LD;               // doubles: the staged block column (n <= 160 rows), then the ten scratches

// threads 0 .. 16 nct - 1: X = E11^{-1} Y for column (tid & 15) of the tile in scratch (tid >> 4)
template <bool LOWER>
__device__ __noinline__ void tri_trsm_all(int k0_, int nb_, int nct_) {
    const int k0 = uni(k0_), nb = uni(nb_), nct = uni(nct_);
    if ((int)threadIdx.x >= 16 * nct) return;
    const double* const panel = g_lds;
    double* const scratch = g_lds + TS_SCR + (threadIdx.x >> 4) * 16 * LU_LD;
    const int lo = threadIdx.x & 15;
    double x[LU_NB];
#pragma unroll
    for (int j = 0; j < LU_NB; ++j) x[j] = j < nb ? scratch[j * LU_LD + lo] : 0.0;
    trsm16<LOWER, false>(x, panel + k0 * LU_LD, nb, g_rdiag);
#pragma unroll
    for (int j = 0; j < LU_NB; ++j) scratch[j * LU_LD + lo] = x[j];
}

// (Two ways of hiding the per-block global round trip of the staging were built and measured slower: the next block column
// prefetched into registers across the block, and waves 3-7 staging while waves 0-2 substitute -- both push this function
// past 256 registers, and a spill reload in the MFMA loop waits for every outstanding load.)
// dm_, mu_ (optional): mu[i] += sum_j X[j][i] dm[j] from the solved tiles while they are still in registers -- the smoother's mean
// update mu_f + G (mu_s - mu-) with X = G^T (standard.py:213-214) without reading G^T back from memory: per tile the four row
// terms of a lane, a shuffle sum over the tile's row quarters, the ten row blocks' partial sums through LDS in a fixed order.
template <bool LOWER>
__device__ __noinline__ void wg_tri_solve_regs(const double* T_, int ldt_, int trans_, double* Bm_, int ldb_, int n_, int nr_,
                                               double* ws_end_, const double* dm_, double* mu_) {
    auto* const T = uni_g(T_);
    auto* const Bm = uni_g(Bm_);
    auto* const dmv = uni_g(dm_);
    auto* const mu = uni_g(mu_);
    auto* const ws_end = uni_g(ws_end_);
    (void)ws_end;
    RK_STAMP_DECL(ws_end);
    const int ldt = uni(ldt_), trans = uni(trans_), ldb = uni(ldb_), n = uni(n_), nr = uni(nr_);
    const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6)), lo = lane & 15, hi = lane >> 4;
    const int nbk = (n + 15) >> 4, nct = (nr + 15) >> 4;
    double* const panel = g_lds;
    double* const scr = g_lds + TS_SCR;
    d4 t[BS_Q];
#pragma unroll
    for (int q = 0; q < BS_Q; ++q) {
        const int e = wave + NWAVE * q, rb = e / BS_T, ct = e - rb * BS_T;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = 16 * rb + 4 * v + hi, col = 16 * ct + lo;
            // clamped, unconditional and UNMASKED: all loads are in flight together (a select or an asm barrier behind a
            // load makes hipcc wait for it at once -- 52 serialised round trips per solve).  A row / column past the end
            // holds a copy of the last one: finite values that only meet zero multipliers and are never stored.
            t[q][v] = Bm[min(row, n - 1) * ldb + min(col, nr - 1)];
        }
    }
    for (int kk = 0; kk < nbk; ++kk) {
        const int k = LOWER ? kk : nbk - 1 - kk;
        const int k0 = 16 * k, nb = min(16, n - k0);
        const int r_lo = LOWER ? k0 : 0, nrow = LOWER ? n - k0 : k0 + nb;
        __syncthreads();                                            // the previous block's panel and scratches are free
        // (all loads of the block column first, clamped instead of masked, then the LDS writes: the rolled loop had ONE load in flight
        //  per thread -- up to five global round trips per block, 71 k of the substitution's 195 k cycles at n = 160)
        {
            constexpr int SU = (16 * BS_T * LU_NB + DT - 1) / DT;   // elements per thread at the largest block column
            const int total = nrow * LU_NB;
            double sv[SU];
            if (!trans) {
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int e = min((int)threadIdx.x + DT * u, total - 1);
                    sv[u] = T[(r_lo + (e >> 4)) * ldt + k0 + min(e & 15, nb - 1)];
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int e = threadIdx.x + DT * u;
                    if (e < total) panel[(r_lo + (e >> 4)) * LU_LD + (e & 15)] = (e & 15) < nb ? sv[u] : 0.0;
                }
            } else {
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int e = min((int)threadIdx.x + DT * u, total - 1);
                    const int c = e / nrow, r = r_lo + (e - c * nrow);
                    sv[u] = T[(k0 + min(c, nb - 1)) * ldt + r];
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int e = threadIdx.x + DT * u;
                    const int c = e / nrow, r = r_lo + (e - c * nrow);
                    if (e < total) panel[r * LU_LD + c] = c < nb ? sv[u] : 0.0;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < BS_Q; ++q) {
            const int e = wave + NWAVE * q, rb = e / BS_T, ct = e - rb * BS_T;
            if (rb == k && ct < nct) {
#pragma unroll
                for (int v = 0; v < 4; ++v) scr[ct * 16 * LU_LD + (4 * v + hi) * LU_LD + lo] = t[q][v];
            }
        }
        __syncthreads();
        RK_STAMP(11);
        if (threadIdx.x < nb) g_rdiag[threadIdx.x] = 1.0 / panel[(k0 + threadIdx.x) * LU_LD + threadIdx.x];
        __syncthreads();
        tri_trsm_all<LOWER>(k0, nb, nct);
        __syncthreads();
        RK_STAMP(12);
#pragma unroll
        for (int q = 0; q < BS_Q; ++q) {
            const int e = wave + NWAVE * q, rb = e / BS_T, ct = e - rb * BS_T;
            if (ct < nct && rb < nbk) {
                const double* const sc = scr + ct * 16 * LU_LD;
                if (rb == k) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) t[q][v] = sc[(4 * v + hi) * LU_LD + lo];
                } else if (LOWER ? rb > k : rb < k) {               // Y_rb -= E(rb, k) X_k
                    const int er = min(16 * rb + lo, n - 1);
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq)
                        t[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-panel[er * LU_LD + 4 * kq + hi],
                                                                    sc[(4 * kq + hi) * LU_LD + lo], t[q], 0, 0, 0);
                }
            }
        }
        RK_STAMP(13);
    }
#pragma unroll
    for (int q = 0; q < BS_Q; ++q) {
        const int e = wave + NWAVE * q, rb = e / BS_T, ct = e - rb * BS_T;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = 16 * rb + 4 * v + hi, col = 16 * ct + lo;
            if (e < BS_T * BS_T && row < n && col < nr) Bm[row * ldb + col] = t[q][v];
        }
    }
    __syncthreads();
    if (mu) {
        double* const part = g_lds;                                 // [row block][column]: BS_T x (16 BS_T) doubles
#pragma unroll
        for (int q = 0; q < BS_Q; ++q) {
            const int e = wave + NWAVE * q, rb = e / BS_T, ct = e - rb * BS_T;
            if (e < BS_T * BS_T) {
                double sacc = 0.0;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * rb + 4 * v + hi;
                    sacc = fma(t[q][v], row < n ? dmv[row] : 0.0, sacc);      // (rows past the end hold copies: zero weight)
                }
                sacc += __shfl_xor(sacc, 16, 64);
                sacc += __shfl_xor(sacc, 32, 64);
                if (hi == 0) part[rb * (16 * BS_T) + 16 * ct + lo] = sacc;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nr; i += DT) {
            double sacc = 0.0;
            for (int rb = 0; rb < nbk; ++rb) sacc += part[rb * (16 * BS_T) + i];
            mu[i] = mu[i] + sacc;
        }
        __syncthreads();
    }
    RK_STAMP(10);
}

// any size: one thread per right-hand-side column, everything in global memory (slow; beyond 160 x 160 only)
template <bool LOWER>
__device__ __noinline__ void wg_tri_solve_slow(const double* T_, int ldt_, int trans_, double* Bm_, int ldb_, int n_, int nr_) {
    auto* const T = uni_g(T_);
    auto* const Bm = uni_g(Bm_);
    const int ldt = uni(ldt_), trans = uni(trans_), ldb = uni(ldb_), n = uni(n_), nr = uni(nr_);
    for (int c = threadIdx.x; c < nr; c += DT) {
        for (int kk = 0; kk < n; ++kk) {
            const int k = LOWER ? kk : n - 1 - kk;
            const double xk = Bm[(size_t)k * ldb + c] / (trans ? T[(size_t)k * ldt + k] : T[(size_t)k * ldt + k]);
            Bm[(size_t)k * ldb + c] = xk;
            if (LOWER) for (int i = k + 1; i < n; ++i)
                Bm[(size_t)i * ldb + c] = fma(-(trans ? T[(size_t)k * ldt + i] : T[(size_t)i * ldt + k]), xk, Bm[(size_t)i * ldb + c]);
            else for (int i = 0; i < k; ++i)
                Bm[(size_t)i * ldb + c] = fma(-(trans ? T[(size_t)k * ldt + i] : T[(size_t)i * ldt + k]), xk, Bm[(size_t)i * ldb + c]);
        }
    }
    __syncthreads();
}

template <bool LOWER>
__device__ __forceinline__ void wg_tri_solve(const double* T, int ldt, int trans, double* Bm, int ldb, int n, int nr,
                                             double* ws_end = nullptr) {
    if (n <= 16 * BS_T && nr <= 16 * BS_T) wg_tri_solve_regs<LOWER>(T, ldt, trans, Bm, ldb, n, nr, ws_end, nullptr, nullptr);
    else wg_tri_solve_slow<LOWER>(T, ldt, trans, Bm, ldb, n, nr);
}

// ---------------------------------------------------------------------------------------------------------------
// Workspace of one trajectory (doubles), shared by the three kernels below.
// ---------------------------------------------------------------------------------------------------------------
struct DenseSqWs {
    double *S, *A1, *A2, *A3, *Wt, *WS, *X, *W2, *Sm, *Vh, *mup, *f, *yhat, *dm;
};
__host__ __device__ inline size_t dense_sq_off_Wt(int p) { return 6 * (size_t)p * p; }
__host__ __device__ inline size_t dense_sq_off_mup(int p, int m) {
    return 6 * (size_t)p * p + 4 * (size_t)m * p + ((size_t)p + m) * m + (size_t)m * m;
}
__device__ __forceinline__ DenseSqWs carve_sq(double* w, int p, int m) {
    DenseSqWs d;
    const size_t pp = (size_t)p * p, mp = (size_t)m * p;
    d.S = w; d.A1 = d.S + 3 * pp; d.A2 = d.A1 + pp; d.A3 = d.A2 + pp;
    d.Wt = d.A3 + pp; d.WS = d.Wt + mp; d.X = d.WS + mp; d.W2 = d.X + mp;
    d.Sm = d.W2 + mp; d.Vh = d.Sm + ((size_t)p + m) * m;
    d.mup = d.Vh + (size_t)m * m; d.f = d.mup + p; d.yhat = d.f + m; d.dm = d.yhat + m;
    return d;
}
size_t dense_sq_ws_doubles(int p, int m) {
    return dense_sq_off_mup(p, m) + 2 * (size_t)p + 2 * (size_t)m + 16;
}

// 1.0 if the predict step's stack has the structure wg_qr_r can use: every entry of Q outside the nd x nd diagonal blocks and
// every entry of R^{1/2} above the diagonal exactly zero, p a multiple of 16 (what prior.indep_init + a Cholesky factor give)
__global__ void dense_sqcheck_kernel(const double* Q, const double* Rh, int p, int nd, double* flag) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    for (int e = threadIdx.x; e < p * p; e += blockDim.x) {
        const int i = e / p, j = e - i * p;
        if (i / nd != j / nd && Q[e] != 0.0) mine = 1;
        if (j > i && Rh[e] != 0.0) mine = 1;
    }
    if (mine) atomicOr(&bad, 1);
    __syncthreads();
    if (threadIdx.x == 0) *flag = (bad || (p & 15) || nd < 1 || p % nd) ? 0.0 : 1.0;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward pass, square-root form (solve.py:31-122 with square_root.py:56-57 and 88-99).  MODE as in dense_fwd_kernel.
// a.var: the filtered factors L_n (lower), a.fac_pred: the predicted factors L^-_n (or null: not kept).
// ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(DT) dense_sqrt_fwd_kernel(DenseArgs a) {
    const int b = blockIdx.x, p = a.p, m = a.m;
    const int nd = p / m;
    const DenseSqWs w = carve_sq(a.ws + (size_t)b * a.ws_stride, p, m);
    double* mean = a.mean + (size_t)b * (a.N + 1) * p;
    double* var = a.var + (size_t)b * (a.N + 1) * p * p;
    double* facp = a.fac_pred ? a.fac_pred + (size_t)b * (a.N + 1) * p * p : nullptr;
    double* meanp = a.mean_pred ? a.mean_pred + (size_t)b * (a.N + 1) * p : nullptr;
    const double* Aode = a.theta;
    const bool noisy = a.itg == RK_INTERROGATE_RODEO;              // var_meas != 0 (interrogate.py:110-113)
    const int kv = noisy ? m : 0;
    double* const ws_end = a.ws + a.ws_stride;                     // (phase stamps of workgroup 0, -DRK_DENSE_STAMPS)
    (void)ws_end;
    // set by dense_sqcheck_kernel: Q block diagonal (nd x nd), R^{1/2} lower triangular, p a multiple of 16 (indep_init priors)
    const bool structured = a.ws[a.ws_stride - 1] != 0.0;
    RK_STAMP_DECL(ws_end);
    if (MODE != 2) {
        RK_STAMP_ZERO_N(15);
        for (int i = threadIdx.x; i < p; i += DT) mean[i] = a.x0_b ? a.x0[(size_t)i * a.B + b] : a.x0[i];
        for (int e = threadIdx.x; e < p * p; e += DT) var[e] = 0.0;
        if (facp) for (int e = threadIdx.x; e < p * p; e += DT) facp[e] = 0.0;
        if (meanp) for (int i = threadIdx.x; i < p; i += DT) meanp[i] = a.x0_b ? a.x0[(size_t)i * a.B + b] : a.x0[i];
    }
    __syncthreads();
    auto predict = [&](int n) {
        const double* mu = mean + (size_t)n * p;
        const double* Ln = var + (size_t)n * p * p;
        double* Lp = facp ? facp + (size_t)(n + 1) * p * p : w.A2;
        // square_root.py:56-57: L^- = add_sqrt(Q L, R^{1/2}) = R^T of qr([ (Q L)^T ; R^{1/2 T} ])
        // (structured: R^{1/2 T}, upper triangular, on top and the block upper triangular (Q L)^T below -- wg_qr_r)
        RK_STAMP_RESET();
        wg_gemm(gemm_op(w.S + (structured ? (size_t)p * p : 0), p, Ln, p, true, a.Q, p, true, p, p, p, nullptr, 0, 0.0, 1.0));
        wg_transpose(w.S + (structured ? 0 : (size_t)p * p), p, a.R, p, p, p, 0);
        RK_STAMP(0);
        wg_qr_r(w.S, p, 2 * p, p, ws_end, structured ? p : 0, structured ? nd : 0);
        RK_STAMP(1);
        wg_transpose(Lp, p, w.S, p, p, p, 1);
        wg_gemv<false>(w.mup, a.Q, p, mu, p, p, nullptr, 0.0, 1.0);
        RK_STAMP(2);
        if (meanp) {
            for (int i = threadIdx.x; i < p; i += DT) meanp[(size_t)(n + 1) * p + i] = w.mup[i];
            __syncthreads();
        }
    };
    auto update = [&](int n) {
        const double* Lp = facp ? facp + (size_t)(n + 1) * p * p : w.A2;
        double* mu_o = mean + (size_t)(n + 1) * p;
        double* L_o = var + (size_t)(n + 1) * p * p;
        // ---- interrogation (as in dense_fwd_kernel): W~ in w.Wt, yhat = W~ mu- + a ----
        if (MODE == 0) {
            for (int e = threadIdx.x; e < m * p; e += DT) {
                const int i = e / p, j = e % p;
                double Jij = 0.0;
                if (a.itg == RK_INTERROGATE_KRAMER && j % nd == 0) {
                    const int v = j / nd;
                    Jij = a.theta_b ? Aode[((size_t)i * m + v) * a.B + b] : Aode[(size_t)i * m + v];
                }
                w.Wt[e] = a.W[e] + (-Jij);
            }
            __syncthreads();
        }
        for (int i = threadIdx.x >> 6; i < m; i += DT / 64) {
            const int lane = threadIdx.x & 63;
            double s = 0.0, jm = 0.0, wm = 0.0;
            if (MODE == 0)
                for (int v = lane; v < m; v += 64) {
                    const double Aiv = a.theta_b ? Aode[((size_t)i * m + v) * a.B + b] : Aode[(size_t)i * m + v];
                    s = fma(Aiv, w.mup[(size_t)v * nd], s);
                }
            for (int j = lane; j < p; j += 64) {
                const double wt = w.Wt[(size_t)i * p + j], mj = w.mup[j];
                if (MODE == 0) jm = fma(a.W[(size_t)i * p + j] - wt, mj, jm);
                wm = fma(wt, mj, wm);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); jm += __shfl_xor(jm, off); wm += __shfl_xor(wm, off); }
            const double am = MODE != 0 ? w.f[i] : (a.itg == RK_INTERROGATE_KRAMER ? -s + jm : -s);
            if (lane == 0) w.yhat[i] = wm + am;
        }
        __syncthreads();
        // ---- update (square_root.py:88-99) ----
        RK_STAMP(3);
        wg_gemm(gemm_op(w.WS, p, w.Wt, p, false, Lp, p, false, m, p, p, nullptr, 0, 0.0, 1.0));            // W~ L^-
        wg_transpose(w.Sm, m, w.WS, p, p, m, 0);
        if (noisy) {
            // the reference hands W L^- W^T to the update as the "factor" of var_meas (interrogate.py:110-113 called with
            // a factor): kept, it defines the reference's numbers in this mode
            wg_gemm(gemm_op(w.Vh, m, w.WS, p, false, a.W, p, true, m, m, p, nullptr, 0, 0.0, 1.0));
            wg_transpose(w.Sm + (size_t)p * m, m, w.Vh, m, m, m, 0);
        }
        RK_STAMP(4);
        wg_qr_r(w.Sm, m, p + kv, m);                                                                       // L_m^T in w.Sm
        RK_STAMP(5);
        for (int e = threadIdx.x; e < m * p; e += DT) w.X[e] = w.Wt[e];
        __syncthreads();
        wg_tri_solve<true>(w.Sm, m, 1, w.X, p, m, p);                                                      // L_m^{-1} W~
        RK_STAMP(6);
        wg_gemm(gemm_op(w.W2, p, w.X, p, false, Lp, p, false, m, p, p, nullptr, 0, 0.0, 1.0));             // . L^-
        wg_gemm(gemm_op(w.X, p, w.W2, p, false, Lp, p, true, m, p, p, nullptr, 0, 0.0, 1.0));              // . L^-^T
        RK_STAMP(7);
        wg_tri_solve<false>(w.Sm, m, 0, w.X, p, m, p);                                                     // L_m^{-T} . = K^T
        RK_STAMP(6);
        for (int i = threadIdx.x; i < p; i += DT) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s = fma(w.X[(size_t)j * p + i], 0.0 - w.yhat[j], s);
            mu_o[i] = w.mup[i] + s;
        }
        wg_gemm(gemm_op(w.A1, p, w.X, p, true, w.WS, p, false, p, p, m, Lp, p, 1.0, -1.0));                // L^- - K (W~ L^-)
        wg_transpose(w.S, p, w.A1, p, p, p, 0);
        if (noisy) wg_gemm(gemm_op(w.S + (size_t)p * p, p, w.Vh, m, true, w.X, p, false, m, p, m, nullptr, 0, 0.0, 1.0));   // (K V^{1/2})^T
        RK_STAMP(8);
        wg_qr_r(w.S, p, p + kv, p, ws_end);
        RK_STAMP(9);
        wg_transpose(L_o, p, w.S, p, p, p, 1);
        RK_STAMP(2);
    };
    if (MODE == 0) {
        for (int n = 0; n < a.N; ++n) { predict(n); update(n); }
    } else if (MODE == 1) {
        predict(0);
    } else {
        update(a.n0);
        if (a.n0 + 1 < a.N) predict(a.n0 + 1);
    }
}

// G^T = L^-^{-T} ( (L^-^{-1} Q) (L_f L_f^T) ) into w.A3 (square_root.py:170-175), mu- into w.mup
__device__ __forceinline__ void dense_sqrt_gain(const DenseArgs& a, const DenseSqWs& w, const double* Lf, const double* Lp,
                                                const double* mu_f, int p) {
    double* const ws_end = a.ws + a.ws_stride;
    (void)ws_end;
    RK_STAMP_DECL(ws_end);
    wg_gemm(gemm_op(w.A1, p, Lf, p, false, Lf, p, true, p, p, p, nullptr, 0, 0.0, 1.0));                   // L_f L_f^T
    for (int e = threadIdx.x; e < p * p; e += DT) w.A2[e] = a.Q[e];
    __syncthreads();
    RK_STAMP(0);
    wg_tri_solve<true>(Lp, p, 0, w.A2, p, p, p);                                                           // L^-^{-1} Q
    RK_STAMP(1);
    wg_gemm(gemm_op(w.A3, p, w.A2, p, false, w.A1, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
    RK_STAMP(2);
    wg_tri_solve<false>(Lp, p, 1, w.A3, p, p, p, RK_STAMP_TRI ? ws_end : nullptr);                              // G^T
    RK_STAMP(3);
    wg_gemv<false>(w.mup, a.Q, p, mu_f, p, p, nullptr, 0.0, 1.0);
    // J^T = I - Q^T G^T into w.A1 (square_root.py:215-216)
    wg_gemm(gemm_op(w.A1, p, a.Q, p, true, w.A3, p, false, p, p, p, nullptr, 0, 0.0, -1.0));
    for (int i = threadIdx.x; i < p; i += DT) w.A1[(size_t)i * p + i] += 1.0;
    __syncthreads();
    RK_STAMP(4);
}
// out = mu_f + G dm (column sums of G^T = w.A3 through LDS)
__device__ __forceinline__ void dense_sqrt_mean(const DenseSqWs& w, const double* mu_f, double* out, int p) {
    double* const lds = g_lds;
    const int ng = DT / 64;
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
        out[i] = mu_f[i] + s;
    }
    __syncthreads();
}

// Backward mean / variance smoother in square-root form (solve.py:257-301 with square_root.py:209-219), in place.
__global__ void __launch_bounds__(DT) dense_sqrt_bwd_mv_kernel(DenseArgs a) {
    const int b = blockIdx.x, p = a.p, m = a.m;
    const DenseSqWs w = carve_sq(a.ws + (size_t)b * a.ws_stride, p, m);
    double* mean = a.mean + (size_t)b * (a.N + 1) * p;
    double* var = a.var + (size_t)b * (a.N + 1) * p * p;
    const double* facp = a.fac_pred + (size_t)b * (a.N + 1) * p * p;
    const size_t pp = (size_t)p * p;
    double* const ws_end = a.ws + a.ws_stride;
    (void)ws_end;
    RK_STAMP_DECL(ws_end);
    RK_STAMP_ZERO_N(15);
    for (int n = a.N - 1; n >= 1; --n) {
        double* mu_f = mean + (size_t)n * p;
        double* Lf = var + (size_t)n * pp;
        const double* mu_s = mean + (size_t)(n + 1) * p;
        const double* Ls = var + (size_t)(n + 1) * pp;
        const double* Lp = facp + (size_t)(n + 1) * pp;
        dense_sqrt_gain(a, w, Lf, Lp, mu_f, p);
        for (int i = threadIdx.x; i < p; i += DT) w.dm[i] = mu_s[i] - w.mup[i];
        __syncthreads();
        // L_s = add_sqrt(G [L_next | R^{1/2}], J L_f): rows of the stacked matrix = (G L_next)^T, (G R^{1/2})^T, (J L_f)^T
        RK_STAMP_RESET();
        wg_gemm(gemm_op(w.S, p, Ls, p, true, w.A3, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
        wg_gemm(gemm_op(w.S + pp, p, a.R, p, true, w.A3, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
        wg_gemm(gemm_op(w.S + 2 * pp, p, Lf, p, true, w.A1, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
        RK_STAMP(5);
        dense_sqrt_mean(w, mu_f, mu_f, p);                                                    // square_root.py:211-212
        RK_STAMP(6);
        wg_qr_r(w.S, p, 3 * p, p, RK_STAMP_TRI ? nullptr : ws_end);
        RK_STAMP(7);
        wg_transpose(Lf, p, w.S, p, p, p, 1);
        RK_STAMP(8);
    }
}

// x = mean + L (sgn . z) with L = R^T for the upper triangular R (row stride p) and sgn_j = sign of r_jj: the factor's
// column signs are normalised (diagonal >= 0) so that the draw does not depend on the signs a QR happens to produce
// (oracle/scan.py draw(), oracle/interrogations.py).  z: p standard normals at g_lds[zoff ..].
__device__ __forceinline__ void dense_sqrt_draw(const double* R, const double* mean, double* x, int p, int zoff, bool lower) {
    const double* const z = g_lds + zoff;
    for (int i = threadIdx.x; i < p; i += DT) {
        double acc = 0.0;
        for (int j = 0; j <= i; ++j) {
            const double d = lower ? R[(size_t)j * p + j] : R[(size_t)j * p + j];
            const double lij = lower ? R[(size_t)i * p + j] : R[(size_t)j * p + i];
            acc = fma(d < 0.0 ? -lij : lij, z[j], acc);
        }
        x[i] = mean[i] + acc;
    }
    __syncthreads();
}

// Backward sampler in square-root form (solve.py:137-205 with square_root.py:252-261); the filtered factors stay untouched.
__global__ void __launch_bounds__(DT) dense_sqrt_bwd_sim_kernel(DenseArgs a) {
    const int b = blockIdx.x, p = a.p, m = a.m;
    const DenseSqWs w = carve_sq(a.ws + (size_t)b * a.ws_stride, p, m);
    const double* mean = a.mean + (size_t)b * (a.N + 1) * p;
    const double* var = a.var + (size_t)b * (a.N + 1) * p * p;
    const double* facp = a.fac_pred + (size_t)b * (a.N + 1) * p * p;
    const size_t pp = (size_t)p * p;
    const unsigned traj = (unsigned)(a.traj_offset + (unsigned long long)b);
    auto put_x = [&](int n, const double* xv) {
        for (int i = threadIdx.x; i < p; i += DT) a.x[((size_t)n * p + i) * a.B + b] = xv[i];
    };
    double* const xn = w.W2;                                   // x_{n+1}: p doubles (w.W2 holds m p >= p and is free here)
    wg_normals(a.seed, traj, (unsigned)a.N, PURPOSE_SMOOTH, p, 0);
    dense_sqrt_draw(var + (size_t)a.N * pp, mean + (size_t)a.N * p, xn, p, 0, true);        // terminal draw (solve.py:182-186)
    put_x(a.N, xn);
    __syncthreads();
    for (int n = a.N - 1; n >= 1; --n) {
        const double* mu_f = mean + (size_t)n * p;
        const double* Lf = var + (size_t)n * pp;
        const double* Lp = facp + (size_t)(n + 1) * pp;
        dense_sqrt_gain(a, w, Lf, Lp, mu_f, p);
        for (int i = threadIdx.x; i < p; i += DT) w.dm[i] = xn[i] - w.mup[i];                 // x_{n+1} - mu-
        __syncthreads();
        wg_gemm(gemm_op(w.S, p, a.R, p, true, w.A3, p, false, p, p, p, nullptr, 0, 0.0, 1.0));              // (G R^{1/2})^T
        wg_gemm(gemm_op(w.S + pp, p, Lf, p, true, w.A1, p, false, p, p, p, nullptr, 0, 0.0, 1.0));          // (J L_f)^T
        dense_sqrt_mean(w, mu_f, w.mup, p);                                                   // mean_sim (square_root.py:254-255)
        wg_qr_r(w.S, p, 2 * p, p);
        wg_normals(a.seed, traj, (unsigned)n, PURPOSE_SMOOTH, p, 0);
        dense_sqrt_draw(w.S, w.mup, xn, p, 0, false);
        put_x(n, xn);
        __syncthreads();
    }
    put_x(0, mean);
}

}  // namespace rk
