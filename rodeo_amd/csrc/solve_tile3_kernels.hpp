// Forward kernel of the p = 3 MFMA-tile path (solve_tile3.hip holds the description of the tile algebra), as a template
// over the right-hand side: instantiated ahead of time for the built-in ODEs (solve_tile3.hip) and at run time by
// hiprtc for user-supplied ones (rhs_jit.hip).  RTC-safe: no host code, no <hip/hip_runtime.h> under __HIPCC_RTC__.
#pragma once
#include "rk_enums.hpp"
#include "kalman_small.hpp"
#include "mfma_tile.hpp"
#include "philox.hpp"
#include "solve_args.hpp"

namespace rk {

constexpr int TILE_DOUBLES = 12;     // 3 rows x [Sigma(3) | mu] per (time step, tile)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct TileCoord {
    int r, g, c;          // row, tile-in-wave, column
    int tau;              // global tile index (clamped to a valid tile)
    int b, blk;           // trajectory, block
    bool valid;           // this lane's tile exists
};

// TPW tiles per wave: 4 in the backward kernels (the flat tile array), Tpw<D> in the forward kernel (whole trajectories)
template <int D, int TPW = 4>
__device__ __forceinline__ TileCoord tile_coord(int wave, int lane, int n_tiles) {
    TileCoord t;
    t.r = lane >> 4; t.g = (lane >> 2) & 3; t.c = lane & 3;
    const int tau = wave * TPW + t.g;
    t.valid = t.g < TPW && tau < n_tiles;
    t.tau = t.valid ? tau : (wave * TPW < n_tiles ? wave * TPW : n_tiles - 1);     // an idle slot repeats the wave's first tile
    t.b = t.tau / D; t.blk = t.tau - t.b * D;
    return t;
}

// ---- forward ---------------------------------------------------------------------------------------------------
// n_block <= 4: one wave per workgroup carries whole trajectories (Tpw<D> tiles).  n_block = 5..64: a workgroup of
// ceil(D / 4) waves (up to 16 = 1024 threads) is ONE trajectory, wave w holds blocks 4 w .. 4 w + 3, and the evaluation
// points of all blocks are exchanged through LDS once per step (one barrier; double-buffered so that no second one is
// needed).  The reference's block solver has no limit on n_block (src/rodeo/solve.py:47-68 vmaps over it); 64 is the
// largest workgroup (round 3 stopped at 16: a 32-variable ODE could only run in the O((d p)^3) non-block form).
constexpr int TILE_MAX_BLOCKS = 64;

template <int D>
struct TileWaves {                               // waves per forward workgroup
    static constexpr int value = D <= 4 ? 1 : (D + 3) / 4;
};

// sqrt(x) for a variance x > 0 in the normal range from v_rsq_f64 and ONE coupled correction: four dependent operations
// where fast_sqrt_pos (two corrections, <= 1 ulp) has eight -- on the chkrebtii step's chain, whose draw x_0 = mu-_0 +
// sqrt(Sigma-_00) z_0 is compared at 1e-7 (relative error of this root <= 1e-14: the square of v_rsq_f64's)
__device__ __forceinline__ double sqrt_pos_1step(double x) {
    const double rs = __builtin_amdgcn_rsq(x);
    const double g = x * rs, h = 0.5 * rs;
    const double s = fma(fma(-g, g, x), h, g);
    return x > 0.0 ? s : 0.0;
}

template <class RHS, int ITG>
__global__ void __launch_bounds__(64 * TileWaves<RHS::D>::value) fwd_tile3_kernel(SolveArgs a, double* __restrict__ tiles) {
    constexpr int D = RHS::D, P = 3, NW = TileWaves<D>::value, TPW = NW > 1 ? 4 : Tpw<D>::value;
    static_assert(D >= 1 && D <= TILE_MAX_BLOCKS, "tile path: n_block in 1..64");
    static_assert(RHS::NDEP == 1, "tile path: right-hand sides that depend on X[b][0] only");
    const int n_tiles = a.B * D;
    const int wave_in_wg = threadIdx.x >> 6, lane = threadIdx.x & 63;
    TileCoord tc;
    if constexpr (NW == 1) {
        tc = tile_coord<D, TPW>(blockIdx.x, lane, n_tiles);
    } else {
        tc.r = lane >> 4; tc.g = (lane >> 2) & 3; tc.c = lane & 3;
        const int blk_w = wave_in_wg * 4 + tc.g;
        tc.valid = blk_w < D;
        tc.b = blockIdx.x; tc.blk = tc.valid ? blk_w : D - 1;       // (idle slots of the last wave repeat the last block)
        tc.tau = tc.b * D + tc.blk;
    }
    const int r = tc.r, c = tc.c, b = tc.b, blk = tc.blk;
    const bool in3 = r < 3 && c < 3;

    // per-lane constants in D layout
    const double Qt = in3 ? ld(a.Q, ((size_t)blk * P + c) * P + r, a.Q_b, a.B, b) : ((r == 3 && c == 3) ? 1.0 : 0.0);
    const double Qt0 = in3 ? Qt : 0.0;                 // Q~^T with the (3,3) one removed: MF(Qt0, U, .) has a zero row 3
    const double Rt = in3 ? ld(a.R, ((size_t)blk * P + r) * P + c, a.R_b, a.B, b) : 0.0;
    const double RtT = in3 ? ld(a.R, ((size_t)blk * P + c) * P + r, a.R_b, a.B, b) : 0.0;   // R~^T
    const double Wr = r < 3 ? ld(a.W, (size_t)blk * P + r, a.W_b, a.B, b) : 0.0;          // W[k] at row k (all columns)
    const double Y0 = r < 3 ? ld(a.Q, ((size_t)blk * P + 0) * P + r, a.Q_b, a.B, b) : 0.0; // Q[0][k] at row k
    const double E0 = r == 0 ? 1.0 : 0.0;                                                  // selects row 0
    const double e3r = r == 3 ? 1.0 : 0.0;
    double th[RHS::NTHETA];
#pragma unroll
    for (int k = 0; k < RHS::NTHETA; ++k) th[k] = a.theta ? ld(a.theta, k, a.theta_b, a.B, b) : 0.0;
    // Tile form of the right-hand side (rhs.hpp): f_b = k0 v + k1 v^3 + k2 v' + k3, J_b0 = k4 + k5 v^2 in the block's own
    // and the other block's first state v, v'.  The measurement row X_w below (W_0 - J0 in row 0, W_r in rows 1 and 2,
    // the offset a = J0 v - f in row 3) is then ONE cubic per lane with lane-dependent coefficients,
    //     X_w = ((c3 v + c2) v + c1) v + (co v' + c0),
    // four fused multiply-adds instead of f, J0, a and two selects (nine): every VALU instruction lengthens the
    // step's dependent chain.  Same polynomials as interrogate.py:76-82, associated differently at rounding level.
    double c3 = 0.0, c2 = 0.0, c1 = 0.0, co = 0.0, c0 = Wr;
    if constexpr (rhs_has_tile_form<RHS>::value && D == 2) {
        double tk[6];
        RHS::tile_consts(blk, th, tk);
        const bool jac = ITG == RK_INTERROGATE_KRAMER;
        const double k4 = jac ? tk[4] : 0.0, k5 = jac ? tk[5] : 0.0;
        if (r == 3) { c3 = k5 - tk[1]; c1 = k4 - tk[0]; co = -tk[2]; c0 = -tk[3]; }
        if (r == 0) { c2 = -k5; c0 = Wr - k4; }
    }
    __shared__ double zbuf_all[NW][4 * 16];            // chkrebtii: z_0 of the next 16 steps for each of the wave's 4 tiles
    double* const zbuf = zbuf_all[wave_in_wg];
    __shared__ double vx[2][NW > 1 ? TILE_MAX_BLOCKS : 1];   // NW > 1: the blocks' evaluation points of this / the next step
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);

    // M_0 = [0 | ode_init ; 0 1]   (solve.py:53-54)
    double M = r < 3 ? (c == 3 ? ld(a.x0, (size_t)blk * P + r, a.x0_b, a.B, b) : 0.0) : (c == 3 ? 1.0 : 0.0);
    // lanes without a slot in the 3 x 4 tile: row 3, or tiles past the end
    const bool st = tc.valid && r < 3;
    const size_t tstride_all = (size_t)n_tiles * TILE_DOUBLES;
    // In the loop every lane stores through a buffer window on this wave's part of the time row (scalar base, no
    // per-lane pointer arithmetic: every VALU instruction lengthens the dependent chain); slot-less lanes are out of
    // range and dropped by the hardware.
    const char* row = (const char*)(tiles + (NW == 1 ? (size_t)blockIdx.x * TPW : (size_t)blockIdx.x * D + (size_t)wave_in_wg * 4) * TILE_DOUBLES);
    const int bvoff = st ? (int)((tc.g * TILE_DOUBLES + r * 4 + c) * sizeof(double)) : (int)0x80000000;
    auto store_row = [&](double v) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, TPW * TILE_DOUBLES * 8, 0x00020000);
        u32x2 bits;
        __builtin_memcpy(&bits, &v, 8);
        __builtin_amdgcn_raw_buffer_store_b64(bits, rsrc, bvoff, 0, 0);
    };
    // The state of time n goes out one step LATE, behind the first MFMA of step n + 1: a store issued right behind the
    // instruction that produced its data holds the wave's (in-order) issue until that result can be read by the memory
    // path -- 22 cycles on a bare add-MFMA-MFMA chain against 8 when the value is a step old
    // (profiles/r02_probe13_store_cost.log); on the headline kernel 0.483 -> 0.432 ms.  The empty asm makes the stored
    // value "depend" on that MFMA, so the scheduler keeps the store behind it; no instruction or wait state is emitted
    // (a macro, not a lambda: with a by-value copy of M hipcc's schedule of the headline step loses the whole gain).
#define RK_STORE_BEHIND(M_, mfma_result)                              \
    do {                                                              \
        asm("" : "+v"(M_) : "v"(mfma_result));                        \
        store_row(M_);                                                \
        row += tstride_all * sizeof(double);                          \
    } while (0)

    if constexpr (rhs_has_tile_form<RHS>::value && D == 2 && ITG != RK_INTERROGATE_CHKREBTII) {
        // The headline path.  A step is ONE dependent chain for its wave (seven MFMAs, thirteen VALU instructions, one
        // store), and with few trajectories (at most one wave per SIMD) its latency is the run time.  What was measured
        // about that chain (profiles/r01_probe2/3/6/8/9*.log): nothing of the same wave overlaps an fp64 MFMA but another,
        // independent MFMA; a dependent MFMA or any VALU instruction behind an MFMA waits its full 29 cycles; a dependent
        // VALU operation costs about 7.  Hence: as few instructions as possible (merged cubic for X_w, cubic reciprocal
        // step, buffer store), and the statement order below, which is the one hipcc's scheduler turns into the
        // shortest stream (240 cycles per step; its choices for other source orders, and every order pinned with
        // sched_barriers, measured 250-300).  A loop rotated to take U and the evaluation point off the chain with two
        // more MFMAs per step measured 344.
        // The order is pinned with empty asm statements that make an input of the next instruction "depend" on the result
        // of the previous one: no instruction is emitted, the variables keep their registers, hipcc still inserts the
        // wait states, and the stream no longer changes with unrelated edits (sched_barriers cost 12-24 cycles per
        // step; of 220 random valid orders in profiles/r01_probe9_fwd_order_search.log this one was the fastest).
#define RK_AFTER(in, res) asm("" : "+v"(in) : "v"(res))
        for (int n = 0; n < a.N; ++n) {
            double U = MF(M, Qt, 0.0);                              // (Q~ M)^T                         (standard.py:57-59)
            RK_STORE_BEHIND(M, U);                                  // the state of time n
            double B0 = MF(Y0, M, 0.0);                             // row 0 of Q~ M in every row: mu-_0 in column 3
            RK_AFTER(U, B0);
            double MpT = MF(Qt0, U, RtT);                           // M-^T with row 3 zeroed: the offset entry of X_w must not enter Z0
            RK_AFTER(B0, MpT);
            const double v_own = quad_bcast3(B0);                   // the point the ODE is evaluated at, X[b][0]
            const double v_oth = pair_other_quad_uniform(v_own);
            const double Xw = fma(fma(fma(c3, v_own, c2), v_own, c1), v_own, fma(co, v_oth, c0));
            RK_AFTER(U, Xw);
            double Mp = MF(U, Qt, Rt);                              // M- = Q~ M Q~^T + R~
            RK_AFTER(MpT, Mp);
            double Z0 = MF(MpT, Xw, 0.0);                           // Sigma- W~^T                      (standard.py:97)
            RK_AFTER(Mp, Z0);
            double WS = MF(Xw, Mp, 0.0);                            // [W~ Sigma- | W~ mu- + a]
            RK_AFTER(Z0, WS);
            double S = MF(Z0, Xw, 0.0);
            RK_AFTER(WS, S);
            if constexpr (ITG == RK_INTERROGATE_RODEO) S = S + S;   // var_meas = W Sigma- W^T (interrogate.py:110-113)
            const double PW = Z0 * WS;
            RK_AFTER(S, PW);
            const double y0 = __builtin_amdgcn_rcp(S);
            const double e = fma(-S, y0, 1.0);
            const double y = fma(y0, fma(e, e, e), y0);             // 1 / S (linalg_small.hpp, fast_rcp_cubic)
            M = fma(-PW, y, Mp);                                    // [Sigma- - K (W~ Sigma-) | mu- - K yhat]  (standard.py:98-102)
        }
        store_row(M);                                               // time N
#undef RK_AFTER
#ifdef RK_PLACEMENT_DEBUG
        // experiment build only (scripts/placement_probe.py): where this wave ran, into its slice of the scratch tail
        if (lane == 0) {
            double* tail = tiles + (size_t)(a.N + 1) * tstride_all + (size_t)blockIdx.x * 64;
            tail[0] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
            tail[1] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
        }
#endif
        return;
    }
    if constexpr (rhs_has_tile_form<RHS>::value && D == 2 && ITG == RK_INTERROGATE_CHKREBTII) {
        // The pseudo-marginal path (BASELINE config 4).  interrogate_chkrebtii has no Jacobian term (wgt_meas = 0), so the
        // measurement row is the constant W and NONE of the step's MFMAs depends on the draw: all seven are issued back to
        // back (U, M-, M-^T, row 0 of M-, Sigma- W^T, W Sigma-, S), and the draw x_0 = mu-_0 + sqrt(Sigma-_00) z_0
        // (interrogate.py:22-34), f(x) and the offset a = -f only touch column 3 of [W Sigma- | W mu- + a] afterwards.
        // What the step costs (round 4, experiment builds scripts/c4_ablation.sh, C4's 1024 draws x 800 steps): 150 us; without
        // the Philox / Box-Muller burst every 16 steps 126; without the square root of Sigma-_00 either 103.  Built on that and
        // measured in round 4, none faster than this kernel: (i) a generator wave per chain wave (two-wave workgroups, and
        // four-wave workgroups of two chains and two generators, LDS-only barriers, placement primer): 155 / 175 us -- the
        // generator's fp64 work lands on SIMDs that carry chains; (ii) covariance and mean on separate waves
        // (interrogate_chkrebtii's gain sequence does not depend on the mean; hand-off of Sigma- W^T, S, Sigma-_00 through LDS,
        // square roots and reciprocals of a chunk taken off the chain) with and without a third generator wave: each chain
        // alone steps in ~270 cycles -- the MFMA dependency chain U -> M-^T -> Sigma- W^T -> S plus the reciprocal is the
        // kramer step's -- and two chains meet on a SIMD: 156-183 us; (iii) the generator as a resumable computation (normal_pair cut
        // into 13 pieces, one per step, for the draws of the NEXT 16 steps, two z buffers): 141 -> 138 us -- a VALU instruction behind
        // an fp64 MFMA waits for it (probe3), so the pieces have no shadow to go into and cost their own issue time wherever they
        // stand; not worth its second code path.  Kept: the one-correction square root below.
        double tk[6];
        RHS::tile_consts(blk, th, tk);
        const double ac3 = -tk[1], ac1 = -tk[0], aco = -tk[2], ac0 = -tk[3];      // a = -f as the cubic of the generic path's row 3
        const double e3c = c == 3 ? 1.0 : 0.0;
        for (int n = 0; n < a.N; ++n) {
            // z_0 of this step first: its LDS latency hides behind the MFMAs
#if defined(RK_T3_ABLATE) && RK_T3_ABLATE >= 1            // experiment builds (scripts/c4_ablation.sh): no generator
            const double zn = 0.25;
#else
            if ((n & 15) == 0) {                            // the 16 lanes of a tile draw z_0 for 16 consecutive steps
                double z0, z1;
                normal_pair(a.seed, traj, (uint32_t)(n + r * 4 + c), (uint32_t)blk, PURPOSE_INTERROGATE, 0u, z0, z1);
                zbuf[tc.g * 16 + r * 4 + c] = z0;
            }
            const double zn = zbuf[tc.g * 16 + (n & 15)];
#endif
            const double U = MF(M, Qt, 0.0);
            RK_STORE_BEHIND(M, U);
            const double Mp = MF(U, Qt, Rt);
            const double MpT = MF(Qt0, U, RtT);
            const double R0 = MF(E0, Mp, 0.0);              // row 0 of M- in every row: [Sigma-_00 .. | mu-_0]
            double Z0 = MF(MpT, Wr, 0.0);                   // Sigma- W^T                          (standard.py:97)
            const double WS0 = MF(Wr, Mp, 0.0);             // [W Sigma- | W mu-]
            asm("" : "+v"(Z0) : "v"(WS0));                  // (keeps the last MFMA behind this one: hipcc moved WS0 to the end of the step)
            double S = MF(Z0, Wr, 0.0);
            S = S + S;                                      // + var_meas = W Sigma- W^T            (interrogate.py:26-29)
            const double rS = fast_rcp_cubic(S);            // (ahead of the draw's chain, which is independent of it: -3 %)
            double r0v = R0;
            asm("" : "+v"(r0v) : "v"(rS));
#if defined(RK_T3_ABLATE) && RK_T3_ABLATE >= 2            // ... and no square root
            const double v_own = fma(quad_bcast0(r0v), zn, quad_bcast3(r0v));
#else
            const double v_own = fma(sqrt_pos_1step(quad_bcast0(r0v)), zn, quad_bcast3(r0v));
#endif
            const double v_oth = pair_other_quad_uniform(v_own);
            const double am = fma(fma(fma(ac3, v_own, 0.0), v_own, ac1), v_own, fma(aco, v_oth, ac0));     // mean_meas = -f(x, t)
            const double WS = fma(e3c, am, WS0);            // yhat = W mu- + a in column 3         (standard.py:93)
            const double PW = Z0 * WS;
            M = fma(-PW, rS, Mp);                           // standard.py:98-102
        }
        store_row(M);
        return;
    }
    double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0, l4 = 0.0, XwL = 0.0;
    if constexpr (rhs_has_tile3_form<RHS>::value && D == 3 && NW == 1) {
        double kk[5];
        RHS::tile3_consts(blk, th, kk);
        const bool jac = ITG == RK_INTERROGATE_KRAMER;
        l0 = jac ? 0.0 : -kk[0]; l1 = -kk[1]; l2 = -kk[2]; l3 = -kk[3]; l4 = -kk[4];          // a = -f + J0 own
        XwL = fma(jac ? -kk[0] : 0.0, E0, Wr);                                                  // W~ = W - J: constant rows
    }
    for (int n = 0; n < a.N; ++n) {
        // ---- predict (standard.py:57-59): U = (Q~ M)^T, M- = Q~ M Q~^T + R~; B0 = row 0 of Q~ M in every row ----
        // (a 4x4x4 fp64 MFMA blocks this wave's issue for ~17 cycles = 4 fp64 VALU ops, and nothing overlaps it --
        //  profiles/r01_probe3_mfma_valu_serialize.log -- so the step is written with the fewest MFMAs: seven)
        const double U = MF(M, Qt, 0.0);
        RK_STORE_BEHIND(M, U);
        double v_own;                                  // the point the ODE is evaluated at: X[b][0] of this tile's block
        if constexpr (ITG != RK_INTERROGATE_CHKREBTII) v_own = quad_bcast3(MF(Y0, M, 0.0));   // mu-_0 in all 16 lanes
        const double Mp = MF(U, Qt, Rt);
        // exact transpose of M- (Q~ M^T Q~^T + R~^T) with its row 3 (= mu-^T) zeroed, so that the offset entry of X_w
        // below does not enter Sigma- W~^T
        const double MpT = MF(Qt0, U, RtT);
        if constexpr (ITG == RK_INTERROGATE_CHKREBTII) {
            // interrogate.py:22-34: x ~ N(mu-, Sigma-) with the lower factor; only x_0 = mu-_0 + sqrt(Sigma-_00) z_0
            // reaches f (RHS::NDEP == 1).  The 16 lanes of a tile draw z_0 for 16 consecutive steps at once.
            if ((n & 15) == 0) {
                double z0, z1;
                normal_pair(a.seed, traj, (uint32_t)(n + r * 4 + c), (uint32_t)blk, PURPOSE_INTERROGATE, 0u, z0, z1);
                zbuf[tc.g * 16 + r * 4 + c] = z0;
            }
            const double zn = zbuf[tc.g * 16 + (n & 15)];
            const double R0 = MF(E0, Mp, 0.0);         // row 0 of M- in every row: [Sigma-_00 .. | mu-_0]
            const double s00 = quad_bcast0(R0);
            v_own = fma(sqrt(s00 > 0.0 ? s00 : 0.0), zn, quad_bcast3(R0));
        }
        // ---- interrogation (interrogate.py): f and the block-diagonal Jacobian at v_own ----
        const double t = a.t_min + (a.t_max - a.t_min) * (double)(n + 1) / (double)a.N;     // solve.py:74
        double Xw;      // X_w[k] (row form): W~_k = W_k - J_k for k < 3 (solve.py:79, interrogate.py:80), a = -f + J mu- at k = 3
        if constexpr (rhs_has_tile_form<RHS>::value && D == 2) {
            const double v_oth = pair_other_quad_uniform(v_own);        // v_own is uniform in each quad
            Xw = fma(fma(fma(c3, v_own, c2), v_own, c1), v_own, fma(co, v_oth, c0));
        } else if constexpr (rhs_has_tile3_form<RHS>::value && D == 3 && NW == 1) {
            // Lorenz63-type right-hand sides (the form of fwd_tile4_kernel's trimmed step): the Jacobian entry is a constant,
            // the offset a bilinear form of this block's and its neighbours' evaluation points (three plain DPP row rotations)
            const double n1 = from_next_tile(v_own), p1 = from_prev_tile(v_own), p2 = dpp64<0x128>(v_own);
            const double a_meas = fma(l4, p2 * p1, fma(l3, p1 * n1, fma(l2, p1, fma(l1, n1, l0 * v_own))));
            Xw = fma(a_meas, e3r, XwL);
        } else {
            // (one column: the tile path takes right-hand sides that read X[b][0] only, NDEP = 1 -- with all P columns the dual
            //  copy of a 32-block system is 3 KB per lane and went to scratch: 45 us per forward step, round 4)
            constexpr int PX = 1;
            double X[D][PX];
            if constexpr (NW == 1) {
                double vals[D];
                gather_blocks<D>(v_own, vals);
#pragma unroll
                for (int bb = 0; bb < D; ++bb) X[bb][0] = vals[bb];
            } else {
                if (tc.valid && r == 0 && c == 0) vx[n & 1][blk] = v_own;
                __syncthreads();
#pragma unroll
                for (int bb = 0; bb < D; ++bb) X[bb][0] = vx[n & 1][bb];
            }
            double fb, J0;
            if constexpr (ITG == RK_INTERROGATE_KRAMER && rhs_has_fjac0<RHS>::value) {
                RHS::template fjac0_block<PX>(X, t, th, blk, fb, J0);    // one evaluation, one dual direction (dual.hpp)
            } else if constexpr (ITG != RK_INTERROGATE_KRAMER && rhs_has_f_block<RHS>::value) {
                fb = RHS::template f_block<PX>(X, t, th, blk);           // this lane's block alone
                J0 = 0.0;
            } else {
                double f[D], J[D][PX];
                if constexpr (ITG == RK_INTERROGATE_KRAMER) {
                    RHS::template fjac<PX>(X, t, th, f, J);
                } else {
                    RHS::template f<PX>(X, t, th, f);
#pragma unroll
                    for (int bb = 0; bb < D; ++bb)
#pragma unroll
                        for (int j = 0; j < PX; ++j) J[bb][j] = 0.0;
                }
                double J0s[D];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) J0s[bb] = J[bb][0];
                fb = pick_block<D>(f, blk); J0 = pick_block<D>(J0s, blk);
            }
            const double a_meas = fma(J0, v_own, -fb);                  // mean_meas (interrogate.py:81-82)
            Xw = fma(-J0, E0, fma(a_meas, e3r, Wr));                    // rows: W_0 - J0, W_1, W_2, a
        }
        // ---- update (standard.py:93-102) ----
        const double WS = MF(Xw, Mp, 0.0);                          // [W~ Sigma- | W~ mu- + a]   (column form)
        const double Z0 = MF(MpT, Xw, 0.0);                         // Sigma- W~^T (standard.py:97; row form, 0 in row 3)
        double S = MF(Z0, Xw, 0.0);
        if constexpr (ITG == RK_INTERROGATE_RODEO || ITG == RK_INTERROGATE_CHKREBTII)
            S = S + S;                                              // var_meas = W Sigma- W^T (interrogate.py:110-113, 26-29)
        const double K = Z0 * fast_rcp_cubic(S);
        M = fma(-K, WS, Mp);
    }
    store_row(M);
#undef RK_STORE_BEHIND
}

}  // namespace rk
