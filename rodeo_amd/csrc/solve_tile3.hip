// MFMA-tile solver kernels for n_bstate = 3, n_bmeas = 1, n_block in {1, 2}: the headline FitzHugh-Nagumo path.
//
//   src/rodeo/solve.py:31-122   _solve_filter -> fwd_tile3_kernel      one wave = 4 (trajectory, block) tiles
//   src/rodeo/solve.py:257-301  solve_mv      -> bwd_mv_tile3_kernel   producer wave (gain, time-parallel)
//                                                                      + consumer wave (carry recursion, MFMA)
//
// State of one block as an augmented 4x4 tile spread over 16 lanes (mfma_tile.hpp):
//        M = [ Sigma  mu ]      predict (standard.py:57-59) in two MFMAs:  U = MF(M, Qt) = (Q~ M)^T,
//            [   0     1 ]                                                  M- = MF(U, Qt, R~) = Q~ M Q~^T + R~
// with Q~ = diag(Q, 1), Qt = Q~^T, R~ = diag(R, 0).  The update (standard.py:93-102) with W~ = W + wgt_meas extended by
// the offset a = mean_meas as a 4th entry (M-[3][:] = e_3) needs three more MFMAs:
//        WS[c] = sum_k X[k] M-[k][c]   (c < 3: W~ Sigma- ; c = 3: W~ mu- + a = yhat)      column form
//        Z[r]  = sum_{k<3} M-[r][k] X[k]  = Sigma- W~^T, row form, from the exact transpose M-^T = MF(Qt, U, R~^T)
//                                       (using W~ Sigma- for it lets the antisymmetric rounding part of Sigma drift)
//        S     = sum_{k<3} Z[k] X[k] + V
//        M     = M- - (Z / S) WS       -> [ Sigma- - K (W~ Sigma-) | mu- - K yhat ]
// HBM format ("tile layout"): per time step and tile the 3 x 4 block [Sigma | mu] row-major = 96 B, exactly the
// algorithmic d*p*(p+1)*8 bytes; a wave's store is 4 x 96 contiguous bytes.
//
// Backward (solve.py:279-301): G_n = Sigma_f Q^T (Sigma-)^{-1} does not depend on the carry, so producer waves
// evaluate it for 16 time steps x 4 tiles at once (one lane per item, register LU with partial pivoting exactly as the
// reference's utils.py:119) and hand [M_f | M- | G~^T] tiles to the consumer wave through LDS.  The consumer's
// per-step dependent chain is then  D = Ms - M- ; V1 = MF(D, Gt) = (G~ D)^T ; Ms = MF(V1, Gt, M_f)
// (standard.py:213-216 for mean and variance at once, G~ = diag(G, 1)).
#include "common.hpp"
#include "kalman_small.hpp"
#include "mfma_tile.hpp"
#include "philox.hpp"
#include "rhs.hpp"
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

template <int D>
__device__ __forceinline__ TileCoord tile_coord(int wave, int lane, int n_tiles) {
    TileCoord t;
    t.r = lane >> 4; t.g = (lane >> 2) & 3; t.c = lane & 3;
    const int tau = wave * 4 + t.g;
    t.valid = tau < n_tiles;
    t.tau = t.valid ? tau : n_tiles - 1;
    t.b = t.tau / D; t.blk = t.tau - t.b * D;
    return t;
}

// ---- forward ---------------------------------------------------------------------------------------------------
template <class RHS, int ITG>
__global__ void __launch_bounds__(64) fwd_tile3_kernel(SolveArgs a, double* __restrict__ tiles) {
    constexpr int D = RHS::D, P = 3;
    static_assert(D == 1 || D == 2, "tile path: n_block in {1, 2}");
    static_assert(RHS::NDEP == 1, "tile path: right-hand sides that depend on X[b][0] only");
    const int n_tiles = a.B * D;
    const TileCoord tc = tile_coord<D>(blockIdx.x, threadIdx.x, n_tiles);
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
    if constexpr (RHS::HAS_TILE_FORM && D == 2) {
        double tk[6];
        RHS::tile_consts(blk, th, tk);
        const bool jac = ITG == RK_INTERROGATE_KRAMER;
        const double k4 = jac ? tk[4] : 0.0, k5 = jac ? tk[5] : 0.0;
        if (r == 3) { c3 = k5 - tk[1]; c1 = k4 - tk[0]; co = -tk[2]; c0 = -tk[3]; }
        if (r == 0) { c2 = -k5; c0 = Wr - k4; }
    }
    __shared__ double zbuf[4 * 16];                    // chkrebtii: z_0 of the next 16 steps for each of the 4 tiles
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);

    // M_0 = [0 | ode_init ; 0 1]   (solve.py:53-54)
    double M = r < 3 ? (c == 3 ? ld(a.x0, (size_t)blk * P + r, a.x0_b, a.B, b) : 0.0) : (c == 3 ? 1.0 : 0.0);
    // lanes without a slot in the 3 x 4 tile: row 3, or tiles past the end
    const bool st = tc.valid && r < 3;
    const size_t tstride_all = (size_t)n_tiles * TILE_DOUBLES;
    double* const dump = tiles + (size_t)(a.N + 1) * tstride_all + (size_t)blockIdx.x * 64;
    dump[threadIdx.x] = r == 3 ? (c == 3 ? 1.0 : 0.0) : 0.0;       // row 3 = e_3 for the backward kernels' slot-less lanes
    // In the loop every lane stores through a 384-byte buffer window on this wave's part of the time row (scalar base,
    // no per-lane pointer arithmetic: every VALU instruction lengthens the dependent chain); slot-less lanes are out of
    // range and dropped by the hardware.
    const char* row = (const char*)(tiles + (size_t)blockIdx.x * 4 * TILE_DOUBLES);
    const int bvoff = st ? (int)((tc.g * TILE_DOUBLES + r * 4 + c) * sizeof(double)) : (int)0x80000000;
    auto store_row = [&](double v) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, 4 * TILE_DOUBLES * 8, 0x00020000);
        u32x2 bits;
        __builtin_memcpy(&bits, &v, 8);
        __builtin_amdgcn_raw_buffer_store_b64(bits, rsrc, bvoff, 0, 0);
    };
    store_row(M);

    if constexpr (RHS::HAS_TILE_FORM && D == 2 && ITG != RK_INTERROGATE_CHKREBTII) {
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
            RK_AFTER(M, U);
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
            row += tstride_all * sizeof(double);
            store_row(M);
        }
#undef RK_AFTER
        return;
    }
    for (int n = 0; n < a.N; ++n) {
        // ---- predict (standard.py:57-59): U = (Q~ M)^T, M- = Q~ M Q~^T + R~; B0 = row 0 of Q~ M in every row ----
        // (a 4x4x4 fp64 MFMA blocks this wave's issue for ~17 cycles = 4 fp64 VALU ops, and nothing overlaps it --
        //  profiles/r01_probe3_mfma_valu_serialize.log -- so the step is written with the fewest MFMAs: seven)
        const double U = MF(M, Qt, 0.0);
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
        if constexpr (RHS::HAS_TILE_FORM && D == 2) {
            const double v_oth = pair_other_quad_uniform(v_own);        // v_own is uniform in each quad
            Xw = fma(fma(fma(c3, v_own, c2), v_own, c1), v_own, fma(co, v_oth, c0));
        } else {
            double X[D][P];
#pragma unroll
            for (int bb = 0; bb < D; ++bb)
#pragma unroll
                for (int j = 0; j < P; ++j) X[bb][j] = 0.0;
            if constexpr (D == 1) {
                X[0][0] = v_own;
            } else {
                X[0][0] = pair_block0(v_own);
                X[1][0] = pair_block1(v_own);
            }
            double f[D], J[D][P];
            if constexpr (ITG == RK_INTERROGATE_KRAMER) {
                RHS::template fjac<P>(X, t, th, f, J);
            } else {
                RHS::template f<P>(X, t, th, f);
#pragma unroll
                for (int bb = 0; bb < D; ++bb)
#pragma unroll
                    for (int j = 0; j < P; ++j) J[bb][j] = 0.0;
            }
            double fb = f[0], J0 = J[0][0];
            if constexpr (D == 2) { fb = blk == 0 ? f[0] : f[1]; J0 = blk == 0 ? J[0][0] : J[1][0]; }
            const double a_meas = fma(J0, v_own, -fb);                  // mean_meas (interrogate.py:81-82)
            Xw = fma(-J0, E0, fma(a_meas, e3r, Wr));                    // rows: W_0 - J0, W_1, W_2, a
        }
        // ---- update (standard.py:93-102) ----
        const double WS = MF(Xw, Mp, 0.0);                          // [W~ Sigma- | W~ mu- + a]   (column form)
        const double Z0 = MF(MpT, Xw, 0.0);                         // Sigma- W~^T (standard.py:97; row form, 0 in row 3)
        double S = MF(Z0, Xw, 0.0);
        if constexpr (ITG == RK_INTERROGATE_RODEO || ITG == RK_INTERROGATE_CHKREBTII)
            S = S + S;                                              // var_meas = W Sigma- W^T (interrogate.py:110-113, 26-29)
        const double K = Z0 * fast_rcp(S);
        M = fma(-K, WS, Mp);
        row += tstride_all * sizeof(double);
        store_row(M);
    }
}

// ---- backward: one consumer wave + three producer waves per 4 tiles -------------------------------------------------
// Time runs in "ticks" separated by workgroup barriers; in tick t the consumer smooths chunk t (16 steps) while the
// producers prepare later chunks.  Producer q (of three) owns the chunks ch = q (mod 3); its work on chunk ch is three
// stages in the ticks ch-3, ch-2, ch-1 (fetch + predict; T^T and the LU's forward sweep; back substitution + hand-off),
// so in every tick the three producers each run a different stage of three different chunks.  Chunk ch is handed over
// in LDS buffer ch & 1 (written during tick ch-1, read during tick ch).
constexpr int CHUNK = 16;                       // time steps per hand-off
constexpr int ITEM_BYTES = 3 * 128;              // per (step, tile): M-, G~^T, M_f tiles of 16 doubles
constexpr int BUF_BYTES = CHUNK * 4 * ITEM_BYTES;   // 24 KiB
constexpr int ZONE_BYTES = 64 * TILE_DOUBLES * 8;   // 6 KiB: one producer's prefetched filt tiles (64 lanes x 96 B)

// byte offset inside a buffer of element idx (= 4 r + c) of tile `which` of item (s, g)
__device__ __forceinline__ int lds_byte(int s, int g, int which, int idx) {
    const int item = s * 4 + g;
    return item * ITEM_BYTES + which * 128 + (tile_slot(s, g, idx) << 3);
}

// Workgroup = 4 waves for 4 tiles: wave 0 consumes, waves 1..3 produce; a workgroup's waves go to the CU's four SIMDs
// one each, and the CU holds two or three such workgroups (48 KiB of LDS each) that run out of step with each
// other, which evens out the load of the SIMDs (a speed consideration only -- any placement gives the same results).
__global__ void __launch_bounds__(256) bwd_mv_tile3_kernel(SolveArgs a, double* __restrict__ tiles, int D) {
    constexpr int P = 3;
    __shared__ __attribute__((aligned(16))) char lds_all[2 * BUF_BYTES];
    __shared__ __attribute__((aligned(16))) char zones[3 * ZONE_BYTES];      // the producers' prefetch landing zones
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;    // 0 = consumer; producers q = wave - 1 own ch = q (mod 3)
    const int n_tiles = a.B * D;
    const size_t tstride = (size_t)n_tiles * TILE_DOUBLES;
    const int n_chunks = (a.N - 1 + CHUNK - 1) / CHUNK;            // steps n = N-1 .. 1

    // constant entries of the hand-off tiles (row 3 = e_3; column 3 of G~^T = e_3) are written once
    for (int i = threadIdx.x; i < 2 * 64 * 3 * 16; i += 256) {
        const int idx = i & 15, which = (i >> 4) % 3, item = ((i >> 4) / 3) & 63, buf = i / (64 * 3 * 16);
        const double v = (idx == 15) ? 1.0 : 0.0;
        *(double*)(lds_all + buf * BUF_BYTES + lds_byte(item >> 2, item & 3, which, idx)) = v;
    }
    __syncthreads();

    const int tw = blockIdx.x;                                     // tile-wave index: tiles 4 tw .. 4 tw + 3
    char* const lds_raw = lds_all;
    // this tile-wave's 64-double slice of the scratch tail (row 3 of its tiles: e_3)
    double* const dump = tiles + (size_t)(a.N + 1) * tstride + (size_t)tw * 64;

    if (wave >= 1) {
        // ---------------- producers: one lane per (step-in-chunk, tile) ----------------
        const int p = wave - 1;
        const int s = lane >> 2, g = lane & 3;
        int tau = tw * 4 + g;
        if (tau >= n_tiles) tau = n_tiles - 1;
        const int b = tau / D, blk = tau - b * D;
        double Q[P][P], R[P][P];
        load_block_consts<P>(a, blk, b, Q, R);
        int woff[12];                                              // LDS byte offsets of the 12 slots this lane writes
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) woff[i * 4 + j] = lds_byte(s, g, 0, i * 4 + j);
        // The filt tiles of this producer's next chunk are prefetched by LDS-DMA into the wave's own 6 KiB landing zone
        // right after stage 1 has read the zone, three ticks before they are needed; no prefetch lives in registers
        // (mfma_tile.hpp, lds_dma16).  The zone is the image of the chunk's 16 rows x 384 contiguous bytes (this
        // tile-wave's 4 tiles in 16 time rows): piece j = 64 i + lane of instruction i is bytes 16 (j % 24) of row
        // j / 24, so one instruction reads 2 2/3 whole rows (measured: 64 scattered 16-byte pieces per instruction
        // cost 185-280 cycles of issue each, whole rows about a quarter of that).
        char* const zone = zones + p * ZONE_BYTES;
        const unsigned zone_lds = __builtin_amdgcn_readfirstlane(lds_addr(zone));
        int frow[6], fcol[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { const int j = 64 * i + lane; frow[i] = j / 24; fcol[i] = (j % 24) * 16; }
        const char* const wave_tiles = (const char*)(tiles + (size_t)tw * 4 * TILE_DOUBLES);
        auto fetch = [&](int ch) {
            const int n_hi = a.N - 1 - ch * CHUNK;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int n = n_hi - frow[i];                        // rows past the start (n < 1) are clamped, never handed over
                lds_dma16(wave_tiles + (size_t)(n < 1 ? 1 : n) * tstride * 8 + fcol[i], zone_lds + 1024 * i);
            }
        };
        lds_dma_wait_all();                                        // retire the loads of Q, R before the first DMA
        if (p < n_chunks) fetch(p);
        // Chunk ch passes through three stages in the ticks ch-3, ch-2, ch-1 (one workgroup barrier per tick), so in
        // every tick the three producers each run a different stage of three different chunks: equal work per SIMD
        // and tick.  All state between stages stays in this wave's registers.
        double mf[P], Sf[P][P], mp[P], Sp[P][P], A[P][P], X[P][P], rpiv[P];
        for (int t = -3; t < n_chunks; ++t) {
            const int ch1 = t + 3, ch2 = t + 2, ch3 = t + 1;
            if (ch1 % 3 == p) {
                // ---- stage 1 of chunk ch1: take the fetched tiles, start the next fetch, predict ----
                if (ch1 < n_chunks) {
                    lds_dma_wait_all();
                    double buf[TILE_DOUBLES];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double2 v = *(const double2*)(zone + 96 * lane + 16 * k);      // lane = 4 s + g
                        buf[2 * k] = v.x; buf[2 * k + 1] = v.y;
                    }
                    lds_reads_done();
                    if (ch1 + 3 < n_chunks) fetch(ch1 + 3);
#pragma unroll
                    for (int i = 0; i < P; ++i) {
#pragma unroll
                        for (int j = 0; j < P; ++j) Sf[i][j] = buf[i * 4 + j];
                        mf[i] = buf[i * 4 + 3];
                    }
                    predict_block<P>(Q, R, mf, Sf, mp, Sp);          // pred[n+1] from filt[n]   (standard.py:57-59)
                }
            } else if (ch2 >= 0 && ch2 % 3 == p) {
                // ---- stage 2 of chunk ch2: T^T = (Sigma_f Q^T)^T (standard.py:175), LU of Sigma-, forward sweep ----
                if (ch2 < n_chunks) {
                    double T[P][P];
                    mm_nt<P, P, P>(Sf, Q, T);
#pragma unroll
                    for (int i = 0; i < P; ++i)
#pragma unroll
                        for (int j = 0; j < P; ++j) { A[i][j] = Sp[i][j]; X[i][j] = T[j][i]; }
                    lu_factor_fwd<P, P>(A, X, rpiv);
                }
            } else if (ch3 >= 0) {
                // ---- stage 3 of chunk ch3: back substitution, X = solve(Sigma-, T^T) = G^T (standard.py:176), hand-off ----
                if (ch3 < n_chunks) {
                    lu_back<P, P>(A, X, rpiv);
                    const int n = a.N - 1 - ch3 * CHUNK - s;
                    if (n >= 1) {
                        char* o = lds_raw + (ch3 & 1) * BUF_BYTES;
#pragma unroll
                        for (int i = 0; i < P; ++i) {
#pragma unroll
                            for (int j = 0; j < P; ++j) {
                                *(double*)(o + woff[i * 4 + j]) = Sp[i][j];             // M-   (which = 0)
                                *(double*)(o + woff[i * 4 + j] + 128) = X[i][j];        // G~^T (which = 1)
                                *(double*)(o + woff[i * 4 + j] + 256) = Sf[i][j];       // M_f  (which = 2)
                            }
                            *(double*)(o + woff[i * 4 + 3]) = mp[i];
                            *(double*)(o + woff[i * 4 + 3] + 256) = mf[i];
                        }
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // ---------------- consumer: the carry recursion on MFMA tiles ----------------
        const TileCoord tc = tile_coord<1>(tw, lane, n_tiles);              // (b, blk) not needed here
        __builtin_amdgcn_s_setprio(3);      // the dependent chain is the critical path: win issue arbitration on this SIMD
        const int r = tc.r, g = tc.g, c = tc.c, idx = r * 4 + c;
        const bool st = tc.valid && r < 3;
        // lanes without a slot (row 3, tiles past the end) read and write the scratch tail with stride 0: row 3 of
        // every tile is e_3 there (stored by the forward kernel and re-stored here), so no masking is needed
        double* base = st ? tiles + (size_t)tc.tau * TILE_DOUBLES + idx : dump + lane;
        const size_t ostride = st ? tstride : 0;
        int roff[4];                                                // per-lane LDS byte offsets for s & 3 = 0..3
#pragma unroll
        for (int k = 0; k < 4; ++k) roff[k] = lds_byte(k, g, 0, idx) - k * 4 * ITEM_BYTES;
        double Ms = base[(size_t)a.N * ostride];                    // carry = filt[N]  (solve.py:279-282)
        // full chunks store with a scalar row pointer + this lane's byte offset in the tile-wave's 384 bytes
        const char* const wave_rows = (const char*)(tiles + (size_t)tw * 4 * TILE_DOUBLES);
        const size_t row_bytes = tstride * sizeof(double);
        const unsigned bvoff = st ? (unsigned)((g * TILE_DOUBLES + idx) * sizeof(double)) : 0x80000000u;   // or: out of range
        const bool buffer_ok = (CHUNK - 1) * row_bytes + 384 < 0x7fffffffull;      // one chunk's rows within a 2 GiB buffer window
        const int chunk_span = (int)((CHUNK - 1) * row_bytes + 384);
        __syncthreads();                                            // tick -3
        __syncthreads();                                            // tick -2
        __syncthreads();                                            // tick -1: chunk 0 is in LDS
        for (int t = 0; t < n_chunks; ++t) {
            const char* in = lds_raw + (t & 1) * BUF_BYTES;
            const int n_hi = a.N - 1 - t * CHUNK;
            const int cnt = __builtin_amdgcn_readfirstlane(n_hi >= CHUNK ? CHUNK : n_hi);   // steps n_hi .. n_hi-cnt+1
            double* o = base + (size_t)n_hi * ostride;
            {
                if (cnt == CHUNK && buffer_ok) {
                    // full chunk, branch-free (immediate LDS offsets): the 16-step dependent chain
                    //     D = Ms - M- ; V = MF(D, G~^T) ; Ms = MF(V, G~^T, M_f)              (standard.py:213-216)
                    // Every instruction of this wave sits on that chain (VALU, LDS and memory instructions do not
                    // overlap its MFMAs: profiles/r01_probe6_bwd_consumer.log), so a step is two MFMAs, one add, three
                    // LDS reads issued LOOKAHEAD steps early (at most 15 LDS operations can be outstanding per wave,
                    // so reading everything up front only stalls) and one buffer store: scalar row offset, no pointer
                    // arithmetic, lanes without a slot dropped by the range check.
                    constexpr int LOOKAHEAD = 4;
                    double Mp[CHUNK], Gt[CHUNK], Mf[CHUNK];
                    auto load = [&](int s) {
                        const char* q = in + roff[s & 3] + s * 4 * ITEM_BYTES;
                        Mp[s] = *(const double*)(q);
                        Gt[s] = *(const double*)(q + 128);
                        Mf[s] = *(const double*)(q + 256);
                    };
#pragma unroll
                    for (int s = 0; s < LOOKAHEAD; ++s) load(s);
                    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(wave_rows + (size_t)(n_hi - (CHUNK - 1)) * row_bytes), 0, chunk_span, 0x00020000);
                    double Dm = Ms - Mp[0];
#pragma unroll
                    for (int s = 0; s < CHUNK; ++s) {
                        if (s + LOOKAHEAD < CHUNK) load(s + LOOKAHEAD);
                        __builtin_amdgcn_sched_barrier(0);
                        const double V1 = MF(Dm, Gt[s], 0.0);           // (G~ D)^T
                        Ms = MF(V1, Gt[s], Mf[s]);                      // G~ D G~^T + M_f
                        if (s + 1 < CHUNK) Dm = Ms - Mp[s + 1];
                        u32x2 bits;
                        __builtin_memcpy(&bits, &Ms, 8);
                        __builtin_amdgcn_raw_buffer_store_b64(bits, rsrc, (int)bvoff, (int)((CHUNK - 1 - s) * row_bytes), 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    for (int s = 0; s < cnt; ++s) {
                        const char* q = in + lds_byte(s, g, 0, idx);
                        const double Mp = *(const double*)(q), Gt = *(const double*)(q + 128), Mf = *(const double*)(q + 256);
                        const double V1 = MF(Ms - Mp, Gt, 0.0);
                        Ms = MF(V1, Gt, Mf);
                        o[0] = Ms;
                        o -= ostride;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- backward sampler (solve.py:162-204): x_n = mu_f + G (x_{n+1} - mu-) + L~ z_n --------------------------------
// Everything but the chain in x is carry-independent, so the producers (same three-stage scheme as above) evaluate per
// (step, tile): G, mu-, and mu_f + L~ z with L~ = psd_factor(Sigma_f - G T^T) (standard.py:248-254, the draw of
// solve.py:179) and the Philox normals; the consumer's dependent chain is ONE MFMA per step:
//     x = MF(G~^T, x - mu-, mu_f + L~ z)        (x, mu-, ... in row form: lane (r, g, c) holds component r)
// The terminal draw x_N ~ N(filt[N]) (solve.py:182-186) is the same step with G = 0.  Hand-off item: 256 B.
constexpr int SIM_ITEM = 256;                   // [G~^T tile 128 B | 128 B vector area: mu-, mu_f + L z (swizzled)]
constexpr int SIM_BUF = CHUNK * 4 * SIM_ITEM;   // 16 KiB

__device__ __forceinline__ int sim_tile_byte(int s, int g, int idx) {
    const int item = s * 4 + g;
    return item * SIM_ITEM + (tile_slot(s, g, idx) << 3);
}
__device__ __forceinline__ int sim_vec_byte(int s, int g, int which, int rr) {
    const int item = s * 4 + g;
    return item * SIM_ITEM + 128 + (vec_slot(s, g, which, rr) << 3);
}

// Same workgroup structure as bwd_mv_tile3_kernel: wave 0 consumes, waves 1..3 produce in three stages per chunk
// (the Philox normals, the bulk of the producers' work, are split over stages 1 and 2).
__global__ void __launch_bounds__(256) bwd_sim_tile3_kernel(SolveArgs a, double* __restrict__ tiles, int D) {
    constexpr int P = 3;
    __shared__ __attribute__((aligned(16))) char lds_all[2 * SIM_BUF];
    __shared__ __attribute__((aligned(16))) char zones[3 * ZONE_BYTES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_tiles = a.B * D;
    const size_t tstride = (size_t)n_tiles * TILE_DOUBLES;
    const int n_chunks = (a.N + CHUNK - 1) / CHUNK;                // steps n = N .. 1 (n = N is the terminal draw)
    for (int i = threadIdx.x; i < 2 * SIM_BUF / 8; i += 256) ((double*)lds_all)[i] = 0.0;
    __syncthreads();
    const int tw = blockIdx.x;
    char* const lds_raw = lds_all;
    double* const dump = tiles + (size_t)(a.N + 1) * tstride + (size_t)tw * 64;

    if (wave >= 1) {
        // ---------------- producers ----------------
        const int p = wave - 1;
        const int s = lane >> 2, g = lane & 3;
        int tau = tw * 4 + g;
        if (tau >= n_tiles) tau = n_tiles - 1;
        const int b = tau / D, blk = tau - b * D;
        const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
        double Q[P][P], R[P][P];
        load_block_consts<P>(a, blk, b, Q, R);
        int woff[9], voff[3], voff1[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) woff[i * 3 + j] = sim_tile_byte(s, g, i * 4 + j);
            voff[i] = sim_vec_byte(s, g, 0, i);
            voff1[i] = sim_vec_byte(s, g, 1, i);
        }
        // LDS-DMA prefetch of the chunk's filt tiles into this wave's landing zone (see bwd_mv_tile3_kernel)
        char* const zone = zones + p * ZONE_BYTES;
        const unsigned zone_lds = __builtin_amdgcn_readfirstlane(lds_addr(zone));
        int frow[6], fcol[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { const int j = 64 * i + lane; frow[i] = j / 24; fcol[i] = (j % 24) * 16; }
        const char* const wave_tiles = (const char*)(tiles + (size_t)tw * 4 * TILE_DOUBLES);
        auto fetch = [&](int ch) {
            const int n_hi = a.N - ch * CHUNK;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int n = n_hi - frow[i];
                lds_dma16(wave_tiles + (size_t)(n < 1 ? 1 : n) * tstride * 8 + fcol[i], zone_lds + 1024 * i);
            }
        };
        lds_dma_wait_all();                                        // retire the loads of Q, R before the first DMA
        if (p < n_chunks) fetch(p);
        double mf[P], Sf[P][P], mp[P], Sp[P][P], T[P][P], A[P][P], X[P][P], rpiv[P], z[P];
        for (int t = -3; t < n_chunks; ++t) {
            const int ch1 = t + 3, ch2 = t + 2, ch3 = t + 1;
            if (ch1 % 3 == p) {
                // ---- stage 1 of chunk ch1: fetched tiles, next fetch, predict, the first two normals ----
                if (ch1 < n_chunks) {
                    lds_dma_wait_all();
                    double buf[TILE_DOUBLES];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double2 v = *(const double2*)(zone + 96 * lane + 16 * k);      // lane = 4 s + g
                        buf[2 * k] = v.x; buf[2 * k + 1] = v.y;
                    }
                    lds_reads_done();
                    if (ch1 + 3 < n_chunks) fetch(ch1 + 3);
#pragma unroll
                    for (int i = 0; i < P; ++i) {
#pragma unroll
                        for (int j = 0; j < P; ++j) Sf[i][j] = buf[i * 4 + j];
                        mf[i] = buf[i * 4 + 3];
                    }
                    predict_block<P>(Q, R, mf, Sf, mp, Sp);              // pred[n+1] from filt[n]   (standard.py:57-59)
                    const int n = a.N - ch1 * CHUNK - s;
                    normal_pair(a.seed, traj, (uint32_t)(n < 1 ? 1 : n), (uint32_t)blk, PURPOSE_SMOOTH, 0u, z[0], z[1]);
                }
            } else if (ch2 >= 0 && ch2 % 3 == p) {
                // ---- stage 2 of chunk ch2: T (standard.py:175), LU of Sigma- with the forward sweep, the third normal ----
                if (ch2 < n_chunks) {
                    mm_nt<P, P, P>(Sf, Q, T);
#pragma unroll
                    for (int i = 0; i < P; ++i)
#pragma unroll
                        for (int j = 0; j < P; ++j) { A[i][j] = Sp[i][j]; X[i][j] = T[j][i]; }
                    lu_factor_fwd<P, P>(A, X, rpiv);
                    const int n = a.N - ch2 * CHUNK - s;
                    double z3;
                    normal_pair(a.seed, traj, (uint32_t)(n < 1 ? 1 : n), (uint32_t)blk, PURPOSE_SMOOTH, 1u, z[2], z3);
                }
            } else if (ch3 >= 0) {
                // ---- stage 3 of chunk ch3: G (standard.py:176), the conditional draw (standard.py:248-254), hand-off ----
                if (ch3 < n_chunks) {
                    const int n = a.N - ch3 * CHUNK - s;
                    double GT[P][P], Ssim[P][P], L[P][P], G[P][P];
                    lu_back<P, P>(A, X, rpiv);                            // X = G^T
                    const bool term = n == a.N;                           // terminal draw: G = 0, var = filt[N]
#pragma unroll
                    for (int i = 0; i < P; ++i)
#pragma unroll
                        for (int j = 0; j < P; ++j) G[i][j] = term ? 0.0 : X[j][i];
                    mm_nt<P, P, P>(G, T, GT);
#pragma unroll
                    for (int i = 0; i < P; ++i)
#pragma unroll
                        for (int j = 0; j < P; ++j) Ssim[i][j] = Sf[i][j] - GT[i][j];   // standard.py:253-254
                    psd_factor<P>(Ssim, L);
                    if (n >= 1) {
                        char* o = lds_raw + (ch3 & 1) * SIM_BUF;
#pragma unroll
                        for (int i = 0; i < P; ++i) {
                            double w = mf[i];
#pragma unroll
                            for (int k = 0; k <= i; ++k) w = fma(L[i][k], z[k], w);
#pragma unroll
                            for (int j = 0; j < P; ++j) *(double*)(o + woff[i * 3 + j]) = G[j][i];      // G~^T
                            *(double*)(o + voff[i]) = term ? 0.0 : mp[i];
                            *(double*)(o + voff1[i]) = w;                                               // mu_f + L z
                        }
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // ---------------- consumer ----------------
        const TileCoord tc = tile_coord<1>(tw, lane, n_tiles);
        const int r = tc.r, g = tc.g, c = tc.c, idx = r * 4 + c;
        const int b = tc.tau / D, blk = tc.tau - b * D;
        const bool st = tc.valid && r < 3 && c == 0;
        const size_t xstride = (size_t)D * P * a.B;
        double* bx = st ? a.x + ((size_t)blk * P + r) * a.B + b : dump + lane;
        const size_t sx = st ? xstride : 0;
        int roff[4], rvec[4], rvec1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            roff[k] = sim_tile_byte(k, g, idx) - k * 4 * SIM_ITEM;
            rvec[k] = sim_vec_byte(k, g, 0, r) - k * 4 * SIM_ITEM;
            rvec1[k] = sim_vec_byte(k, g, 1, r) - k * 4 * SIM_ITEM;
        }
        double x = 0.0;
        __syncthreads();
        __syncthreads();
        __syncthreads();
        for (int t = 0; t < n_chunks; ++t) {
            const char* in = lds_raw + (t & 1) * SIM_BUF;
            const int n_hi = a.N - t * CHUNK;
            const int cnt = __builtin_amdgcn_readfirstlane(n_hi >= CHUNK ? CHUNK : n_hi);
            double* o = bx + (size_t)n_hi * sx;
            auto step = [&](const char* q, const char* qv, const char* qw) {
                const double Gt = *(const double*)(q), mp = *(const double*)(qv), mfw = *(const double*)(qw);
                x = MF(Gt, x - mp, mfw);
                o[0] = x;
                o -= sx;
            };
            if (cnt == CHUNK) {
#pragma unroll
                for (int s = 0; s < CHUNK; ++s) step(in + roff[s & 3] + s * 4 * SIM_ITEM, in + rvec[s & 3] + s * 4 * SIM_ITEM,
                                                     in + rvec1[s & 3] + s * 4 * SIM_ITEM);
            } else {
                for (int s = 0; s < cnt; ++s) step(in + sim_tile_byte(s, g, idx), in + sim_vec_byte(s, g, 0, r), in + sim_vec_byte(s, g, 1, r));
            }
            __syncthreads();
        }
        // x[0] = ode_init exactly (solve.py:196-204): the mean column of tile time 0
        if (st) bx[0] = tiles[(size_t)tc.tau * TILE_DOUBLES + r * 4 + 3];
    }
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
// Note on occupancy: the forward recursion is one dependent chain per wave (about 290 cycles per step: 7 MFMAs + 27
// VALU operations issued in order), so a launch takes n_steps x that latency however few waves there are; fewer tiles
// per wave or more waves per workgroup change nothing (measured, round 1) -- only more trajectories raise throughput.
template <class RHS>
static int launch_fwd_tile(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles) {
    const dim3 grid(div_up(a.B * RHS::D, 4)), block(64);
    LaunchTimer t(h, "fwd_tile3_kernel");
    switch (c->interrogate) {
        case RK_INTERROGATE_KRAMER:
            hipLaunchKernelGGL((fwd_tile3_kernel<RHS, RK_INTERROGATE_KRAMER>), grid, block, 0, h->stream, a, tiles); break;
        case RK_INTERROGATE_SCHOBER:
            hipLaunchKernelGGL((fwd_tile3_kernel<RHS, RK_INTERROGATE_SCHOBER>), grid, block, 0, h->stream, a, tiles); break;
        case RK_INTERROGATE_RODEO:
            hipLaunchKernelGGL((fwd_tile3_kernel<RHS, RK_INTERROGATE_RODEO>), grid, block, 0, h->stream, a, tiles); break;
        case RK_INTERROGATE_CHKREBTII:
            hipLaunchKernelGGL((fwd_tile3_kernel<RHS, RK_INTERROGATE_CHKREBTII>), grid, block, 0, h->stream, a, tiles); break;
        default:
            set_error("tile path: interrogate id %d not supported", c->interrogate);
            return RK_ERR_UNSUPPORTED;
    }
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

bool tile3_supported(const rk_solve_cfg* c, int mode) {
    if (c->flags & (RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR)) return false;
    if (c->kalman_type != RK_KALMAN_STANDARD || c->n_bstate != 3 || c->n_bmeas != 1) return false;
    if (c->interrogate < RK_INTERROGATE_RODEO || c->interrogate > RK_INTERROGATE_CHKREBTII) return false;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) return c->n_block == 2;
    if (c->rhs_id == RK_RHS_HIGHER_ORDER) return c->n_block == 1;
    return false;
}

int tile3_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int mode) {
    int rc;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) rc = launch_fwd_tile<FitzHughNagumo>(h, c, a, tiles);
    else rc = launch_fwd_tile<HigherOrder>(h, c, a, tiles);
    if (rc || mode == RK_MODE_FILTER) return rc;
    if (mode == RK_MODE_SIM) {
        LaunchTimer t(h, "bwd_sim_tile3_kernel");
        hipLaunchKernelGGL(bwd_sim_tile3_kernel, dim3(div_up(a.B * a.D, 4)), dim3(256), 0, h->stream, a, tiles, a.D);
        t.stop();
        RK_HIP(hipGetLastError());
        return RK_OK;
    }
    if (a.N < 2) return rc;
    LaunchTimer t(h, "bwd_mv_tile3_kernel");
    hipLaunchKernelGGL(bwd_mv_tile3_kernel, dim3(div_up(a.B * a.D, 4)), dim3(256), 0, h->stream, a, tiles, a.D);
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

}  // namespace rk
