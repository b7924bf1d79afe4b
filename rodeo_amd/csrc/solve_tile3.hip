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
//        Z[r]  = sum_k M-[k][r] X[k]   (= W~ Sigma- again, row form; stands in for Sigma- W~^T -- Sigma- is symmetric up
//                                       to rounding, see DESIGN.md "symmetry use")
//        S     = sum_{k<3} Z[k] X[k] + V
//        M     = M- - (Z / S) WS       -> [ Sigma- - K (W~ Sigma-) | mu- - K yhat ]
// HBM format ("tile layout"): per time step and tile the 3 x 4 block [Sigma | mu] row-major = 96 B, exactly the
// algorithmic d*p*(p+1)*8 bytes; a wave's store is 4 x 96 contiguous bytes.
//
// Backward (solve.py:279-301): G_n = Sigma_f Q^T (Sigma-)^{-1} does not depend on the carry, so a producer wave
// evaluates it for 16 time steps x 4 tiles at once (one lane per item, register LU with partial pivoting exactly as the
// reference's utils.py:119) and hands [M_f | M- | G~^T] tiles to the consumer wave through LDS.  The consumer's
// per-step dependent chain is then  D = Ms - M- ; V1 = MF(D, Gt) = (G~ D)^T ; Ms = MF(V1, Gt, M_f)
// (standard.py:213-216 for mean and variance at once, G~ = diag(G, 1)).
#include "common.hpp"
#include "kalman_small.hpp"
#include "mfma_tile.hpp"
#include "rhs.hpp"
#include "solve_args.hpp"

namespace rk {

constexpr int TILE_DOUBLES = 12;     // 3 rows x [Sigma(3) | mu] per (time step, tile)

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
    const double Rt = in3 ? ld(a.R, ((size_t)blk * P + r) * P + c, a.R_b, a.B, b) : 0.0;
    const double Wr = r < 3 ? ld(a.W, (size_t)blk * P + r, a.W_b, a.B, b) : 0.0;          // row form
    const double Y0 = r < 3 ? ld(a.Q, ((size_t)blk * P + 0) * P + r, a.Q_b, a.B, b) : 0.0; // Q[0][k] at row k
    double th[RHS::NTHETA];
#pragma unroll
    for (int k = 0; k < RHS::NTHETA; ++k) th[k] = a.theta ? ld(a.theta, k, a.theta_b, a.B, b) : 0.0;

    // M_0 = [0 | ode_init ; 0 1]   (solve.py:53-54)
    double M = r < 3 ? (c == 3 ? ld(a.x0, (size_t)blk * P + r, a.x0_b, a.B, b) : 0.0) : (c == 3 ? 1.0 : 0.0);
    // lanes without a slot in the 3 x 4 tile (row 3, or tiles past the end) store to the 64-double scratch tail of the
    // buffer instead of being masked off: no exec-mask branch in the time loop
    const bool st = tc.valid && r < 3;
    const size_t tstride_all = (size_t)n_tiles * TILE_DOUBLES;
    double* out = st ? tiles + (size_t)tc.tau * TILE_DOUBLES + r * 4 + c
                     : tiles + (size_t)(a.N + 1) * tstride_all + threadIdx.x;
    const size_t tstride = st ? tstride_all : 0;
    out[0] = M;

    for (int n = 0; n < a.N; ++n) {
        // predict, and row 0 of Q~ M broadcast to all rows (its column 3 is mu-_0)
        const double U = MF(M, Qt, 0.0);
        const double B0 = MF(Y0, M, 0.0);
        const double Mp = MF(U, Qt, Rt);
        const double v_own = quad_bcast3(B0);
        // interrogation (interrogate.py): f and the block-diagonal Jacobian at mu-
        double X[D][P];
#pragma unroll
        for (int bb = 0; bb < D; ++bb)
#pragma unroll
            for (int j = 0; j < P; ++j) X[bb][j] = 0.0;
        if constexpr (D == 1) {
            X[0][0] = v_own;
        } else {
            const double v_prev = from_prev_tile(v_own), v_next = from_next_tile(v_own);
            const double v_oth = blk == 0 ? v_next : v_prev;
            X[0][0] = blk == 0 ? v_own : v_oth;
            X[1][0] = blk == 0 ? v_oth : v_own;
        }
        const double t = a.t_min + (a.t_max - a.t_min) * (double)(n + 1) / (double)a.N;     // solve.py:74
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
        // X_w[k] (row form): W~_k = W_k - J_k for k < 3 (solve.py:79, interrogate.py:80), a = -f + J mu- at k = 3
        const double a_meas = fma(J0, v_own, -fb);
        const double Xw = r == 0 ? Wr - J0 : (r == 3 ? a_meas : Wr);
        // update
        const double WS = MF(Xw, Mp, 0.0);
        const double Zr = MF(Mp, Xw, 0.0);
        const double Z0 = r == 3 ? 0.0 : Zr;
        double S = MF(Z0, Xw, 0.0);
        if constexpr (ITG == RK_INTERROGATE_RODEO) S = S + S;      // var_meas = W Sigma- W^T (interrogate.py:110-113)
        const double K = Z0 * fast_rcp(S);
        M = fma(-K, WS, Mp);
        out += tstride;
        out[0] = M;
    }
}

// ---- backward: producer / consumer -----------------------------------------------------------------------------
constexpr int CHUNK = 16;                       // time steps per hand-off
constexpr int SLOT = 3 * 16;                    // doubles per (step, tile): M_f, M-, G~^T tiles

__device__ __forceinline__ int lds_off(int s, int g, int which, int idx) {
    const int item = s * 4 + g;
    return (item * 3 + which) * 16 + (idx ^ (item & 15));      // XOR swizzle: conflict-free producer writes
}

__global__ void __launch_bounds__(128) bwd_mv_tile3_kernel(SolveArgs a, double* __restrict__ tiles, int D) {
    constexpr int P = 3;
    __shared__ double lds[2][CHUNK * 4 * SLOT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_tiles = a.B * D;
    const size_t tstride = (size_t)n_tiles * TILE_DOUBLES;
    const int n_back = a.N - 1;                                   // steps n = N-1 .. 1
    const int n_chunks = (n_back + CHUNK - 1) / CHUNK;

    // constant entries of the hand-off tiles (row 3 = e_3, zero padding) are written once
    for (int i = threadIdx.x; i < 2 * CHUNK * 4 * SLOT; i += 128) {
        const int buf = i / (CHUNK * 4 * SLOT), rem = i % (CHUNK * 4 * SLOT);
        const int item = rem / SLOT, which = (rem % SLOT) / 16, idx = rem % 16;
        const int rr = idx >> 2, cc = idx & 3;
        const double v = (rr == 3 && cc == 3) ? 1.0 : 0.0;
        lds[buf][(item * 3 + which) * 16 + (idx ^ (item & 15))] = v;
    }
    __syncthreads();

    if (wave == 1) {
        // ---------------- producer: one lane per (step-in-chunk, tile) ----------------
        const int s = lane >> 2, g = lane & 3;
        int tau = blockIdx.x * 4 + g;
        if (tau >= n_tiles) tau = n_tiles - 1;
        const int b = tau / D, blk = tau - b * D;
        double Q[P][P], R[P][P];
        load_block_consts<P>(a, blk, b, Q, R);
        const double* tin = tiles + (size_t)tau * TILE_DOUBLES;
        double cur[TILE_DOUBLES], nxt[TILE_DOUBLES];
        auto fetch = [&](int ch, double (&dst)[TILE_DOUBLES]) {
            int n = a.N - 1 - ch * CHUNK - s;
            n = n < 1 ? 1 : n;                                   // clamped loads are never used (n >= 1 check below)
            const double* in = tin + (size_t)n * tstride;
#pragma unroll
            for (int i = 0; i < TILE_DOUBLES; ++i) dst[i] = in[i];
        };
        fetch(0, cur);
        for (int ch = 0; ch <= n_chunks; ++ch) {
            if (ch + 1 < n_chunks) fetch(ch + 1, nxt);           // prefetch the next chunk under this chunk's compute
            if (ch < n_chunks) {
                const int n = a.N - 1 - ch * CHUNK - s;
                if (n >= 1) {
                    double mf[P], Sf[P][P];
#pragma unroll
                    for (int i = 0; i < P; ++i) {
#pragma unroll
                        for (int j = 0; j < P; ++j) Sf[i][j] = cur[i * 4 + j];
                        mf[i] = cur[i * 4 + 3];
                    }
                    double mp[P], Sp[P][P], T[P][P], G[P][P];
                    predict_block<P>(Q, R, mf, Sf, mp, Sp);          // pred[n+1] from filt[n]
                    smooth_gain<P>(Q, Sf, Sp, T, G);                 // standard.py:175-176 (pivoted LU)
                    double* o = lds[ch & 1];
#pragma unroll
                    for (int i = 0; i < P; ++i) {
#pragma unroll
                        for (int j = 0; j < P; ++j) {
                            o[lds_off(s, g, 0, i * 4 + j)] = Sf[i][j];
                            o[lds_off(s, g, 1, i * 4 + j)] = Sp[i][j];
                            o[lds_off(s, g, 2, i * 4 + j)] = G[j][i];      // G~^T
                        }
                        o[lds_off(s, g, 0, i * 4 + 3)] = mf[i];
                        o[lds_off(s, g, 1, i * 4 + 3)] = mp[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TILE_DOUBLES; ++i) cur[i] = nxt[i];
            __syncthreads();
        }
    } else {
        // ---------------- consumer: the carry recursion on MFMA tiles ----------------
        const TileCoord tc = tile_coord<1>(blockIdx.x, lane, n_tiles);      // (b, blk) not needed here
        const int r = tc.r, g = tc.g, c = tc.c, idx = r * 4 + c;
        const bool st = tc.valid && r < 3;
        // non-storing lanes write to the scratch tail (stride 0) instead of being masked off
        double* base = st ? tiles + (size_t)tc.tau * TILE_DOUBLES + idx : tiles + (size_t)(a.N + 1) * tstride + lane;
        const size_t ostride = st ? tstride : 0;
        // carry = filt[N]  (solve.py:279-282); row 3 = e_3
        double Ms = r < 3 ? tiles[(size_t)a.N * tstride + (size_t)tc.tau * TILE_DOUBLES + idx] : (c == 3 ? 1.0 : 0.0);
        __syncthreads();                                              // chunk 0 produced
        for (int ch = 0; ch < n_chunks; ++ch) {
            const double* in = lds[ch & 1];
            const int n_hi = a.N - 1 - ch * CHUNK;
            const int cnt = __builtin_amdgcn_readfirstlane(n_hi >= CHUNK ? CHUNK : n_hi);   // steps n_hi .. n_hi-cnt+1
            double* o = base + (size_t)n_hi * ostride;
            double Mf = in[lds_off(0, g, 0, idx)], Mp = in[lds_off(0, g, 1, idx)], Gt = in[lds_off(0, g, 2, idx)];
            for (int s = 0; s < cnt; ++s) {
                // software pipeline: next step's hand-off tiles are read from LDS under this step's MFMAs
                const int sn = s + 1 < CHUNK ? s + 1 : s;
                const double nMf = in[lds_off(sn, g, 0, idx)], nMp = in[lds_off(sn, g, 1, idx)],
                             nGt = in[lds_off(sn, g, 2, idx)];
                const double Dm = Ms - Mp;
                const double V1 = MF(Dm, Gt, 0.0);                    // (G~ D)^T
                Ms = MF(V1, Gt, Mf);                                  // G~ D G~^T + M_f   (standard.py:213-216)
                o[0] = Ms;
                o -= ostride;
                Mf = nMf; Mp = nMp; Gt = nGt;
            }
            __syncthreads();
        }
    }
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
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
    if (mode == 2) return false;                                   // solve_sim: batch-minor kernels
    if (c->kalman_type != RK_KALMAN_STANDARD || c->n_bstate != 3 || c->n_bmeas != 1) return false;
    if (c->interrogate != RK_INTERROGATE_KRAMER && c->interrogate != RK_INTERROGATE_SCHOBER &&
        c->interrogate != RK_INTERROGATE_RODEO)
        return false;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) return c->n_block == 2;
    if (c->rhs_id == RK_RHS_HIGHER_ORDER) return c->n_block == 1;
    return false;
}

int tile3_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int mode) {
    int rc;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) rc = launch_fwd_tile<FitzHughNagumo>(h, c, a, tiles);
    else rc = launch_fwd_tile<HigherOrder>(h, c, a, tiles);
    if (rc || mode == 0 || a.N < 2) return rc;
    LaunchTimer t(h, "bwd_mv_tile3_kernel");
    hipLaunchKernelGGL(bwd_mv_tile3_kernel, dim3(div_up(a.B * a.D, 4)), dim3(128), 0, h->stream, a, tiles, a.D);
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

}  // namespace rk
