// MFMA-tile solver kernels for n_bstate = 3, n_bmeas = 1, n_block <= 4: the headline FitzHugh-Nagumo path.
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
#include <cstdlib>
#include "common.hpp"
#include "kalman_small.hpp"
#include "mfma_tile.hpp"
#include "philox.hpp"
#include "rhs.hpp"
#include "solve_args.hpp"
#include "solve_tile3_kernels.hpp"

namespace rk {

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
// The producer waves of the backward kernels (bwd_mv_tile3_kernel, fenrir_bwd_tile3_kernel): wave 1 + q owns the chunks
// ch = q (mod 3) and hands [M-, G~^T, M_f] of every (step, tile) of a chunk over in LDS buffer ch & 1.
__device__ __forceinline__ void tile3_gain_producers(const SolveArgs& a, const double* __restrict__ tiles, int D, int tw, int wave,
                                                     int lane, int n_tiles, size_t tstride, int n_chunks, char* lds_raw,
                                                     char* zones) {
    constexpr int P = 3;
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
            // (INVARIANT of lds_barrier(): this landing zone is read by the wave that issued the DMA and by no other)
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
        lds_barrier();   
    }
}

__global__ void __launch_bounds__(256) bwd_mv_tile3_kernel(SolveArgs a, double* __restrict__ tiles, int D) {
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
    // (Tried: the chain wave elected by HW_ID -- the one on SIMD 0, so that the two chain waves of a CU's two workgroups share a
    // SIMD with each other and with no producer: 0.281 against 0.264 ms; two MFMA chains on one SIMD cost more than a chain and
    // a producer.)
    const int role = wave;

    if (role >= 1) {
        tile3_gain_producers(a, tiles, D, tw, role, lane, n_tiles, tstride, n_chunks, lds_raw, zones);
    } else {
        // ---------------- consumer: the carry recursion on MFMA tiles ----------------
        const TileCoord tc = tile_coord<1>(tw, lane, n_tiles);              // (b, blk) not needed here
        __builtin_amdgcn_s_setprio(3);      // the dependent chain is the critical path: win issue arbitration on this SIMD
        const int r = tc.r, g = tc.g, c = tc.c, idx = r * 4 + c;
        const bool st = tc.valid && r < 3;
        // lanes without a slot (row 3, tiles past the end) write the scratch tail with stride 0 in the pointer path
        double* base = st ? tiles + (size_t)tc.tau * TILE_DOUBLES + idx : dump + lane;
        const size_t ostride = st ? tstride : 0;
        int roff[4];                                                // per-lane LDS byte offsets for s & 3 = 0..3
#pragma unroll
        for (int k = 0; k < 4; ++k) roff[k] = lds_byte(k, g, 0, idx) - k * 4 * ITEM_BYTES;
        // carry = filt[N] (solve.py:279-282); lanes without a slot hold row 3 = e_3 of the augmented tile (or zeros)
        double Ms = st ? base[(size_t)a.N * ostride] : ((r == 3 && c == 3) ? 1.0 : 0.0);
        // full chunks store with a scalar row pointer + this lane's byte offset in the tile-wave's 384 bytes
        const char* const wave_rows = (const char*)(tiles + (size_t)tw * 4 * TILE_DOUBLES);
        const size_t row_bytes = tstride * sizeof(double);
        const unsigned bvoff = st ? (unsigned)((g * TILE_DOUBLES + idx) * sizeof(double)) : 0x80000000u;   // or: out of range
        const bool buffer_ok = (CHUNK - 1) * row_bytes + 384 < 0x7fffffffull;      // one chunk's rows within a 2 GiB buffer window
        const int chunk_span = (int)((CHUNK - 1) * row_bytes + 384);
        lds_barrier();                                               // tick -3
        lds_barrier();                                               // tick -2
        lds_barrier();                                               // tick -1: chunk 0 is in LDS
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
            lds_barrier();   
        }
    }
}

// ---- Fenrir backward pass on the tile path (src/rodeo/inference/fenrir.py:86-259) -------------------------------------
// The backward Markov chain of smooth_cond maps (A = G, b = mu_f - G mu-, C = Sigma_f - G T^T, standard.py:366-370) run
// as a Kalman filter backwards in time IS the carry recursion of smooth_mv started from filt[N]:
//     A M A^T + C = G M G^T + (Sigma_f - G Sigma- G^T)      (T^T = Sigma- G^T),
// interrupted by an observation update wherever an observation sits on the grid.  So the kernel is bwd_mv_tile3_kernel
// with the same producers, a consumer that stores nothing, conditions on the observations (rare: a slow per-step path
// for the chunks that contain one) and accumulates their log-densities.  Predicted moments are re-evaluated from the
// filtered ones like everywhere on the tile path, so the forward pass is the MFMA-tile forward kernel.
struct FenrirObs {
    const double *obs, *obs_w, *obs_v;      // (n_obs, D), (n_obs, D, 3), (n_obs, D)
    const int32_t* obs_ind;                 // (n_obs,) ascending grid indices
    int n_obs;
    double* logdens;                        // (B,), zeroed by the caller
};

__global__ void __launch_bounds__(256) fenrir_bwd_tile3_kernel(SolveArgs a, const double* __restrict__ tiles, int D, FenrirObs ob) {
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

    if (wave >= 1) {
        tile3_gain_producers(a, tiles, D, tw, wave, lane, n_tiles, tstride, n_chunks, lds_raw, zones);
    } else {
        // ---------------- consumer: the backward filter on MFMA tiles ----------------
        __builtin_amdgcn_s_setprio(3);
        const int r = lane >> 4, g = (lane >> 2) & 3, c = lane & 3, idx = r * 4 + c;
        const int tau_raw = tw * 4 + g;
        const bool valid = tau_raw < n_tiles;
        const int tau = valid ? tau_raw : n_tiles - 1;
        const int b = tau / D, blk = tau - b * D;
        const bool in_tile = valid && r < 3;
        const double e3 = (r == 3 && c == 3) ? 1.0 : 0.0, I4 = r == c ? 1.0 : 0.0;
        const double* const my = tiles + (size_t)tau * TILE_DOUBLES + idx;
        int roff[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) roff[k] = lds_byte(k, g, 0, idx) - k * 4 * ITEM_BYTES;
        double Ms = in_tile ? my[(size_t)a.N * tstride] : e3;       // terminal point filt[N]   (fenrir.py:186-188)
        double acc = 0.0;
        int i = ob.n_obs - 1;
        int next = i >= 0 ? ob.obs_ind[i] : -1;                     // grid index of the next observation (backwards in time)
        // Conditioning on observation i (forecast standard.py:333-335, log-density, update standard.py:93-102) is the
        // forward update in tile form with the measurement row X_w = [D_0, D_1, D_2 | -y] and var_meas = Omega:
        //     WS = MF(X_w, M) = [D Sigma | D mu - y] ; Z = MF(M^T with row 3 zeroed, X_w) = Sigma D^T ; w = MF(Z, X_w) + Omega
        // (the transpose is one MFMA with the identity).  Observations are rare, so this path is not tuned.
        auto observe = [&]() {
            const double xw = valid ? (r < 3 ? ob.obs_w[((size_t)i * D + blk) * 3 + r] : -ob.obs[(size_t)i * D + blk]) : 0.0;
            const double Om = valid ? ob.obs_v[(size_t)i * D + blk] : 1.0;
            double MsT = MF(Ms, I4, 0.0);
            MsT = r == 3 ? 0.0 : MsT;
            const double WS = MF(xw, Ms, 0.0);
            const double Z = MF(MsT, xw, 0.0);
            const double w = MF(Z, xw, 0.0) + Om;                               // var_fore
            const double z = -quad_bcast3(WS);                                  // y - D mu
            if (fabs(w) > 1e-8) acc += -0.5 * (z * z / w + log(w)) - 0.5 * 1.83787706640934548356;   // utils.py:60-78
            const double K = Z / w;                                             // solve_var with a 1 x 1 system
            Ms = fma(-K, WS, Ms);
            --i;
            next = i >= 0 ? ob.obs_ind[i] : -1;
        };
        if (next >= a.N) observe();                                 // fenrir.py:189-209
        __syncthreads();                                            // tick -3
        __syncthreads();                                            // tick -2
        __syncthreads();                                            // tick -1: chunk 0 is in LDS
        for (int t = 0; t < n_chunks; ++t) {
            const char* in = lds_raw + (t & 1) * BUF_BYTES;
            const int n_hi = a.N - 1 - t * CHUNK;
            const int cnt = __builtin_amdgcn_readfirstlane(n_hi >= CHUNK ? CHUNK : n_hi);   // steps n_hi .. n_hi-cnt+1
            const bool has_obs = next > n_hi - cnt;
            if (cnt == CHUNK && !has_obs) {
                // the 16-step dependent chain of bwd_mv_tile3_kernel, nothing stored:
                //     D = M - M- ; V = MF(D, G~^T) ; M = MF(V, G~^T, M_f)   = G M G^T + (M_f - G M- G^T)  (standard.py:366-370)
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
                double Dm = Ms - Mp[0];
#pragma unroll
                for (int s = 0; s < CHUNK; ++s) {
                    if (s + LOOKAHEAD < CHUNK) load(s + LOOKAHEAD);
                    __builtin_amdgcn_sched_barrier(0);
                    const double V1 = MF(Dm, Gt[s], 0.0);
                    Ms = MF(V1, Gt[s], Mf[s]);
                    if (s + 1 < CHUNK) Dm = Ms - Mp[s + 1];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int s = 0; s < cnt; ++s) {
                    const char* q = in + lds_byte(s, g, 0, idx);
                    const double Mp = *(const double*)(q), Gt = *(const double*)(q + 128), Mf = *(const double*)(q + 256);
                    const double V1 = MF(Ms - Mp, Gt, 0.0);
                    Ms = MF(V1, Gt, Mf);
                    if (next == n_hi - s) observe();                            // fenrir.py:155-170
                }
            }
            __syncthreads();
        }
        // n = 0: filt[0] = (ode_init, 0) has G = 0, so the state there is filt[0] itself whatever came before
        if (next == 0) {
            Ms = in_tile ? my[0] : e3;
            observe();
        }
        if (valid && r == 0 && c == 0) atomicAdd(&ob.logdens[b], acc);
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
// Log-posterior tail of the user-level log-density (docs/examples/parameter.md:188-210, 331-354) folded into the sampler's
// consumer wave (LP): it holds x_n of every step anyway, so the terms  norm.logpdf(obs_k, x_{n(k)}[blk][0], noise_sd)  are added
// where the step index meets the next observation index (ascending indices walked from the end; observations and indices in
// LDS), the blocks of a trajectory are summed across their tiles, the N(0, prior_sd^2) terms of the first n_prior unconstrained
// parameters are added, and ONE double per trajectory leaves -- no path (a.x = NULL: nothing of x is stored), no second
// launch, no host round trip between sampler and reduction (C4: 275 -> 240 us per 1024 draws, of which 30 were the upload
// of the parameters between the two launches).  Needs the tiles of a trajectory inside one wave: n_block in {1, 2, 4}.
struct SimLogpost {
    const double* obs;                      // (n_obs, D) row-major
    const int32_t* obs_ind;                 // (n_obs,) ascending grid indices (clamped to [0, N])
    int n_obs;
    double noise_sd;
    const double* upars;                    // (n_prior, B) batch-minor or NULL
    int n_prior;
    double prior_sd;
    double* logpost;                        // (B,)
};
constexpr int LP_MAX_OBS = 512, LP_MAX_VALS = 1024;

template <bool LP>
__global__ void __launch_bounds__(256, 2) bwd_sim_tile3_kernel(SolveArgs a, double* __restrict__ tiles, int D, SimLogpost lp) {
    constexpr int P = 3;
    __shared__ double lp_obs[LP ? LP_MAX_VALS : 1];
    __shared__ int lp_ind[LP ? LP_MAX_OBS : 1];
    if constexpr (LP) {
        for (int i = threadIdx.x; i < lp.n_obs * D; i += 256) lp_obs[i] = lp.obs[i];
        for (int i = threadIdx.x; i < lp.n_obs; i += 256) { const int ni = lp.obs_ind[i]; lp_ind[i] = ni < 0 ? 0 : (ni > a.N ? a.N : ni); }
    }
    __shared__ __attribute__((aligned(16))) char lds_all[2 * SIM_BUF];
    __shared__ __attribute__((aligned(16))) char zones[3 * ZONE_BYTES];
    // Q | R of the workgroup's four tiles, read by the producers where they are used instead of living in 36 registers
    // across the ticks (the kernel had 257 registers, i.e. one workgroup per CU; capped at 256 it spilled two of them,
    // and a spill reload waits for every outstanding LDS-DMA load)
    constexpr int QR_STRIDE = 2 * 9 + 2;
    __shared__ __attribute__((aligned(16))) double qr[4 * QR_STRIDE];
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
        if (p == 0 && s == 0) {
            double Q0[P][P], R0[P][P];
            load_block_consts<P>(a, blk, b, Q0, R0);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int j = 0; j < P; ++j) { qr[g * QR_STRIDE + i * P + j] = Q0[i][j]; qr[g * QR_STRIDE + 9 + i * P + j] = R0[i][j]; }
        }
        lds_barrier();                                           // (matched by the consumer's first barrier)
        const double* const my_qr = qr + g * QR_STRIDE;
        auto load_q = [&](double (&Qv)[P][P]) {
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int j = 0; j < P; ++j) Qv[i][j] = my_qr[i * P + j];
        };
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
                // (INVARIANT of lds_barrier(): this landing zone is read by the wave that issued the DMA and by no other)
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
                    {
                        double Q[P][P], R[P][P];
                        load_q(Q);
#pragma unroll
                        for (int i = 0; i < P; ++i)
#pragma unroll
                            for (int j = 0; j < P; ++j) R[i][j] = my_qr[9 + i * P + j];
                        predict_block<P>(Q, R, mf, Sf, mp, Sp);          // pred[n+1] from filt[n]   (standard.py:57-59)
                    }
                    const int n = a.N - ch1 * CHUNK - s;
#ifdef RK_T3_ABLATE_SIMZ                                       // experiment builds (scripts/c4_ablation.sh): the sampler without its generator
                    z[0] = 0.25 + 1e-3 * n; z[1] = -0.5;
#else
                    normal_pair(a.seed, traj, (uint32_t)(n < 1 ? 1 : n), (uint32_t)blk, PURPOSE_SMOOTH, 0u, z[0], z[1]);
#endif
                }
            } else if (ch2 >= 0 && ch2 % 3 == p) {
                // ---- stage 2 of chunk ch2: T (standard.py:175), LU of Sigma- with the forward sweep, the third normal ----
                if (ch2 < n_chunks) {
                    {
                        double Q[P][P];
                        load_q(Q);
                        mm_nt<P, P, P>(Sf, Q, T);
                    }
#pragma unroll
                    for (int i = 0; i < P; ++i)
#pragma unroll
                        for (int j = 0; j < P; ++j) { A[i][j] = Sp[i][j]; X[i][j] = T[j][i]; }
                    lu_factor_fwd<P, P>(A, X, rpiv);
                    const int n = a.N - ch2 * CHUNK - s;
                    double z3;
#ifdef RK_T3_ABLATE_SIMZ
                    z[2] = 0.125 + 1e-3 * n; z3 = 0.0;
#else
                    normal_pair(a.seed, traj, (uint32_t)(n < 1 ? 1 : n), (uint32_t)blk, PURPOSE_SMOOTH, 1u, z[2], z3);
#endif
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
#ifdef RK_T3_ABLATE_SIMPSD                                     // ... and without the factor of the conditional variance
#pragma unroll
                    for (int i = 0; i < P; ++i)
#pragma unroll
                        for (int j = 0; j < P; ++j) L[i][j] = Ssim[i][j];
#else
                    psd_factor<P>(Ssim, L);
#endif
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
            lds_barrier();                                          // (LDS only: the DMA prefetch of three ticks ahead stays in flight)
        }
    } else {
        // ---------------- consumer ----------------
        const TileCoord tc = tile_coord<1>(tw, lane, n_tiles);
        const int r = tc.r, g = tc.g, c = tc.c, idx = r * 4 + c;
        const int b = tc.tau / D, blk = tc.tau - b * D;
        const bool st = tc.valid && r < 3 && c == 0 && a.x != nullptr;       // (a.x = NULL: only the log-posterior is wanted)
        const size_t xstride = (size_t)D * P * a.B;
        double* bx = st ? a.x + ((size_t)blk * P + r) * a.B + b : dump + lane;
        // log-posterior: the next observation to meet (indices ascending, walked from the end), this lane's running sum
        const double LOG_SQRT_2PI = 0.91893853320467274178;
        const double lsd = LP ? log(lp.noise_sd) : 0.0;
        int io = LP ? lp.n_obs - 1 : -1;
        double acc = 0.0;
        auto next_index = [&]() { return io >= 0 ? __builtin_amdgcn_readfirstlane(lp_ind[io]) : -1; };
        auto observe = [&](double xv) {                             // scipy.stats.norm.logpdf(obs, loc = x, scale = noise_sd)
            const double zz = (lp_obs[io * D + blk] - xv) / lp.noise_sd;
            acc += -0.5 * zz * zz - lsd - LOG_SQRT_2PI;
            --io;
        };
        const size_t sx = st ? xstride : 0;
        const bool xbuf_ok = (size_t)CHUNK * xstride * sizeof(double) < 0x7fffffffull;
        const int xvo = st && xbuf_ok ? (int)((((size_t)blk * P + r) * a.B + b) * sizeof(double)) : (int)0x80000000;
        int roff[4], rvec[4], rvec1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            roff[k] = sim_tile_byte(k, g, idx) - k * 4 * SIM_ITEM;
            rvec[k] = sim_vec_byte(k, g, 0, r) - k * 4 * SIM_ITEM;
            rvec1[k] = sim_vec_byte(k, g, 1, r) - k * 4 * SIM_ITEM;
        }
        double x = 0.0;
        lds_barrier();                                            // Q | R of the tiles (and the observations) are in LDS
        int next_n = LP ? next_index() : -1;
        lds_barrier();
        lds_barrier();
        lds_barrier();
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
            int n_cur = n_hi;                                       // time index of the step being drawn
            if (cnt == CHUNK && xbuf_ok) {
                // whole chunk: the draws leave through a buffer window on the chunk's 16 time rows of x (scalar base,
                // constant per-lane offset, lanes without a slot out of range) -- no 64-bit pointer arithmetic on the chain
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(a.x + (size_t)(n_hi - (CHUNK - 1)) * xstride), 0, (int)(CHUNK * xstride * sizeof(double)), 0x00020000);
                double xs[LP ? CHUNK : 1];
#pragma unroll
                for (int s = 0; s < CHUNK; ++s) {
                    const double Gt = *(const double*)(in + roff[s & 3] + s * 4 * SIM_ITEM);
                    const double mp = *(const double*)(in + rvec[s & 3] + s * 4 * SIM_ITEM);
                    const double mfw = *(const double*)(in + rvec1[s & 3] + s * 4 * SIM_ITEM);
                    x = MF(Gt, x - mp, mfw);
                    u32x2 bits;
                    __builtin_memcpy(&bits, &x, 8);
                    // (only the log-posterior wanted, a.x = NULL: the store stays, out of range -- a uniform branch around it cost 6 us)
                    __builtin_amdgcn_raw_buffer_store_b64(bits, rsrc, xvo, (int)((CHUNK - 1 - s) * xstride * sizeof(double)), 0);
                    if constexpr (LP) xs[s] = x;
                }
                if constexpr (LP) {
                    // the chunk's observation terms BEHIND its chain (a compare-and-branch per step inside it cost the sampler
                    // 12 of its 85 us: it breaks the chunk's schedule of LDS reads ahead of the MFMAs)
                    if (next_n > n_hi - CHUNK) {
#pragma unroll
                        for (int s = 0; s < CHUNK; ++s)
                            while (n_hi - s == next_n) { observe(xs[s]); next_n = next_index(); }
                    }
                }
            } else if (cnt == CHUNK) {
#pragma unroll
                for (int s = 0; s < CHUNK; ++s) {
                    step(in + roff[s & 3] + s * 4 * SIM_ITEM, in + rvec[s & 3] + s * 4 * SIM_ITEM, in + rvec1[s & 3] + s * 4 * SIM_ITEM);
                    if constexpr (LP) {
                        while (n_cur == next_n) { observe(x); next_n = next_index(); }
                    }
                    --n_cur;
                }
            } else {
                for (int s = 0; s < cnt; ++s) {
                    step(in + sim_tile_byte(s, g, idx), in + sim_vec_byte(s, g, 0, r), in + sim_vec_byte(s, g, 1, r));
                    if constexpr (LP) {
                        while (n_cur == next_n) { observe(x); next_n = next_index(); }
                    }
                    --n_cur;
                }
            }
            lds_barrier();
        }
        // x[0] = ode_init exactly (solve.py:196-204): the mean column of tile time 0
        const double x_init = tiles[(size_t)tc.tau * TILE_DOUBLES + r * 4 + 3];
        if (st) bx[0] = x_init;
        if constexpr (LP) {
            while (next_n == 0) { observe(x_init); next_n = next_index(); }
            // the row-0 lanes of a tile hold its block's sum; the blocks of a trajectory are neighbouring tiles of this wave
            if (D >= 2) acc += __shfl_xor(acc, 4, 64);
            if (D >= 4) acc += __shfl_xor(acc, 8, 64);
            if (tc.valid && r == 0 && c == 0 && blk == 0) {
                if (lp.upars) {
                    const double lps = log(lp.prior_sd);
                    for (int k = 0; k < lp.n_prior; ++k) {
                        const double zz = lp.upars[(size_t)k * a.B + b] / lp.prior_sd;
                        acc += -0.5 * zz * zz - lps - LOG_SQRT_2PI;
                    }
                }
                lp.logpost[b] = acc;
            }
        }
    }
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
// Note on occupancy: the forward recursion is one dependent chain per wave (about 290 cycles per step: 7 MFMAs + 27
// VALU operations issued in order), so a launch takes n_steps x that latency however few waves there are; fewer tiles
// per wave or more waves per workgroup change nothing (measured, round 1) -- only more trajectories raise throughput.
template <class RHS>
static int launch_fwd_tile(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles) {
    const dim3 grid(div_up(a.B * RHS::D, Tpw<RHS::D>::value)), block(64);
    launch_placement_primer(h, grid, block);           // (common.hpp: exact one-wave-per-SIMD placement behind any kernel)
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

bool is_user_rhs(int rhs_id);
bool user_tile_available(const rk_solve_cfg* c, int which);
int user_forward_tile(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int which);

bool tile3_supported(const rk_solve_cfg* c, int mode) {
    if (c->flags & (RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR)) return false;
    if (c->kalman_type != RK_KALMAN_STANDARD || c->n_bstate != 3 || c->n_bmeas != 1) return false;
    if (c->interrogate < RK_INTERROGATE_RODEO || c->interrogate > RK_INTERROGATE_CHKREBTII) return false;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) return c->n_block == 2;
    if (c->rhs_id == RK_RHS_LORENZ63) return c->n_block == 3;
    if (c->rhs_id == RK_RHS_HIGHER_ORDER) return c->n_block == 1;
    if (is_user_rhs(c->rhs_id)) return user_tile_available(c, 3);       // hiprtc build of fwd_tile3_kernel (rhs_jit.hip)
    return false;
}

bool tile3_sim_logpost_supported(const rk_solve_cfg* c, int n_obs) {
    return tile3_supported(c, RK_MODE_SIM) && (c->n_block == 1 || c->n_block == 2 || c->n_block == 4) && n_obs <= LP_MAX_OBS &&
           n_obs * c->n_block <= LP_MAX_VALS;
}

int tile3_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int mode, const SimLogpost* lp = nullptr);

int tile3_solve_sim_logpost(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, const double* obs,
                            const int32_t* obs_ind, int n_obs, double noise_sd, const double* upars, int n_prior, double prior_sd,
                            double* logpost) {
    SimLogpost lp;
    lp.obs = obs; lp.obs_ind = obs_ind; lp.n_obs = n_obs; lp.noise_sd = noise_sd; lp.upars = upars; lp.n_prior = n_prior;
    lp.prior_sd = prior_sd; lp.logpost = logpost;
    return tile3_solve(h, c, a, tiles, RK_MODE_SIM, &lp);
}

int tile3_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int mode, const SimLogpost* lp) {
    int rc;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) rc = launch_fwd_tile<FitzHughNagumo>(h, c, a, tiles);
    else if (c->rhs_id == RK_RHS_LORENZ63) rc = launch_fwd_tile<Lorenz63>(h, c, a, tiles);
    else if (is_user_rhs(c->rhs_id)) rc = user_forward_tile(h, c, a, tiles, 3);
    else rc = launch_fwd_tile<HigherOrder>(h, c, a, tiles);
    if (rc || mode == RK_MODE_FILTER) return rc;
    if (mode == RK_MODE_SIM) {
        LaunchTimer t(h, "bwd_sim_tile3_kernel");
        if (lp) hipLaunchKernelGGL(bwd_sim_tile3_kernel<true>, dim3(div_up(a.B * a.D, 4)), dim3(256), 0, h->stream, a, tiles, a.D, *lp);
        else hipLaunchKernelGGL(bwd_sim_tile3_kernel<false>, dim3(div_up(a.B * a.D, 4)), dim3(256), 0, h->stream, a, tiles, a.D, SimLogpost{});
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

// rk_fenrir_backward on the tile layout: `tiles` = the filtered moments of rk_solve_filter in RK_LAYOUT_TILE3
int tile3_fenrir_backward(rk_handle h, const SolveArgs& a, const double* tiles, const double* obs, const double* obs_w,
                          const double* obs_v, const int32_t* obs_ind, int n_obs, double* logdens) {
    FenrirObs ob;
    ob.obs = obs; ob.obs_w = obs_w; ob.obs_v = obs_v; ob.obs_ind = obs_ind; ob.n_obs = n_obs; ob.logdens = logdens;
    LaunchTimer t(h, "fenrir_bwd_tile3_kernel");
    hipLaunchKernelGGL(fenrir_bwd_tile3_kernel, dim3(div_up(a.B * a.D, 4)), dim3(256), 0, h->stream, a, tiles, a.D, ob);
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

}  // namespace rk
