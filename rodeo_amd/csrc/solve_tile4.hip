// MFMA-tile solver kernels for n_bstate = 4, n_bmeas = 1, n_block <= 4: BASELINE config 3 (Lorenz63) and
// the second-order example of docs/examples/higher_order.md.  Same machinery as solve_tile3.hip (read that header
// first); with p = 4 the 4 x 4 tile is filled by Sigma, so the mean travels as a second per-lane value in
// "row form" (lane (r, g, c) holds mu_r for every c):
//     predict  (standard.py:57-59):  U = MF(S, Qt) = (Q Sigma)^T ;  S- = MF(U, Qt, R) ;  m- = MF(Qt, m) = Q mu
//     update   (standard.py:93-102): X_w[k] = W_k - J_k (row form);  yhat = MF(X_w, m-, a) ;  WS = MF(X_w, S-) ;
//                                    Z = MF(S-^T, X_w) = Sigma- W~^T ;  s = MF(Z, X_w) (+V) ;  K = Z / s ;
//                                    S = S- - K WS ;  m = m- - K yhat
//     smooth   (standard.py:213-216): D = Ss - S- ; V = MF(D, Gt) ; Ss = MF(V, Gt, S_f) ; ms = MF(Gt, ms - m-, m_f)
// HBM format (RK_LAYOUT_TILE4): per time step and (trajectory, block) 20 doubles [Sigma row-major (16) | mu (4)]
// = 160 B = the algorithmic p (p + 1) 8 bytes.  With n_block = 3 a wave carries the three blocks of ONE trajectory in
// tiles g = 0..2 (g = 3 idles); with n_block = 2 two trajectories; with n_block = 1 four.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"
#include "kalman_small.hpp"
#include "mfma_tile.hpp"
#include "rhs.hpp"
#include "solve_args.hpp"
#include "solve_tile4_kernels.hpp"

namespace rk {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Same structure as bwd_mv_tile3_kernel (read its comments first): 4-wave workgroups, wave 0 consumes the 16-step
// chunks, waves 1..3 produce them in three pipeline stages; LDS-DMA prefetch of the filtered tiles, hand-off through
// LDS with the conflict-free swizzles of mfma_tile.hpp, consumer with software-pipelined LDS reads and buffer stores.
// A wave carries TPW = 3 (n_block = 3: the three blocks of one trajectory) or 4 tiles; items are packed TPW to a time
// step and the chunk is 16 (TPW = 3) or 12 (TPW = 4) steps, so a workgroup needs 48 + 24 (+ 1) KiB of LDS and at most
// 256 registers per lane: two workgroups share a CU (measured with per-workgroup time stamps, -DRK_T4_STAMPS: with
// 260 registers the two workgroups of a CU ran one after the other).
constexpr int ITEM4 = 512;                       // bytes per (step, tile): S- | G^T | S_f tiles, then m- (4), m_f (4)

template <int TPW>
__device__ __forceinline__ int lds4_tile(int s, int g, int which, int idx) {
    return (s * TPW + g) * ITEM4 + which * 128 + (tile_slot(s, g, idx) << 3);
}
template <int TPW>
__device__ __forceinline__ int lds4_vec(int s, int g, int which, int rr) {
    return (s * TPW + g) * ITEM4 + 384 + (vec_slot(s, g, which, rr) << 3);
}

// The consumer's image of a time row (round 4): every lane drops its Sigma element into a 64-double array and its mean element
// into a second one, 512 bytes behind, inside the step's LDS bytes -- two unmasked stores off one address that go out as ONE
// ds_write2_b64, where two exec-masked ds_write_b64 to the row format stood (probe15: the two masked writes cost the chain 57
// cycles per step) -- and whoever sends the row to memory gathers the 16-byte pieces of the row format (per tile 16 Sigma doubles,
// then 4 mean doubles) from them: a Sigma piece is two neighbouring doubles of the first array, a mean piece two doubles 128 bytes
// apart in the second.  Lane = 16 r + 4 g + c (t4_coord); the mean is in row form (the lanes c = 0 hold the row's value).
// o0, o1: byte offsets of the piece's two doubles inside the step's image.
constexpr int T4_IMG_MEAN = 512;
__device__ __forceinline__ void t4_piece_offsets(int fcol_bytes, int& o0, int& o1) {
    const int gq = fcol_bytes / (T4_DOUBLES * 8), off = fcol_bytes - gq * (T4_DOUBLES * 8);
    if (off < 128) {
        const int idx0 = off >> 3, r = idx0 >> 2, c0 = idx0 & 3;
        o0 = (16 * r + 4 * gq + c0) * 8;
        o1 = o0 + 8;
    } else {
        const int r0 = (off - 128) >> 3;
        o0 = T4_IMG_MEAN + (16 * r0 + 4 * gq) * 8;
        o1 = o0 + 128;
    }
}
typedef double t4_d2 __attribute__((ext_vector_type(2)));

#ifdef RK_T4_STAMPS   // experiment build: per-workgroup cycle sums (consumer work / barrier wait, producer stages / wait)
#define T4_STAMP_ARG , long long* __restrict__ dbg
#define T4_NOW() __builtin_amdgcn_s_memtime()
#else
#define T4_STAMP_ARG
#endif
// SETS = 1, NP = 3: the 4-wave workgroup described above, two of them on a CU.
// SETS = 2, NP = 2 (round 4, "a SIMD of its own for the chain"): ONE 8-wave workgroup per CU carries two such sets.  The waves of a
// workgroup go to the SIMDs in turn (wave w on SIMD w mod 4, HW_ID probe), so waves 0 and 1 -- the two consumers -- sit on
// SIMD 0 and SIMD 1, waves 4 and 5 leave at once and keep those SIMDs free of anything else, and the producers are waves
// 2, 6 (set 0, SIMD 2) and 3, 7 (set 1, SIMD 3): two per set, each taking every other chunk through stages 1 + 2 in one tick
// and stage 3 in the next.
template <int D, int NP, int SETS>
__global__ void __launch_bounds__(256 * SETS, SETS == 1 ? 2 : 1) bwd_mv_tile4_kernel(SolveArgs a, double* __restrict__ tiles T4_STAMP_ARG) {
    constexpr int P = 4, P4 = 4, TPW = Tpw<D>::value;
    static_assert((SETS == 1 && NP == 3) || (SETS == 2 && NP == 2), "wave roles are written for these two shapes");
    // time steps per hand-off: 16 with three tiles per wave, 12 with four -- 74 KiB of LDS either way, so that TWO
    // workgroups share a CU (with 16 steps x 4 tiles: 96 KiB, one workgroup per CU, the chain waves idle half the time)
    constexpr int CH4 = TPW == 3 ? 16 : 12;
    // the consumer's image of a time row: two dense arrays (t4_piece_offsets) with three tiles per wave; the row format itself, written
    // by exec-masked stores a step late, with four (measured: the dense image costs the four-tile shape 5 %, the masked one the
    // three-tile shape 10 %)
    constexpr bool DENSE_IMG = TPW == 3;
    constexpr int BUF = CH4 * TPW * ITEM4;                         // 24 KiB
    constexpr int ROW_BYTES = TPW * T4_DOUBLES * 8;                // this tile-wave's bytes per time row: 480 / 640
    constexpr int N_DMA = (CH4 * ROW_BYTES / 16 + 63) / 64;        // 1-KiB LDS-DMA pieces per chunk: 8 / 10
    constexpr int ZONE = N_DMA * 1024;
    __shared__ __attribute__((aligned(16))) char lds_all_[SETS][2 * BUF];
    __shared__ __attribute__((aligned(16))) char zones_[SETS][NP * ZONE];
    // Q | R of this workgroup's TPW blocks: 64 registers per producer lane if they were kept across ticks, which is
    // what pushed the kernel over 256 registers = one workgroup per CU (the stride keeps the four blocks' rows in
    // different banks; lanes of one block read the same address)
    constexpr int QR_STRIDE = 2 * P4 * P4 + 2;
    __shared__ __attribute__((aligned(16))) double qr_[SETS][4 * QR_STRIDE];
    const int hw_wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // role: 0 = consumer; producers q = role - 1 own the chunks ch = q (mod NP)
    const int set = SETS == 1 ? 0 : (hw_wave & 1);
    const int wave = SETS == 1 ? hw_wave : (hw_wave < 2 ? 0 : (hw_wave < 4 ? 1 : (hw_wave < 6 ? -1 : 2)));
#ifdef RK_T4_STAMPS
    if (lane == 0) {                                               // where the dispatcher put each of the workgroup's waves: slot 19 of its first tile-wave
        unsigned hw_;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
        atomicOr((unsigned long long*)&dbg[(size_t)blockIdx.x * SETS * 20 + 19], (unsigned long long)(((hw_ >> 4) & 3) | 4) << (4 * hw_wave));
    }
#endif
    if (wave < 0) return;                                          // (SETS = 2: waves 4, 5 only keep SIMD 0, 1 free)
#if defined(RK_T4_EXP) && RK_T4_EXP == 1                      // experiment builds (results wrong by construction): the second set without its producers,
    if (SETS == 2 && set == 1 && wave >= 1) return;
#elif defined(RK_T4_EXP) && RK_T4_EXP == 2                    // ... without its consumer,
    if (SETS == 2 && set == 1 && wave == 0) return;
#elif defined(RK_T4_EXP) && RK_T4_EXP == 3                    // ... not at all (one set per 8-wave workgroup, the other half of the tile-waves unprocessed)
    if (SETS == 2 && set == 1) return;
#endif
    char* const lds_all = lds_all_[set];
    char* const zones = zones_[set];
    double* const qr = qr_[set];
    const int n_tiles = a.B * D;
    const size_t tstride = (size_t)n_tiles * T4_DOUBLES;
    const size_t row_bytes = tstride * sizeof(double);
    const int n_chunks = (a.N - 1 + CH4 - 1) / CH4;                // steps n = N-1 .. 1
    const int tw = blockIdx.x * SETS + set;                        // tile-wave: tiles TPW tw .. TPW tw + TPW - 1
    if (SETS > 1 && tw * TPW >= n_tiles) return;                   // (an odd number of tile-waves: the whole second set leaves before any barrier)
    const char* const wave_rows = (const char*)(tiles + (size_t)tw * TPW * T4_DOUBLES);

    if (wave >= 1) {
        // ---------------- producers: one lane per (step-in-chunk, tile) ----------------
        const int p = wave - 1;
        const int s = lane >> 2, g = lane & 3;
        const bool active = g < TPW && s < CH4;
        int tau = tw * TPW + (active ? g : 0);
        if (tau >= n_tiles) tau = n_tiles - 1;
        const int b = tau / D, blk = tau - b * D;
        const bool buffer_ok = (CH4 - 1) * row_bytes + ROW_BYTES < 0x7fffffffull;
        const int chunk_span = (int)((CH4 - 1) * row_bytes + ROW_BYTES);
        if (p == 0 && s == 0) {
            double Q0[P][P], R0[P][P];
            load_block_consts<P>(a, blk, b, Q0, R0);
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int j = 0; j < P; ++j) { qr[g * QR_STRIDE + i * P + j] = Q0[i][j]; qr[g * QR_STRIDE + P * P + i * P + j] = R0[i][j]; }
        }
        lds_barrier();                                              // (the consumer's matching barrier is its first one)
        int woff[16], voff[4], voff1[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) woff[i] = lds4_tile<TPW>(s, g, 0, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) { voff[i] = lds4_vec<TPW>(s, g, 0, i); voff1[i] = lds4_vec<TPW>(s, g, 1, i); }
        // landing zone = image of the chunk's 16 time rows x ROW_BYTES; piece j = 64 i + lane (clamped in the last one)
        char* const zone = zones + p * ZONE;
        const unsigned zone_lds = __builtin_amdgcn_readfirstlane(lds_addr(zone));
        int frow[N_DMA], fcol[N_DMA], po0[N_DMA], po1[N_DMA];
#pragma unroll
        for (int i = 0; i < N_DMA; ++i) {
            int j = 64 * i + lane;
            j = j < CH4 * ROW_BYTES / 16 ? j : CH4 * ROW_BYTES / 16 - 1;
            frow[i] = j / (ROW_BYTES / 16); fcol[i] = (j % (ROW_BYTES / 16)) * 16;
            t4_piece_offsets(fcol[i], po0[i], po1[i]);
            po0[i] += frow[i] * (TPW * ITEM4); po1[i] += frow[i] * (TPW * ITEM4);
        }
        auto fetch = [&](int ch) {
            const int n_hi = a.N - 1 - ch * CH4;
#pragma unroll
            for (int i = 0; i < N_DMA; ++i) {
                const int n = n_hi - frow[i];
                // (INVARIANT of lds_barrier(): this landing zone is read by the wave that issued the DMA and by no other)
                lds_dma16(wave_rows + (size_t)(n < 1 ? 1 : n) * row_bytes + fcol[i], zone_lds + 1024 * i);
            }
        };
        lds_dma_wait_all();                                        // retire the loads of Q, R before the first DMA
        const double* const my_qr = qr + g * QR_STRIDE;
        if (p < n_chunks) fetch(p);
        double mf[P], Sf[P][P], mp[P], Sp[P][P], A[P][P], X[P][P], rpiv[P];
#if RK_T4_STAMPS >= 2
        long long acc[4] = {0, 0, 0, 0};
#endif
        for (int t = -NP; t < n_chunks; ++t) {
            // NP = 3: one stage per tick (chunk t + 3, t + 2 or t + 1); NP = 2: stages 1 and 2 of chunk t + 2 in one tick
            const int ch1 = t + NP, ch2 = NP == 3 ? t + 2 : t + 2, ch3 = t + 1;
#if RK_T4_STAMPS >= 2
            const long long tA = T4_NOW();
            const int stg = ch1 % NP == p ? 0 : ((NP == 3 && ch2 >= 0 && ch2 % NP == p) ? 1 : 2);
#endif
            if (ch1 % NP == p) {
                // ---- stage 1 of chunk ch1: fetched tiles, next fetch, predict, T^T ----
                if (ch1 < n_chunks) {
                    lds_dma_wait_all();
                    double buf[T4_DOUBLES];
                    const char* mine = zone + (active ? s * ROW_BYTES + g * (T4_DOUBLES * 8) : 0);
#pragma unroll
                    for (int k = 0; k < T4_DOUBLES / 2; ++k) {
                        const double2 v = *(const double2*)(mine + 16 * k);
                        buf[2 * k] = v.x; buf[2 * k + 1] = v.y;
                    }
                    lds_reads_done();
                    if (ch1 + NP < n_chunks) fetch(ch1 + NP);
#pragma unroll
                    for (int i = 0; i < P; ++i) {
#pragma unroll
                        for (int j = 0; j < P; ++j) Sf[i][j] = buf[i * 4 + j];
                        mf[i] = buf[16 + i];
                    }
                    double T[P][P];
                    double Q[P][P], R[P][P];
#pragma unroll
                    for (int k = 0; k < P * P / 2; ++k) {
                        const double2 q2 = *(const double2*)(my_qr + 2 * k), r2 = *(const double2*)(my_qr + P * P + 2 * k);
                        Q[(2 * k) / P][(2 * k) % P] = q2.x; Q[(2 * k + 1) / P][(2 * k + 1) % P] = q2.y;
                        R[(2 * k) / P][(2 * k) % P] = r2.x; R[(2 * k + 1) / P][(2 * k + 1) % P] = r2.y;
                    }
                    predict_block<P>(Q, R, mf, Sf, mp, Sp);          // pred[n+1] from filt[n]   (standard.py:57-59)
                    if constexpr (NP != 3) {
                        mm_nt<P, P, P>(Sf, Q, T);                    // T = Sigma_f Q^T          (standard.py:175)
#pragma unroll
                        for (int i = 0; i < P; ++i)
#pragma unroll
                            for (int j = 0; j < P; ++j) X[i][j] = T[j][i];
                    }
                }
            }
            if (NP == 3 ? (ch1 % NP != p && ch2 >= 0 && ch2 % NP == p) : (ch1 % NP == p)) {
                // ---- stage 2 of chunk ch2: T^T, LU of Sigma- with the forward sweep on T^T, back substitution ----
                // (round 4: with the consumer's tick down to ~2.9 k cycles the LONGEST producer stage sets the tick -- three stages of
                //  3.3 / 1.5 / 4.4 k; T^T moved here from stage 1 and the back substitution from stage 3: ~2.7 / 2.9 / 3.0 k)
                if (ch2 < n_chunks) {
                    if constexpr (NP == 3) {
                        double T[P][P], Q[P][P];
#pragma unroll
                        for (int k = 0; k < P * P / 2; ++k) {
                            const double2 q2 = *(const double2*)(my_qr + 2 * k);
                            Q[(2 * k) / P][(2 * k) % P] = q2.x; Q[(2 * k + 1) / P][(2 * k + 1) % P] = q2.y;
                        }
                        mm_nt<P, P, P>(Sf, Q, T);                    // T = Sigma_f Q^T          (standard.py:175)
#pragma unroll
                        for (int i = 0; i < P; ++i)
#pragma unroll
                            for (int j = 0; j < P; ++j) X[i][j] = T[j][i];
                    }
#pragma unroll
                    for (int i = 0; i < P; ++i)
#pragma unroll
                        for (int j = 0; j < P; ++j) A[i][j] = Sp[i][j];
                    lu_factor_fwd<P, P>(A, X, rpiv);
                    lu_back<P, P>(A, X, rpiv);                       // X = G^T (standard.py:176)
                }
            } else if (ch1 % NP != p && ch3 >= 0 && (NP == 3 || ch3 % NP == p)) {
                // ---- stage 3 of chunk ch3: the finished image of chunk ch3 - 2 to memory, hand-off of this chunk ----
                if (ch3 < n_chunks) {
                    // the smoothed rows of chunk ch3 - 2: the consumer built their image in the buffer this wave is
                    // about to refill (LDS operations of one wave execute in order); 16 bytes per lane, whole rows
                    if (ch3 >= 2 && buffer_ok) {
                        const char* img = lds_all + (ch3 & 1) * BUF;
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                            (void*)(wave_rows + (size_t)(a.N - 1 - (ch3 - 2) * CH4 - (CH4 - 1)) * row_bytes), 0, chunk_span, 0x00020000);
#pragma unroll
                        for (int i0 = 0; i0 < N_DMA; i0 += 4) {
                            u32x4 v[4];
#pragma unroll
                            for (int i = i0; i < i0 + 4 && i < N_DMA; ++i) {
                                if constexpr (DENSE_IMG) {
                                    t4_d2 pr;
                                    pr.x = *(const double*)(img + po0[i]); pr.y = *(const double*)(img + po1[i]);
                                    v[i - i0] = __builtin_bit_cast(u32x4, pr);
                                } else {
                                    v[i - i0] = *(const u32x4*)(img + frow[i] * (TPW * ITEM4) + fcol[i]);
                                }
                            }
#pragma unroll
                            for (int i = i0; i < i0 + 4 && i < N_DMA; ++i) {
                                const int vo = tw * TPW + fcol[i] / (T4_DOUBLES * 8) < n_tiles
                                                   ? (CH4 - 1 - frow[i]) * (int)row_bytes + fcol[i] : (int)0x80000000;
                                __builtin_amdgcn_raw_buffer_store_b128(v[i - i0], rs, vo, 0, 0);
                            }
                        }
                    }
                    const int n = a.N - 1 - ch3 * CH4 - s;
                    if (n >= 1 && active) {
                        char* o = lds_all + (ch3 & 1) * BUF;
#pragma unroll
                        for (int i = 0; i < P; ++i) {
#pragma unroll
                            for (int j = 0; j < P; ++j) {
                                *(double*)(o + woff[i * 4 + j]) = Sp[i][j];
                                *(double*)(o + woff[i * 4 + j] + 128) = X[i][j];
                                *(double*)(o + woff[i * 4 + j] + 256) = Sf[i][j];
                            }
                            *(double*)(o + voff[i]) = mp[i];
                            *(double*)(o + voff1[i]) = mf[i];
                        }
                    }
                }
            }
#if RK_T4_STAMPS >= 2
            const long long tB = T4_NOW();
            lds_barrier();                                      // (the barrier the shipped kernel uses: the stamps must time ITS waits)
            const long long tC = T4_NOW();
            acc[stg] += tB - tA; acc[3] += tC - tB;
#else
            lds_barrier();   
#endif
        }
#if RK_T4_STAMPS >= 2
        if (lane == 0) for (int k = 0; k < 4; ++k) dbg[tw * 20 + 4 + p * 4 + k] = acc[k];
#endif
    } else {
        // ---------------- consumer ----------------
        const T4Coord tc = t4_coord<D>(tw, lane, n_tiles);
        __builtin_amdgcn_s_setprio(3);
        const int r = tc.r, g = tc.g, c = tc.c, idx = r * 4 + c;
        const int gl = g < TPW ? g : 0;                             // (idle tile slot: reads slot 0's items, stores nothing)
        const bool st_m = tc.valid && c == 0;
        // carry = filt[N]  (solve.py:279-282); the mean in row form
        double Ss = tc.valid ? tiles[(size_t)a.N * tstride + (size_t)tc.tau * T4_DOUBLES + idx] : 0.0;
        double ms = tc.valid ? tiles[(size_t)a.N * tstride + (size_t)tc.tau * T4_DOUBLES + 16 + r] : 0.0;
        const bool buffer_ok = (CH4 - 1) * row_bytes + ROW_BYTES < 0x7fffffffull;
        const int chunk_span = (int)((CH4 - 1) * row_bytes + ROW_BYTES);
        // row images for the 16-byte stores: lane = piece fj of row fr of a group of RPF rows (10 pieces per tile)
        constexpr int PIECES = ROW_BYTES / 16, RPF = 64 / PIECES, NFL = CH4 / RPF;
        static_assert(CH4 % RPF == 0 && 64 * 16 <= TPW * ITEM4, "the 64 pair slots of a step live inside the step's items");
        const int fj = lane % PIECES, fr = lane / PIECES;
        const bool fvalid = fr < RPF && tw * TPW + fj / 10 < n_tiles;
        int f_o0, f_o1;
        t4_piece_offsets(fj * 16, f_o0, f_o1);
        f_o0 = fvalid ? fr * TPW * ITEM4 + f_o0 : 0; f_o1 = fvalid ? fr * TPW * ITEM4 + f_o1 : 8;
        const int f_lds = fvalid ? fr * TPW * ITEM4 + fj * 16 : 0;
        const int wP = lane * 8;                                    // this lane's Sigma element in a step's image; its mean element T4_IMG_MEAN behind
        const int wS = gl * (T4_DOUBLES * 8) + idx * 8, wM = gl * (T4_DOUBLES * 8) + 128 + r * 8;      // (row format)
        const int f_vo = fvalid && buffer_ok ? (int)((RPF - 1 - fr) * row_bytes) + fj * 16 : (int)0x80000000;
        int roff[4], rvec[4], rvec1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            roff[k] = lds4_tile<TPW>(k, gl, 0, idx) - k * TPW * ITEM4;
            rvec[k] = lds4_vec<TPW>(k, gl, 0, r) - k * TPW * ITEM4;
            rvec1[k] = lds4_vec<TPW>(k, gl, 1, r) - k * TPW * ITEM4;
        }
        lds_barrier();                                               // Q | R of the blocks are in LDS
#pragma unroll
        for (int k = 0; k < NP; ++k) lds_barrier();                  // ticks -NP .. -1: chunk 0 is in LDS
#ifdef RK_T4_STAMPS
        long long cacc[2] = {0, 0};
        const long long rt0 = __builtin_amdgcn_s_memrealtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
#endif
        for (int t = 0; t < n_chunks; ++t) {
#ifdef RK_T4_STAMPS
            const long long tA = T4_NOW();
#endif
            const char* in = lds_all + (t & 1) * BUF;
            const int n_hi = a.N - 1 - t * CH4;
            const int cnt = __builtin_amdgcn_readfirstlane(n_hi >= CH4 ? CH4 : n_hi);
            if (cnt == CH4 && buffer_ok) {
                // branch-free chunk: LDS reads LOOKAHEAD steps ahead (five per step), three MFMAs and two subtractions
                // per step.  The results do not go to memory lane by lane (an 8-byte-per-lane store occupies the CU's
                // address path for ~40 cycles whatever it carries: two of them per step and two workgroups per CU
                // were the kernel's bound): each lane drops its element into the image of the time row, built in the
                // LDS bytes of the step's own items (read by now), and every RPF steps one 16-byte-per-lane store
                // writes RPF whole rows.
#ifdef RK_T4_LOOKAHEAD
                constexpr int LOOKAHEAD = RK_T4_LOOKAHEAD;
#else
                constexpr int LOOKAHEAD = 2;
#endif
                double Sp[CH4], Gt[CH4], Sf[CH4], mp[CH4], mf[CH4];
                auto load = [&](int s) {
                    const char* q = in + roff[s & 3] + s * TPW * ITEM4;
                    Sp[s] = *(const double*)(q);
                    Gt[s] = *(const double*)(q + 128);
                    Sf[s] = *(const double*)(q + 256);
                    mp[s] = *(const double*)(in + rvec[s & 3] + s * TPW * ITEM4);
                    mf[s] = *(const double*)(in + rvec1[s & 3] + s * TPW * ITEM4);
                };
#pragma unroll
                for (int s = 0; s < LOOKAHEAD; ++s) load(s);
                char* const img = lds_all + (t & 1) * BUF;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(wave_rows + (size_t)(n_hi - (CH4 - 1)) * row_bytes), 0, chunk_span, 0x00020000);
                const bool self_flush = t + 2 >= n_chunks;          // (no producer refills this buffer any more)
                u32x4 fl[NFL];
#pragma unroll
                for (int s = 0; s < CH4; ++s) {
                    if (s + LOOKAHEAD < CH4) load(s + LOOKAHEAD);
                    __builtin_amdgcn_sched_barrier(0);
                    const double V1 = MF(Ss - Sp[s], Gt[s], 0.0);       // (G D)^T
                    if constexpr (TPW == 4) {
                        // four tiles per wave: the previous step's results go into the image here, a step old and behind
                        // an MFMA (profiles/r02_probe13_store_cost.log); measured -5 % with four tiles, +4 % with three
                        if (s >= 1) {
                            double So = Ss, mo = ms;
                            asm("" : "+v"(So), "+v"(mo) : "v"(V1));
                            if (tc.valid) *(double*)(img + (s - 1) * TPW * ITEM4 + wS) = So;
                            if (st_m) *(double*)(img + (s - 1) * TPW * ITEM4 + wM) = mo;
                        }
                    }
                    ms = MF(Gt[s], ms - mp[s], mf[s]);                  // mu_f + G (mu_s - mu-)     (standard.py:213-214)
                    Ss = MF(V1, Gt[s], Sf[s]);                          // Sigma_f + G D G^T         (standard.py:215-216)
                    if constexpr (DENSE_IMG) {
                        *(double*)(img + s * TPW * ITEM4 + wP) = Ss;             // all lanes, both stores in one ds_write2_b64
                        *(double*)(img + s * TPW * ITEM4 + wP + T4_IMG_MEAN) = ms;  // (idle tile slots land in unused bytes)
                    } else if (s == CH4 - 1) {
                        if (tc.valid) *(double*)(img + s * TPW * ITEM4 + wS) = Ss;
                        if (st_m) *(double*)(img + s * TPW * ITEM4 + wM) = ms;
                    }
                    if (TPW != 4 && (s + 1) % RPF == 0 && self_flush) {
                        const int k = s / RPF;
                        {
                            t4_d2 pr;
                            pr.x = *(const double*)(img + (s - (RPF - 1)) * TPW * ITEM4 + f_o0);
                            pr.y = *(const double*)(img + (s - (RPF - 1)) * TPW * ITEM4 + f_o1);
                            fl[k] = __builtin_bit_cast(u32x4, pr);
                        }
                        if (k >= 1)
                            __builtin_amdgcn_raw_buffer_store_b128(fl[k - 1], rsrc, f_vo, (int)((CH4 - s + RPF - 1) * row_bytes), 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (self_flush) {
                    if constexpr (TPW == 4) {                       // (rows are complete only now: their writes run a step late)
#pragma unroll
                        for (int k = 0; k < NFL; ++k) {
                            const u32x4 v = *(const u32x4*)(img + k * RPF * TPW * ITEM4 + f_lds);      // (TPW = 4: the row format)
                            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, f_vo, (int)((CH4 - (k + 1) * RPF) * row_bytes), 0);
                        }
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(fl[NFL - 1], rsrc, f_vo, 0, 0);
                    }
                }
            } else {
                for (int s = 0; s < cnt; ++s) {
                    const char* q = in + lds4_tile<TPW>(s, gl, 0, idx);
                    const double Sp = *(const double*)(q), Gt = *(const double*)(q + 128), Sf = *(const double*)(q + 256);
                    const double mp = *(const double*)(in + lds4_vec<TPW>(s, gl, 0, r));
                    const double mf = *(const double*)(in + lds4_vec<TPW>(s, gl, 1, r));
                    const double V1 = MF(Ss - Sp, Gt, 0.0);
                    ms = MF(Gt, ms - mp, mf);
                    Ss = MF(V1, Gt, Sf);
                    double* row = tiles + (size_t)(n_hi - s) * tstride + (size_t)tc.tau * T4_DOUBLES;
                    if (tc.valid) row[idx] = Ss;
                    if (st_m) row[16 + r] = ms;
                }
            }
#ifdef RK_T4_STAMPS
            const long long tB = T4_NOW();
            lds_barrier();                                      // (the barrier the shipped kernel uses: the stamps must time ITS waits)
            const long long tC = T4_NOW();
            cacc[0] += tB - tA; cacc[1] += tC - tB;
#if RK_T4_STAMPS >= 3      // per-tick consumer work of the first 8 workgroups, behind the per-workgroup sums
            if (tw < 8 && t < 2048 && lane == 0) dbg[(size_t)gridDim.x * SETS * 20 + (size_t)tw * 2048 + t] = tB - tA;
#endif
#else
            lds_barrier();   
#endif
        }
#ifdef RK_T4_STAMPS
        if (lane == 0) { dbg[tw * 20] = cacc[0]; dbg[tw * 20 + 1] = cacc[1]; dbg[tw * 20 + 2] = n_chunks; dbg[tw * 20 + 16] = rt0; dbg[tw * 20 + 17] = __builtin_amdgcn_s_memrealtime(); dbg[tw * 20 + 18] = ((long long)xcc << 32) | hwid; }
#endif
    }
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
template <class RHS>
static int launch_fwd_tile4(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles) {
    const dim3 grid(div_up(a.B * RHS::D, Tpw<RHS::D>::value)), block(64);
    launch_placement_primer(h, grid, block);           // (common.hpp: exact one-wave-per-SIMD placement behind any kernel)
    LaunchTimer t(h, "fwd_tile4_kernel");
    switch (c->interrogate) {
        case RK_INTERROGATE_KRAMER:
            hipLaunchKernelGGL((fwd_tile4_kernel<RHS, RK_INTERROGATE_KRAMER>), grid, block, 0, h->stream, a, tiles); break;
        case RK_INTERROGATE_SCHOBER:
            hipLaunchKernelGGL((fwd_tile4_kernel<RHS, RK_INTERROGATE_SCHOBER>), grid, block, 0, h->stream, a, tiles); break;
        case RK_INTERROGATE_RODEO:
            hipLaunchKernelGGL((fwd_tile4_kernel<RHS, RK_INTERROGATE_RODEO>), grid, block, 0, h->stream, a, tiles); break;
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

bool tile4_supported(const rk_solve_cfg* c, int mode) {
    if (c->flags & (RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR)) return false;
    if (mode == RK_MODE_SIM) return false;
    if (c->kalman_type != RK_KALMAN_STANDARD || c->n_bstate != 4 || c->n_bmeas != 1) return false;
    if (c->interrogate != RK_INTERROGATE_KRAMER && c->interrogate != RK_INTERROGATE_SCHOBER &&
        c->interrogate != RK_INTERROGATE_RODEO)
        return false;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) return c->n_block == 2;
    if (c->rhs_id == RK_RHS_LORENZ63) return c->n_block == 3;
    if (c->rhs_id == RK_RHS_HIGHER_ORDER) return c->n_block == 1;
    if (is_user_rhs(c->rhs_id)) return user_tile_available(c, 4);       // hiprtc build of fwd_tile4_kernel (rhs_jit.hip)
    return false;
}

size_t tile4_doubles(const rk_solve_cfg* c) {
    const size_t n_tiles = (size_t)c->n_block * (size_t)c->n_traj;
    const size_t tpw = c->n_block == 3 ? 3 : 4;
    const size_t waves = (n_tiles + tpw - 1) / tpw;
    return (size_t)(c->n_steps + 1) * n_tiles * T4_DOUBLES + ((waves + 1) / 2) * 2 * 128;
}

int tile4_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int mode) {
    int rc;
    if (c->rhs_id == RK_RHS_FITZHUGH_NAGUMO) rc = launch_fwd_tile4<FitzHughNagumo>(h, c, a, tiles);
    else if (c->rhs_id == RK_RHS_LORENZ63) rc = launch_fwd_tile4<Lorenz63>(h, c, a, tiles);
    else if (is_user_rhs(c->rhs_id)) rc = user_forward_tile(h, c, a, tiles, 4);
    else rc = launch_fwd_tile4<HigherOrder>(h, c, a, tiles);
    if (rc || mode == RK_MODE_FILTER || a.N < 2) return rc;
    const int tpw = a.D == 3 ? 3 : 4;
    // Two shapes of workgroup (RK_T4_BWD=quad|ds forces one).  quad: 4 waves, two workgroups per CU.  ds: 8 waves carrying two sets, each
    // consumer alone on its SIMD -- built in round 4 to see whether the chain wave suffers from its SIMD partner: it does not (HW_ID stamps
    // confirm the placement, and the consumer's tick stayed at 3925 cycles against 3945, 245 per step, even with ONE set per CU,
    // profiles/r04_c3_dedicated_simd.txt).  What shortened the step was the dense image (one ds_write2_b64 per step: 245 -> 190 cycles);
    // with it the three-tile shape runs 2.22 ms in ds against 2.79 in quad (C3), the four-tile shape 0.57 in quad against 0.68 in ds
    // (FitzHugh-Nagumo, n_deriv 4): the default follows the shape.  (A third producer per set on waves 4, 5 of the two-set layout: 2.30-2.32
    // against 2.29 ms -- at 4.3 TB/s of reads and writes in equal parts the pass is at the rate the memory system gives this stream.)
    static const int forced = [] { const char* e = getenv("RK_T4_BWD"); return e && e[0] == 'd' ? 2 : (e && e[0] == 'q' ? 1 : 0); }();
    const bool quad = forced ? forced == 1 : tpw != 3;
    const int n_tw = div_up(a.B * a.D, tpw);
    const dim3 grid(quad ? n_tw : div_up(n_tw, 2)), block(quad ? 256 : 512);
    LaunchTimer t(h, "bwd_mv_tile4_kernel");
#ifdef RK_T4_STAMPS
    long long* dbg = nullptr;
    const unsigned n_slot = grid.x * (quad ? 1 : 2);            // one slot of 20 per tile-wave (set)
    RK_HIP(hipMalloc(&dbg, ((size_t)n_slot * 20 + 8 * 2048) * sizeof(long long)));
    RK_HIP(hipMemsetAsync(dbg, 0, ((size_t)n_slot * 20 + 8 * 2048) * sizeof(long long), h->stream));
#define T4_DBG , dbg
#else
#define T4_DBG
#endif
#define RK_T4_BWD(D_)                                                                                                   \
    do {                                                                                                                \
        if (quad) hipLaunchKernelGGL((bwd_mv_tile4_kernel<D_, 3, 1>), grid, block, 0, h->stream, a, tiles T4_DBG);      \
        else hipLaunchKernelGGL((bwd_mv_tile4_kernel<D_, 2, 2>), grid, block, 0, h->stream, a, tiles T4_DBG);           \
    } while (0)
    if (a.D == 1) RK_T4_BWD(1);
    else if (a.D == 2) RK_T4_BWD(2);
    else if (a.D == 3) RK_T4_BWD(3);
    else RK_T4_BWD(4);
#undef RK_T4_BWD
    t.stop();
    RK_HIP(hipGetLastError());
#ifdef RK_T4_STAMPS
    {
        std::vector<long long> hd((size_t)n_slot * 20 + 8 * 2048);
        RK_HIP(hipMemcpyAsync(hd.data(), dbg, hd.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
        RK_HIP(hipStreamSynchronize(h->stream));
        RK_HIP(hipFree(dbg));
        double m[16] = {0};
        unsigned n_live = 0;
        for (unsigned w = 0; w < n_slot; ++w) n_live += hd[w * 20 + 2] > 0;
        for (unsigned w = 0; w < n_slot; ++w) for (int k = 0; k < 16; ++k) m[k] += (double)hd[w * 20 + k] / (n_live ? n_live : 1);
        const double nc = m[2] > 0 ? m[2] : 1;
        fprintf(stderr, "[t4 stamps] per tick (mean over %u workgroups, %g ticks): consumer work %.0f wait %.0f |", n_slot, nc, m[0] / nc, m[1] / nc);
        for (int p = 0; p < (quad ? 3 : 2); ++p) fprintf(stderr, " producer %d: stage1 %.0f stage2 %.0f stage3 %.0f wait %.0f |", p, m[4 + 4 * p] * (quad ? 3 : 2) / nc, m[5 + 4 * p] * (quad ? 3 : 2) / nc, m[6 + 4 * p] * (quad ? 3 : 2) / nc, m[7 + 4 * p] / nc);
        fprintf(stderr, "\n");
        long long t_min = hd[16], t_max = hd[17]; double life = 0;
        std::vector<int> per_cu(8 * 128, 0), simd_of(8 * 128, -1); int same_simd = 0, pairs = 0;
        for (unsigned w = 0; w < n_slot; ++w) {
            t_min = std::min(t_min, hd[w * 20 + 16]); t_max = std::max(t_max, hd[w * 20 + 17]);
            life += (double)(hd[w * 20 + 17] - hd[w * 20 + 16]) / n_slot;
            const unsigned hw = (unsigned)hd[w * 20 + 18], xc = (unsigned)(hd[w * 20 + 18] >> 32) & 7;
            const unsigned cu = (hw >> 8) & 15, se = (hw >> 13) & 7;    // HW_ID: cu_id [11:8], sh [12], se [15:13]
            const unsigned simd = (hw >> 4) & 3;
            const int key = xc * 128 + se * 16 + cu;
            if (simd_of[key] >= 0) { pairs++; same_simd += simd_of[key] == (int)simd; }
            simd_of[key] = (int)simd;
            per_cu[key] += 1;
        }
        int hist[8] = {0};
        for (int v : per_cu) if (v < 8) hist[v]++;
        fprintf(stderr, "[t4 stamps] kernel span %.1f us (100 MHz clock), mean workgroup lifetime %.1f us; workgroups per (xcc, se, cu): ", (t_max - t_min) / 100.0, life / 100.0);
        for (int k = 1; k < 8; ++k) if (hist[k]) fprintf(stderr, "%d CUs with %d, ", hist[k], k);
        fprintf(stderr, "consumer waves of a CU on the same SIMD: %d of %d pairs", same_simd, pairs);
        {
            int sh[4] = {0, 0, 0, 0};
            for (unsigned w = 0; w < n_slot; ++w) if (hd[w * 20 + 2] > 0) sh[((unsigned)hd[w * 20 + 18] >> 4) & 3]++;
            fprintf(stderr, "; consumers per SIMD id: %d %d %d %d", sh[0], sh[1], sh[2], sh[3]);
            // SIMD of wave w relative to the SIMD of wave 0, over all workgroups (SETS = 2: waves 0-7)
            const int nwv = quad ? 4 : 8;
            std::vector<int> rel(nwv * 4, 0);
            for (unsigned w = 0; w < n_slot; w += (quad ? 1 : 2)) {
                const unsigned long long v = (unsigned long long)hd[w * 20 + 19];
                if (!(v & 4)) continue;
                for (int k = 0; k < nwv; ++k) if ((v >> (4 * k)) & 4) rel[k * 4 + (((v >> (4 * k)) & 3) - (v & 3) + 4) % 4]++;
            }
            fprintf(stderr, "\n[t4 stamps] SIMD of wave w minus SIMD of wave 0 (mod 4), counts of 0 1 2 3:");
            for (int k = 0; k < nwv; ++k) fprintf(stderr, " w%d: %d %d %d %d |", k, rel[k * 4], rel[k * 4 + 1], rel[k * 4 + 2], rel[k * 4 + 3]);
        }
#if RK_T4_STAMPS >= 3
        for (int w8 = 0; w8 < 8 && w8 < (int)n_slot; ++w8) {
            std::vector<long long> v;
            for (int t = 0; t < 2048; ++t) if (hd[(size_t)n_slot * 20 + w8 * 2048 + t] > 0) v.push_back(hd[(size_t)n_slot * 20 + w8 * 2048 + t]);
            if (v.empty()) continue;
            std::vector<long long> q = v; std::sort(q.begin(), q.end());
            fprintf(stderr, "\n[t4 stamps] workgroup %d consumer work per tick: min %lld p10 %lld p50 %lld p90 %lld max %lld; first ticks:", w8,
                    q.front(), q[q.size() / 10], q[q.size() / 2], q[q.size() * 9 / 10], q.back());
            for (int t = 100; t < 124 && t < (int)v.size(); ++t) fprintf(stderr, " %lld", v[t]);
        }
#endif
        fprintf(stderr, "\n");
    }
#endif
    return RK_OK;
}

}  // namespace rk
