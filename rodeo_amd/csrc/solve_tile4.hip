// MFMA-tile solver kernels for n_bstate = 4, n_bmeas = 1, n_block in {1, 2, 3}: BASELINE config 3 (Lorenz63) and
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
#include "common.hpp"
#include "kalman_small.hpp"
#include "mfma_tile.hpp"
#include "rhs.hpp"
#include "solve_args.hpp"

namespace rk {

constexpr int T4_DOUBLES = 20;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int D>
struct Tpw {                                     // tiles per wave
    static constexpr int value = D == 3 ? 3 : 4;
};

struct T4Coord {
    int r, g, c, tau, b, blk;
    bool valid;
};

template <int D>
__device__ __forceinline__ T4Coord t4_coord(int wave, int lane, int n_tiles) {
    T4Coord t;
    t.r = lane >> 4; t.g = (lane >> 2) & 3; t.c = lane & 3;
    const int tau = wave * Tpw<D>::value + t.g;
    t.valid = t.g < Tpw<D>::value && tau < n_tiles;
    t.tau = t.valid ? tau : (wave * Tpw<D>::value < n_tiles ? wave * Tpw<D>::value : n_tiles - 1);
    t.b = t.tau / D; t.blk = t.tau - t.b * D;
    return t;
}

// value of block bb of this lane's trajectory, for every bb, given each tile's own value (tiles of one trajectory
// are adjacent 4-lane banks of the DPP row): masked row rotations, no selects
template <int D>
__device__ __forceinline__ void gather_blocks(double own, double (&vals)[D]) {
    if constexpr (D == 1) {
        vals[0] = own;
    } else if constexpr (D == 2) {
        vals[0] = pair_block0(own);
        vals[1] = pair_block1(own);
    } else {
        static_assert(D == 3, "gather_blocks: n_block in {1, 2, 3}");
        // tiles g = 0, 1, 2 are blocks 0, 1, 2;  ror:4k moves a value k tiles up
        vals[0] = dpp64_banks<0x128, 0x4>(dpp64_banks<0x124, 0x2>(own, own), own);   // g=1 <- g-1, g=2 <- g-2
        vals[1] = dpp64_banks<0x124, 0x4>(dpp64_banks<0x12C, 0x1>(own, own), own);   // g=0 <- g+1, g=2 <- g-1
        vals[2] = dpp64_banks<0x12C, 0x2>(dpp64_banks<0x128, 0x1>(own, own), own);   // g=0 <- g+2, g=1 <- g+1
    }
}

template <int D>
__device__ __forceinline__ double pick_block(const double (&v)[D], int blk) {
    if constexpr (D == 1) return v[0];
    else if constexpr (D == 2) return blk == 0 ? v[0] : v[1];
    else return blk == 0 ? v[0] : (blk == 1 ? v[1] : v[2]);
}

// ---- forward ---------------------------------------------------------------------------------------------------
template <class RHS, int ITG>
__global__ void __launch_bounds__(64) fwd_tile4_kernel(SolveArgs a, double* __restrict__ tiles) {
    constexpr int D = RHS::D, P = 4;
    static_assert(RHS::NDEP == 1, "tile path: right-hand sides that depend on X[b][0] only");
    const int n_tiles = a.B * D;
    const T4Coord tc = t4_coord<D>(blockIdx.x, threadIdx.x, n_tiles);
    const int r = tc.r, c = tc.c, b = tc.b, blk = tc.blk;

    const double Qt = ld(a.Q, ((size_t)blk * P + c) * P + r, a.Q_b, a.B, b);          // Q^T in D layout
    const double Rt = ld(a.R, ((size_t)blk * P + r) * P + c, a.R_b, a.B, b);
    const double RtT = ld(a.R, ((size_t)blk * P + c) * P + r, a.R_b, a.B, b);         // R^T in D layout
    const double Wr = ld(a.W, (size_t)blk * P + r, a.W_b, a.B, b);                    // row form
    const double Y0 = ld(a.Q, ((size_t)blk * P + 0) * P + r, a.Q_b, a.B, b);          // Q[0][k] at row k
    const double E0 = r == 0 ? 1.0 : 0.0;
    double th[RHS::NTHETA];
#pragma unroll
    for (int k = 0; k < RHS::NTHETA; ++k) th[k] = a.theta ? ld(a.theta, k, a.theta_b, a.B, b) : 0.0;

    double S = 0.0;                                                                    // solve.py:54
    double m = ld(a.x0, (size_t)blk * P + r, a.x0_b, a.B, b);                         // solve.py:53, row form
    const size_t tstride_all = (size_t)n_tiles * T4_DOUBLES;
    double* const dump = tiles + (size_t)(a.N + 1) * tstride_all + (size_t)blockIdx.x * 128;
    const bool st_m = tc.valid && c == 0;
    if constexpr (RHS::HAS_TILE3_FORM && D == 3) {
        // Lorenz63-type right-hand sides (config C3) in hand-trimmed form, after fwd_tile3_kernel: every instruction of
        // the wave lies on the step's dependent chain, so -- per-lane coefficients instead of three blocks + selects
        // (the measurement row W~ = W - J is constant, the offset a = J mu- - f is a bilinear form of this block's and
        // its neighbours' evaluation points, fetched with three plain DPP row rotations), cubic reciprocal step,
        // buffer stores with a scalar row base (slot-less lanes dropped by the range check).
        double kk[5];
        RHS::tile3_consts(blk, th, kk);
        const bool jac = ITG == RK_INTERROGATE_KRAMER;
        const double c0 = jac ? 0.0 : -kk[0], c1 = -kk[1], c2 = -kk[2], c3 = -kk[3], c4 = -kk[4];   // a = -f + J0 own
        const double Xw = fma(jac ? -kk[0] : 0.0, E0, Wr);          // W~ = W - J, row form (solve.py:79): constant
        const char* row = (const char*)(tiles + (size_t)blockIdx.x * Tpw<D>::value * T4_DOUBLES);
        const int voS = tc.valid ? (int)((tc.g * T4_DOUBLES + r * 4 + c) * sizeof(double)) : (int)0x80000000;
        const int voM = st_m ? (int)((tc.g * T4_DOUBLES + 16 + r) * sizeof(double)) : (int)0x80000000;
        auto store_row = [&](double vS, double vM) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, Tpw<D>::value * T4_DOUBLES * 8, 0x00020000);
            u32x2 bS, bM;
            __builtin_memcpy(&bS, &vS, 8);
            __builtin_memcpy(&bM, &vM, 8);
            __builtin_amdgcn_raw_buffer_store_b64(bS, rsrc, voS, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(bM, rsrc, voM, 0, 0);
        };
        store_row(S, m);
        for (int n = 0; n < a.N; ++n) {
            const double U = MF(S, Qt, 0.0);
            const double v_own = MF(Y0, m, 0.0);                      // (Q mu)_0 in all 16 lanes of the tile
            const double mp = MF(Qt, m, 0.0);                         // Q mu, row form
            const double SpT = MF(Qt, U, RtT);                        // exact transpose of S-: Q Sigma^T Q^T + R^T
            const double Sp = MF(U, Qt, Rt);                          // Q Sigma Q^T + R
            const double n1 = from_next_tile(v_own), p1 = from_prev_tile(v_own), p2 = dpp64<0x128>(v_own);
            const double a_meas = fma(c4, p2 * p1, fma(c3, p1 * n1, fma(c2, p1, fma(c1, n1, c0 * v_own))));
            const double Z = MF(SpT, Xw, 0.0);                        // Sigma- W~^T (standard.py:97), row form
            const double WS = MF(Xw, Sp, 0.0);
            const double yhat = MF(Xw, mp, a_meas);
            double Sc = MF(Z, Xw, 0.0);
            if constexpr (ITG == RK_INTERROGATE_RODEO) Sc = Sc + Sc;  // var_meas = W Sigma- W^T (interrogate.py:110-113)
            const double K = -Z * fast_rcp_cubic(Sc);
            S = fma(K, WS, Sp);
            m = fma(K, yhat, mp);
            row += tstride_all * sizeof(double);
            store_row(S, m);
        }
        return;
    }
    double* oS = tc.valid ? tiles + (size_t)tc.tau * T4_DOUBLES + r * 4 + c : dump + threadIdx.x;
    double* oM = st_m ? tiles + (size_t)tc.tau * T4_DOUBLES + 16 + r : dump + 64 + threadIdx.x;
    const size_t sS = tc.valid ? tstride_all : 0, sM = st_m ? tstride_all : 0;
    oS[0] = S;
    oM[0] = m;

    for (int n = 0; n < a.N; ++n) {
        const double U = MF(S, Qt, 0.0);
        const double v_own = MF(Y0, m, 0.0);                      // (Q mu)_0 in all 16 lanes of the tile
        const double mp = MF(Qt, m, 0.0);                         // Q mu, row form
        const double Sp = MF(U, Qt, Rt);                          // Q Sigma Q^T + R
        const double SpT = MF(Qt, U, RtT);                        // its exact transpose: Q Sigma^T Q^T + R^T
        // ---- interrogation (interrogate.py) ----
        double X[D][P], vals[D];
        gather_blocks<D>(v_own, vals);
#pragma unroll
        for (int bb = 0; bb < D; ++bb) {
#pragma unroll
            for (int j = 0; j < P; ++j) X[bb][j] = 0.0;
            X[bb][0] = vals[bb];
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
        double J0s[D];
#pragma unroll
        for (int bb = 0; bb < D; ++bb) J0s[bb] = J[bb][0];
        const double fb = pick_block<D>(f, blk), J0 = pick_block<D>(J0s, blk);
        const double a_meas = fma(J0, v_own, -fb);                // mean_meas = -f + J mu-   (interrogate.py:81-82)
        const double Xw = fma(-J0, E0, Wr);                       // W~ = W - J, row form     (solve.py:79)
        // ---- update (standard.py:93-102) ----
        const double yhat = MF(Xw, mp, a_meas);
        const double WS = MF(Xw, Sp, 0.0);
        const double Z = MF(SpT, Xw, 0.0);                        // Sigma- W~^T (standard.py:97), row form
        double Sc = MF(Z, Xw, 0.0);
        if constexpr (ITG == RK_INTERROGATE_RODEO) Sc = Sc + Sc;  // var_meas = W Sigma- W^T (interrogate.py:110-113)
        const double K = Z * fast_rcp(Sc);
        S = fma(-K, WS, Sp);
        m = fma(-K, yhat, mp);
        oS += sS; oM += sM;
        oS[0] = S;
        oM[0] = m;
    }
}

// ---- backward ----------------------------------------------------------------------------------------------------
constexpr int CH4 = 16;                          // time steps per hand-off
constexpr int ITEM4 = 512;                       // bytes per (step, tile): S- | G^T | S_f tiles, then m- (4), m_f (4)
constexpr int BUF4 = CH4 * 4 * ITEM4;            // 32 KiB

__device__ __forceinline__ int lds4_tile(int s, int g, int which, int idx) {
    const int item = s * 4 + g;
    return item * ITEM4 + which * 128 + (tile_slot(s, g, idx) << 3);
}
__device__ __forceinline__ int lds4_vec(int s, int g, int which, int rr) {
    const int item = s * 4 + g;
    return item * ITEM4 + 384 + (vec_slot(s, g, which, rr) << 3);
}

template <int D>
__global__ void __launch_bounds__(512) bwd_mv_tile4_kernel(SolveArgs a, double* __restrict__ tiles) {
    constexpr int P = 4, TPW = Tpw<D>::value;
    __shared__ __attribute__((aligned(16))) char lds_all[2 * 2 * BUF4];
    const int wave_id = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave_id == 4 || wave_id == 5) return;                      // placeholders: keep the consumers' SIMDs free
    const int n_tiles = a.B * D;
    const size_t tstride = (size_t)n_tiles * T4_DOUBLES;
    const int n_chunks = (a.N - 1 + CH4 - 1) / CH4;

    const int grp = wave_id & 1;
    const int role = wave_id < 2 ? 0 : (wave_id < 4 ? 1 : 2);      // 0 consumer, 1 / 2 producers of parity 0 / 1
    const int tw = blockIdx.x * 2 + grp;                           // tile-wave: tiles TPW tw .. TPW tw + TPW - 1
    char* const lds_raw = lds_all + grp * 2 * BUF4;
    double* const dump = tiles + (size_t)(a.N + 1) * tstride + (size_t)tw * 128;

    if (role >= 1) {
        // ---------------- producers: one lane per (step-in-chunk, tile) ----------------
        const int p = role - 1;
        const int s = lane >> 2, g = lane & 3;
        int tau = tw * TPW + (g < TPW ? g : 0);
        if (tau >= n_tiles) tau = n_tiles - 1;
        const int b = tau / D, blk = tau - b * D;
        double Q[P][P], R[P][P];
        load_block_consts<P>(a, blk, b, Q, R);
        const double* tin = tiles + (size_t)tau * T4_DOUBLES;
        int woff[16], voff[4], voff1[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) woff[i] = lds4_tile(s, g, 0, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) { voff[i] = lds4_vec(s, g, 0, i); voff1[i] = lds4_vec(s, g, 1, i); }
        double bufE[T4_DOUBLES], bufO[T4_DOUBLES];
        auto fetch = [&](int ch, double (&dst)[T4_DOUBLES]) {
            int n = a.N - 1 - ch * CH4 - s;
            n = n < 1 ? 1 : n;
            const double* in = tin + (size_t)n * tstride;
#pragma unroll
            for (int i = 0; i < T4_DOUBLES; ++i) dst[i] = in[i];
        };
        if (p < n_chunks) fetch(p, bufE);
        if (p + 2 < n_chunks) fetch(p + 2, bufO);
        double mf[P], Sf[P][P], mp[P], Sp[P][P], T[P][P];
        auto phaseA = [&](int chA, double (&buf)[T4_DOUBLES]) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
#pragma unroll
                for (int j = 0; j < P; ++j) Sf[i][j] = buf[i * 4 + j];
                mf[i] = buf[16 + i];
            }
            if (chA + 4 < n_chunks) fetch(chA + 4, buf);
            __builtin_amdgcn_sched_barrier(0);
            predict_block<P>(Q, R, mf, Sf, mp, Sp);              // pred[n+1] from filt[n]   (standard.py:57-59)
            mm_nt<P, P, P>(Sf, Q, T);                            // T = Sigma_f Q^T          (standard.py:175)
        };
        for (int t = -2; t < n_chunks; ++t) {
            const int chA = t + 2, chB = t + 1;
            if ((chA & 1) == p) {
                if (chA < n_chunks) {
                    if ((chA >> 1) & 1) phaseA(chA, bufO); else phaseA(chA, bufE);
                }
            } else if (chB >= 0 && chB < n_chunks) {
                double A[P][P], X[P][P];
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int j = 0; j < P; ++j) { A[i][j] = Sp[i][j]; X[i][j] = T[j][i]; }
                lu_solve<P, P>(A, X);                            // X = G^T                  (standard.py:176)
                const int n = a.N - 1 - chB * CH4 - s;
                if (n >= 1 && g < TPW) {
                    char* o = lds_raw + (chB & 1) * BUF4;
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
            __syncthreads();
        }
    } else {
        // ---------------- consumer ----------------
        const T4Coord tc = t4_coord<D>(tw, lane, n_tiles);
        const int r = tc.r, g = tc.g, c = tc.c, idx = r * 4 + c;
        const bool st_m = tc.valid && c == 0;
        double* bS = tc.valid ? tiles + (size_t)tc.tau * T4_DOUBLES + idx : dump + lane;
        double* bM = st_m ? tiles + (size_t)tc.tau * T4_DOUBLES + 16 + r : dump + 64 + lane;
        const size_t sS = tc.valid ? tstride : 0, sM = st_m ? tstride : 0;
        // carry = filt[N]  (solve.py:279-282); the mean in row form
        double Ss = tc.valid ? tiles[(size_t)a.N * tstride + (size_t)tc.tau * T4_DOUBLES + idx] : 0.0;
        double ms = tc.valid ? tiles[(size_t)a.N * tstride + (size_t)tc.tau * T4_DOUBLES + 16 + r] : 0.0;
        int roff[4], rvec[4], rvec1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            roff[k] = lds4_tile(k, g, 0, idx) - k * 4 * ITEM4;
            rvec[k] = lds4_vec(k, g, 0, r) - k * 4 * ITEM4;
            rvec1[k] = lds4_vec(k, g, 1, r) - k * 4 * ITEM4;
        }
        __syncthreads();                                            // tick -2
        __syncthreads();                                            // tick -1: chunk 0 is in LDS
        for (int t = 0; t < n_chunks; ++t) {
            const char* in = lds_raw + (t & 1) * BUF4;
            const int n_hi = a.N - 1 - t * CH4;
            const int cnt = __builtin_amdgcn_readfirstlane(n_hi >= CH4 ? CH4 : n_hi);
            double* oS = bS + (size_t)n_hi * sS;
            double* oM = bM + (size_t)n_hi * sM;
            auto step = [&](const char* q, const char* qv, const char* qw) {
                const double Sp = *(const double*)(q), Gt = *(const double*)(q + 128), Sf = *(const double*)(q + 256);
                const double mp = *(const double*)(qv), mf = *(const double*)(qw);
                const double V1 = MF(Ss - Sp, Gt, 0.0);             // (G D)^T
                ms = MF(Gt, ms - mp, mf);                           // mu_f + G (mu_s - mu-)     (standard.py:213-214)
                Ss = MF(V1, Gt, Sf);                                // Sigma_f + G D G^T         (standard.py:215-216)
                oS[0] = Ss; oM[0] = ms;
                oS -= sS; oM -= sM;
            };
            if (cnt == CH4) {
#pragma unroll
                for (int s = 0; s < CH4; ++s) step(in + roff[s & 3] + s * 4 * ITEM4, in + rvec[s & 3] + s * 4 * ITEM4,
                                                   in + rvec1[s & 3] + s * 4 * ITEM4);
            } else {
                for (int s = 0; s < cnt; ++s) step(in + lds4_tile(s, g, 0, idx), in + lds4_vec(s, g, 0, r), in + lds4_vec(s, g, 1, r));
            }
            __syncthreads();
        }
    }
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
template <class RHS>
static int launch_fwd_tile4(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles) {
    const dim3 grid(div_up(a.B * RHS::D, Tpw<RHS::D>::value)), block(64);
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
    else rc = launch_fwd_tile4<HigherOrder>(h, c, a, tiles);
    if (rc || mode == RK_MODE_FILTER || a.N < 2) return rc;
    const int tpw = a.D == 3 ? 3 : 4;
    const dim3 grid(div_up(div_up(a.B * a.D, tpw), 2)), block(512);
    LaunchTimer t(h, "bwd_mv_tile4_kernel");
    if (a.D == 1) hipLaunchKernelGGL((bwd_mv_tile4_kernel<1>), grid, block, 0, h->stream, a, tiles);
    else if (a.D == 2) hipLaunchKernelGGL((bwd_mv_tile4_kernel<2>), grid, block, 0, h->stream, a, tiles);
    else hipLaunchKernelGGL((bwd_mv_tile4_kernel<3>), grid, block, 0, h->stream, a, tiles);
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

}  // namespace rk
