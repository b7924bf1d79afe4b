// Forward kernel of the p = 4 MFMA-tile path (see solve_tile4.hip), as a template over the right-hand side: built ahead
// of time for the built-in ODEs and by hiprtc for user-supplied ones (rhs_jit.hip).  RTC-safe.
#pragma once
#include "rk_enums.hpp"
#include "kalman_small.hpp"
#include "mfma_tile.hpp"
#include "solve_args.hpp"

namespace rk {

constexpr int T4_DOUBLES = 20;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

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
    if constexpr (rhs_has_tile3_form<RHS>::value && D == 3) {
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
        // (the state of time n is stored one step late, behind the first MFMA of step n + 1: solve_tile3_kernels.hpp)
        // (round 4: the step written in "issue order" -- the covariance chain U -> S-^T -> Z -> S_c -> 1 / S_c first, every MFMA as soon as
        //  its operand is back, the offset's DPP moves and FMAs between the links of the reciprocal, pinned with sched_barrier(0) -- took
        //  372 cycles per step against hipcc's own order's 307: the in-order model "17 cycles of issue, 29 of latency per MFMA" that
        //  explains the chains of probe15 does not predict the mixed MFMA / VALU stream; left to the compiler)
        for (int n = 0; n < a.N; ++n) {
            const double U = MF(S, Qt, 0.0);
            asm volatile("" :: "v"(U) : "memory");                   // (the stores stay behind U; no copy of m, which lives on)
            store_row(S, m);
            row += tstride_all * sizeof(double);
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
        }
        store_row(S, m);
        return;
    }
    if constexpr (rhs_has_tile_form<RHS>::value && D == 2) {
        // Right-hand sides with the two-block tile form of the p = 3 kernel (FitzHugh-Nagumo: f and J0 of a block are
        // polynomials in its own and the other block's evaluation point, RHS::tile_eval) without the generic path's
        // gathers, block arrays and selects: the other block's point by one row_half_mirror move, per-lane coefficients,
        // buffer stores with a scalar row base.  Same arithmetic as the generic path below.
        double tk[RHS::NTILEK];
        RHS::tile_consts(blk, th, tk);
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
        for (int n = 0; n < a.N; ++n) {
            const double U = MF(S, Qt, 0.0);
            asm volatile("" :: "v"(U) : "memory");                   // (the state of time n leaves behind U: solve_tile3_kernels.hpp)
            store_row(S, m);
            row += tstride_all * sizeof(double);
            const double v_own = MF(Y0, m, 0.0);                      // (Q mu)_0 in all 16 lanes of the tile
            const double mp = MF(Qt, m, 0.0);                         // Q mu, row form
            const double Sp = MF(U, Qt, Rt);                          // Q Sigma Q^T + R
            const double SpT = MF(Qt, U, RtT);                        // its exact transpose
            const double v_oth = pair_other_quad_uniform(v_own);      // the other block's evaluation point
            const double t = a.t_min + (a.t_max - a.t_min) * (double)(n + 1) / (double)a.N;     // solve.py:74
            double fb, J0;
            RHS::tile_eval(tk, v_own, v_oth, t, fb, J0);
            if constexpr (ITG != RK_INTERROGATE_KRAMER) J0 = 0.0;
            const double a_meas = fma(J0, v_own, -fb);                // mean_meas = -f + J mu-   (interrogate.py:81-82)
            const double Xw = fma(-J0, E0, Wr);                       // W~ = W - J, row form     (solve.py:79)
            const double yhat = MF(Xw, mp, a_meas);
            const double WS = MF(Xw, Sp, 0.0);
            const double Z = MF(SpT, Xw, 0.0);                        // Sigma- W~^T (standard.py:97), row form
            double Sc = MF(Z, Xw, 0.0);
            if constexpr (ITG == RK_INTERROGATE_RODEO) Sc = Sc + Sc;  // var_meas = W Sigma- W^T (interrogate.py:110-113)
            const double K = Z * fast_rcp_cubic(Sc);
            S = fma(-K, WS, Sp);
            m = fma(-K, yhat, mp);
        }
        store_row(S, m);
        return;
    }
    double* oS = tc.valid ? tiles + (size_t)tc.tau * T4_DOUBLES + r * 4 + c : dump + threadIdx.x;
    double* oM = st_m ? tiles + (size_t)tc.tau * T4_DOUBLES + 16 + r : dump + 64 + threadIdx.x;
    const size_t sS = tc.valid ? tstride_all : 0, sM = st_m ? tstride_all : 0;
    for (int n = 0; n < a.N; ++n) {
        const double U = MF(S, Qt, 0.0);
        asm volatile("" :: "v"(U) : "memory");
        oS[0] = S;
        oM[0] = m;
        oS += sS; oM += sM;
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
        double fb, J0;
        if constexpr (ITG == RK_INTERROGATE_KRAMER && rhs_has_fjac0<RHS>::value) {
            RHS::template fjac0_block<P>(X, t, th, blk, fb, J0);    // one evaluation, one dual direction (dual.hpp)
        } else {
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
            fb = pick_block<D>(f, blk); J0 = pick_block<D>(J0s, blk);
        }
        const double a_meas = fma(J0, v_own, -fb);                // mean_meas = -f + J mu-   (interrogate.py:81-82)
        const double Xw = fma(-J0, E0, Wr);                       // W~ = W - J, row form     (solve.py:79)
        // ---- update (standard.py:93-102) ----
        const double yhat = MF(Xw, mp, a_meas);
        const double WS = MF(Xw, Sp, 0.0);
        const double Z = MF(SpT, Xw, 0.0);                        // Sigma- W~^T (standard.py:97), row form
        double Sc = MF(Z, Xw, 0.0);
        if constexpr (ITG == RK_INTERROGATE_RODEO) Sc = Sc + Sc;  // var_meas = W Sigma- W^T (interrogate.py:110-113)
        const double K = Z * fast_rcp_cubic(Sc);
        S = fma(-K, WS, Sp);
        m = fma(-K, yhat, mp);
    }
    oS[0] = S;
    oM[0] = m;
}

}  // namespace rk
