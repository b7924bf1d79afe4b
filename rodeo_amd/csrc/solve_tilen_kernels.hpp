// Forward kernel of the BLOCKED MFMA-tile path: n_bstate = 4 .. 4 NB (NB = 1, 2: up to 8 derivatives per variable), any of
// the four interrogations, as a template over the right-hand side -- built ahead of time for the built-in ODEs
// (solve_tilen.hip) and by hiprtc for user-supplied ones (rhs_jit.hip).  RTC-safe.
//
// The reference treats n_deriv as a free argument (src/rodeo/solve.py:48, prior/ibm.py:65-88).  The hand-trimmed tile
// kernels exist for p = 3 and p = 4 only; here a p x p block (zero-padded to 4 NB) is an NB x NB array of 4 x 4 tiles,
// each tile one double per lane in the D layout of v_mfma_f64_4x4x4_4b_f64 (mfma_tile.hpp: MF(X, Y, Z) = X^T Y + Z per
// tile, four (trajectory, block) units per wave), and every product of the step is the blocked form of the tile4 step
// (solve_tile4_kernels.hpp), term for term the reference's predict / update (standard.py:57-59, 93-102):
//     U  [a][b] = sum_k S [k][a]^T Qt[k][b]            = (Q Sigma)^T
//     S- [a][b] = sum_k U [k][a]^T Qt[k][b] + R [a][b] = Q Sigma Q^T + R
//     S-T[a][b] = sum_k Qt[k][a]^T U [k][b] + R^T[a][b] (the exact transpose, for Sigma- W~^T in row form)
//     m- [a]    = sum_k Qt[k][a]^T m[k]                 (row form: lane (r, ., c) holds component 4 a + r)
//     yhat = sum_k Xw[k]^T m-[k] + a ;  WS[b] = sum_k Xw[k]^T S-[k][b] ;  Z[a] = sum_k S-T[k][a]^T Xw[k] ;  s = sum_k Z[k]^T Xw[k]
//     Sigma = S- - (Z / s) WS ;  mu = m- - (Z / s) yhat
// Zero padding is self-consistent: padded rows / columns of Q, R, W, Sigma stay exact zeros through every product.
// 9 MFMAs per step at NB = 1, 42 at NB = 2 (+2 for interrogate_chkrebtii's draw).
//
// HBM format RK_LAYOUT_TILEP (for p = 4 identical to RK_LAYOUT_TILE4): per time step and unit the p*p + p doubles
// [Sigma row-major | mu] -- the algorithmic d p (p+1) 8 bytes per trajectory-step.
#pragma once
#include "rk_enums.hpp"
#include "kalman_small.hpp"
#include "mfma_tile.hpp"
#include "philox.hpp"
#include "solve_args.hpp"
#include "solve_tile3_kernels.hpp"        // TileWaves, TILE_MAX_BLOCKS, gather through LDS for n_block > 4

namespace rk {

// C[a][b] = sum_k X[k][a]^T Y[k][b] + Z[a][b]   (blocked X^T Y + Z)
template <int NB>
__device__ __forceinline__ void bmm_tn(const double (&X)[NB][NB], const double (&Y)[NB][NB], const double (&Z)[NB][NB],
                                       double (&C)[NB][NB]) {
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double acc = MF(X[0][a], Y[0][b], Z[a][b]);
#pragma unroll
            for (int k = 1; k < NB; ++k) acc = MF(X[k][a], Y[k][b], acc);
            C[a][b] = acc;
        }
}
template <int NB>
__device__ __forceinline__ void bmm_tn0(const double (&X)[NB][NB], const double (&Y)[NB][NB], double (&C)[NB][NB]) {
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double acc = MF(X[0][a], Y[0][b], 0.0);
#pragma unroll
            for (int k = 1; k < NB; ++k) acc = MF(X[k][a], Y[k][b], acc);
            C[a][b] = acc;
        }
}
// y[a] = sum_k X[k][a]^T v[k] + z[a]   (blocked X^T v for a row-form vector v)
template <int NB>
__device__ __forceinline__ void bmv_t(const double (&X)[NB][NB], const double (&v)[NB], const double (&z)[NB], double (&y)[NB]) {
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        double acc = MF(X[0][a], v[0], z[a]);
#pragma unroll
        for (int k = 1; k < NB; ++k) acc = MF(X[k][a], v[k], acc);
        y[a] = acc;
    }
}
// scalar (uniform in the 16 lanes of a unit) sum_k u[k]^T v[k] + z for row-form u, v
template <int NB>
__device__ __forceinline__ double bdot(const double (&u)[NB], const double (&v)[NB], double z) {
    double acc = MF(u[0], v[0], z);
#pragma unroll
    for (int k = 1; k < NB; ++k) acc = MF(u[k], v[k], acc);
    return acc;
}

template <class RHS, int ITG, int NB>
__global__ void __launch_bounds__(64 * TileWaves<RHS::D>::value) fwd_tilen_kernel(SolveArgs a, double* __restrict__ tiles, int P) {
    constexpr int D = RHS::D, NW = TileWaves<D>::value, TPW = NW > 1 ? 4 : Tpw<D>::value;
    static_assert(D >= 1 && D <= TILE_MAX_BLOCKS, "tile path: n_block in 1..64");
    static_assert(RHS::NDEP == 1, "tile path: right-hand sides that depend on X[b][0] only");
    const int n_units = a.B * D, PP = P * P + P;
    const int wave_in_wg = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane >> 4, g = (lane >> 2) & 3, c = lane & 3;
    int b, blk;
    bool valid;
    if constexpr (NW == 1) {
        const int tau = blockIdx.x * TPW + g;
        valid = g < TPW && tau < n_units;
        const int tc = valid ? tau : (blockIdx.x * TPW < n_units ? blockIdx.x * TPW : n_units - 1);   // idle slots repeat a real unit
        b = tc / D; blk = tc - b * D;
    } else {
        const int blk_w = wave_in_wg * 4 + g;
        valid = blk_w < D;
        b = blockIdx.x; blk = valid ? blk_w : D - 1;
    }

    // ---- per-lane constants, zero-padded to 4 NB ----
    double Qt[NB][NB], Rt[NB][NB], RtT[NB][NB], Wr[NB], Y0[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int i = 4 * k + r;
        Wr[k] = i < P ? ld(a.W, (size_t)blk * P + i, a.W_b, a.B, b) : 0.0;
        Y0[k] = i < P ? ld(a.Q, ((size_t)blk * P + 0) * P + i, a.Q_b, a.B, b) : 0.0;        // Q[0][i] at row i
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int j = 4 * bb + c;
            const bool in = i < P && j < P;
            Qt[k][bb] = in ? ld(a.Q, ((size_t)blk * P + j) * P + i, a.Q_b, a.B, b) : 0.0;     // (Q^T)[i][j]
            Rt[k][bb] = in ? ld(a.R, ((size_t)blk * P + i) * P + j, a.R_b, a.B, b) : 0.0;
            RtT[k][bb] = in ? ld(a.R, ((size_t)blk * P + j) * P + i, a.R_b, a.B, b) : 0.0;
        }
    }
    const double E0 = r == 0 ? 1.0 : 0.0;
    double th[RHS::NTHETA];
#pragma unroll
    for (int k = 0; k < RHS::NTHETA; ++k) th[k] = a.theta ? ld(a.theta, k, a.theta_b, a.B, b) : 0.0;
    __shared__ double zbuf_all[NW][4 * 16];            // chkrebtii: z_0 of the next 16 steps for each of the wave's 4 units
    double* const zbuf = zbuf_all[wave_in_wg];
    __shared__ double vx[2][NW > 1 ? TILE_MAX_BLOCKS : 1];
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);

    // ---- state: Sigma = 0, mu = ode_init (solve.py:53-54) ----
    double S[NB][NB], m[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int i = 4 * k + r;
        m[k] = i < P ? ld(a.x0, (size_t)blk * P + i, a.x0_b, a.B, b) : 0.0;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) S[k][bb] = 0.0;
    }
    // ---- output slots of this lane inside its unit's [Sigma | mu] record ----
    // Stores go through a buffer window on this wave's units of the current time row (scalar base, per-lane byte offset):
    // lanes without a slot (padding, idle units) carry an out-of-range offset and are dropped by the hardware, so the
    // step has no exec-mask branches around its stores.
    const size_t tstride = (size_t)n_units * PP;
    const size_t base_unit = NW == 1 ? (size_t)blockIdx.x * TPW : (size_t)blockIdx.x * D + (size_t)wave_in_wg * 4;
    const char* row = (const char*)(tiles + base_unit * PP);
    int offS[NB][NB], offM[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int i = 4 * k + r;
        offM[k] = (valid && c == 0 && i < P) ? (int)((g * PP + P * P + i) * sizeof(double)) : (int)0x80000000;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int j = 4 * bb + c;
            offS[k][bb] = (valid && i < P && j < P) ? (int)((g * PP + i * P + j) * sizeof(double)) : (int)0x80000000;
        }
    }
    // (called one step late, behind the first MFMAs of the next step -- `after` is one of their results: an empty asm
    // that reads it and clobbers memory keeps the stores behind it; solve_tile3_kernels.hpp, RK_STORE_BEHIND)
    auto store = [&](double after) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, TPW * PP * 8, 0x00020000);
        asm volatile("" :: "v"(after) : "memory");               // (the stores stay behind that MFMA; no copies of the state)
        auto put = [&](double v, int off) {
            u32x2 bits;
            __builtin_memcpy(&bits, &v, 8);
            __builtin_amdgcn_raw_buffer_store_b64(bits, rsrc, off, 0, 0);
        };
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            put(m[k], offM[k]);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) put(S[k][bb], offS[k][bb]);
        }
    };

    double tk[tile_form_consts<RHS>::N];
    if constexpr (rhs_has_tile_form<RHS>::value && D == 2) RHS::tile_consts(blk, th, tk);
    double kk3[5];
    if constexpr (rhs_has_tile3_form<RHS>::value && D == 3) RHS::tile3_consts(blk, th, kk3);
    for (int n = 0; n < a.N; ++n) {
        if constexpr (ITG == RK_INTERROGATE_CHKREBTII) {
            if ((n & 15) == 0) {                          // the 16 lanes of a unit draw z_0 for 16 consecutive steps
                double z0, z1;
                normal_pair(a.seed, traj, (uint32_t)(n + r * 4 + c), (uint32_t)blk, PURPOSE_INTERROGATE, 0u, z0, z1);
                zbuf[g * 16 + r * 4 + c] = z0;
            }
        }
        // ---- predict (standard.py:57-59) ----
        double U[NB][NB], Sp[NB][NB], SpT[NB][NB], mp[NB], zero[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) zero[k] = 0.0;
        bmm_tn0<NB>(S, Qt, U);
        store(U[0][0]);                                   // the state of time n
        row += tstride * sizeof(double);
        double v_own = bdot<NB>(Y0, m, 0.0);             // (Q mu)_0 in all 16 lanes of the unit: the evaluation point X[b][0]
        bmv_t<NB>(Qt, m, zero, mp);
        bmm_tn<NB>(U, Qt, Rt, Sp);
        bmm_tn<NB>(Qt, U, RtT, SpT);
        if constexpr (ITG == RK_INTERROGATE_CHKREBTII) {
            // interrogate.py:22-34: x ~ N(mu-, Sigma-) through the lower factor; only x_0 = mu-_0 + sqrt(Sigma-_00) z_0 reaches f
            const double zn = zbuf[g * 16 + (n & 15)];
            const double R0 = MF(E0, Sp[0][0], 0.0);     // row 0 of Sigma- in every row
            const double s00 = quad_bcast0(R0);
            v_own = fma(sqrt(s00 > 0.0 ? s00 : 0.0), zn, v_own);
        }
        // ---- interrogation (interrogate.py): f and the block-diagonal Jacobian entry at the evaluation points ----
        const double t = a.t_min + (a.t_max - a.t_min) * (double)(n + 1) / (double)a.N;     // solve.py:74
        double fb, J0;
        if constexpr (rhs_has_tile_form<RHS>::value && D == 2 && NW == 1) {
            // two-block tile form (FitzHugh-Nagumo): the other block's point by one DPP move, per-lane coefficients
            RHS::tile_eval(tk, v_own, pair_other_quad_uniform(v_own), t, fb, J0);
            if constexpr (ITG != RK_INTERROGATE_KRAMER) J0 = 0.0;
        } else if constexpr (rhs_has_tile3_form<RHS>::value && D == 3 && NW == 1) {
            // Lorenz63-type right-hand sides: f is a bilinear form of this block's and its neighbours' evaluation points
            // (three plain DPP row rotations), the Jacobian entry a constant (fwd_tile4_kernel's trimmed step)
            const double n1 = from_next_tile(v_own), p1 = from_prev_tile(v_own), p2 = dpp64<0x128>(v_own);
            fb = fma(kk3[4], p2 * p1, fma(kk3[3], p1 * n1, fma(kk3[2], p1, fma(kk3[1], n1, kk3[0] * v_own))));
            J0 = ITG == RK_INTERROGATE_KRAMER ? kk3[0] : 0.0;
        } else {
        double X[D][1];
        if constexpr (NW == 1) {
            double vals[D];
            gather_blocks<D>(v_own, vals);
#pragma unroll
            for (int bb = 0; bb < D; ++bb) X[bb][0] = vals[bb];
        } else {
            if (valid && r == 0 && c == 0) vx[n & 1][blk] = v_own;
            __syncthreads();
#pragma unroll
            for (int bb = 0; bb < D; ++bb) X[bb][0] = vx[n & 1][bb];
        }
        if constexpr (ITG == RK_INTERROGATE_KRAMER && rhs_has_fjac0<RHS>::value) {
            RHS::template fjac0_block<1>(X, t, th, blk, fb, J0);
        } else if constexpr (ITG != RK_INTERROGATE_KRAMER && rhs_has_f_block<RHS>::value) {
            fb = RHS::template f_block<1>(X, t, th, blk);       // this lane's block alone
            J0 = 0.0;
        } else {
            double f[D], J[D][1];
            if constexpr (ITG == RK_INTERROGATE_KRAMER) {
                RHS::template fjac<1>(X, t, th, f, J);
            } else {
                RHS::template f<1>(X, t, th, f);
#pragma unroll
                for (int bb = 0; bb < D; ++bb) J[bb][0] = 0.0;
            }
            double J0s[D];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) J0s[bb] = J[bb][0];
            fb = pick_block<D>(f, blk); J0 = pick_block<D>(J0s, blk);
        }
        }
        // kramer: mean_meas = -f + J mu- (interrogate.py:81-82; J has only its first entry); the others: -f(x)
        const double a_meas = ITG == RK_INTERROGATE_KRAMER ? fma(J0, v_own, -fb) : -fb;
        double Xw[NB];                                    // W~ = W - J, row form (solve.py:79)
#pragma unroll
        for (int k = 0; k < NB; ++k) Xw[k] = Wr[k];
        Xw[0] = fma(-J0, E0, Wr[0]);
        // ---- update (standard.py:93-102) ----
        const double yhat = bdot<NB>(Xw, mp, a_meas);
        double WS[NB], Z[NB];
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            double acc = MF(Xw[0], Sp[0][bb], 0.0);
#pragma unroll
            for (int k = 1; k < NB; ++k) acc = MF(Xw[k], Sp[k][bb], acc);
            WS[bb] = acc;                                 // column form: (W~ Sigma-)_j
        }
        bmv_t<NB>(SpT, Xw, zero, Z);                      // Sigma- W~^T (standard.py:97), row form
        double Sc = bdot<NB>(Z, Xw, 0.0);
        if constexpr (ITG == RK_INTERROGATE_RODEO || ITG == RK_INTERROGATE_CHKREBTII)
            Sc = Sc + Sc;                                 // + var_meas = W Sigma- W^T (interrogate.py:110-113, 26-29)
        const double rS = fast_rcp(Sc);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const double K = Z[k] * rS;
            m[k] = fma(-K, yhat, mp[k]);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) S[k][bb] = fma(-K, WS[bb], Sp[k][bb]);
        }
    }
    store(0.0);                                           // time N
}

}  // namespace rk
