// The column step of the dense LU's panel factorisation (solve_dense.hip: lu_panel_col, solve_dense_lu_regs.hpp: rl_panel_cols) and
// the wave reductions it uses.  A header of its own so that scripts/probe/probe16.hip times exactly this code in isolation.
// Needs LU_NB, fast_rcp (linalg_small.hpp) in scope.
#pragma once

namespace rk {

template <int CTRL>
__device__ __forceinline__ double dense_dpp64(double x) {
    int lo_ = __double2loint(x), hi_ = __double2hiint(x);
    lo_ = __builtin_amdgcn_update_dpp(lo_, lo_, CTRL, 0xF, 0xF, false);
    hi_ = __builtin_amdgcn_update_dpp(hi_, hi_, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi_, lo_);
}
// maximum over the wave, in every lane (values >= 0 or -1)
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dense_dpp64<0xB1>(v));                    // quad_perm [1,0,3,2]
    v = fmax(v, dense_dpp64<0x4E>(v));                    // quad_perm [2,3,0,1]
    v = fmax(v, dense_dpp64<0x141>(v));                   // row_half_mirror
    v = fmax(v, dense_dpp64<0x140>(v));                   // row_mirror
    double m = -1.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lo_ = __builtin_amdgcn_readlane(__double2loint(v), 16 * r), hi_ = __builtin_amdgcn_readlane(__double2hiint(v), 16 * r);
        m = fmax(m, __hiloint2double(hi_, lo_));
    }
    return m;
}
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    int m = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 4; ++r) m = min(m, __builtin_amdgcn_readlane(v, 16 * r));
    return m;
}

__device__ __forceinline__ double rl_readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {          // maximum over the wave, uniform
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) m = max(m, (unsigned)__builtin_amdgcn_readlane((int)v, 16 * r));
    return m;
}

// maximum over the wave (uniform): four DPP steps with the lane exchange INSIDE the max (v_max_u32_dpp: hipcc's update_dpp
// builtin makes a move, two wait states and a max of each), then the four rows' values through readlane
__device__ __forceinline__ unsigned wave_max_u32_fused(unsigned v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    const unsigned m0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), m1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned m2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), m3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(m0, m1), max(m2, m3));
}

// One column step of the panel.  The rows that were still active when the panel began are dealt to the lanes of wave 0 in
// compact order (RS rows per lane: 3, 2, 1 as the elimination proceeds), in place; pos[s] = LAPACK position of the row
// (0x7fffffff in an empty slot), gpos = 16 k + J the position this step fills (the panel loop over memory: positions
// relative to the panel, gpos = J, n = its rows).  Returns the pivot's position.
// Written for the VECTOR unit alone: the first versions (compound conditions -> s_and / s_or on compare masks, exec-masked
// updates, a branch per slot) took 1.5 k cycles per column for ~120 executed instructions -- every hop compare -> scalar
// logic -> select is a pipeline round trip (stamps of round 4: 23.6 k cycles per 160-row panel even with its code warm).
// Here every select hangs on ONE compare (VCC), the rows' |a| are compared as 64-bit integers (|a| >= 0: same order), rows
// out of play carry the key 0, the update multiplies them by an exact zero instead of masking them, and the scalar unit is
// left with the wave reduction, the uniqueness test and the pivot row's readlanes.
template <int RS, int J>
__device__ __forceinline__ int rl_panel_col(double (&a)[RS][LU_NB], int (&pos)[RS], int gpos, int n) {
    typedef unsigned long long u64;
    const unsigned span = (unsigned)(n - gpos);
    // ---- pivot: largest |a|, smallest position, over the rows whose position is >= gpos ----
    u64 bk = (unsigned)(pos[0] - gpos) < span ? (u64)__double_as_longlong(fabs(a[0][J])) + 1ull : 0ull;
    int bs = 0;
    unsigned tie = 0u;                                          // != 0: two candidates of this lane have the same |a| (exact path)
#pragma unroll
    for (int s = 1; s < RS; ++s) {
        const u64 ks = (unsigned)(pos[s] - gpos) < span ? (u64)__double_as_longlong(fabs(a[s][J])) + 1ull : 0ull;
        tie |= ks == bk ? (unsigned)(ks | (ks >> 32)) : 0u;
        const bool better = ks > bk;
        bk = better ? ks : bk;
        bs = better ? s : bs;
    }
    int bp = pos[0];
#pragma unroll
    for (int s = 1; s < RS; ++s) bp = bs == s ? pos[s] : bp;
    // The wave maximum is found on the high words; one matching lane is the rule, and then its candidate is the pivot.
    // Ties (on the high word across lanes, or exact ones inside a lane) take the exact path.
    const unsigned key = bk != 0ull ? (unsigned)(bk >> 32) + 1u : 0u;
    const unsigned mkey = wave_max_u32_fused(key);
    int pj, ol, os;
    const u64 mm = __builtin_amdgcn_ballot_w64(key == mkey);
    const u64 tt = RS > 1 ? __builtin_amdgcn_ballot_w64(tie != 0u) : 0ull;
    if (mkey != 0u && __builtin_popcountll(mm) == 1 && tt == 0ull) {
        ol = (int)__builtin_ctzll(mm);
        pj = __builtin_amdgcn_readlane(bp, ol);
        os = RS > 1 ? __builtin_amdgcn_readlane(bs, ol) : 0;
    } else {
        if (mkey == 0u) {
            pj = gpos;                                          // nothing in play (n reached): keep the row
        } else {
            double best = -1.0;
            int bq = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < RS; ++s) {
                const double v = fabs(a[s][J]);
                if ((unsigned)(pos[s] - gpos) < span && (v > best || (v == best && pos[s] < bq))) { best = v; bq = pos[s]; }
            }
            const double m2 = wave_max_f64(best);
            pj = wave_min_i32(best == m2 ? bq : 0x7fffffff);
            if (pj == 0x7fffffff) pj = gpos;
        }
        int ms = -1;
#pragma unroll
        for (int s = RS - 1; s >= 0; --s) ms = pos[s] == pj ? s : ms;
        ol = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(ms >= 0));
        os = __builtin_amdgcn_readlane(ms, ol);
    }
    // ---- the pivot row to every lane (scalars) ----
    double prow[LU_NB];
    if (RS == 1 || os == 0) {
#pragma unroll
        for (int c = J; c < LU_NB; ++c) prow[c] = rl_readlane_f64(a[0][c], ol);
    } else if (RS == 2 || os == 1) {
#pragma unroll
        for (int c = J; c < LU_NB; ++c) prow[c] = rl_readlane_f64(a[RS > 1 ? 1 : 0][c], ol);
    } else {
#pragma unroll
        for (int c = J; c < LU_NB; ++c) prow[c] = rl_readlane_f64(a[RS > 2 ? 2 : 0][c], ol);
    }
    const double rinv = fast_rcp(prow[J]);
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        int t = pos[s] == pj ? gpos : pos[s];                    // interchange gpos <-> pj
        t = pos[s] == gpos ? pj : t;
        pos[s] = t;
        const bool below = (unsigned)(t - gpos - 1) < span - 1u;             // still below the diagonal
        const double lm = a[s][J] * rinv;
        const double lmz = below ? lm : 0.0;                    // (rows out of play: a - 0 * u = a exactly)
        a[s][J] = below ? lm : a[s][J];
#pragma unroll
        for (int c = J + 1; c < LU_NB; ++c) a[s][c] = fma(-lmz, prow[c], a[s][c]);
    }
    return pj;
}

}  // namespace rk
