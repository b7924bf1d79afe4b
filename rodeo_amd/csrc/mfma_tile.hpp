// 4x4 fp64 tiles spread over the lanes of a wavefront, multiplied with v_mfma_f64_4x4x4_4b_f64.
//
// Why: with one lane per trajectory the N-step recursion is VALU-issue-bound (~335 dependent-ish fp64 ops per step on
// 16 waves for the headline config).  Spreading each (trajectory, block) state over 16 lanes cuts the per-wave
// instruction count ~10x, but then every matrix product needs cross-lane operands; the 4-block 4x4x4 fp64 MFMA does
// that data movement inside the matrix pipe (no DPP / LDS shuffles) at the same FLOP rate as the VALU.
//
// Lane map (measured on gfx950, profiles/r01_probe_f64_mfma_layout_and_latency.log):
//     lane = 16*r + 4*g + c        r = row (0..3), g = tile within the wave (0..3), c = column (0..3)
//   A operand: lane holds A_g[i = c][k = r]   i.e. a register holding matrix X "in D layout" is read as X^T
//   B operand: lane holds B_g[k = r][j = c]   (D layout)
//   C / D    : lane holds D_g[i = r][j = c]   (D layout)
// so for registers X, Y, Z holding one matrix element per lane in D layout:   MF(X, Y, Z) = X^T Y + Z   per tile.
// Latency 29 cycles through A/B, 21 through C; issue 16 cycles (profiles/r01_probe2_fp64_valu_mfma_latency.log).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include "linalg_small.hpp"

namespace rk {

__device__ __forceinline__ double MF(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// (for controls that give EVERY lane a source -- quad_perm, row_ror, the mirrors: the move then has no `old` operand and
// hipcc does not copy x first when x stays live: six v_mov per step of the Lorenz63 p = 4 kernel)
template <int CTRL>
__device__ __forceinline__ double dpp64(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// DPP move restricted to the banks (= tiles g, 4 lanes each) in BANK_MASK; other lanes keep `old`
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ double dpp64_banks(double old, double x) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, BANK_MASK, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
// For n_block = 2 (tiles g = 0,1 and 2,3 are the two blocks of one trajectory): the value of block 0 / block 1 of
// this lane's trajectory, given each tile's own value x.  One masked DPP move per 32-bit half, no selects.
__device__ __forceinline__ double pair_block0(double x) { return dpp64_banks<0x124, 0xA>(x, x); }   // odd g <- g-1
__device__ __forceinline__ double pair_block1(double x) { return dpp64_banks<0x12C, 0x5>(x, x); }   // even g <- g+1

// Value of the OTHER block of this lane's trajectory (n_block = 2) for a value that is already the same in the four
// lanes of every (r, g) quad: row_half_mirror reverses each
// 8-lane half of a DPP row, lane (g, c) <- (g ^ 1, 3 - c), one move per 32-bit half instead of two moves + a copy
__device__ __forceinline__ double pair_other_quad_uniform(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x141, 0xf, 0xf, false);   // no `old`: every lane has a source
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x141, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// broadcast column 3 of every (r, g) quad to its four lanes
__device__ __forceinline__ double quad_bcast3(double x) { return dpp64<0xFF>(x); }
__device__ __forceinline__ double quad_bcast0(double x) { return dpp64<0x00>(x); }
// value held by the same (r, c) lane of tile g-1 / g+1 (cyclic within the 16-lane row)
__device__ __forceinline__ double from_prev_tile(double x) { return dpp64<0x124>(x); }   // row_ror:4
__device__ __forceinline__ double from_next_tile(double x) { return dpp64<0x12C>(x); }   // row_ror:12

// ---- the blocks of one trajectory inside a wave (forward kernels) --------------------------------------------------
// A forward wave carries whole trajectories: with n_block = D the tiles g = 0..TPW-1 of a wave are 4 / D trajectories
// (D = 1, 2, 4) or the three blocks of one (D = 3, tile slot 3 idles).
template <int D>
struct Tpw {                                     // tiles per forward wave
    static constexpr int value = D == 3 ? 3 : 4;
};

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
        static_assert(D == 3 || D == 4, "gather_blocks: n_block in {1, 2, 3, 4} (more blocks go through LDS)");
        // three or four blocks: the wave carries ONE trajectory (tile g = block g) and a tile's value is the same in its 16
        // lanes, so block k's value is lane 4 k of the wave, read into scalar registers: two v_readlane per value instead
        // of two or three masked DPP rotations with a copy each (the generic p = 3 step of a three-variable ODE spent more
        // time here than in its seven MFMAs)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(own), 4 * k), hi = __builtin_amdgcn_readlane(__double2hiint(own), 4 * k);
            vals[k] = __hiloint2double(hi, lo);
        }
    }
}

template <int D>
__device__ __forceinline__ double pick_block(const double (&v)[D], int blk) {
    if constexpr (D == 1) return v[0];
    else if constexpr (D == 2) return blk == 0 ? v[0] : v[1];
    else if constexpr (D == 3) return blk == 0 ? v[0] : (blk == 1 ? v[1] : v[2]);
    else if constexpr (D == 4) return blk < 2 ? (blk == 0 ? v[0] : v[1]) : (blk == 2 ? v[2] : v[3]);
    else {
        double x = v[0];
#pragma unroll
        for (int k = 1; k < D; ++k) x = blk == k ? v[k] : x;
        return x;
    }
}

// ---- LDS hand-off swizzles of the backward kernels (producer lanes: one per item; consumer lanes: tile elements) ----
// Slot (0..15, in doubles) of element (r, c) inside a 128-byte hand-off tile of item (s, g).  Two access patterns
// must both be free of LDS bank conflicts (the tiles of all items start at the same bank):
//   producer stores: 16 consecutive lanes = 16 items (s & 3, g), one element (r, c)   -> slot bijective in (s & 3, g)
//   consumer loads : 16 / 32 consecutive lanes = one item row s, lanes (g, c), fixed r -> slot bijective in (g, c)
// slot = 4 (g ^ r) + (c ^ (s & 3)) satisfies both (measured: the earlier idx ^ item swizzle served only the stores
// and made every consumer load a 4-way conflict, 1300 of 2800 cycles per 16-step tick).
__device__ __forceinline__ int tile_slot(int s, int g, int idx) {
    const int r = idx >> 2, c = idx & 3;
    return 4 * ((g ^ r) & 3) + ((c ^ s) & 3);
}
// Same for a 4-vector `which` (0, 1) of item (s, g) inside the item's 128-byte vector area; the consumer's lanes
// (r, g, c) load element r (c broadcasts).
__device__ __forceinline__ int vec_slot(int s, int g, int which, int rr) {
    return 4 * ((s ^ (2 * which) ^ (g & 1)) & 3) + ((rr ^ g) & 3);
}

// ---- global -> LDS prefetch without registers (LDS-DMA) ----
// A prefetch held in VGPRs across loop iterations makes hipcc land the loads in temporaries and copy them into the
// loop-carried registers behind s_waitcnt vmcnt(0) at the end of the issuing block, i.e. it exposes the whole HBM
// latency (measured in bwd_mv_tile3_kernel: 2,400 instead of 600 cycles per phase).  global_load_lds_dwordx4 has no
// register destination: each lane's 16 bytes at `src` go to LDS byte address lds_dst + 16 * lane.  hipcc does not
// count these loads: the issuing wave waits with lds_dma_wait_all() before it reads the zone (cdna_hip_programming.md
// section 8, "What hipcc does not do"), and a zone is private to its wave, so no barrier is involved.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_dst /* wave-uniform */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
// s_waitcnt vmcnt(0) as the builtin (0x0F70 = vmcnt 0, expcnt 7, lgkmcnt 15), which hipcc's own wait bookkeeping sees.
// Call it once after the kernel's ordinary global loads and before the first lds_dma16: otherwise hipcc, which does
// not count the DMA instructions, later waits for "its" outstanding loads with a counted vmcnt(N) that in fact drains
// the DMAs in flight (measured: 900-1,400 cycles per stage in bwd_mv_tile3_kernel).
__device__ __forceinline__ void lds_dma_wait_all() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Workgroup barrier for producer / consumer kernels whose waves exchange data through LDS only: waits for this wave's LDS
// accesses, not for its global stores and LDS-DMA loads (__syncthreads() waits vmcnt(0) too: the chain wave then stands at every
// tick until its own stores of the tick have been written, and a producer until the prefetch it issued for three ticks later
// has landed).  Nobody in such a kernel reads what another wave stored to global memory.
// INVARIANT the callers keep (noted at every lds_dma16 call site): an LDS-DMA landing zone is read only by the wave that
// issued the DMA, behind that wave's own lds_dma_wait_all() -- this barrier does not wait for DMAs in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace rk
