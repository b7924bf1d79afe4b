// Counter-based normal stream for solve_sim / interrogate_chkrebtii draws (device side).
// Philox4x32-10 (Salmon et al., SC'11) + Box-Muller; the NumPy mirror used by the tests is
// oracle/counter_rng.py and both are pinned by the Random123 known-answer vectors.
//
// Stands in for the JAX threefry keys of src/rodeo/solve.py:147,179 and src/rodeo/interrogate.py:23,30-34
// (whose bit-stream is not reproducible without JAX).
//
//   key     = (seed & 0xffffffff, seed >> 32)
//   counter = (global trajectory index, step, block | purpose << 16, chunk)
//   one call -> 4 x u32 -> 2 uniforms in (0,1) -> 2 normals (cos, sin)
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace rk {

constexpr uint32_t PURPOSE_INTERROGATE = 0;
constexpr uint32_t PURPOSE_SMOOTH = 1;

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        if (r < 9) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    const uint64_t x = ((uint64_t)hi << 21) | ((uint64_t)lo >> 11);
    return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
}

// two standard normals for one counter
__device__ __forceinline__ void normal_pair(uint64_t seed, uint32_t traj, uint32_t step, uint32_t block,
                                            uint32_t purpose, uint32_t chunk, double& z0, double& z1) {
    uint32_t r[4];
    philox4x32_10(traj, step, block | (purpose << 16), chunk, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    // sqrt without the scaling branches of sqrt() (the argument is in [1.1e-16, 74]); sin / cos of 2 pi u2 through
    // sincospi: exact range reduction instead of the generic large-argument path of sincos() -- together about half
    // the instructions; the results differ from the NumPy mirror's cos(fl(2 pi u2)) by the rounding of the angle (1e-16)
    const double t = -2.0 * log(u1);
    const double rs = __builtin_amdgcn_rsq(t);
    double g = t * rs, h = 0.5 * rs;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double rad = fma(fma(-g, g, t), h, g);
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    z0 = rad * c;
    z1 = rad * s;
}

// P standard normals for (traj, step, block, purpose)
template <int P>
__device__ __forceinline__ void normals(uint64_t seed, uint32_t traj, uint32_t step, uint32_t block,
                                        uint32_t purpose, double (&z)[P]) {
#pragma unroll
    for (int c = 0; c < (P + 1) / 2; ++c) {
        double a, b;
        normal_pair(seed, traj, step, block, purpose, (uint32_t)c, a, b);
        z[2 * c] = a;
        if (2 * c + 1 < P) z[2 * c + 1] = b;
    }
}

}  // namespace rk
