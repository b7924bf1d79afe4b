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

// log(u) for u in (0, 1) -- the uniform of Box-Muller -- in ~35 instructions where the library's log() takes about twice
// that: frexp, m in [sqrt(1/2), sqrt 2), s = f / (2 + f) with the reciprocal of linalg_small.hpp, fdlibm's degree-7 polynomial
// in s^2 (e_log.c: |error| < 2^-58 on that interval) and e ln 2 added by one FMA.  Absolute error ~1e-16 (1 + |log u|): the
// draws are compared with the NumPy mirror (oracle/counter_rng.py) at 1e-7 and below along whole sample paths.
__device__ __forceinline__ double bm_log01(double u) {
    int e = __builtin_amdgcn_frexp_exp(u);                          // u = m 2^e, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(u);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double rr = __builtin_amdgcn_rcp(2.0 + f);
    const double d = 2.0 + f;
    double r2 = fma(fma(-d, rr, 1.0), rr, rr);
    r2 = fma(fma(-d, r2, 1.0), r2, r2);
    const double sv = f * r2;
    const double z = sv * sv, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double hfsq = 0.5 * f * f;
    const double lm = f - (hfsq - sv * (hfsq + (t1 + t2)));
    return fma((double)e, 0.69314718055994530942, lm);
}
// sin(pi x), cos(pi x) for x in [0, 2]: quarter-turn reduction (exact: x - q / 2 is a difference of doubles with a common
// exponent range), fdlibm's kernel polynomials on |a| <= pi / 4 (k_sin.c, k_cos.c: < 2^-58), the quadrant by selects.
__device__ __forceinline__ void bm_sincospi02(double x, double& sn, double& cs) {
    const double qd = __builtin_rint(x + x);                        // 0 .. 4
    const double a = fma(qd, -0.5, x) * 3.14159265358979323846;
    const int q = (int)qd;
    const double z = a * a;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                   2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                    8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double s0 = fma(a * z, ps, a);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                   -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                    -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double c0 = fma(z * z, pc, fma(z, -0.5, 1.0));
    const bool sw = (q & 1) != 0;                                   // odd quarter turns swap sine and cosine
    const double s1 = sw ? c0 : s0, c1 = sw ? s0 : c0;
    sn = ((q + 0) & 2) ? -s1 : s1;                                  // q = 2, 3: sin < 0
    cs = ((q + 1) & 2) ? -c1 : c1;                                  // q = 1, 2: cos < 0
}

// two standard normals for one counter
__device__ __forceinline__ void normal_pair(uint64_t seed, uint32_t traj, uint32_t step, uint32_t block,
                                            uint32_t purpose, uint32_t chunk, double& z0, double& z1) {
    uint32_t r[4];
    philox4x32_10(traj, step, block | (purpose << 16), chunk, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    // Box-Muller with its own log and sincospi (round 4: the generator was 72 of the 446 cycles of C4's forward step and most
    // of its backward sampler's producer work; the library's log + sincospi are about twice the instructions of these two,
    // which need neither special cases nor large arguments); sqrt without the scaling branches of sqrt() (the argument is in
    // [1.1e-16, 74]).  The results differ from the NumPy mirror's at the 1e-15 level.
    const double t = -2.0 * bm_log01(u1);
    const double rs = __builtin_amdgcn_rsq(t);
    double g = t * rs, h = 0.5 * rs;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double rad = fma(fma(-g, g, t), h, g);
    double s, c;
    bm_sincospi02(2.0 * u2, s, c);
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
