// Fused solver kernels for kalman_type = "square-root" (src/rodeo/solve.py:140-141 selecting
// src/rodeo/kalmantv/square_root.py): same time loops as solve_small.hip, with every variance replaced by a lower
// square-root factor and add_sqrt (Householder QR, src/rodeo/utils.py:10-24) in place of the variance sums.
//   prior_pars = (Q, chol(R))  as in docs/examples/higher_order.md:106-125
//   outputs: mean (N+1, d, p, B) and the FACTORS (N+1, d, p, p, B), batch-minor, like the reference's return values.
// One lane per (trajectory, block), forward (the blocks of a trajectory in neighbouring lanes: the interrogation couples them)
// and backward.
// Reference quirks kept because they define the reference's numbers in this mode (oracle/interrogations.py has the
// same): interrogate_rodeo hands W L- W^T and interrogate_chkrebtii hands W L- (1 x p) to the update as the
// "factor" of var_meas (src/rodeo/interrogate.py:36-42, 110-113).  One quirk is NOT kept: solve_sim's draws use
// N(mean, L L^T) (the reference passes the factor where jax expects a covariance, solve.py:179).
// LAW OF THE SAMPLER (solve_sim in this mode): draws are x = mean + L z, i.e. N(mean, L L^T), L = the conditional factor of
// square_root.smooth_sim.  The reference passes that factor to jax.random.multivariate_normal(method="svd") in the
// COVARIANCE slot (src/rodeo/solve.py:179,182-186 with square_root.py:259), i.e. samples N(mean, L): not reproduced (not a law
// when L is not symmetric PSD; MIGRATION.md).  tests/test_gpu_solver.py::test_square_root_sim_law_is_L_Lt pins it by moments.
#include <cstdlib>
#include "common.hpp"
#include "kalman_small.hpp"
#include "philox.hpp"
#include "rhs.hpp"
#include "solve_args.hpp"
#include "sqrt_small.hpp"
#include "solve_sqrt_kernels.hpp"

namespace rk {

template <int P>
__device__ __forceinline__ void load_state(const SolveArgs& a, int n, int blk, int b, double (&mf)[P],
                                           double (&Lf)[P][P]) {
    const size_t B = (size_t)a.B;
    const double* mi = a.mean + ((size_t)n * a.D + blk) * P * B + b;
    const double* vi = a.var + ((size_t)n * a.D + blk) * P * P * B + b;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        mf[i] = mi[(size_t)i * B];
#pragma unroll
        for (int j = 0; j < P; ++j) Lf[i][j] = vi[((size_t)i * P + j) * B];
    }
}

// backward: SIM = false -> smooth_mv (square_root.py:209-219), SIM = true -> smooth_sim + draw (square_root.py:252-261)
template <int P, bool SIM>
__global__ void __launch_bounds__(64) bwd_sqrt_kernel(SolveArgs a) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.B * a.D) return;
    const int blk = l / a.B, b = l - blk * a.B;
    const size_t B = (size_t)a.B;
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
    double Q[P][P], LR[P][P];
    load_block_consts<P>(a, blk, b, Q, LR);
    double ms[P], Ls[P][P], xn[P];
    load_state<P>(a, a.N, blk, b, ms, Ls);                       // carry = filt[N]
    if (SIM) {
        double z[P];
        normals<P>(a.seed, traj, (uint32_t)a.N, (uint32_t)blk, PURPOSE_SMOOTH, z);
#pragma unroll
        for (int j = 0; j < P; ++j) z[j] = Ls[j][j] < 0.0 ? -z[j] : z[j];        // draws use the factor with diag >= 0
#pragma unroll
        for (int i = 0; i < P; ++i) {
            double s = ms[i];
#pragma unroll
            for (int k = 0; k < P; ++k) s = fma(Ls[i][k], z[k], s);
            xn[i] = s;
            a.x[(((size_t)a.N * a.D + blk) * P + i) * B + b] = s;
        }
    }
    for (int n = a.N - 1; n >= 1; --n) {
        double mf[P], Lf[P][P], mp[P], Lp[P][P], G[P][P], JL[P][P];
        load_state<P>(a, n, blk, b, mf, Lf);
        sqrt_predict<P>(Q, LR, mf, Lf, mp, Lp);                  // pred[n+1] re-evaluated from filt[n]
        sqrt_gain<P>(Q, Lf, Lp, G, JL);
        double dm[P], gm[P];
#pragma unroll
        for (int i = 0; i < P; ++i) dm[i] = (SIM ? xn[i] : ms[i]) - mp[i];
        mv<P, P>(G, dm, gm);
        if (!SIM) {
            double both[P][2 * P], GA[P][2 * P];
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int j = 0; j < P; ++j) { both[i][j] = Ls[i][j]; both[i][P + j] = LR[i][j]; }
            mm<P, P, 2 * P>(G, both, GA);
            add_sqrt<P, 2 * P, P>(GA, JL, Ls);                   // square_root.py:217-218
            double* mo = a.mean + ((size_t)n * a.D + blk) * P * B + b;
            double* vo = a.var + ((size_t)n * a.D + blk) * P * P * B + b;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                ms[i] = mf[i] + gm[i];
                mo[(size_t)i * B] = ms[i];
#pragma unroll
                for (int j = 0; j < P; ++j) vo[((size_t)i * P + j) * B] = Ls[i][j];
            }
        } else {
            double GR[P][P], Lsim[P][P], z[P];
            mm<P, P, P>(G, LR, GR);
            add_sqrt<P, P, P>(GR, JL, Lsim);                     // square_root.py:259-260
            normals<P>(a.seed, traj, (uint32_t)n, (uint32_t)blk, PURPOSE_SMOOTH, z);
#pragma unroll
            for (int j = 0; j < P; ++j) z[j] = Lsim[j][j] < 0.0 ? -z[j] : z[j];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                double s = mf[i] + gm[i];
#pragma unroll
                for (int k = 0; k < P; ++k) s = fma(Lsim[i][k], z[k], s);
                xn[i] = s;
            }
#pragma unroll
            for (int i = 0; i < P; ++i) a.x[(((size_t)n * a.D + blk) * P + i) * B + b] = xn[i];
        }
    }
    if (SIM) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            a.x[((size_t)blk * P + i) * B + b] = a.mean[((size_t)blk * P + i) * B + b];   // x[0] = ode_init
    }
}

// ---- solve_mv backward pass in two kernels (the blocked tile path's split, solve_tilen.hip) -------------------------------------
// Everything of a smoothing step that does not depend on the carry -- the re-evaluated prediction (a Householder QR), the gain
// G (two triangular solves), J L_f = (I - G Q) L_f and G R^{1/2} (square_root.py:170-175, 214-217) -- is evaluated for ALL time
// steps at once, one lane per (step, block, trajectory), into a batch-minor workspace record [G | J L_f | G R^{1/2} | mu-] of
// 3 p^2 + p doubles; the sequential part that remains per step is G L_s, one add_sqrt (the QR of the 3 p x p stack) and the
// mean: ~250 instead of ~850 instructions on the one-lane-per-block chain whose length is the run time at small batches
// (7.3 -> 2.x ms at n_deriv = 3 on the headline shape).  Same arithmetic in the same order as bwd_sqrt_kernel<P, false>, which
// stays as the path without a workspace.
__host__ __device__ inline size_t sqrt_rec_doubles(int p) { return 3 * (size_t)p * p + p; }

template <int P>
__global__ void __launch_bounds__(64) sqrt_gain_kernel(SolveArgs a, double* __restrict__ ws) {
    const size_t B = (size_t)a.B, per_step = (size_t)a.D * B;
    const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= (size_t)(a.N - 1) * per_step) return;
    const int n = 1 + (int)(l / per_step), blk = (int)((l % per_step) / B), b = (int)(l % B);
    double Q[P][P], LR[P][P], mf[P], Lf[P][P], mp[P], Lp[P][P], G[P][P], JL[P][P], GR[P][P];
    load_block_consts<P>(a, blk, b, Q, LR);
    load_state<P>(a, n, blk, b, mf, Lf);
    sqrt_predict<P>(Q, LR, mf, Lf, mp, Lp);                      // pred[n+1] re-evaluated from filt[n]
    sqrt_gain<P>(Q, Lf, Lp, G, JL);
    mm<P, P, P>(G, LR, GR);
    double* rec = ws + ((size_t)n * a.D + blk) * sqrt_rec_doubles(P) * B + b;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        rec[(size_t)(3 * P * P + i) * B] = mp[i];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            rec[(size_t)(i * P + j) * B] = G[i][j];
            rec[(size_t)(P * P + i * P + j) * B] = JL[i][j];
            rec[(size_t)(2 * P * P + i * P + j) * B] = GR[i][j];
        }
    }
}

template <int P>
__global__ void __launch_bounds__(64) bwd_sqrt_chain_kernel(SolveArgs a, const double* __restrict__ ws) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.B * a.D) return;
    const int blk = l / a.B, b = l - blk * a.B;
    const size_t B = (size_t)a.B;
    double ms[P], Ls[P][P];
    load_state<P>(a, a.N, blk, b, ms, Ls);                       // carry = filt[N]
    struct Rec { double G[P][P], JL[P][P], GR[P][P], mp[P], mf[P]; };
    auto load = [&](int n, Rec& q) {
        const double* rec = ws + ((size_t)n * a.D + blk) * sqrt_rec_doubles(P) * B + b;
        const double* mi = a.mean + ((size_t)n * a.D + blk) * P * B + b;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            q.mp[i] = rec[(size_t)(3 * P * P + i) * B];
            q.mf[i] = mi[(size_t)i * B];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                q.G[i][j] = rec[(size_t)(i * P + j) * B];
                q.JL[i][j] = rec[(size_t)(P * P + i * P + j) * B];
                q.GR[i][j] = rec[(size_t)(2 * P * P + i * P + j) * B];
            }
        }
    };
    // RD records in flight (static register names in an unrolled ring): a record comes from HBM / the memory-side cache, two
    // or three chain steps away at p = 3
    constexpr int RD = P <= 3 ? 4 : (P == 4 ? 3 : 2);
    Rec q[RD];
#pragma unroll
    for (int s = 0; s < RD; ++s) load(a.N - 1 - s >= 1 ? a.N - 1 - s : 1, q[s]);
    auto step = [&](int n, Rec& r) {
        double dm[P], gm[P], GA[P][2 * P];
#pragma unroll
        for (int i = 0; i < P; ++i) dm[i] = ms[i] - r.mp[i];
        mv<P, P>(r.G, dm, gm);
        // G [L_s | R^{1/2}]: the left half in the order of mm<P, P, 2 P>, the right half from the record
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int j = 0; j < P; ++j) {
                double acc = r.G[i][0] * Ls[0][j];
#pragma unroll
                for (int k = 1; k < P; ++k) acc = fma(r.G[i][k], Ls[k][j], acc);
                GA[i][j] = acc;
                GA[i][P + j] = r.GR[i][j];
            }
        add_sqrt<P, 2 * P, P>(GA, r.JL, Ls);                     // square_root.py:217-218
        double* mo = a.mean + ((size_t)n * a.D + blk) * P * B + b;
        double* vo = a.var + ((size_t)n * a.D + blk) * P * P * B + b;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            ms[i] = r.mf[i] + gm[i];
            mo[(size_t)i * B] = ms[i];
#pragma unroll
            for (int j = 0; j < P; ++j) vo[((size_t)i * P + j) * B] = Ls[i][j];
        }
        const int nn = n - RD;                                   // refill the slot (indices below 1: a harmless reload)
        load(nn >= 1 ? nn : 1, r);
    };
    int n = a.N - 1;
    while (n >= RD) {
#pragma unroll
        for (int s = 0; s < RD; ++s) step(n - s, q[s]);
        n -= RD;
    }
#pragma unroll
    for (int s = 0; s < RD; ++s)
        if (s < n) step(n - s, q[s]);                            // (uniform)
}

// ---- solve_sim the same way: of a sampling step (square_root.py:252-261, solve.py:179) only G (x_{n+1} - mu-) depends on the
// carry; the gain, the conditional factor L_sim = add_sqrt(G R^{1/2}, J L_f) and the draw's own part mu_f + L_sim z are
// evaluated for all steps at once into a record [G | mu- | mu_f + L_sim z] of p^2 + 2 p doubles, and the chain is one
// matrix-vector product per step.  Same arithmetic and the same normals as bwd_sqrt_kernel<P, true>.
__host__ __device__ inline size_t sqrt_sim_rec_doubles(int p) { return (size_t)p * p + 2 * p; }

template <int P>
__global__ void __launch_bounds__(64) sqrt_sim_gain_kernel(SolveArgs a, double* __restrict__ ws) {
    const size_t B = (size_t)a.B, per_step = (size_t)a.D * B;
    const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= (size_t)(a.N - 1) * per_step) return;
    const int n = 1 + (int)(l / per_step), blk = (int)((l % per_step) / B), b = (int)(l % B);
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
    double Q[P][P], LR[P][P], mf[P], Lf[P][P], mp[P], Lp[P][P], G[P][P], JL[P][P], GR[P][P], Lsim[P][P], z[P];
    load_block_consts<P>(a, blk, b, Q, LR);
    load_state<P>(a, n, blk, b, mf, Lf);
    sqrt_predict<P>(Q, LR, mf, Lf, mp, Lp);
    sqrt_gain<P>(Q, Lf, Lp, G, JL);
    mm<P, P, P>(G, LR, GR);
    add_sqrt<P, P, P>(GR, JL, Lsim);                             // square_root.py:259-260
    normals<P>(a.seed, traj, (uint32_t)n, (uint32_t)blk, PURPOSE_SMOOTH, z);
#pragma unroll
    for (int j = 0; j < P; ++j) z[j] = Lsim[j][j] < 0.0 ? -z[j] : z[j];
    double* rec = ws + ((size_t)n * a.D + blk) * sqrt_sim_rec_doubles(P) * B + b;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        double w = 0.0;                                          // (the order of bwd_sqrt_kernel: (mu_f + G dx) + L z, L z summed first)
#pragma unroll
        for (int k = 0; k < P; ++k) w = fma(Lsim[i][k], z[k], w);
        rec[(size_t)(P * P + i) * B] = mp[i];
        rec[(size_t)(P * P + P + i) * B] = w;
#pragma unroll
        for (int j = 0; j < P; ++j) rec[(size_t)(i * P + j) * B] = G[i][j];
    }
}

template <int P>
__global__ void __launch_bounds__(64) bwd_sqrt_sim_chain_kernel(SolveArgs a, const double* __restrict__ ws) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.B * a.D) return;
    const int blk = l / a.B, b = l - blk * a.B;
    const size_t B = (size_t)a.B;
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
    double xn[P];
    {
        double ms[P], Ls[P][P], z[P];
        load_state<P>(a, a.N, blk, b, ms, Ls);                   // x_N ~ N(filt[N])
        normals<P>(a.seed, traj, (uint32_t)a.N, (uint32_t)blk, PURPOSE_SMOOTH, z);
#pragma unroll
        for (int j = 0; j < P; ++j) z[j] = Ls[j][j] < 0.0 ? -z[j] : z[j];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            double sacc = ms[i];
#pragma unroll
            for (int k = 0; k < P; ++k) sacc = fma(Ls[i][k], z[k], sacc);
            xn[i] = sacc;
            a.x[(((size_t)a.N * a.D + blk) * P + i) * B + b] = sacc;
        }
    }
    struct Rec { double G[P][P], mp[P], w[P], mf[P]; };
    auto load = [&](int n, Rec& q) {
        const double* rec = ws + ((size_t)n * a.D + blk) * sqrt_sim_rec_doubles(P) * B + b;
        const double* mi = a.mean + ((size_t)n * a.D + blk) * P * B + b;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            q.mp[i] = rec[(size_t)(P * P + i) * B];
            q.w[i] = rec[(size_t)(P * P + P + i) * B];
            q.mf[i] = mi[(size_t)i * B];
#pragma unroll
            for (int j = 0; j < P; ++j) q.G[i][j] = rec[(size_t)(i * P + j) * B];
        }
    };
    constexpr int RD = P <= 4 ? 4 : 2;
    Rec q[RD];
#pragma unroll
    for (int s = 0; s < RD; ++s) load(a.N - 1 - s >= 1 ? a.N - 1 - s : 1, q[s]);
    auto step = [&](int n, Rec& r) {
        double dm[P], gm[P];
#pragma unroll
        for (int i = 0; i < P; ++i) dm[i] = xn[i] - r.mp[i];
        mv<P, P>(r.G, dm, gm);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            // bwd_sqrt_kernel: s = mu_f + gm, then fma over L z term by term; here L z arrives summed: rounding-level difference
            xn[i] = (r.mf[i] + gm[i]) + r.w[i];
            a.x[(((size_t)n * a.D + blk) * P + i) * B + b] = xn[i];
        }
        const int nn = n - RD;
        load(nn >= 1 ? nn : 1, r);
    };
    int n = a.N - 1;
    while (n >= RD) {
#pragma unroll
        for (int s = 0; s < RD; ++s) step(n - s, q[s]);
        n -= RD;
    }
#pragma unroll
    for (int s = 0; s < RD; ++s)
        if (s < n) step(n - s, q[s]);
#pragma unroll
    for (int i = 0; i < P; ++i)
        a.x[((size_t)blk * P + i) * B + b] = a.mean[((size_t)blk * P + i) * B + b];   // x[0] = ode_init
}

size_t sqrt_ws_doubles(const rk_solve_cfg* c, int mode) {
    if (mode == RK_MODE_FILTER || c->n_steps < 2) return 0;
    const size_t rec = mode == RK_MODE_MV ? sqrt_rec_doubles(c->n_bstate) : sqrt_sim_rec_doubles(c->n_bstate);
    return (size_t)c->n_steps * c->n_block * rec * (size_t)c->n_traj;
}

// RK_FLAG_STORE_PRED (_solve_filter's "state_pred" in square-root form): the predicted means and FACTORS are a function of the
// filtered ones and the prior alone, pred[n + 1] = sqrt_predict(filt[n]) (square_root.py:56-57), so they are re-evaluated for
// all time steps at once behind the forward pass -- the same arithmetic as inside it, one lane per (time, block, trajectory);
// a store inside the time loop cost the filter 8 % whether or not it was taken.  Index 0 = (ode_init, 0) (solve.py:114-121).
template <int P>
__global__ void __launch_bounds__(64) sqrt_pred_kernel(SolveArgs a) {
    const size_t B = (size_t)a.B, per_step = (size_t)a.D * B;
    const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= (size_t)(a.N + 1) * per_step) return;
    const int n = (int)(l / per_step), blk = (int)((l - (size_t)n * per_step) / B), b = (int)(l % B);
    double mup[P], Lp[P][P];
    if (n == 0) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            mup[i] = a.mean[((size_t)blk * P + i) * B + b];
#pragma unroll
            for (int j = 0; j < P; ++j) Lp[i][j] = 0.0;
        }
    } else {
        double Q[P][P], LR[P][P], mf[P], Lf[P][P];
        load_block_consts<P>(a, blk, b, Q, LR);
        load_state<P>(a, n - 1, blk, b, mf, Lf);
        sqrt_predict<P>(Q, LR, mf, Lf, mup, Lp);
    }
    double* mpo = a.mean_pred + ((size_t)n * a.D + blk) * P * B + b;
    double* vpo = a.var_pred + ((size_t)n * a.D + blk) * P * P * B + b;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        mpo[(size_t)i * B] = mup[i];
#pragma unroll
        for (int j = 0; j < P; ++j) vpo[((size_t)i * P + j) * B] = Lp[i][j];
    }
}

template <class RHS, int P>
static int launch_fwd_sqrt_p(rk_handle h, const SolveArgs& a, int itg) {
    const dim3 grid(div_up(a.B, 64 / RHS::D)), block(64);      // 64 / D trajectories per wave (solve_sqrt_kernels.hpp)
    LaunchTimer t(h, "fwd_sqrt_kernel");
    switch (itg) {
        case RK_INTERROGATE_RODEO: hipLaunchKernelGGL((fwd_sqrt_kernel<RHS, P, RK_INTERROGATE_RODEO>), grid, block, 0, h->stream, a); break;
        case RK_INTERROGATE_SCHOBER: hipLaunchKernelGGL((fwd_sqrt_kernel<RHS, P, RK_INTERROGATE_SCHOBER>), grid, block, 0, h->stream, a); break;
        case RK_INTERROGATE_KRAMER: hipLaunchKernelGGL((fwd_sqrt_kernel<RHS, P, RK_INTERROGATE_KRAMER>), grid, block, 0, h->stream, a); break;
        case RK_INTERROGATE_CHKREBTII: hipLaunchKernelGGL((fwd_sqrt_kernel<RHS, P, RK_INTERROGATE_CHKREBTII>), grid, block, 0, h->stream, a); break;
        default: set_error("unknown interrogate id %d", itg); return RK_ERR_UNSUPPORTED;
    }
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

template <class RHS>
static int launch_fwd_sqrt(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a) {
    RK_REQUIRE(c->n_block == RHS::D && c->n_bmeas == 1, RK_ERR_UNSUPPORTED,
               "rhs %d needs n_block=%d, n_bmeas=1 (got %d, %d)", c->rhs_id, RHS::D, c->n_block, c->n_bmeas);
    switch (c->n_bstate) {
        case 2: return launch_fwd_sqrt_p<RHS, 2>(h, a, c->interrogate);
        case 3: return launch_fwd_sqrt_p<RHS, 3>(h, a, c->interrogate);
        case 4: return launch_fwd_sqrt_p<RHS, 4>(h, a, c->interrogate);
        case 5: return launch_fwd_sqrt_p<RHS, 5>(h, a, c->interrogate);
        case 6: return launch_fwd_sqrt_p<RHS, 6>(h, a, c->interrogate);
        case 7: return launch_fwd_sqrt_p<RHS, 7>(h, a, c->interrogate);
        case 8: return launch_fwd_sqrt_p<RHS, 8>(h, a, c->interrogate);
    }
    set_error("square-root solver supports n_bstate in [2, 8], got %d", c->n_bstate);
    return RK_ERR_UNSUPPORTED;
}

bool is_user_rhs(int rhs_id);
int user_forward_sqrt(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a);

int sqrt_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a_, int mode, double* ws, size_t ws_bytes) {
    RK_REQUIRE(c->n_bstate >= 2 && c->n_bstate <= 8, RK_ERR_UNSUPPORTED, "square-root solver supports n_bstate in [2, 8], got %d",
               c->n_bstate);
    const SolveArgs& a = a_;
    int rc;
    if (is_user_rhs(c->rhs_id)) rc = user_forward_sqrt(h, c, a);      // hiprtc build of fwd_sqrt_kernel (rhs_jit.hip)
    else switch (c->rhs_id) {
        case RK_RHS_FITZHUGH_NAGUMO: rc = launch_fwd_sqrt<FitzHughNagumo>(h, c, a); break;
        case RK_RHS_LORENZ63: rc = launch_fwd_sqrt<Lorenz63>(h, c, a); break;
        case RK_RHS_HIGHER_ORDER: rc = launch_fwd_sqrt<HigherOrder>(h, c, a); break;
        default: set_error("unknown rhs_id %d for the square-root solver", c->rhs_id); return RK_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    if (c->flags & RK_FLAG_STORE_PRED) {
        const size_t lanes = (size_t)(a.N + 1) * a.D * (size_t)a.B;
        RK_REQUIRE(lanes < 0x7fffffffull * 64, RK_ERR_UNSUPPORTED, "square-root solver: too many (time, block, trajectory) items for one launch");
        const dim3 pgrid((unsigned)((lanes + 63) / 64)), pblock(64);
        LaunchTimer t(h, "sqrt_pred_kernel");
#define RK_SQP(P_) case P_: hipLaunchKernelGGL((sqrt_pred_kernel<P_>), pgrid, pblock, 0, h->stream, a); break;
        switch (c->n_bstate) { RK_SQP(2) RK_SQP(3) RK_SQP(4) RK_SQP(5) RK_SQP(6) RK_SQP(7) RK_SQP(8) }
#undef RK_SQP
        t.stop();
        RK_HIP(hipGetLastError());
    }
    if (mode == RK_MODE_FILTER) return RK_OK;
    const dim3 grid(div_up(a.B * a.D, 64)), block(64);
    const size_t need = sqrt_ws_doubles(c, mode) * sizeof(double);
    const char* const bsel = getenv("RK_SQRT_BWD");                   // "single": the one-kernel form (the parity test's other leg)
    if (mode == RK_MODE_MV && need && ws && ws_bytes >= need && !(bsel && bsel[0] == 's')) {       // (no workspace: the one-kernel form below)
        const size_t lanes = (size_t)(a.N - 1) * a.D * (size_t)a.B;
        RK_REQUIRE(lanes < 0x7fffffffull * 64, RK_ERR_UNSUPPORTED, "square-root solver: too many (time, block, trajectory) items for one launch");
        const dim3 ggrid((unsigned)((lanes + 63) / 64));
        {
            LaunchTimer t(h, "sqrt_gain_kernel");
#define RK_SQG(P_) case P_: hipLaunchKernelGGL((sqrt_gain_kernel<P_>), ggrid, block, 0, h->stream, a, ws); break;
            switch (c->n_bstate) { RK_SQG(2) RK_SQG(3) RK_SQG(4) RK_SQG(5) RK_SQG(6) RK_SQG(7) RK_SQG(8) }
#undef RK_SQG
            t.stop();
            RK_HIP(hipGetLastError());
        }
        LaunchTimer t(h, "bwd_sqrt_chain_kernel");
#define RK_SQC(P_) case P_: hipLaunchKernelGGL((bwd_sqrt_chain_kernel<P_>), grid, block, 0, h->stream, a, ws); break;
        switch (c->n_bstate) { RK_SQC(2) RK_SQC(3) RK_SQC(4) RK_SQC(5) RK_SQC(6) RK_SQC(7) RK_SQC(8) }
#undef RK_SQC
        t.stop();
        RK_HIP(hipGetLastError());
        return RK_OK;
    }
    if (mode == RK_MODE_SIM && need && ws && ws_bytes >= need && !(bsel && bsel[0] == 's')) {
        const size_t lanes = (size_t)(a.N - 1) * a.D * (size_t)a.B;
        RK_REQUIRE(lanes < 0x7fffffffull * 64, RK_ERR_UNSUPPORTED, "square-root solver: too many (time, block, trajectory) items for one launch");
        const dim3 ggrid((unsigned)((lanes + 63) / 64));
        {
            LaunchTimer t(h, "sqrt_sim_gain_kernel");
#define RK_SQG(P_) case P_: hipLaunchKernelGGL((sqrt_sim_gain_kernel<P_>), ggrid, block, 0, h->stream, a, ws); break;
            switch (c->n_bstate) { RK_SQG(2) RK_SQG(3) RK_SQG(4) RK_SQG(5) RK_SQG(6) RK_SQG(7) RK_SQG(8) }
#undef RK_SQG
            t.stop();
            RK_HIP(hipGetLastError());
        }
        LaunchTimer t(h, "bwd_sqrt_sim_chain_kernel");
#define RK_SQC(P_) case P_: hipLaunchKernelGGL((bwd_sqrt_sim_chain_kernel<P_>), grid, block, 0, h->stream, a, ws); break;
        switch (c->n_bstate) { RK_SQC(2) RK_SQC(3) RK_SQC(4) RK_SQC(5) RK_SQC(6) RK_SQC(7) RK_SQC(8) }
#undef RK_SQC
        t.stop();
        RK_HIP(hipGetLastError());
        return RK_OK;
    }
    LaunchTimer t(h, mode == RK_MODE_SIM ? "bwd_sqrt_sim_kernel" : "bwd_sqrt_mv_kernel");
#define RK_SQ(P_)                                                                                        \
    case P_:                                                                                             \
        if (mode == RK_MODE_SIM) hipLaunchKernelGGL((bwd_sqrt_kernel<P_, true>), grid, block, 0, h->stream, a);  \
        else hipLaunchKernelGGL((bwd_sqrt_kernel<P_, false>), grid, block, 0, h->stream, a);             \
        break;
    switch (c->n_bstate) { RK_SQ(2) RK_SQ(3) RK_SQ(4) RK_SQ(5) RK_SQ(6) RK_SQ(7) RK_SQ(8) }
#undef RK_SQ
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

}  // namespace rk
