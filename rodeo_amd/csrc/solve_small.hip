// Small-block solver kernels (n_bmeas = 1, n_bstate <= 6): the fused time loops of
//   src/rodeo/solve.py:31-122   _solve_filter   -> fwd_kernel
//   src/rodeo/solve.py:257-301  solve_mv        -> bwd_mv_kernel   (in place over the filtered moments)
//   src/rodeo/solve.py:162-204  solve_sim       -> bwd_sim_kernel
// The whole N-step recursion of a trajectory runs inside ONE kernel launch; the state lives in VGPRs and only the
// filtered / smoothed moments touch HBM, in the batch-minor layout of include/rodeo_kalman.h (512 B per wave per
// element row).  Predicted moments are re-evaluated from the filtered ones in the backward pass instead of being
// stored (they are a pure function of filt[n], Q, R: standard.py:57-59).
#include "common.hpp"
#include "kalman_small.hpp"
#include "philox.hpp"
#include "rhs.hpp"
#include "solve_args.hpp"
#include "solve_small_kernels.hpp"

namespace rk {

// ---- backward passes: one lane per (block, trajectory) -----------------------------------------------------------
template <int P>
__device__ __forceinline__ void load_filt(const SolveArgs& a, int n, int blk, int b, double (&mf)[P],
                                          double (&Sf)[P][P]) {
    const size_t B = (size_t)a.B;
    const double* mi = a.mean + ((size_t)n * a.D + blk) * P * B + b;
    const double* vi = a.var + ((size_t)n * a.D + blk) * P * P * B + b;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        mf[i] = mi[(size_t)i * B];
#pragma unroll
        for (int j = 0; j < P; ++j) Sf[i][j] = vi[((size_t)i * P + j) * B];
    }
}

// the same from the blocked tile path's records [Sigma row-major (p^2) | mu (p)] per (time, trajectory, block)
// (RK_LAYOUT_TILE4 / RK_LAYOUT_TILEP, solve_tilen_kernels.hpp)
template <int P>
__device__ __forceinline__ void load_filt_tiles(const SolveArgs& a, const double* tiles, int n, int blk, int b, double (&mf)[P],
                                                double (&Sf)[P][P]) {
    const double* rec = tiles + ((size_t)n * a.B * a.D + (size_t)b * a.D + blk) * (P * P + P);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        mf[i] = rec[P * P + i];
#pragma unroll
        for (int j = 0; j < P; ++j) Sf[i][j] = rec[i * P + j];
    }
}

template <int P>
__global__ void __launch_bounds__(64) bwd_mv_kernel(SolveArgs a) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.B * a.D) return;
    const int blk = l / a.B, b = l - blk * a.B;
    const size_t B = (size_t)a.B;
    double Q[P][P], R[P][P];
    load_block_consts<P>(a, blk, b, Q, R);

    double ms[P], Ss[P][P];                    // carry: smoothed moments at n+1 (solve.py:279-282)
    load_filt<P>(a, a.N, blk, b, ms, Ss);
    double mf[P], Sf[P][P];
    if (a.N >= 2) load_filt<P>(a, a.N - 1, blk, b, mf, Sf);
    for (int n = a.N - 1; n >= 1; --n) {
        // software prefetch of the next (earlier) filtered state while this step computes
        double mfn[P], Sfn[P][P];
        if (n >= 2) load_filt<P>(a, n - 1, blk, b, mfn, Sfn);
        double mp[P], Sp[P][P], T[P][P], G[P][P];
        predict_block<P>(Q, R, mf, Sf, mp, Sp);          // pred[n+1] re-evaluated from filt[n]
        smooth_gain<P>(Q, Sf, Sp, T, G);
        smooth_mv_block<P>(G, mf, Sf, mp, Sp, ms, Ss);
        double* mo = a.mean + ((size_t)n * a.D + blk) * P * B + b;
        double* vo = a.var + ((size_t)n * a.D + blk) * P * P * B + b;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            mo[(size_t)i * B] = ms[i];
#pragma unroll
            for (int j = 0; j < P; ++j) vo[((size_t)i * P + j) * B] = Ss[i][j];
        }
        if (n >= 2) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                mf[i] = mfn[i];
#pragma unroll
                for (int j = 0; j < P; ++j) Sf[i][j] = Sfn[i][j];
            }
        }
    }
    // time 0 keeps (ode_init, 0) written by the forward pass (solve.py:295-301)
}

template <int P>
__global__ void __launch_bounds__(64) bwd_sim_kernel(SolveArgs a) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.B * a.D) return;
    const int blk = l / a.B, b = l - blk * a.B;
    const size_t B = (size_t)a.B;
    const uint32_t traj = (uint32_t)(a.traj_offset + (uint64_t)b);
    double Q[P][P], R[P][P];
    load_block_consts<P>(a, blk, b, Q, R);

    auto store_x = [&](int n, const double (&x)[P]) {
        double* xo = a.x + ((size_t)n * a.D + blk) * P * B + b;
#pragma unroll
        for (int i = 0; i < P; ++i) xo[(size_t)i * B] = x[i];
    };

    double mf[P], Sf[P][P], xn[P], z[P];
    // terminal draw x_N ~ N(filt[N]) (solve.py:182-186)
    load_filt<P>(a, a.N, blk, b, mf, Sf);
    normals<P>(a.seed, traj, (uint32_t)a.N, (uint32_t)blk, PURPOSE_SMOOTH, z);
    mvn_draw<P>(mf, Sf, z, xn);
    store_x(a.N, xn);
    if (a.N >= 2) load_filt<P>(a, a.N - 1, blk, b, mf, Sf);
    for (int n = a.N - 1; n >= 1; --n) {
        double mfn[P], Sfn[P][P];
        if (n >= 2) load_filt<P>(a, n - 1, blk, b, mfn, Sfn);
        double mp[P], Sp[P][P], T[P][P], G[P][P], msim[P], Ssim[P][P];
        predict_block<P>(Q, R, mf, Sf, mp, Sp);
        smooth_gain<P>(Q, Sf, Sp, T, G);
        smooth_sim_block<P>(G, T, mf, Sf, mp, xn, msim, Ssim);
        normals<P>(a.seed, traj, (uint32_t)n, (uint32_t)blk, PURPOSE_SMOOTH, z);
        mvn_draw<P>(msim, Ssim, z, xn);
        store_x(n, xn);
        if (n >= 2) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                mf[i] = mfn[i];
#pragma unroll
                for (int j = 0; j < P; ++j) Sf[i][j] = Sfn[i][j];
            }
        }
    }
    // x[0] = ode_init exactly (solve.py:196-204)
    double x0[P];
#pragma unroll
    for (int i = 0; i < P; ++i) x0[i] = a.mean[((size_t)blk * P + i) * B + b];
    store_x(0, x0);
}

// ---- Gaussian observation log-posterior reduction (docs/examples/parameter.md:188-210) ---------------------------
// One wave per trajectory: the n_obs x n_block observation terms are spread over the 64 lanes (a fixed stride-64 assignment
// and a fixed xor tree, so the value does not depend on the launch), where one lane per trajectory walked its 82 strided
// loads one after the other on 16 waves in all (C4: 36 us of a 306 us evaluation; now 6).
__global__ void __launch_bounds__(64) gauss_logpost_kernel(int B, int n_steps, int D, int P, int tile, int tile_off, const double* x,
                                                           const double* obs, const int32_t* obs_ind, int n_obs, double noise_sd,
                                                           const double* upars, int n_prior, double prior_sd, double* out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const double LOG_SQRT_2PI = 0.91893853320467274178;
    const double lsd = log(noise_sd);
    double acc = 0.0;
    for (int e = lane; e < n_obs * D; e += 64) {
        const int k = e / D, blk = e - k * D;
        int ni = obs_ind[k];
        ni = ni < 0 ? 0 : (ni > n_steps ? n_steps : ni);     // never index outside the (N+1)-long path
        const size_t n = (size_t)ni;
        const double xv = tile ? x[((n * B + b) * D + blk) * tile + tile_off]      // mu_0 inside the tile
                               : x[((n * D + blk) * P + 0) * (size_t)B + b];
        const double zz = (obs[e] - xv) / noise_sd;
        acc += -0.5 * zz * zz - lsd - LOG_SQRT_2PI;           // scipy.stats.norm.logpdf
    }
    if (upars) {
        const double lps = log(prior_sd);
        for (int k = lane; k < n_prior; k += 64) {
            const double zz = upars[(size_t)k * B + b] / prior_sd;
            acc += -0.5 * zz * zz - lps - LOG_SQRT_2PI;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) out[b] = acc;
}

// ---- Fenrir backward pass (src/rodeo/inference/fenrir.py:86-259), scalar observations per block --------------------
// One lane per (block, trajectory): the backward Markov chain X_n = A_n X_{n+1} + b_n + C_n^{1/2} eps
// (smooth_cond, standard.py:366-370) is run as a Kalman filter backwards in time over the stored forward moments
// (batch-minor filt and pred), conditioning on the observations y_i = D_i X_{n(i)} + N(0, Omega_i) at their grid
// indices; every observation contributes log N(y_i; D m, D S D^T + Omega) with utils.py:60-78's rule that a variance
// with |w| <= 1e-8 contributes nothing (jnp.isclose(w, 0, rtol=1e-300)).  Block values are summed per trajectory.
// With STORE (fenrir.py:236-258, for fenrir's solve_mv) the backward filter's predicted and updated moments of every
// time and the Markov weights A_n are kept in `states`: per (time n, block) an item of 3 P^2 + 2 P doubles
// [m_pred (P), S_pred (P^2), m_filt (P), S_filt (P^2), A (P^2)], batch-minor.
// MO = n_bobs (observations per block, fenrir.py:106-122): obs (n_obs, d, MO), obs_w (n_obs, d, MO, P), obs_v (n_obs, d, MO, MO)
// TILES: the forward pass ran on the blocked MFMA tiles (n_bstate 4 .. 8): `tiles` holds its records and the predicted
// moments are re-evaluated from the filtered ones (standard.py:57-59, what the forward pass computed) instead of read.
template <int P, bool STORE, int MO, bool TILES = false>
__global__ void __launch_bounds__(64) fenrir_bwd_kernel(SolveArgs a, const double* __restrict__ obs,
                                                        const double* __restrict__ obs_w, const double* __restrict__ obs_v,
                                                        const int32_t* __restrict__ obs_ind, int n_obs,
                                                        double* __restrict__ logdens, double* __restrict__ states,
                                                        const double* __restrict__ tiles = nullptr) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.B * a.D) return;
    const int blk = l / a.B, b = l - blk * a.B;
    const size_t B = (size_t)a.B;
    const double LOG_2PI = 1.83787706640934548356;
    double Q[P][P], R[P][P];
    load_block_consts<P>(a, blk, b, Q, R);
    double bm[P], bS[P][P];
    if constexpr (TILES) load_filt_tiles<P>(a, tiles, a.N, blk, b, bm, bS);
    else load_filt<P>(a, a.N, blk, b, bm, bS);                               // terminal point (fenrir.py:186-188)
    double acc = 0.0;
    int i = n_obs - 1;
    // forecast (standard.py:333-335) + log-density + update (standard.py:93-102) with observation i
    auto observe = [&](double (&m)[P], double (&S)[P][P]) {
        if constexpr (MO == 1) {
            double D[P], SD[P];
#pragma unroll
            for (int k = 0; k < P; ++k) D[k] = obs_w[((size_t)i * a.D + blk) * P + k];
            const double y = obs[(size_t)i * a.D + blk], Om = obs_v[(size_t)i * a.D + blk];
            const double mean_fore = dot<P>(D, m);
#pragma unroll
            for (int r = 0; r < P; ++r) SD[r] = dot<P>(S[r], D);                // Sigma D^T
            double DS[P];
#pragma unroll
            for (int c = 0; c < P; ++c) {
                double t = D[0] * S[0][c];
#pragma unroll
                for (int k = 1; k < P; ++k) t = fma(D[k], S[k][c], t);
                DS[c] = t;                                                      // D Sigma
            }
            const double w = dot<P>(DS, D) + Om;                                // var_fore
            const double z = y - mean_fore;
            if (fabs(w) > 1e-8) acc += -0.5 * (z * z / w + log(w)) - 0.5 * LOG_2PI;
#pragma unroll
            for (int r = 0; r < P; ++r) {
                const double K = SD[r] / w;                                     // solve_var with a 1 x 1 system
                m[r] = fma(K, z, m[r]);
#pragma unroll
                for (int c = 0; c < P; ++c) S[r][c] = fma(-K, DS[c], S[r][c]);
            }
        } else {
            // vector observation of this block: y (MO), D (MO x P), Omega (MO x MO)
            double D[MO][P], y[MO], Wf[MO][MO], DS[MO][P], X[MO][P], z[MO];
            const size_t ib = (size_t)i * a.D + blk;
#pragma unroll
            for (int j = 0; j < MO; ++j) {
                y[j] = obs[ib * MO + j];
#pragma unroll
                for (int k = 0; k < P; ++k) D[j][k] = obs_w[(ib * MO + j) * P + k];
            }
#pragma unroll
            for (int j = 0; j < MO; ++j) {
                z[j] = y[j] - dot<P>(D[j], m);                                   // x_meas - (D mu + 0)
#pragma unroll
                for (int c = 0; c < P; ++c) {
                    double t = D[j][0] * S[0][c];
#pragma unroll
                    for (int k = 1; k < P; ++k) t = fma(D[j][k], S[k][c], t);
                    DS[j][c] = t;                                               // D Sigma  (var_meas_state_pred)
                }
            }
#pragma unroll
            for (int j = 0; j < MO; ++j)
#pragma unroll
                for (int l2 = 0; l2 < MO; ++l2) Wf[j][l2] = dot<P>(DS[j], D[l2]) + obs_v[(ib * MO + j) * MO + l2];   // var_fore
            {   // log N(y; D mu, var_fore) by eigendecomposition (utils.py:60-78)
                double Aw[MO][MO], w[MO], V[MO][MO];
#pragma unroll
                for (int j = 0; j < MO; ++j)
#pragma unroll
                    for (int l2 = 0; l2 < MO; ++l2) Aw[j][l2] = 0.5 * (Wf[j][l2] + Wf[l2][j]);
                sym_eig_jacobi<MO>(Aw, w, V);
#pragma unroll
                for (int k = 0; k < MO; ++k) {
                    double zk = 0.0;
#pragma unroll
                    for (int j = 0; j < MO; ++j) zk = fma(V[j][k], z[j], zk);
                    if (fabs(w[k]) > 1e-8) acc += -0.5 * (zk * zk / w[k] + log(w[k])) - 0.5 * LOG_2PI;
                }
            }
            // K^T = solve(var_fore, (Sigma D^T)^T) by LU with partial pivoting (utils.py:119)
#pragma unroll
            for (int j = 0; j < MO; ++j)
#pragma unroll
                for (int r = 0; r < P; ++r) X[j][r] = dot<P>(S[r], D[j]);        // (Sigma D^T)^T, row j
            lu_solve<MO, P>(Wf, X);
            double dm[P], dS[P][P];
#pragma unroll
            for (int r = 0; r < P; ++r) {
                double t = X[0][r] * z[0];
#pragma unroll
                for (int j = 1; j < MO; ++j) t = fma(X[j][r], z[j], t);
                dm[r] = t;
#pragma unroll
                for (int c = 0; c < P; ++c) {
                    double u = X[0][r] * DS[0][c];
#pragma unroll
                    for (int j = 1; j < MO; ++j) u = fma(X[j][r], DS[j][c], u);
                    dS[r][c] = u;
                }
            }
#pragma unroll
            for (int r = 0; r < P; ++r) {
                m[r] = m[r] + dm[r];                                            // mu + K (y - D mu)
#pragma unroll
                for (int c = 0; c < P; ++c) S[r][c] = S[r][c] - dS[r][c];       // Sigma - K (D Sigma)
            }
        }
        --i;
    };
    constexpr int ITEM = 3 * P * P + 2 * P;
    auto keep = [&](int n, int off, const double (&m)[P], const double (&S)[P][P]) {
        double* o = states + (((size_t)n * a.D + blk) * ITEM + off) * B + b;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            o[(size_t)r * B] = m[r];
#pragma unroll
            for (int c = 0; c < P; ++c) o[(size_t)(P + r * P + c) * B] = S[r][c];
        }
    };
    if constexpr (STORE) keep(a.N, 0, bm, bS);                              // "prediction" at N = the terminal point
    if (i >= 0 && obs_ind[i] >= a.N) observe(bm, bS);                      // fenrir.py:189-209
    if constexpr (STORE) keep(a.N, P + P * P, bm, bS);
    for (int n = a.N - 1; n >= 0; --n) {
        double mf[P], Sf[P][P], mp[P], Sp[P][P], T[P][P], G[P][P];
        if constexpr (TILES) {
            load_filt_tiles<P>(a, tiles, n, blk, b, mf, Sf);
            predict_block<P>(Q, R, mf, Sf, mp, Sp);                         // pred[n + 1] from filt[n]
        } else {
            load_filt<P>(a, n, blk, b, mf, Sf);
            // stored predicted moments of time n + 1 (solve.py:93-96)
            const double* mi = a.mean_pred + ((size_t)(n + 1) * a.D + blk) * P * B + b;
            const double* vi = a.var_pred + ((size_t)(n + 1) * a.D + blk) * P * P * B + b;
#pragma unroll
            for (int r = 0; r < P; ++r) {
                mp[r] = mi[(size_t)r * B];
#pragma unroll
                for (int c = 0; c < P; ++c) Sp[r][c] = vi[((size_t)r * P + c) * B];
            }
        }
        smooth_gain<P>(Q, Sf, Sp, T, G);                                    // A = G            (standard.py:175-176)
        double bb[P], Cc[P][P], GT[P][P];
        mm_nt<P, P, P>(G, T, GT);
#pragma unroll
        for (int r = 0; r < P; ++r) {
            bb[r] = mf[r] - dot<P>(G[r], mp);                               // b = mu_f - G mu-   (standard.py:368)
#pragma unroll
            for (int c = 0; c < P; ++c) Cc[r][c] = Sf[r][c] - GT[r][c];     // C = Sigma_f - G T^T (standard.py:369-370)
        }
        double nm[P], nS[P][P];
        predict_block<P>(G, Cc, bm, bS, nm, nS);                            // A m + 0, A S A^T + C (standard.py:57-59)
#pragma unroll
        for (int r = 0; r < P; ++r) {
            bm[r] = nm[r] + bb[r];
#pragma unroll
            for (int c = 0; c < P; ++c) bS[r][c] = nS[r][c];
        }
        if constexpr (STORE) {
            keep(n, 0, bm, bS);
            double* o = states + (((size_t)n * a.D + blk) * ITEM + 2 * (P + P * P)) * B + b;
#pragma unroll
            for (int r = 0; r < P; ++r)
#pragma unroll
                for (int c = 0; c < P; ++c) o[(size_t)(r * P + c) * B] = G[r][c];
        }
        if (i >= 0 && obs_ind[i] == n) observe(bm, bS);                     // fenrir.py:155-170
        if constexpr (STORE) keep(n, P + P * P, bm, bS);
    }
    if (logdens) atomicAdd(&logdens[b], acc);
}

// fenrir.py:333-402: the smoothing pass over the backward filter's stored moments, a forward sweep in time -- times 0 and
// 1 keep the backward filter's own estimates; for k = 0 .. N-2
//     (m, S)_{k+2} = smooth_mv(next = (m, S)_{k+1}, wgt_state = A_{k+1}, filt = bfilt_{k+2}, pred = bpred_{k+1}).
// Results go to a.mean / a.var (batch-minor), which the backward filter no longer needs.
template <int P>
__global__ void __launch_bounds__(64) fenrir_smooth_kernel(SolveArgs a, const double* __restrict__ states) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.B * a.D) return;
    const int blk = l / a.B, b = l - blk * a.B;
    const size_t B = (size_t)a.B;
    constexpr int ITEM = 3 * P * P + 2 * P;
    auto item = [&](int n, int off, double (&m)[P], double (&S)[P][P]) {
        const double* o = states + (((size_t)n * a.D + blk) * ITEM + off) * B + b;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            m[r] = o[(size_t)r * B];
#pragma unroll
            for (int c = 0; c < P; ++c) S[r][c] = o[(size_t)(P + r * P + c) * B];
        }
    };
    auto put = [&](int n, const double (&m)[P], const double (&S)[P][P]) {
        double* mo = a.mean + ((size_t)n * a.D + blk) * P * B + b;
        double* vo = a.var + ((size_t)n * a.D + blk) * P * P * B + b;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            mo[(size_t)r * B] = m[r];
#pragma unroll
            for (int c = 0; c < P; ++c) vo[((size_t)r * P + c) * B] = S[r][c];
        }
    };
    double cm[P], cS[P][P];
    item(0, P + P * P, cm, cS);
    put(0, cm, cS);
    if (a.N < 1) return;
    item(1, P + P * P, cm, cS);
    put(1, cm, cS);
    for (int k = 0; k + 2 <= a.N; ++k) {
        double fm[P], fS[P][P], pm[P], pS[P][P], A[P][P], T[P][P], G[P][P];
        item(k + 2, P + P * P, fm, fS);
        item(k + 1, 0, pm, pS);
        {
            const double* o = states + (((size_t)(k + 1) * a.D + blk) * ITEM + 2 * (P + P * P)) * B + b;
#pragma unroll
            for (int r = 0; r < P; ++r)
#pragma unroll
                for (int c = 0; c < P; ++c) A[r][c] = o[(size_t)(r * P + c) * B];
        }
        smooth_gain<P>(A, fS, pS, T, G);                                    // standard.py:175-176 with wgt_state = A
        double dm[P], dS[P][P], GD[P][P], nS[P][P];
#pragma unroll
        for (int r = 0; r < P; ++r) {
            dm[r] = cm[r] - pm[r];
#pragma unroll
            for (int c = 0; c < P; ++c) dS[r][c] = cS[r][c] - pS[r][c];
        }
        mm<P, P, P>(G, dS, GD);
        mm_nt<P, P, P>(GD, G, nS);
#pragma unroll
        for (int r = 0; r < P; ++r) {
            cm[r] = fm[r] + dot<P>(G[r], dm);                               // standard.py:213-214
#pragma unroll
            for (int c = 0; c < P; ++c) cS[r][c] = fS[r][c] + nS[r][c];     // standard.py:215-216
        }
        put(k + 2, cm, cS);
    }
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
static int make_args(const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out, SolveArgs& a) {
    a.B = c->n_traj; a.N = c->n_steps; a.D = c->n_block;
    a.t_min = c->t_min; a.t_max = c->t_max; a.seed = c->seed; a.traj_offset = c->traj_offset;
    a.W = in->ode_weight; a.W_b = in->ode_weight_batched;
    a.x0 = in->ode_init; a.x0_b = in->ode_init_batched;
    a.Q = in->prior_weight; a.Q_b = in->prior_weight_batched;
    a.R = in->prior_var; a.R_b = in->prior_var_batched;
    a.theta = in->theta; a.theta_b = in->theta_batched;
    a.mean = out ? out->mean_state : nullptr; a.var = out ? out->var_state : nullptr;
    a.mean_pred = out ? out->mean_pred : nullptr; a.var_pred = out ? out->var_pred : nullptr;
    a.x = out ? out->x_state : nullptr;
    return RK_OK;
}

static int check_cfg(const rk_solve_cfg* c, const rk_solve_in* in) {
    RK_REQUIRE(c && in, RK_ERR_INVALID, "null cfg / in");
    RK_REQUIRE(c->n_traj >= 1 && c->n_steps >= 1 && c->n_block >= 1 && c->n_bstate >= 1 && c->n_bmeas >= 1,
               RK_ERR_INVALID, "non-positive dimension (n_traj=%d n_steps=%d n_block=%d n_bstate=%d n_bmeas=%d)",
               c->n_traj, c->n_steps, c->n_block, c->n_bstate, c->n_bmeas);
    RK_REQUIRE(in->ode_weight && in->ode_init && in->prior_weight && in->prior_var, RK_ERR_INVALID,
               "ode_weight / ode_init / prior_weight / prior_var must not be NULL");
    RK_REQUIRE(c->kalman_type == RK_KALMAN_STANDARD || c->kalman_type == RK_KALMAN_SQRT, RK_ERR_UNSUPPORTED,
               "unknown kalman_type %d", c->kalman_type);
    RK_REQUIRE(c->interrogate >= 0 && c->interrogate <= 3, RK_ERR_UNSUPPORTED, "unknown interrogate id %d",
               c->interrogate);
    return RK_OK;
}

template <class RHS, int P, int ITG>
static int launch_fwd_t(rk_handle h, const SolveArgs& a, bool store_pred) {
    const dim3 grid(div_up(a.B, 64)), block(64);
    LaunchTimer t(h, "fwd_kernel");
    if (store_pred) hipLaunchKernelGGL((fwd_kernel<RHS, P, ITG, true>), grid, block, 0, h->stream, a);
    else hipLaunchKernelGGL((fwd_kernel<RHS, P, ITG, false>), grid, block, 0, h->stream, a);
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

template <class RHS, int P>
static int launch_fwd_p(rk_handle h, const SolveArgs& a, int itg, bool sp) {
    switch (itg) {
        case RK_INTERROGATE_RODEO: return launch_fwd_t<RHS, P, RK_INTERROGATE_RODEO>(h, a, sp);
        case RK_INTERROGATE_SCHOBER: return launch_fwd_t<RHS, P, RK_INTERROGATE_SCHOBER>(h, a, sp);
        case RK_INTERROGATE_KRAMER: return launch_fwd_t<RHS, P, RK_INTERROGATE_KRAMER>(h, a, sp);
        case RK_INTERROGATE_CHKREBTII: return launch_fwd_t<RHS, P, RK_INTERROGATE_CHKREBTII>(h, a, sp);
    }
    set_error("unknown interrogate id %d", itg);
    return RK_ERR_UNSUPPORTED;
}

template <class RHS>
static int launch_fwd_rhs(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a) {
    RK_REQUIRE(c->n_block == RHS::D && c->n_bmeas == 1, RK_ERR_UNSUPPORTED,
               "rhs %d needs n_block=%d, n_bmeas=1 (got %d, %d)", c->rhs_id, RHS::D, c->n_block, c->n_bmeas);
    RK_REQUIRE(c->n_theta == 0 || c->n_theta >= RHS::NTHETA || !a.theta, RK_ERR_INVALID,
               "rhs %d needs %d parameters, got n_theta=%d", c->rhs_id, RHS::NTHETA, c->n_theta);
    const bool sp = (c->flags & RK_FLAG_STORE_PRED) != 0;
    switch (c->n_bstate) {
        case 2: return launch_fwd_p<RHS, 2>(h, a, c->interrogate, sp);
        case 3: return launch_fwd_p<RHS, 3>(h, a, c->interrogate, sp);
        case 4: return launch_fwd_p<RHS, 4>(h, a, c->interrogate, sp);
        case 5: return launch_fwd_p<RHS, 5>(h, a, c->interrogate, sp);
        case 6: return launch_fwd_p<RHS, 6>(h, a, c->interrogate, sp);
    }
    set_error("lane-per-trajectory path supports n_bstate in [2, 6] (the blocked tile path 4 .. 8 without RK_FLAG_STORE_PRED / "
              "RK_FLAG_BATCH_MINOR), got %d", c->n_bstate);
    return RK_ERR_UNSUPPORTED;
}

int user_forward(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a);
bool is_user_rhs(int rhs_id);

int small_forward(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a) {
    if (is_user_rhs(c->rhs_id)) return user_forward(h, c, a);
    switch (c->rhs_id) {
        case RK_RHS_FITZHUGH_NAGUMO: return launch_fwd_rhs<FitzHughNagumo>(h, c, a);
        case RK_RHS_LORENZ63: return launch_fwd_rhs<Lorenz63>(h, c, a);
        case RK_RHS_HIGHER_ORDER: return launch_fwd_rhs<HigherOrder>(h, c, a);
    }
    set_error("unknown rhs_id %d for the small-block path", c->rhs_id);
    return RK_ERR_UNSUPPORTED;
}

template <bool SIM>
static int small_backward(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a) {
    const dim3 grid(div_up(a.B * a.D, 64)), block(64);
    LaunchTimer t(h, SIM ? "bwd_sim_kernel" : "bwd_mv_kernel");
#define RK_BWD(P_)                                                                              \
    case P_:                                                                                    \
        if (SIM) hipLaunchKernelGGL((bwd_sim_kernel<P_>), grid, block, 0, h->stream, a);        \
        else hipLaunchKernelGGL((bwd_mv_kernel<P_>), grid, block, 0, h->stream, a);             \
        break;
    switch (c->n_bstate) {
        RK_BWD(2) RK_BWD(3) RK_BWD(4) RK_BWD(5) RK_BWD(6) RK_BWD(7) RK_BWD(8) RK_BWD(9)
        default:
            set_error("small-block path supports n_bstate in [2, 9] for the backward pass, got %d", c->n_bstate);
            return RK_ERR_UNSUPPORTED;
    }
#undef RK_BWD
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

template <class RHS>
static int launch_itg_rhs(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double t, int step,
                          const double* mp, const double* vp, double* wm, double* mm_, double* vm) {
    RK_REQUIRE(c->n_block == RHS::D && c->n_bmeas == 1, RK_ERR_UNSUPPORTED,
               "rhs %d needs n_block=%d, n_bmeas=1 (got %d, %d)", c->rhs_id, RHS::D, c->n_block, c->n_bmeas);
    const dim3 grid(div_up(a.B, 64)), block(64);
    const int sqrt_mode = c->kalman_type == RK_KALMAN_SQRT ? 1 : 0;      // only interrogate_chkrebtii reads it
#define RK_ITG2(P_, I_)                                                                                          \
    hipLaunchKernelGGL((interrogate_kernel<RHS, P_, I_>), grid, block, 0, h->stream, a, t, step, mp, vp, wm, mm_, vm, sqrt_mode)
#define RK_ITG(P_)                                                                      \
    case P_:                                                                            \
        switch (c->interrogate) {                                                       \
            case RK_INTERROGATE_RODEO: RK_ITG2(P_, RK_INTERROGATE_RODEO); break;        \
            case RK_INTERROGATE_SCHOBER: RK_ITG2(P_, RK_INTERROGATE_SCHOBER); break;    \
            case RK_INTERROGATE_KRAMER: RK_ITG2(P_, RK_INTERROGATE_KRAMER); break;      \
            case RK_INTERROGATE_CHKREBTII: RK_ITG2(P_, RK_INTERROGATE_CHKREBTII); break; \
        }                                                                               \
        break;
    switch (c->n_bstate) {
        RK_ITG(2) RK_ITG(3) RK_ITG(4) RK_ITG(5) RK_ITG(6)
        default:
            set_error("rk_interrogate_batched supports n_bstate in [2, 6], got %d", c->n_bstate);
            return RK_ERR_UNSUPPORTED;
    }
#undef RK_ITG
#undef RK_ITG2
    RK_HIP(hipGetLastError());
    return RK_OK;
}

// dense large-block path (solve_dense.hip)
bool dense_supported(const rk_solve_cfg* c, int mode);
int dense_check(const rk_solve_cfg* c, const rk_solve_in* in, int mode);
int dense_solve(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out, int mode);
size_t dense_ws_bytes(const rk_solve_cfg* c, int mode);

// user-supplied right-hand sides (rhs_jit.hip)
bool is_user_rhs(int rhs_id);
int user_forward(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a);
int user_interrogate(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double t, int step, const double* mp,
                     const double* vp, double* wm, double* mm_, double* vm);

// fused square-root solver (solve_sqrt.hip)
int sqrt_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, int mode, double* ws, size_t ws_bytes);
size_t sqrt_ws_doubles(const rk_solve_cfg* c, int mode);

// fenrir with kalman_type = square-root (fenrir_sqrt.hip)
size_t fenrir_sqrt_item_doubles(int p);
int fenrir_sqrt_launch(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, const double* obs, const double* obs_w,
                       const double* obs_v, const int32_t* obs_ind, int n_obs, int n_bobs, double* logdens, double* states);

// MFMA-tile path for n_bstate = 4 (solve_tile4.hip)
bool tile4_supported(const rk_solve_cfg* c, int mode);
int tile4_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int mode);
size_t tile4_doubles(const rk_solve_cfg* c);

// blocked MFMA-tile path for n_bstate = 4 .. 8 (solve_tilen.hip)
bool tilen_supported(const rk_solve_cfg* c, int mode);
int tilen_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, double* ws, size_t ws_bytes, int mode);
size_t tilen_tile_doubles(const rk_solve_cfg* c);
size_t tilen_ws_doubles(const rk_solve_cfg* c, int mode);

// MFMA-tile path (solve_tile3.hip)
bool tile3_supported(const rk_solve_cfg* c, int mode);
struct SimLogpost;
int tile3_solve(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int mode, const SimLogpost* lp = nullptr);
bool tile3_sim_logpost_supported(const rk_solve_cfg* c, int n_obs);
int tile3_solve_sim_logpost(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, const double* obs,
                            const int32_t* obs_ind, int n_obs, double noise_sd, const double* upars, int n_prior, double prior_sd,
                            double* logpost);
int tile3_fenrir_backward(rk_handle h, const SolveArgs& a, const double* tiles, const double* obs, const double* obs_w,
                          const double* obs_v, const int32_t* obs_ind, int n_obs, double* logdens);

}  // namespace rk

using namespace rk;

extern "C" {

int rk_solve_layout(const rk_solve_cfg* c, int32_t mode, int32_t* layout) {
    RK_REQUIRE(c && layout, RK_ERR_INVALID, "rk_solve_layout: null argument");
    RK_REQUIRE(mode >= RK_MODE_FILTER && mode <= RK_MODE_SIM, RK_ERR_INVALID, "rk_solve_layout: bad mode %d", mode);
    *layout = dense_supported(c, mode) ? RK_LAYOUT_TRAJ_MAJOR
              : (tile3_supported(c, mode) ? RK_LAYOUT_TILE3
                 : (tile4_supported(c, mode) ? RK_LAYOUT_TILE4
                    : (tilen_supported(c, mode) ? (c->n_bstate == 4 ? RK_LAYOUT_TILE4 : RK_LAYOUT_TILEP) : RK_LAYOUT_BATCH_MINOR)));
    return RK_OK;
}

int rk_solve_workspace_bytes(const rk_solve_cfg* c, int32_t mode, size_t* bytes) {
    RK_REQUIRE(c && bytes, RK_ERR_INVALID, "rk_solve_workspace_bytes: null argument");
    if (dense_supported(c, mode)) *bytes = dense_ws_bytes(c, mode);
    else if (c->kalman_type == RK_KALMAN_SQRT) *bytes = sqrt_ws_doubles(c, mode) * sizeof(double);     // optional (solve_sqrt.hip)
    else if (!tile3_supported(c, mode) && !tile4_supported(c, mode) && tilen_supported(c, mode))
        *bytes = tilen_ws_doubles(c, mode) * sizeof(double);
    else *bytes = 0;
    return RK_OK;
}

int rk_solve_sizes(const rk_solve_cfg* c, int32_t layout, size_t* mean_bytes, size_t* var_bytes) {
    RK_REQUIRE(c, RK_ERR_INVALID, "rk_solve_sizes: null cfg");
    const size_t m = (size_t)(c->n_steps + 1) * c->n_block * c->n_bstate * (size_t)c->n_traj * sizeof(double);
    if (layout == RK_LAYOUT_TILE3) {
        RK_REQUIRE(c->n_bstate == 3, RK_ERR_INVALID, "RK_LAYOUT_TILE3 needs n_bstate = 3");
        if (mean_bytes) *mean_bytes = 0;
        if (var_bytes) {
            const size_t n_tiles = (size_t)c->n_block * (size_t)c->n_traj;
            *var_bytes = ((size_t)(c->n_steps + 1) * n_tiles * 12 + ((n_tiles + 7) / 8) * 128) * sizeof(double);
        }
        return RK_OK;
    }
    if (layout == RK_LAYOUT_TILE4) {
        RK_REQUIRE(c->n_bstate == 4, RK_ERR_INVALID, "RK_LAYOUT_TILE4 needs n_bstate = 4");
        if (mean_bytes) *mean_bytes = 0;
        if (var_bytes) *var_bytes = tile4_doubles(c) * sizeof(double);
        return RK_OK;
    }
    if (layout == RK_LAYOUT_TILEP) {
        RK_REQUIRE(c->n_bstate >= 4 && c->n_bstate <= 8, RK_ERR_INVALID, "RK_LAYOUT_TILEP needs n_bstate in 4..8");
        if (mean_bytes) *mean_bytes = 0;
        if (var_bytes) *var_bytes = tilen_tile_doubles(c) * sizeof(double);
        return RK_OK;
    }
    RK_REQUIRE(layout == RK_LAYOUT_BATCH_MINOR || layout == RK_LAYOUT_TRAJ_MAJOR, RK_ERR_INVALID, "unknown layout %d", layout);
    if (mean_bytes) *mean_bytes = m;
    if (var_bytes) *var_bytes = m * c->n_bstate;
    return RK_OK;
}

static int solve_common(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out,
                        int mode /*0 filter, 1 mv, 2 sim*/) {
    RK_REQUIRE(h, RK_ERR_INVALID, "null handle");
    int rc = check_cfg(c, in);
    if (rc) return rc;
    const bool dense = dense_supported(c, mode);
    const bool tile4 = !dense && tile4_supported(c, mode);
    const bool tile3 = !dense && !tile4 && tile3_supported(c, mode);
    const bool tilen = !dense && !tile4 && !tile3 && tilen_supported(c, mode);
    const bool tile = tile4 || tile3 || tilen;
    RK_REQUIRE(out && out->var_state && (tile || out->mean_state), RK_ERR_INVALID,
               "out->mean_state / var_state must not be NULL");
    if (dense) {
        rc = dense_check(c, in, mode);
        if (rc) return rc;
        RK_HIP(hipSetDevice(h->device));
        if (!h->profile_keep) { h->prof.clear(); h->event_used = 0; }
        return dense_solve(h, c, in, out, mode);
    }
    RK_REQUIRE(!(c->flags & RK_FLAG_STORE_PRED) || (out->mean_pred && out->var_pred), RK_ERR_INVALID,
               "RK_FLAG_STORE_PRED needs out->mean_pred / var_pred");
    RK_REQUIRE(mode != 2 || out->x_state, RK_ERR_INVALID, "rk_solve_sim needs out->x_state");
    RK_HIP(hipSetDevice(h->device));
    if (!h->profile_keep) { h->prof.clear(); h->event_used = 0; }
    SolveArgs a;
    make_args(c, in, out, a);
    if (c->kalman_type == RK_KALMAN_SQRT) return sqrt_solve(h, c, a, mode, (double*)out->workspace, out->workspace_bytes);
    if (tile4) return tile4_solve(h, c, a, out->var_state, mode);
    if (tile3) return tile3_solve(h, c, a, out->var_state, mode);
    if (tilen) return tilen_solve(h, c, a, out->var_state, (double*)out->workspace, out->workspace_bytes, mode);
    rc = small_forward(h, c, a);
    if (rc) return rc;
    if (mode == 1) rc = small_backward<false>(h, c, a);
    if (mode == 2) rc = small_backward<true>(h, c, a);
    return rc;
}

int rk_solve_filter(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out) {
    return solve_common(h, c, in, out, 0);
}
int rk_solve_mv(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out) {
    return solve_common(h, c, in, out, 1);
}
int rk_solve_sim(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out) {
    return solve_common(h, c, in, out, 2);
}

int rk_gauss_obs_logpost(rk_handle h, int32_t n_traj, int32_t n_steps, int32_t n_block, int32_t n_bstate,
                         int32_t layout, const double* x_state, const double* obs, const int32_t* obs_ind,
                         int32_t n_obs, double noise_sd, const double* upars, int32_t n_prior, double prior_sd,
                         double* logpost);

int rk_solve_sim_logpost(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out,
                         const double* obs, const int32_t* obs_ind, int32_t n_obs, double noise_sd,
                         const double* upars, int32_t n_prior, double prior_sd, double* logpost) {
    RK_REQUIRE(h, RK_ERR_INVALID, "null handle");
    RK_REQUIRE(obs && obs_ind && logpost && n_obs >= 0, RK_ERR_INVALID, "rk_solve_sim_logpost: null obs / obs_ind / logpost");
    RK_REQUIRE(!upars || n_prior >= 0, RK_ERR_INVALID, "rk_solve_sim_logpost: n_prior < 0");
    int rc = check_cfg(c, in);
    if (rc) return rc;
    const bool dense = dense_supported(c, 2);
    if (!dense && !tile4_supported(c, 2) && tile3_sim_logpost_supported(c, n_obs) && c->kalman_type == RK_KALMAN_STANDARD) {
        // the sampler's consumer wave reduces the log-posterior itself; out->x_state may be NULL (no path is stored)
        RK_REQUIRE(out && out->var_state, RK_ERR_INVALID, "out->var_state must not be NULL");
        RK_HIP(hipSetDevice(h->device));
        if (!h->profile_keep) { h->prof.clear(); h->event_used = 0; }
        SolveArgs a;
        make_args(c, in, out, a);
        return tile3_solve_sim_logpost(h, c, a, out->var_state, obs, obs_ind, n_obs, noise_sd, upars, n_prior, prior_sd, logpost);
    }
    // any other configuration: the sampler, then the reduction kernel on its path (one call, no host work in between)
    RK_REQUIRE(out && out->x_state, RK_ERR_INVALID, "rk_solve_sim_logpost: this configuration needs out->x_state");
    rc = solve_common(h, c, in, out, 2);
    if (rc) return rc;
    const bool keep = h->profile_keep;
    h->profile_keep = true;                            // (keep the sampler's kernel times next to the reduction's)
    rc = rk_gauss_obs_logpost(h, c->n_traj, c->n_steps, c->n_block, c->n_bstate, RK_LAYOUT_BATCH_MINOR, out->x_state, obs,
                              obs_ind, n_obs, noise_sd, upars, n_prior, prior_sd, logpost);
    h->profile_keep = keep;
    return rc;
}

int rk_interrogate_batched(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, double t, int32_t step,
                           const double* mean_state_pred, const double* var_state_pred, double* wgt_meas,
                           double* mean_meas, double* var_meas) {
    RK_REQUIRE(h, RK_ERR_INVALID, "null handle");
    RK_REQUIRE(c && in && in->ode_weight, RK_ERR_INVALID, "rk_interrogate_batched: null cfg / in / ode_weight");
    RK_REQUIRE(mean_state_pred && var_state_pred && wgt_meas && mean_meas && var_meas, RK_ERR_INVALID,
               "rk_interrogate_batched: null array");
    RK_REQUIRE(c->n_traj >= 1, RK_ERR_INVALID, "rk_interrogate_batched: n_traj must be positive");
    RK_HIP(hipSetDevice(h->device));
    SolveArgs a;
    make_args(c, in, nullptr, a);
    if (is_user_rhs(c->rhs_id))
        return user_interrogate(h, c, a, t, step, mean_state_pred, var_state_pred, wgt_meas, mean_meas, var_meas);
    switch (c->rhs_id) {
        case RK_RHS_FITZHUGH_NAGUMO:
            return launch_itg_rhs<FitzHughNagumo>(h, c, a, t, step, mean_state_pred, var_state_pred, wgt_meas, mean_meas, var_meas);
        case RK_RHS_LORENZ63:
            return launch_itg_rhs<Lorenz63>(h, c, a, t, step, mean_state_pred, var_state_pred, wgt_meas, mean_meas, var_meas);
        case RK_RHS_HIGHER_ORDER:
            return launch_itg_rhs<HigherOrder>(h, c, a, t, step, mean_state_pred, var_state_pred, wgt_meas, mean_meas, var_meas);
    }
    set_error("unknown rhs_id %d", c->rhs_id);
    return RK_ERR_UNSUPPORTED;
}

int rk_gauss_obs_logpost(rk_handle h, int32_t n_traj, int32_t n_steps, int32_t n_block, int32_t n_bstate,
                         int32_t layout, const double* x_state, const double* obs, const int32_t* obs_ind,
                         int32_t n_obs, double noise_sd, const double* upars, int32_t n_prior, double prior_sd,
                         double* logpost) {
    RK_REQUIRE(h && x_state && obs && obs_ind && logpost, RK_ERR_INVALID, "rk_gauss_obs_logpost: null argument");
    RK_REQUIRE(layout == RK_LAYOUT_BATCH_MINOR || (layout == RK_LAYOUT_TILE3 && n_bstate == 3) ||
                   (layout == RK_LAYOUT_TILE4 && n_bstate == 4) || (layout == RK_LAYOUT_TILEP && n_bstate >= 4), RK_ERR_INVALID,
               "rk_gauss_obs_logpost: bad layout %d for n_bstate %d", layout, n_bstate);
    RK_REQUIRE(n_traj >= 1 && n_obs >= 0 && n_block >= 1 && n_bstate >= 1 && n_steps >= 1, RK_ERR_INVALID,
               "rk_gauss_obs_logpost: bad dimension");
    RK_HIP(hipSetDevice(h->device));
    LaunchTimer t(h, "gauss_logpost_kernel");
    hipLaunchKernelGGL(gauss_logpost_kernel, dim3(n_traj), dim3(64), 0, h->stream, n_traj, n_steps, n_block,
                       n_bstate, layout == RK_LAYOUT_TILE3 ? 12 : (layout == RK_LAYOUT_BATCH_MINOR ? 0 : n_bstate * (n_bstate + 1)),
                       layout == RK_LAYOUT_TILE3 ? 3 : n_bstate * n_bstate, x_state, obs, obs_ind, n_obs, noise_sd, upars, n_prior, prior_sd, logpost);
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

int rk_fenrir_backward(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out,
                       const double* obs, const double* obs_weight, const double* obs_var, const int32_t* obs_ind,
                       int32_t n_obs, int32_t n_bobs, double* logdens) {
    RK_REQUIRE(h && c && in && out && obs && obs_weight && obs_var && obs_ind && logdens, RK_ERR_INVALID,
               "rk_fenrir_backward: null argument");
    RK_REQUIRE(n_bobs >= 1 && n_bobs <= 3, RK_ERR_UNSUPPORTED, "rk_fenrir_backward: n_bobs in 1..3, got %d", n_bobs);
    if (c->kalman_type == RK_KALMAN_SQRT) {
        // fenrir.py:292-296: every step map from square_root.py; the filtered FACTORS of the square-root forward pass
        // (batch-minor, solve_sqrt.hip) are all it needs -- the predicted ones are re-evaluated (fenrir_sqrt.hip)
        RK_REQUIRE(out->mean_state && out->var_state && n_obs >= 0, RK_ERR_INVALID,
                   "rk_fenrir_backward (square-root): out->mean_state / var_state of rk_solve_filter are null");
        SolveArgs as;
        int rcs = make_args(c, in, out, as);
        if (rcs) return rcs;
        RK_HIP(hipSetDevice(h->device));
        RK_HIP(hipMemsetAsync(logdens, 0, sizeof(double) * (size_t)c->n_traj, h->stream));
        return fenrir_sqrt_launch(h, c, as, obs, obs_weight, obs_var, obs_ind, n_obs, n_bobs, logdens, nullptr);
    }
    RK_REQUIRE(c->kalman_type == RK_KALMAN_STANDARD, RK_ERR_UNSUPPORTED, "rk_fenrir_backward: unknown kalman_type %d", c->kalman_type);
    if (n_bobs == 1 && !(c->flags & (RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR)) && tile3_supported(c, RK_MODE_FILTER)) {
        // the filter ran on the MFMA-tile path: out->var_state holds the RK_LAYOUT_TILE3 tiles, predicted moments are
        // re-evaluated from the filtered ones (solve_tile3.hip, fenrir_bwd_tile3_kernel)
        RK_REQUIRE(out->var_state && n_obs >= 0, RK_ERR_INVALID, "rk_fenrir_backward: out->var_state (tiles) is null");
        SolveArgs at;
        int rct = make_args(c, in, out, at);
        if (rct) return rct;
        RK_HIP(hipSetDevice(h->device));
        RK_HIP(hipMemsetAsync(logdens, 0, sizeof(double) * (size_t)c->n_traj, h->stream));
        return tile3_fenrir_backward(h, at, out->var_state, obs, obs_weight, obs_var, obs_ind, n_obs, logdens);
    }
    if (!(c->flags & (RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR)) && c->n_bstate >= 4 && c->n_bstate <= 8 &&
        (tile4_supported(c, RK_MODE_FILTER) || tilen_supported(c, RK_MODE_FILTER))) {
        // the filter ran on the blocked MFMA tiles (n_bstate 4 .. 8): out->var_state holds its records [Sigma | mu]; the
        // backward filter runs one lane per (block, trajectory) on them and re-evaluates the predicted moments
        RK_REQUIRE(out->var_state && n_obs >= 0, RK_ERR_INVALID, "rk_fenrir_backward: out->var_state (tiles) is null");
        SolveArgs at;
        int rct = make_args(c, in, out, at);
        if (rct) return rct;
        RK_HIP(hipSetDevice(h->device));
        RK_HIP(hipMemsetAsync(logdens, 0, sizeof(double) * (size_t)c->n_traj, h->stream));
        const dim3 tgrid(div_up(at.B * at.D, 64)), tblock(64);
        LaunchTimer t(h, "fenrir_bwd_kernel<tiles>");
#define RK_FT(P_, M_) if (c->n_bstate == P_ && n_bobs == M_) hipLaunchKernelGGL((fenrir_bwd_kernel<P_, false, M_, true>), tgrid, tblock, 0, h->stream, at, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr, (const double*)out->var_state);
        RK_FT(4, 1) RK_FT(4, 2) RK_FT(4, 3) RK_FT(5, 1) RK_FT(5, 2) RK_FT(5, 3) RK_FT(6, 1) RK_FT(6, 2) RK_FT(6, 3)
        RK_FT(7, 1) RK_FT(7, 2) RK_FT(7, 3) RK_FT(8, 1) RK_FT(8, 2) RK_FT(8, 3)
#undef RK_FT
        t.stop();
        RK_HIP(hipGetLastError());
        return RK_OK;
    }
    RK_REQUIRE(out->mean_state && out->var_state && out->mean_pred && out->var_pred, RK_ERR_INVALID,
               "rk_fenrir_backward needs the batch-minor filtered AND predicted moments of rk_solve_filter "
               "(RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR), or a configuration of the tile paths (n_bstate = 3 .. 8) without them");
    RK_REQUIRE(c->n_bstate >= 2 && c->n_bstate <= 6 && n_obs >= 0, RK_ERR_UNSUPPORTED,
               "rk_fenrir_backward: n_bstate in 2..6");
    SolveArgs a;
    int rc = make_args(c, in, out, a);
    if (rc) return rc;
    RK_HIP(hipSetDevice(h->device));
    RK_HIP(hipMemsetAsync(logdens, 0, sizeof(double) * (size_t)c->n_traj, h->stream));
    const dim3 grid(div_up(a.B * a.D, 64)), block(64);
    LaunchTimer t(h, "fenrir_bwd_kernel");
    {
            if (c->n_bstate == 2 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<2, false, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 2 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<2, false, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 2 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<2, false, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 3 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<3, false, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 3 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<3, false, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 3 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<3, false, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 4 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<4, false, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 4 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<4, false, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 4 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<4, false, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 5 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<5, false, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 5 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<5, false, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 5 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<5, false, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 6 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<6, false, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 6 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<6, false, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
            if (c->n_bstate == 6 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<6, false, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, logdens, (double*)nullptr);
    }
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

int rk_fenrir_workspace_bytes(const rk_solve_cfg* c, size_t* bytes) {
    RK_REQUIRE(c && bytes, RK_ERR_INVALID, "rk_fenrir_workspace_bytes: null argument");
    const size_t p = (size_t)c->n_bstate;
    const size_t item = c->kalman_type == RK_KALMAN_SQRT ? fenrir_sqrt_item_doubles(c->n_bstate) : 3 * p * p + 2 * p;
    *bytes = sizeof(double) * (size_t)(c->n_steps + 1) * c->n_block * item * c->n_traj;
    return RK_OK;
}

// fenrir's solve_mv on the records of the blocked-tile forward pass (n_bstate 4 .. 8, rk_solve_filter without RK_FLAG_STORE_PRED |
// RK_FLAG_BATCH_MINOR): out->var_state holds the records [Sigma | mu] per (time, trajectory, block); the backward filter re-evaluates
// the predicted moments from them (fenrir_bwd_kernel<.., STORE, .., TILES>), the smoothing pass writes mean_out (N+1, d, p, B) and
// var_out (N+1, d, p, p, B), batch-minor.
int rk_fenrir_solve_mv_tiles(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out,
                             const double* obs, const double* obs_weight, const double* obs_var, const int32_t* obs_ind,
                             int32_t n_obs, int32_t n_bobs, void* workspace, double* mean_out, double* var_out) {
    RK_REQUIRE(h && c && in && out && obs && obs_weight && obs_var && obs_ind && workspace && mean_out && var_out, RK_ERR_INVALID,
               "rk_fenrir_solve_mv_tiles: null argument");
    RK_REQUIRE(c->kalman_type == RK_KALMAN_STANDARD && !(c->flags & (RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR)) && c->n_bstate >= 4 &&
               c->n_bstate <= 8 && (tile4_supported(c, RK_MODE_FILTER) || tilen_supported(c, RK_MODE_FILTER)), RK_ERR_UNSUPPORTED,
               "rk_fenrir_solve_mv_tiles: a configuration of the blocked-tile forward pass (kalman_type standard, n_bstate 4..8, no "
               "RK_FLAG_STORE_PRED / RK_FLAG_BATCH_MINOR)");
    RK_REQUIRE(out->var_state && n_obs >= 0 && n_bobs >= 1 && n_bobs <= 3, RK_ERR_INVALID,
               "rk_fenrir_solve_mv_tiles: out->var_state (tile records) is null, or n_bobs outside 1..3");
    SolveArgs a;
    int rc = make_args(c, in, out, a);
    if (rc) return rc;
    a.mean = mean_out; a.var = var_out;                             // (the smoothing pass's output; the backward filter reads the records)
    RK_HIP(hipSetDevice(h->device));
    const dim3 grid(div_up(a.B * a.D, 64)), block(64);
    double* st = (double*)workspace;
    {
        LaunchTimer t(h, "fenrir_bwd_kernel<tiles>");
#define RK_FT(P_, M_) if (c->n_bstate == P_ && n_bobs == M_) hipLaunchKernelGGL((fenrir_bwd_kernel<P_, true, M_, true>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st, (const double*)out->var_state);
        RK_FT(4, 1) RK_FT(4, 2) RK_FT(4, 3) RK_FT(5, 1) RK_FT(5, 2) RK_FT(5, 3) RK_FT(6, 1) RK_FT(6, 2) RK_FT(6, 3)
        RK_FT(7, 1) RK_FT(7, 2) RK_FT(7, 3) RK_FT(8, 1) RK_FT(8, 2) RK_FT(8, 3)
#undef RK_FT
        t.stop();
    }
    RK_HIP(hipGetLastError());
    LaunchTimer t(h, "fenrir_smooth_kernel");
    switch (c->n_bstate) {
        case 4: hipLaunchKernelGGL(fenrir_smooth_kernel<4>, grid, block, 0, h->stream, a, st); break;
        case 5: hipLaunchKernelGGL(fenrir_smooth_kernel<5>, grid, block, 0, h->stream, a, st); break;
        case 6: hipLaunchKernelGGL(fenrir_smooth_kernel<6>, grid, block, 0, h->stream, a, st); break;
        case 7: hipLaunchKernelGGL(fenrir_smooth_kernel<7>, grid, block, 0, h->stream, a, st); break;
        default: hipLaunchKernelGGL(fenrir_smooth_kernel<8>, grid, block, 0, h->stream, a, st); break;
    }
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

int rk_fenrir_solve_mv(rk_handle h, const rk_solve_cfg* c, const rk_solve_in* in, const rk_solve_out* out,
                       const double* obs, const double* obs_weight, const double* obs_var, const int32_t* obs_ind,
                       int32_t n_obs, int32_t n_bobs, void* workspace) {
    RK_REQUIRE(h && c && in && out && obs && obs_weight && obs_var && obs_ind && workspace, RK_ERR_INVALID,
               "rk_fenrir_solve_mv: null argument");
    if (c->kalman_type == RK_KALMAN_SQRT) {                          // fenrir.py:421-426 (fenrir_sqrt.hip)
        RK_REQUIRE(out->mean_state && out->var_state && n_obs >= 0, RK_ERR_INVALID,
                   "rk_fenrir_solve_mv (square-root): out->mean_state / var_state of rk_solve_filter are null");
        SolveArgs as;
        int rcs = make_args(c, in, out, as);
        if (rcs) return rcs;
        RK_HIP(hipSetDevice(h->device));
        return fenrir_sqrt_launch(h, c, as, obs, obs_weight, obs_var, obs_ind, n_obs, n_bobs, nullptr, (double*)workspace);
    }
    RK_REQUIRE(c->kalman_type == RK_KALMAN_STANDARD, RK_ERR_UNSUPPORTED, "rk_fenrir_solve_mv: unknown kalman_type %d", c->kalman_type);
    RK_REQUIRE((c->flags & RK_FLAG_STORE_PRED) && (c->flags & RK_FLAG_BATCH_MINOR) && out->mean_state && out->var_state &&
               out->mean_pred && out->var_pred, RK_ERR_INVALID,
               "rk_fenrir_solve_mv needs the batch-minor filtered AND predicted moments of rk_solve_filter "
               "(RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR)");
    RK_REQUIRE(c->n_bstate >= 2 && c->n_bstate <= 6 && n_obs >= 0 && n_bobs >= 1 && n_bobs <= 3, RK_ERR_UNSUPPORTED,
               "rk_fenrir_solve_mv: n_bstate in 2..6, n_bobs in 1..3");
    SolveArgs a;
    int rc = make_args(c, in, out, a);
    if (rc) return rc;
    RK_HIP(hipSetDevice(h->device));
    const dim3 grid(div_up(a.B * a.D, 64)), block(64);
    double* st = (double*)workspace;
    {
        LaunchTimer t(h, "fenrir_bwd_kernel");
        {
            if (c->n_bstate == 2 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<2, true, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 2 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<2, true, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 2 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<2, true, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 3 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<3, true, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 3 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<3, true, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 3 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<3, true, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 4 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<4, true, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 4 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<4, true, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 4 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<4, true, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 5 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<5, true, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 5 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<5, true, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 5 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<5, true, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 6 && n_bobs == 1) hipLaunchKernelGGL((fenrir_bwd_kernel<6, true, 1>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 6 && n_bobs == 2) hipLaunchKernelGGL((fenrir_bwd_kernel<6, true, 2>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
            if (c->n_bstate == 6 && n_bobs == 3) hipLaunchKernelGGL((fenrir_bwd_kernel<6, true, 3>), grid, block, 0, h->stream, a, obs, obs_weight, obs_var, obs_ind, n_obs, (double*)nullptr, st);
        }
        t.stop();
    }
    RK_HIP(hipGetLastError());
    LaunchTimer t(h, "fenrir_smooth_kernel");
    switch (c->n_bstate) {
        case 2: hipLaunchKernelGGL(fenrir_smooth_kernel<2>, grid, block, 0, h->stream, a, st); break;
        case 3: hipLaunchKernelGGL(fenrir_smooth_kernel<3>, grid, block, 0, h->stream, a, st); break;
        case 4: hipLaunchKernelGGL(fenrir_smooth_kernel<4>, grid, block, 0, h->stream, a, st); break;
        case 5: hipLaunchKernelGGL(fenrir_smooth_kernel<5>, grid, block, 0, h->stream, a, st); break;
        default: hipLaunchKernelGGL(fenrir_smooth_kernel<6>, grid, block, 0, h->stream, a, st); break;
    }
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

}  // extern "C"
