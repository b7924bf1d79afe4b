// Fenrir's backward pass and data-adaptive smoother with kalman_type = "square-root"
// (src/rodeo/inference/fenrir.py:292-296 and 421-426 select src/rodeo/kalmantv/square_root.py for every step map of
// fenrir.py:86-259 and 333-402).  One lane per (block, trajectory), like the covariance-form kernels in solve_small.hip;
// every `var` is a lower square-root factor: prior_pars[1] = chol(R), obs_var = chol(Omega), and the filtered factors of
// the square-root forward pass (solve_sqrt.hip), from which the predicted ones are re-evaluated (square_root.py:56-57).
// square_root.forecast returns the full forecast VARIANCE (square_root.py:343-344: var_fore.dot(var_fore.T)), so the value
// handed to multivariate_normal_logpdf (utils.py:60-78) is a covariance and the result a log-likelihood.
// NumPy mirror: oracle/fenrir.py with funs = oracle/sqrt_ops.py.
#include "common.hpp"
#include "kalman_small.hpp"
#include "solve_args.hpp"
#include "sqrt_small.hpp"

namespace rk {

template <int P>
__device__ __forceinline__ void fs_load_state(const SolveArgs& a, int n, int blk, int b, double (&mf)[P], double (&Lf)[P][P]) {
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

// forecast + log-density + update with one observation of M components (square_root.py:342-345, utils.py:60-78,
// square_root.py:88-99 with x_meas = y, mean_meas = 0, wgt_meas = D, var_meas = the factor Vh of Omega)
template <int P, int M>
__device__ __forceinline__ void fs_observe(const double (&D)[M][P], const double (&y)[M], const double (&Vh)[M][M],
                                           double (&m)[P], double (&L)[P][P], double& acc) {
    const double LOG_2PI = 1.83787706640934548356;
    double DL[M][P], Lm[M][M], z[M];
    mm<M, P, P>(D, L, DL);
    add_sqrt<M, P, M>(DL, Vh, Lm);                                      // factor of D L L^T D^T + Omega
#pragma unroll
    for (int j = 0; j < M; ++j) z[j] = y[j] - dot<P>(D[j], m);
    {
        double Wf[M][M], w[M], V[M][M];
        mm_nt<M, M, M>(Lm, Lm, Wf);                                     // var_fore (square_root.py:344)
        if constexpr (M == 1) {
            if (fabs(Wf[0][0]) > 1e-8) acc += -0.5 * (z[0] * z[0] / Wf[0][0] + log(Wf[0][0])) - 0.5 * LOG_2PI;
        } else {
#pragma unroll
            for (int j = 0; j < M; ++j)
#pragma unroll
                for (int l2 = j + 1; l2 < M; ++l2) Wf[j][l2] = Wf[l2][j] = 0.5 * (Wf[j][l2] + Wf[l2][j]);
            sym_eig_jacobi<M>(Wf, w, V);
#pragma unroll
            for (int k = 0; k < M; ++k) {
                double zk = 0.0;
#pragma unroll
                for (int j = 0; j < M; ++j) zk = fma(V[j][k], z[j], zk);
                if (fabs(w[k]) > 1e-8) acc += -0.5 * (zk * zk / w[k] + log(w[k])) - 0.5 * LOG_2PI;
            }
        }
    }
    // K^T = L_m^{-T} ((L_m^{-1} D) L L^T)   (square_root.py:90-95)
    double I1[M][P], I2[M][P], Kt[M][P];
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int k = 0; k < P; ++k) I1[j][k] = D[j][k];
    solve_lower<M, P>(Lm, I1);
    mm<M, P, P>(I1, L, I2);
    mm_nt<M, P, P>(I2, L, Kt);
    solve_upper_t<M, P>(Lm, Kt);
    double A1[P][P], B1[P][M];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        double s = m[i];
#pragma unroll
        for (int j = 0; j < M; ++j) s = fma(Kt[j][i], z[j], s);
        m[i] = s;                                                       // mu + K (y - D mu)
#pragma unroll
        for (int c = 0; c < P; ++c) {
            double u = L[i][c];
#pragma unroll
            for (int j = 0; j < M; ++j) u = fma(-Kt[j][i], DL[j][c], u);
            A1[i][c] = u;                                               // L - K (D L)
        }
#pragma unroll
        for (int c = 0; c < M; ++c) {
            double u = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) u = fma(Kt[j][i], Vh[j][c], u);
            B1[i][c] = u;                                               // K Omega^{1/2}
        }
    }
    add_sqrt<P, P, M>(A1, B1, L);                                       // square_root.py:98-99
}

// STORE: per (time n, block) an item of 4 P^2 + 2 P doubles [m_pred, L_pred, m_filt, L_filt, A, C^{1/2}], batch-minor
// (fenrir.py:236-258; the square-root smoother needs the Markov chain's factor too, square_root.py:217-218).
template <int P, bool STORE, int MO>
__global__ void __launch_bounds__(64) fenrir_bwd_sqrt_kernel(SolveArgs a, const double* __restrict__ obs,
                                                             const double* __restrict__ obs_w, const double* __restrict__ obs_v,
                                                             const int32_t* __restrict__ obs_ind, int n_obs,
                                                             double* __restrict__ logdens, double* __restrict__ states) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.B * a.D) return;
    const int blk = l / a.B, b = l - blk * a.B;
    const size_t B = (size_t)a.B;
    double Q[P][P], LR[P][P];
    load_block_consts<P>(a, blk, b, Q, LR);
    double bm[P], bL[P][P];
    fs_load_state<P>(a, a.N, blk, b, bm, bL);                               // terminal point (fenrir.py:186-188)
    double acc = 0.0;
    int i = n_obs - 1;
    auto observe = [&](double (&m)[P], double (&L)[P][P]) {
        double D[MO][P], y[MO], Vh[MO][MO];
        const size_t ib = (size_t)i * a.D + blk;
#pragma unroll
        for (int j = 0; j < MO; ++j) {
            y[j] = obs[ib * MO + j];
#pragma unroll
            for (int k = 0; k < P; ++k) D[j][k] = obs_w[(ib * MO + j) * P + k];
#pragma unroll
            for (int k = 0; k < MO; ++k) Vh[j][k] = obs_v[(ib * MO + j) * MO + k];
        }
        fs_observe<P, MO>(D, y, Vh, m, L, acc);
        --i;
    };
    constexpr int ITEM = 4 * P * P + 2 * P;
    auto keep = [&](int n, int off, const double (&m)[P], const double (&S)[P][P]) {
        double* o = states + (((size_t)n * a.D + blk) * ITEM + off) * B + b;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            o[(size_t)r * B] = m[r];
#pragma unroll
            for (int c = 0; c < P; ++c) o[(size_t)(P + r * P + c) * B] = S[r][c];
        }
    };
    if constexpr (STORE) keep(a.N, 0, bm, bL);
    if (i >= 0 && obs_ind[i] >= a.N) observe(bm, bL);                      // fenrir.py:189-209
    if constexpr (STORE) keep(a.N, P + P * P, bm, bL);
    for (int n = a.N - 1; n >= 0; --n) {
        double mf[P], Lf[P][P], mp[P], Lp[P][P], G[P][P], JL[P][P];
        fs_load_state<P>(a, n, blk, b, mf, Lf);
        sqrt_predict<P>(Q, LR, mf, Lf, mp, Lp);                             // pred[n+1] from filt[n] (square_root.py:56-57)
        sqrt_gain<P>(Q, Lf, Lp, G, JL);                                     // A = G (square_root.py:376-377), J L_f
        double bb[P], GR[P][P], Cc[P][P], AL[P][P], nm[P], nL[P][P];
        mm<P, P, P>(G, LR, GR);
        add_sqrt<P, P, P>(GR, JL, Cc);                                      // C^{1/2} (square_root.py:382-383)
        mv<P, P>(G, bm, nm);
        mm<P, P, P>(G, bL, AL);
        add_sqrt<P, P, P>(AL, Cc, nL);                                      // predict with (A, b, C^{1/2}) (square_root.py:56-57)
#pragma unroll
        for (int r = 0; r < P; ++r) {
            bb[r] = mf[r] - dot<P>(G[r], mp);                               // b = mu_f - G mu-   (square_root.py:378)
            bm[r] = nm[r] + bb[r];
#pragma unroll
            for (int c = 0; c < P; ++c) bL[r][c] = nL[r][c];
        }
        if constexpr (STORE) {
            keep(n, 0, bm, bL);
            double* o = states + (((size_t)n * a.D + blk) * ITEM + 2 * (P + P * P)) * B + b;
#pragma unroll
            for (int r = 0; r < P; ++r)
#pragma unroll
                for (int c = 0; c < P; ++c) {
                    o[(size_t)(r * P + c) * B] = G[r][c];
                    o[(size_t)(P * P + r * P + c) * B] = Cc[r][c];
                }
        }
        if (i >= 0 && obs_ind[i] == n) observe(bm, bL);                     // fenrir.py:155-170
        if constexpr (STORE) keep(n, P + P * P, bm, bL);
    }
    if (logdens) atomicAdd(&logdens[b], acc);
}

// fenrir.py:333-402 with square_root.smooth_mv (square_root.py:209-219): forward sweep over the backward filter's stored
// items; times 0 and 1 keep the backward filter's own estimates.
template <int P>
__global__ void __launch_bounds__(64) fenrir_smooth_sqrt_kernel(SolveArgs a, const double* __restrict__ states) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.B * a.D) return;
    const int blk = l / a.B, b = l - blk * a.B;
    const size_t B = (size_t)a.B;
    constexpr int ITEM = 4 * P * P + 2 * P;
    auto item = [&](int n, int off, double (&m)[P], double (&S)[P][P]) {
        const double* o = states + (((size_t)n * a.D + blk) * ITEM + off) * B + b;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            m[r] = o[(size_t)r * B];
#pragma unroll
            for (int c = 0; c < P; ++c) S[r][c] = o[(size_t)(P + r * P + c) * B];
        }
    };
    auto mat = [&](int n, int off, double (&S)[P][P]) {
        const double* o = states + (((size_t)n * a.D + blk) * ITEM + off) * B + b;
#pragma unroll
        for (int r = 0; r < P; ++r)
#pragma unroll
            for (int c = 0; c < P; ++c) S[r][c] = o[(size_t)(r * P + c) * B];
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
    double cm[P], cL[P][P];
    item(0, P + P * P, cm, cL);
    put(0, cm, cL);
    if (a.N < 1) return;
    item(1, P + P * P, cm, cL);
    put(1, cm, cL);
    for (int k = 0; k + 2 <= a.N; ++k) {
        double fm[P], fL[P][P], pm[P], pL[P][P], A[P][P], Ch[P][P], G[P][P], JL[P][P];
        item(k + 2, P + P * P, fm, fL);
        item(k + 1, 0, pm, pL);
        mat(k + 1, 2 * (P + P * P), A);
        mat(k + 1, 2 * (P + P * P) + P * P, Ch);
        sqrt_gain<P>(A, fL, pL, G, JL);                                     // square_root.py:170-175 with wgt_state = A
        double dm[P], gm[P], both[P][2 * P], GA[P][2 * P];
#pragma unroll
        for (int r = 0; r < P; ++r) {
            dm[r] = cm[r] - pm[r];
#pragma unroll
            for (int c = 0; c < P; ++c) { both[r][c] = cL[r][c]; both[r][P + c] = Ch[r][c]; }
        }
        mv<P, P>(G, dm, gm);
        mm<P, P, 2 * P>(G, both, GA);
        add_sqrt<P, 2 * P, P>(GA, JL, cL);                                  // square_root.py:217-218
#pragma unroll
        for (int r = 0; r < P; ++r) cm[r] = fm[r] + gm[r];                  // square_root.py:211-212
        put(k + 2, cm, cL);
    }
}

size_t fenrir_sqrt_item_doubles(int p) { return 4 * (size_t)p * p + 2 * (size_t)p; }

// states == nullptr: the log-density only (rk_fenrir_backward); else the stored backward filter + the smoothing sweep
int fenrir_sqrt_launch(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, const double* obs, const double* obs_w,
                       const double* obs_v, const int32_t* obs_ind, int n_obs, int n_bobs, double* logdens, double* states) {
    RK_REQUIRE(c->n_bstate >= 2 && c->n_bstate <= 8 && n_bobs >= 1 && n_bobs <= 3, RK_ERR_UNSUPPORTED,
               "fenrir, kalman_type=square-root: n_bstate in 2..8, n_bobs in 1..3 (got %d, %d)", c->n_bstate, n_bobs);
    const dim3 grid(div_up(a.B * a.D, 64)), block(64);
    {
        LaunchTimer t(h, "fenrir_bwd_sqrt_kernel");
#define RK_FS(P_, M_)                                                                                                      \
    if (c->n_bstate == P_ && n_bobs == M_) {                                                                             \
        if (states) hipLaunchKernelGGL((fenrir_bwd_sqrt_kernel<P_, true, M_>), grid, block, 0, h->stream, a, obs, obs_w, obs_v, obs_ind, n_obs, logdens, states); \
        else hipLaunchKernelGGL((fenrir_bwd_sqrt_kernel<P_, false, M_>), grid, block, 0, h->stream, a, obs, obs_w, obs_v, obs_ind, n_obs, logdens, states); \
    }
        RK_FS(2, 1) RK_FS(2, 2) RK_FS(2, 3) RK_FS(3, 1) RK_FS(3, 2) RK_FS(3, 3) RK_FS(4, 1) RK_FS(4, 2) RK_FS(4, 3)
        RK_FS(5, 1) RK_FS(5, 2) RK_FS(5, 3) RK_FS(6, 1) RK_FS(6, 2) RK_FS(6, 3)
        RK_FS(7, 1) RK_FS(7, 2) RK_FS(7, 3) RK_FS(8, 1) RK_FS(8, 2) RK_FS(8, 3)        // (functional: the stacks spill from n_bstate = 7 on)
#undef RK_FS
        t.stop();
    }
    RK_HIP(hipGetLastError());
    if (!states) return RK_OK;
    LaunchTimer t(h, "fenrir_smooth_sqrt_kernel");
    switch (c->n_bstate) {
        case 2: hipLaunchKernelGGL(fenrir_smooth_sqrt_kernel<2>, grid, block, 0, h->stream, a, states); break;
        case 3: hipLaunchKernelGGL(fenrir_smooth_sqrt_kernel<3>, grid, block, 0, h->stream, a, states); break;
        case 4: hipLaunchKernelGGL(fenrir_smooth_sqrt_kernel<4>, grid, block, 0, h->stream, a, states); break;
        case 5: hipLaunchKernelGGL(fenrir_smooth_sqrt_kernel<5>, grid, block, 0, h->stream, a, states); break;
        case 6: hipLaunchKernelGGL(fenrir_smooth_sqrt_kernel<6>, grid, block, 0, h->stream, a, states); break;
        case 7: hipLaunchKernelGGL(fenrir_smooth_sqrt_kernel<7>, grid, block, 0, h->stream, a, states); break;
        default: hipLaunchKernelGGL(fenrir_smooth_sqrt_kernel<8>, grid, block, 0, h->stream, a, states); break;
    }
    t.stop();
    RK_HIP(hipGetLastError());
    return RK_OK;
}

}  // namespace rk
