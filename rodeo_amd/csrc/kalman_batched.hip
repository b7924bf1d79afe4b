// Batched per-step operators: the nine functions of src/rodeo/kalmantv/standard.py (and square_root.py) for a batch
// of independent (state, model) tuples -- the reference's `jax.vmap(kalman_funs.<op>)` boundary
// (src/rodeo/solve.py:62-68,81-88,170-178,263-272).  One lane per batch item, runtime (n_state, n_meas) up to MAXN,
// batch-minor arrays (element e of item b at ptr[e * n + b]) so a wave's loads/stores are 512-B rows.
// These entry points serve unit parity and ad-hoc use; the timed solver path is the fused kernels of
// solve_small.hip / solve_dense.hip.  NumPy mirrors: oracle/kalman_ops.py, oracle/sqrt_ops.py.
#include "common.hpp"
#include "kalman_op_args.hpp"

namespace rk {

template <int MAXN>
struct Mat {
    double a[MAXN][MAXN];
};
template <int MAXN>
struct Vec {
    double a[MAXN];
};

template <int MAXN>
__device__ void ldm(Mat<MAXN>& M, const double* p, int r, int c, int n, int b) {
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) M.a[i][j] = p ? p[((size_t)i * c + j) * n + b] : 0.0;
}
template <int MAXN>
__device__ void ldv(Vec<MAXN>& v, const double* p, int r, int n, int b) {
    for (int i = 0; i < r; ++i) v.a[i] = p ? p[(size_t)i * n + b] : 0.0;
}
template <int MAXN>
__device__ void stm(const Mat<MAXN>& M, double* p, int r, int c, int n, int b) {
    if (!p) return;
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) p[((size_t)i * c + j) * n + b] = M.a[i][j];
}
template <int MAXN>
__device__ void stv(const Vec<MAXN>& v, double* p, int r, int n, int b) {
    if (!p) return;
    for (int i = 0; i < r; ++i) p[(size_t)i * n + b] = v.a[i];
}

// C (r x c) = A (r x k) B (k x c)
template <int MAXN>
__device__ void gemm(const Mat<MAXN>& A, const Mat<MAXN>& B, Mat<MAXN>& C, int r, int k, int c) {
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s = fma(A.a[i][l], B.a[l][j], s);
            C.a[i][j] = s;
        }
}
// C (r x c) = A (r x k) B^T, B is (c x k)
template <int MAXN>
__device__ void gemm_nt(const Mat<MAXN>& A, const Mat<MAXN>& B, Mat<MAXN>& C, int r, int k, int c) {
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s = fma(A.a[i][l], B.a[j][l], s);
            C.a[i][j] = s;
        }
}
template <int MAXN>
__device__ void gemv(const Mat<MAXN>& A, const Vec<MAXN>& x, Vec<MAXN>& y, int r, int k) {
    for (int i = 0; i < r; ++i) {
        double s = 0.0;
        for (int l = 0; l < k; ++l) s = fma(A.a[i][l], x.a[l], s);
        y.a[i] = s;
    }
}

// X = A^{-1} B (A k x k, B k x nr) by LU with partial pivoting (utils.py:119); A, B destroyed, B <- X.
template <int MAXN>
__device__ void lu_solve_rt(Mat<MAXN>& A, Mat<MAXN>& B, int k, int nr) {
    for (int c = 0; c < k; ++c) {
        int piv = c;
        double best = fabs(A.a[c][c]);
        for (int i = c + 1; i < k; ++i) {
            const double v = fabs(A.a[i][c]);
            if (v > best) { best = v; piv = i; }
        }
        if (piv != c) {
            for (int j = 0; j < k; ++j) { const double t = A.a[c][j]; A.a[c][j] = A.a[piv][j]; A.a[piv][j] = t; }
            for (int j = 0; j < nr; ++j) { const double t = B.a[c][j]; B.a[c][j] = B.a[piv][j]; B.a[piv][j] = t; }
        }
        const double r = 1.0 / A.a[c][c];
        for (int i = c + 1; i < k; ++i) {
            const double l = A.a[i][c] * r;
            for (int j = c + 1; j < k; ++j) A.a[i][j] = fma(-l, A.a[c][j], A.a[i][j]);
            for (int j = 0; j < nr; ++j) B.a[i][j] = fma(-l, B.a[c][j], B.a[i][j]);
        }
    }
    for (int c = k - 1; c >= 0; --c)
        for (int j = 0; j < nr; ++j) {
            double s = B.a[c][j];
            for (int i = c + 1; i < k; ++i) s = fma(-A.a[c][i], B.a[i][j], s);
            B.a[c][j] = s / A.a[c][c];
        }
}

// ---- square-root helpers (src/rodeo/utils.py:10-24, kalmantv/square_root.py) ---------------------------------
// add_sqrt: lower factor F (n x n) with F F^T = A A^T + B B^T for A (n x ka), B (n x kb): Householder QR of the
// stacked [A^T; B^T] ((ka+kb) x n), F = R^T.  Like jnp.linalg.qr there is no sign normalisation of R's diagonal.
// `stack` is caller-provided scratch of (2*MAXN + MAXN) rows... we bound ka + kb <= 3*MAXN.
template <int MAXN>
struct Tall {
    double a[3 * MAXN][MAXN];
};
template <int MAXN>
__device__ void householder_r(Tall<MAXN>& S, int rows, int n, Mat<MAXN>& F) {
    for (int c = 0; c < n; ++c) {
        double nrm = 0.0;
        for (int i = c; i < rows; ++i) nrm = fma(S.a[i][c], S.a[i][c], nrm);
        nrm = sqrt(nrm);
        if (nrm != 0.0 && c < rows) {
            const double alpha = S.a[c][c] >= 0.0 ? -nrm : nrm;      // LAPACK dlarfg sign choice: beta = -sign(alpha) norm
            const double v0 = S.a[c][c] - alpha;
            double vnorm2 = v0 * v0;
            for (int i = c + 1; i < rows; ++i) vnorm2 = fma(S.a[i][c], S.a[i][c], vnorm2);
            if (vnorm2 != 0.0) {
                const double tau = 2.0 / vnorm2;
                for (int j = c + 1; j < n; ++j) {
                    double d = v0 * S.a[c][j];
                    for (int i = c + 1; i < rows; ++i) d = fma(S.a[i][c], S.a[i][j], d);
                    d *= tau;
                    S.a[c][j] = fma(-d, v0, S.a[c][j]);
                    for (int i = c + 1; i < rows; ++i) S.a[i][j] = fma(-d, S.a[i][c], S.a[i][j]);
                }
            }
            S.a[c][c] = alpha;
            for (int i = c + 1; i < rows; ++i) S.a[i][c] = 0.0;
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) F.a[i][j] = (j <= i && j < rows) ? S.a[j][i] : 0.0;
}
template <int MAXN>
__device__ void add_sqrt(const Mat<MAXN>& A, int ka, const Mat<MAXN>& B, int kb, int n, Mat<MAXN>& F,
                         Tall<MAXN>& S, const Mat<MAXN>* A2 = nullptr, int ka2 = 0) {
    // rows: A^T (ka), optional A2^T (ka2), B^T (kb)
    int r = 0;
    for (int i = 0; i < ka; ++i, ++r)
        for (int j = 0; j < n; ++j) S.a[r][j] = A.a[j][i];
    if (A2)
        for (int i = 0; i < ka2; ++i, ++r)
            for (int j = 0; j < n; ++j) S.a[r][j] = A2->a[j][i];
    for (int i = 0; i < kb; ++i, ++r)
        for (int j = 0; j < n; ++j) S.a[r][j] = B.a[j][i];
    householder_r<MAXN>(S, r, n, F);
}
// solve L X = B (lower) or U X = B (upper, U = L^T given as L); B (k x nr) overwritten
template <int MAXN>
__device__ void trsm_lower(const Mat<MAXN>& L, Mat<MAXN>& B, int k, int nr) {
    for (int j = 0; j < nr; ++j)
        for (int i = 0; i < k; ++i) {
            double s = B.a[i][j];
            for (int l = 0; l < i; ++l) s = fma(-L.a[i][l], B.a[l][j], s);
            B.a[i][j] = s / L.a[i][i];
        }
}
template <int MAXN>
__device__ void trsm_upper_of_lower_t(const Mat<MAXN>& L, Mat<MAXN>& B, int k, int nr) {   // (L^T) X = B
    for (int j = 0; j < nr; ++j)
        for (int i = k - 1; i >= 0; --i) {
            double s = B.a[i][j];
            for (int l = i + 1; l < k; ++l) s = fma(-L.a[l][i], B.a[l][j], s);
            B.a[i][j] = s / L.a[i][i];
        }
}

// ---- the ops ------------------------------------------------------------------------------------------------------
template <int MAXN>
__device__ void op_predict(const OpArgs& a, int b, Vec<MAXN>& mp, Mat<MAXN>& Vp, Tall<MAXN>& scratch) {
    const int p = a.p, n = a.n;
    Mat<MAXN> Q, V, R, T1;
    Vec<MAXN> m, c;
    ldm(Q, a.wgt_state, p, p, n, b); ldm(V, a.var_state_past, p, p, n, b); ldm(R, a.var_state, p, p, n, b);
    ldv(m, a.mean_state_past, p, n, b); ldv(c, a.mean_state, p, n, b);
    gemv(Q, m, mp, p, p);
    for (int i = 0; i < p; ++i) mp.a[i] += c.a[i];                                    // standard.py:57
    gemm(Q, V, T1, p, p, p);
    if (!a.sqrt_form) {
        gemm_nt(T1, Q, Vp, p, p, p);                                                  // standard.py:58-59
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < p; ++j) Vp.a[i][j] += R.a[i][j];
    } else {
        add_sqrt<MAXN>(T1, p, R, p, p, Vp, scratch);                                  // square_root.py:57
    }
}

template <int MAXN>
__device__ void op_update(const OpArgs& a, int b, const Vec<MAXN>& mp, const Mat<MAXN>& Vp, Vec<MAXN>& mf,
                          Mat<MAXN>& Vf, Tall<MAXN>& scratch) {
    const int p = a.p, m = a.m, n = a.n;
    Mat<MAXN> W, V, WS, S, X, K;
    Vec<MAXN> x, am, yhat, innov, corr;
    ldm(W, a.wgt_meas, m, p, n, b); ldm(V, a.var_meas, m, m, n, b);
    ldv(x, a.x_meas, m, n, b); ldv(am, a.mean_meas, m, n, b);
    gemv(W, mp, yhat, m, p);
    for (int i = 0; i < m; ++i) { yhat.a[i] += am.a[i]; innov.a[i] = x.a[i] - yhat.a[i]; }   // standard.py:93
    gemm(W, Vp, WS, m, p, p);                                                                 // W Sigma  /  W L
    if (!a.sqrt_form) {
        gemm_nt(WS, W, S, m, p, m);                                                           // standard.py:95-96
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) S.a[i][j] += V.a[i][j];
        // K = solve(S, (Sigma W^T)^T)^T  (standard.py:97-98):  X = S^{-1} (W Sigma^T)  [m x p],  K = X^T
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < p; ++j) {
                double s = 0.0;                                                               // (Sigma W^T)[j][i]
                for (int l = 0; l < p; ++l) s = fma(Vp.a[j][l], W.a[i][l], s);
                X.a[i][j] = s;
            }
        lu_solve_rt<MAXN>(S, X, m, p);
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < m; ++j) K.a[i][j] = X.a[j][i];
        gemv(K, innov, corr, p, m);
        for (int i = 0; i < p; ++i) mf.a[i] = mp.a[i] + corr.a[i];                            // standard.py:99-100
        Mat<MAXN> KWS;
        gemm(K, WS, KWS, p, m, p);
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < p; ++j) Vf.a[i][j] = Vp.a[i][j] - KWS.a[i][j];                // standard.py:101-102
    } else {
        // square_root.py:90-99
        Mat<MAXN> Ls, I1, I2, Var;
        add_sqrt<MAXN>(WS, p, V, m, m, Ls, scratch);                  // factor of W L L^T W^T + V V^T
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < p; ++j) I1.a[i][j] = W.a[i][j];
        trsm_lower<MAXN>(Ls, I1, m, p);                               // Ls^{-1} W
        gemm(I1, Vp, I2, m, p, p);
        gemm_nt(I2, Vp, I1, m, p, p);                                 // ... L L^T
        trsm_upper_of_lower_t<MAXN>(Ls, I1, m, p);                    // Ls^{-T} (...)  -> K^T  (m x p)
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < m; ++j) K.a[i][j] = I1.a[j][i];
        gemv(K, innov, corr, p, m);
        for (int i = 0; i < p; ++i) mf.a[i] = mp.a[i] + corr.a[i];
        Mat<MAXN> KW, A1, KV;
        gemm(K, W, KW, p, m, p);
        gemm(KW, Vp, A1, p, p, p);
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < p; ++j) A1.a[i][j] = Vp.a[i][j] - A1.a[i][j];     // (I - K W) L
        gemm(K, V, KV, p, m, m);
        add_sqrt<MAXN>(A1, p, KV, m, p, Vf, scratch);
        (void)Var;
    }
}

// _smooth: standard.py:175-176 / square_root.py:170-175.  Returns T (standard only) and G.
template <int MAXN>
__device__ void op_gain(const OpArgs& a, const Mat<MAXN>& Q, const Mat<MAXN>& Vf, const Mat<MAXN>& Vp,
                        Mat<MAXN>& T, Mat<MAXN>& G) {
    const int p = a.p;
    if (!a.sqrt_form) {
        gemm_nt(Vf, Q, T, p, p, p);
        Mat<MAXN> A, X;
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < p; ++j) { A.a[i][j] = Vp.a[i][j]; X.a[i][j] = T.a[j][i]; }
        lu_solve_rt<MAXN>(A, X, p, p);
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < p; ++j) G.a[i][j] = X.a[j][i];
    } else {
        Mat<MAXN> Sf, I1, I2;
        gemm_nt(Vf, Vf, Sf, p, p, p);                                 // L_f L_f^T
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < p; ++j) I1.a[i][j] = Q.a[i][j];
        trsm_lower<MAXN>(Vp, I1, p, p);                               // L_p^{-1} Q
        gemm(I1, Sf, I2, p, p, p);
        trsm_upper_of_lower_t<MAXN>(Vp, I2, p, p);                    // L_p^{-T} (...)
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < p; ++j) G.a[i][j] = I2.a[j][i];
    }
}

template <int MAXN>
__global__ void __launch_bounds__(64) kalman_op_kernel(OpArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.n) return;
    const int p = a.p, m = a.m, n = a.n;
    Tall<MAXN> scratch;
    if (a.op == OP_PREDICT || a.op == OP_FILTER) {
        Vec<MAXN> mp; Mat<MAXN> Vp;
        op_predict<MAXN>(a, b, mp, Vp, scratch);
        stv(mp, a.o_mean_pred, p, n, b); stm(Vp, a.o_var_pred, p, p, n, b);
        if (a.op == OP_FILTER) {
            Vec<MAXN> mf; Mat<MAXN> Vf;
            op_update<MAXN>(a, b, mp, Vp, mf, Vf, scratch);
            stv(mf, a.o_mean_filt, p, n, b); stm(Vf, a.o_var_filt, p, p, n, b);
        }
        return;
    }
    if (a.op == OP_UPDATE || a.op == OP_FORECAST) {
        Vec<MAXN> mp; Mat<MAXN> Vp;
        ldv(mp, a.mean_state_pred, p, n, b); ldm(Vp, a.var_state_pred, p, p, n, b);
        if (a.op == OP_UPDATE) {
            Vec<MAXN> mf; Mat<MAXN> Vf;
            op_update<MAXN>(a, b, mp, Vp, mf, Vf, scratch);
            stv(mf, a.o_mean_filt, p, n, b); stm(Vf, a.o_var_filt, p, p, n, b);
        } else {
            // standard.py:333-335 / square_root.py:342-345 (the latter returns the FULL variance)
            Mat<MAXN> W, V, WS, S; Vec<MAXN> am, yhat;
            ldm(W, a.wgt_meas, m, p, n, b); ldm(V, a.var_meas, m, m, n, b); ldv(am, a.mean_meas, m, n, b);
            gemv(W, mp, yhat, m, p);
            for (int i = 0; i < m; ++i) yhat.a[i] += am.a[i];
            gemm(W, Vp, WS, m, p, p);
            if (!a.sqrt_form) {
                gemm_nt(WS, W, S, m, p, m);
                for (int i = 0; i < m; ++i)
                    for (int j = 0; j < m; ++j) S.a[i][j] += V.a[i][j];
            } else {
                Mat<MAXN> F;
                add_sqrt<MAXN>(WS, p, V, m, m, F, scratch);
                gemm_nt(F, F, S, m, m, m);
            }
            stv(yhat, a.o_mean_fore, m, n, b); stm(S, a.o_var_fore, m, m, n, b);
        }
        return;
    }
    // smoothers
    Mat<MAXN> Q, Vf, Vp, T, G, R;
    Vec<MAXN> mf, mp;
    ldm(Q, a.wgt_state, p, p, n, b); ldm(Vf, a.var_state_filt, p, p, n, b); ldm(Vp, a.var_state_pred, p, p, n, b);
    ldv(mf, a.mean_state_filt, p, n, b); ldv(mp, a.mean_state_pred, p, n, b);
    if (a.sqrt_form) ldm(R, a.var_state, p, p, n, b);
    op_gain<MAXN>(a, Q, Vf, Vp, T, G);
    Mat<MAXN> JL;       // sqrt form: (I - G Q) L_f   (square_root.py:214-215)
    if (a.sqrt_form) {
        Mat<MAXN> GQ;
        gemm(G, Q, GQ, p, p, p);
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < p; ++j) GQ.a[i][j] = (i == j ? 1.0 : 0.0) - GQ.a[i][j];
        gemm(GQ, Vf, JL, p, p, p);
    }
    if (a.op == OP_SMOOTH_MV || a.op == OP_SMOOTH) {
        Vec<MAXN> mn, dm, gm, ms; Mat<MAXN> Vn, Vs;
        ldv(mn, a.mean_state_next, p, n, b); ldm(Vn, a.var_state_next, p, p, n, b);
        for (int i = 0; i < p; ++i) dm.a[i] = mn.a[i] - mp.a[i];
        gemv(G, dm, gm, p, p);
        for (int i = 0; i < p; ++i) ms.a[i] = mf.a[i] + gm.a[i];                              // standard.py:213-214
        if (!a.sqrt_form) {
            Mat<MAXN> D, GD, GDG;
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) D.a[i][j] = Vn.a[i][j] - Vp.a[i][j];
            gemm(G, D, GD, p, p, p);
            gemm_nt(GD, G, GDG, p, p, p);
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) Vs.a[i][j] = Vf.a[i][j] + GDG.a[i][j];            // standard.py:215-216
        } else {
            Mat<MAXN> GN, GR;
            gemm(G, Vn, GN, p, p, p);
            gemm(G, R, GR, p, p, p);
            add_sqrt<MAXN>(GN, p, JL, p, p, Vs, scratch, &GR, p);                             // square_root.py:217-218
        }
        stv(ms, a.o_mean_smooth, p, n, b); stm(Vs, a.o_var_smooth, p, p, n, b);
    }
    if (a.op == OP_SMOOTH_SIM || a.op == OP_SMOOTH || a.op == OP_SMOOTH_COND) {
        Mat<MAXN> Vsim;
        if (!a.sqrt_form) {
            Mat<MAXN> GT;
            gemm_nt(G, T, GT, p, p, p);
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) Vsim.a[i][j] = Vf.a[i][j] - GT.a[i][j];           // standard.py:253-254,370
        } else {
            Mat<MAXN> GR;
            gemm(G, R, GR, p, p, p);
            add_sqrt<MAXN>(GR, p, JL, p, p, Vsim, scratch);                                   // square_root.py:259-260
        }
        if (a.op == OP_SMOOTH_COND) {
            Vec<MAXN> gm, mc;
            gemv(G, mp, gm, p, p);
            for (int i = 0; i < p; ++i) mc.a[i] = mf.a[i] - gm.a[i];                          // standard.py:369
            stm(G, a.o_wgt_cond, p, p, n, b); stv(mc, a.o_mean_cond, p, n, b); stm(Vsim, a.o_var_cond, p, p, n, b);
        } else {
            Vec<MAXN> xn, dm, gm, msim;
            ldv(xn, a.x_state_next, p, n, b);
            for (int i = 0; i < p; ++i) dm.a[i] = xn.a[i] - mp.a[i];
            gemv(G, dm, gm, p, p);
            for (int i = 0; i < p; ++i) msim.a[i] = mf.a[i] + gm.a[i];                        // standard.py:251-252
            stv(msim, a.o_mean_sim, p, n, b); stm(Vsim, a.o_var_sim, p, p, n, b);
        }
    }
}

static int launch_op(rk_handle h, const rk_op_cfg* c, OpArgs& a, int op) {
    RK_REQUIRE(h && c, RK_ERR_INVALID, "null handle / cfg");
    RK_REQUIRE(c->n >= 1 && c->n_state >= 1 && c->n_meas >= 0, RK_ERR_INVALID, "bad dims n=%d n_state=%d n_meas=%d",
               c->n, c->n_state, c->n_meas);
    RK_REQUIRE(c->kalman_type == RK_KALMAN_STANDARD || c->kalman_type == RK_KALMAN_SQRT, RK_ERR_UNSUPPORTED,
               "unknown kalman_type %d", c->kalman_type);
    RK_REQUIRE(c->n_meas <= c->n_state || op >= OP_SMOOTH_MV, RK_ERR_UNSUPPORTED, "n_meas > n_state is not supported");
    a.n = c->n; a.p = c->n_state; a.m = c->n_meas; a.op = op; a.sqrt_form = c->kalman_type == RK_KALMAN_SQRT;
    RK_HIP(hipSetDevice(h->device));
    const dim3 grid(div_up(a.n, 64)), block(64);
    if (a.p <= 8) hipLaunchKernelGGL((kalman_op_kernel<8>), grid, block, 0, h->stream, a);
    else if (a.p <= 16) hipLaunchKernelGGL((kalman_op_kernel<16>), grid, block, 0, h->stream, a);
    else return dense_op_launch(h, a);       // one workgroup per item on the dense building blocks (solve_dense_ops.hpp)
    RK_HIP(hipGetLastError());
    return RK_OK;
}

}  // namespace rk

using namespace rk;

extern "C" {

int rk_kalman_predict_batched(rk_handle h, const rk_op_cfg* c, const double* mean_state_past,
                              const double* var_state_past, const double* mean_state, const double* wgt_state,
                              const double* var_state, double* mean_state_pred, double* var_state_pred) {
    RK_REQUIRE(mean_state_past && var_state_past && wgt_state && var_state && mean_state_pred && var_state_pred,
               RK_ERR_INVALID, "rk_kalman_predict_batched: null array");
    OpArgs a{};
    a.mean_state_past = mean_state_past; a.var_state_past = var_state_past; a.mean_state = mean_state;
    a.wgt_state = wgt_state; a.var_state = var_state; a.o_mean_pred = mean_state_pred; a.o_var_pred = var_state_pred;
    return launch_op(h, c, a, OP_PREDICT);
}

int rk_kalman_update_batched(rk_handle h, const rk_op_cfg* c, const double* mean_state_pred,
                             const double* var_state_pred, const double* x_meas, const double* mean_meas,
                             const double* wgt_meas, const double* var_meas, double* mean_state_filt,
                             double* var_state_filt) {
    RK_REQUIRE(mean_state_pred && var_state_pred && wgt_meas && mean_state_filt && var_state_filt, RK_ERR_INVALID,
               "rk_kalman_update_batched: null array");
    OpArgs a{};
    a.mean_state_pred = mean_state_pred; a.var_state_pred = var_state_pred; a.x_meas = x_meas;
    a.mean_meas = mean_meas; a.wgt_meas = wgt_meas; a.var_meas = var_meas;
    a.o_mean_filt = mean_state_filt; a.o_var_filt = var_state_filt;
    return launch_op(h, c, a, OP_UPDATE);
}

int rk_kalman_filter_batched(rk_handle h, const rk_op_cfg* c, const double* mean_state_past,
                             const double* var_state_past, const double* mean_state, const double* wgt_state,
                             const double* var_state, const double* x_meas, const double* mean_meas,
                             const double* wgt_meas, const double* var_meas, double* mean_state_pred,
                             double* var_state_pred, double* mean_state_filt, double* var_state_filt) {
    RK_REQUIRE(mean_state_past && var_state_past && wgt_state && var_state && wgt_meas && mean_state_pred &&
                   var_state_pred && mean_state_filt && var_state_filt,
               RK_ERR_INVALID, "rk_kalman_filter_batched: null array");
    OpArgs a{};
    a.mean_state_past = mean_state_past; a.var_state_past = var_state_past; a.mean_state = mean_state;
    a.wgt_state = wgt_state; a.var_state = var_state; a.x_meas = x_meas; a.mean_meas = mean_meas;
    a.wgt_meas = wgt_meas; a.var_meas = var_meas;
    a.o_mean_pred = mean_state_pred; a.o_var_pred = var_state_pred;
    a.o_mean_filt = mean_state_filt; a.o_var_filt = var_state_filt;
    return launch_op(h, c, a, OP_FILTER);
}

static int smooth_inputs(OpArgs& a, const rk_op_cfg* c, const double* mean_state_filt, const double* var_state_filt,
                         const double* mean_state_pred, const double* var_state_pred, const double* wgt_state,
                         const double* var_state) {
    RK_REQUIRE(c, RK_ERR_INVALID, "null cfg");
    RK_REQUIRE(mean_state_filt && var_state_filt && mean_state_pred && var_state_pred && wgt_state, RK_ERR_INVALID,
               "smoother: null array");
    // square_root.py:185,228: the square-root smoothers REQUIRE var_state; the standard ones ignore it
    RK_REQUIRE(c->kalman_type != RK_KALMAN_SQRT || var_state, RK_ERR_INVALID,
               "square-root smoothers need var_state (the factor of R)");
    a.mean_state_filt = mean_state_filt; a.var_state_filt = var_state_filt; a.mean_state_pred = mean_state_pred;
    a.var_state_pred = var_state_pred; a.wgt_state = wgt_state; a.var_state = var_state;
    return RK_OK;
}

int rk_kalman_smooth_mv_batched(rk_handle h, const rk_op_cfg* c, const double* mean_state_next,
                                const double* var_state_next, const double* mean_state_filt,
                                const double* var_state_filt, const double* mean_state_pred,
                                const double* var_state_pred, const double* wgt_state, const double* var_state,
                                double* mean_state_smooth, double* var_state_smooth) {
    OpArgs a{};
    int rc = smooth_inputs(a, c, mean_state_filt, var_state_filt, mean_state_pred, var_state_pred, wgt_state, var_state);
    if (rc) return rc;
    RK_REQUIRE(mean_state_next && var_state_next && mean_state_smooth && var_state_smooth, RK_ERR_INVALID,
               "rk_kalman_smooth_mv_batched: null array");
    a.mean_state_next = mean_state_next; a.var_state_next = var_state_next;
    a.o_mean_smooth = mean_state_smooth; a.o_var_smooth = var_state_smooth;
    return launch_op(h, c, a, OP_SMOOTH_MV);
}

int rk_kalman_smooth_sim_batched(rk_handle h, const rk_op_cfg* c, const double* x_state_next,
                                 const double* mean_state_filt, const double* var_state_filt,
                                 const double* mean_state_pred, const double* var_state_pred, const double* wgt_state,
                                 const double* var_state, double* mean_state_sim, double* var_state_sim) {
    OpArgs a{};
    int rc = smooth_inputs(a, c, mean_state_filt, var_state_filt, mean_state_pred, var_state_pred, wgt_state, var_state);
    if (rc) return rc;
    RK_REQUIRE(x_state_next && mean_state_sim && var_state_sim, RK_ERR_INVALID, "rk_kalman_smooth_sim_batched: null array");
    a.x_state_next = x_state_next; a.o_mean_sim = mean_state_sim; a.o_var_sim = var_state_sim;
    return launch_op(h, c, a, OP_SMOOTH_SIM);
}

int rk_kalman_smooth_batched(rk_handle h, const rk_op_cfg* c, const double* x_state_next,
                             const double* mean_state_next, const double* var_state_next,
                             const double* mean_state_filt, const double* var_state_filt,
                             const double* mean_state_pred, const double* var_state_pred, const double* wgt_state,
                             const double* var_state, double* mean_state_sim, double* var_state_sim,
                             double* mean_state_smooth, double* var_state_smooth) {
    OpArgs a{};
    int rc = smooth_inputs(a, c, mean_state_filt, var_state_filt, mean_state_pred, var_state_pred, wgt_state, var_state);
    if (rc) return rc;
    RK_REQUIRE(x_state_next && mean_state_next && var_state_next && mean_state_sim && var_state_sim &&
                   mean_state_smooth && var_state_smooth,
               RK_ERR_INVALID, "rk_kalman_smooth_batched: null array");
    a.x_state_next = x_state_next; a.mean_state_next = mean_state_next; a.var_state_next = var_state_next;
    a.o_mean_sim = mean_state_sim; a.o_var_sim = var_state_sim;
    a.o_mean_smooth = mean_state_smooth; a.o_var_smooth = var_state_smooth;
    return launch_op(h, c, a, OP_SMOOTH);
}

int rk_kalman_forecast_batched(rk_handle h, const rk_op_cfg* c, const double* mean_state_pred,
                               const double* var_state_pred, const double* mean_meas, const double* wgt_meas,
                               const double* var_meas, double* mean_fore, double* var_fore) {
    RK_REQUIRE(mean_state_pred && var_state_pred && wgt_meas && mean_fore && var_fore, RK_ERR_INVALID,
               "rk_kalman_forecast_batched: null array");
    OpArgs a{};
    a.mean_state_pred = mean_state_pred; a.var_state_pred = var_state_pred; a.mean_meas = mean_meas;
    a.wgt_meas = wgt_meas; a.var_meas = var_meas; a.o_mean_fore = mean_fore; a.o_var_fore = var_fore;
    return launch_op(h, c, a, OP_FORECAST);
}

int rk_kalman_smooth_cond_batched(rk_handle h, const rk_op_cfg* c, const double* mean_state_filt,
                                  const double* var_state_filt, const double* mean_state_pred,
                                  const double* var_state_pred, const double* wgt_state, const double* var_state,
                                  double* wgt_state_cond, double* mean_state_cond, double* var_state_cond) {
    OpArgs a{};
    int rc = smooth_inputs(a, c, mean_state_filt, var_state_filt, mean_state_pred, var_state_pred, wgt_state, var_state);
    if (rc) return rc;
    RK_REQUIRE(wgt_state_cond && mean_state_cond && var_state_cond, RK_ERR_INVALID,
               "rk_kalman_smooth_cond_batched: null array");
    a.o_wgt_cond = wgt_state_cond; a.o_mean_cond = mean_state_cond; a.o_var_cond = var_state_cond;
    return launch_op(h, c, a, OP_SMOOTH_COND);
}

}  // extern "C"
