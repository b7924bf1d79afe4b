// The nine per-step operators of src/rodeo/kalmantv/standard.py:31-371 and square_root.py:30-385 for blocks LARGER than the
// lane-per-item kernels take (n_state > 16, e.g. BASELINE config 5's 160 x 160 state with 32 measurements): one 512-thread
// workgroup per batch item, every product / LU / QR / triangular solve from the dense solver's building blocks
// (solve_dense.hip, solve_dense_sqrt.hpp).  The reference's operators are size-agnostic (standard.py:31-60 takes any
// n_state); this is the large-block half of `rk_kalman_*_batched`.  Unit-parity path (tests/test_gpu_kalman_ops.py), not timed.
// Arrays are batch-minor like the small-block operators (element e of item b at ptr[e * n + b]); an item's operands are
// gathered into a contiguous per-item workspace, the results scattered back.
#pragma once
#include "kalman_op_args.hpp"

namespace rk {

__device__ __forceinline__ void op_gather(double* dst, const double* src, int count, int n, int b) {
    for (int e = threadIdx.x; e < count; e += DT) dst[e] = src ? src[(size_t)e * n + b] : 0.0;
}
__device__ __forceinline__ void op_scatter(double* dst, const double* src, int count, int n, int b) {
    if (dst) for (int e = threadIdx.x; e < count; e += DT) dst[(size_t)e * n + b] = src[e];
}
// y = A^T x through LDS partial sums (A is k x r row-major: the G^T of the smoothers), y_i += base_i
__device__ __forceinline__ void op_gemv_t(double* y, const double* At, const double* x, const double* base, double sign, int r, int k) {
    for (int i = threadIdx.x; i < r; i += DT) {
        double s = 0.0;
        for (int l = 0; l < k; ++l) s = fma(At[(size_t)l * r + i], x[l], s);
        y[i] = fma(sign, s, base ? base[i] : 0.0);
    }
    __syncthreads();
}

size_t dense_op_ws_doubles(int p, int m) {
    const size_t pp = (size_t)p * p, mp = (size_t)m * p, mm = (size_t)m * m, v = (size_t)(p > m ? p : m);
    return 13 * pp + 6 * mp + 3 * mm + ((size_t)p + m) * m + 12 * v + 64;
}

__global__ void __launch_bounds__(DT) dense_op_kernel(OpArgs a, double* ws, size_t ws_stride) {
    const int b = blockIdx.x, p = a.p, m = a.m, n = a.n;
    const size_t pp = (size_t)p * p, mp = (size_t)m * p, mm = (size_t)m * m;
    const int vmax = p > m ? p : m;
    double* cur = ws + (size_t)b * ws_stride;
    auto take = [&](size_t cnt) { double* r = cur; cur += cnt; return r; };
    double* const S = take(3 * pp);                                      // stacked QR input (square-root form)
    double *const Q = take(pp), *const R = take(pp), *const Va = take(pp), *const Vb = take(pp), *const Vc = take(pp);
    double *const A1 = take(pp), *const A2 = take(pp), *const A3 = take(pp), *const A4 = take(pp), *const Vo = take(pp);
    double *const W = take(mp), *const WS = take(mp), *const X = take(mp), *const W2 = take(mp), *const Wt2 = take(2 * mp);
    double *const V = take(mm), *const Sm = take(((size_t)p + m) * m), *const Smm = take(mm), *const Vh = take(mm);
    double *const v0 = take(vmax), *const v1 = take(vmax), *const v2 = take(vmax), *const v3 = take(vmax), *const v4 = take(vmax),
           *const v5 = take(vmax), *const v6 = take(vmax), *const v7 = take(vmax);
    int* const piv = (int*)take((size_t)(vmax + 1) / 2 + 1);
    (void)Wt2; (void)Vh;
    const bool sq = a.sqrt_form != 0;

    // ---- predict (standard.py:57-59 / square_root.py:56-57): (mean_past, var_past) -> mp = v4, Vp = Vb ----
    auto predict = [&]() {
        op_gather(Q, a.wgt_state, (int)pp, n, b); op_gather(Va, a.var_state_past, (int)pp, n, b);
        op_gather(R, a.var_state, (int)pp, n, b);
        op_gather(v0, a.mean_state_past, p, n, b); op_gather(v1, a.mean_state, p, n, b);
        __syncthreads();
        wg_gemv<false>(v4, Q, p, v0, p, p, v1, 1.0, 1.0);
        if (!sq) {
            wg_gemm(gemm_op(A1, p, Q, p, false, Va, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
            wg_gemm(gemm_op(Vb, p, A1, p, false, Q, p, true, p, p, p, R, p, 1.0, 1.0));
        } else {
            wg_gemm(gemm_op(S, p, Va, p, true, Q, p, true, p, p, p, nullptr, 0, 0.0, 1.0));          // (Q L)^T
            wg_transpose(S + pp, p, R, p, p, p, 0);
            wg_qr_r(S, p, 2 * p, p);
            wg_transpose(Vb, p, S, p, p, p, 1);
        }
        op_scatter(a.o_mean_pred, v4, p, n, b); op_scatter(a.o_var_pred, Vb, (int)pp, n, b);
        __syncthreads();
    };
    // ---- update / forecast (standard.py:93-102, 333-335 / square_root.py:88-99, 342-345) from (v4, Vb) ----
    auto update = [&](bool fore_only) {
        op_gather(W, a.wgt_meas, (int)mp, n, b); op_gather(V, a.var_meas, (int)mm, n, b);
        op_gather(v0, a.x_meas, m, n, b); op_gather(v1, a.mean_meas, m, n, b);
        __syncthreads();
        wg_gemv<false>(v2, W, p, v4, m, p, v1, 1.0, 1.0);                                          // yhat
        wg_gemm(gemm_op(WS, p, W, p, false, Vb, p, false, m, p, p, nullptr, 0, 0.0, 1.0));          // W Sigma- / W L-
        if (!sq) {
            wg_gemm(gemm_op(Smm, m, WS, p, false, W, p, true, m, m, p, V, m, 1.0, 1.0));           // S
            if (fore_only) { op_scatter(a.o_mean_fore, v2, m, n, b); op_scatter(a.o_var_fore, Smm, (int)mm, n, b); return; }
            wg_gemm(gemm_op(X, p, W, p, false, Vb, p, true, m, p, p, nullptr, 0, 0.0, 1.0));        // (Sigma- W^T)^T
            wg_lu_solve(Smm, m, X, p, m, p, piv);                                                   // K^T (utils.py:119)
        } else {
            wg_transpose(Sm, m, WS, p, p, m, 0);
            wg_transpose(Sm + (size_t)p * m, m, V, m, m, m, 0);
            wg_qr_r(Sm, m, p + m, m);                                                               // L_m^T
            if (fore_only) {
                wg_transpose(Smm, m, Sm, m, m, m, 1);                                               // L_m, clean
                wg_gemm(gemm_op(A1, m, Smm, m, false, Smm, m, true, m, m, m, nullptr, 0, 0.0, 1.0)); // L_m L_m^T (square_root.py:344)
                op_scatter(a.o_mean_fore, v2, m, n, b); op_scatter(a.o_var_fore, A1, (int)mm, n, b);
                return;
            }
            for (int e = threadIdx.x; e < (int)mp; e += DT) X[e] = W[e];
            __syncthreads();
            wg_tri_solve<true>(Sm, m, 1, X, p, m, p);
            wg_gemm(gemm_op(W2, p, X, p, false, Vb, p, false, m, p, p, nullptr, 0, 0.0, 1.0));
            wg_gemm(gemm_op(X, p, W2, p, false, Vb, p, true, m, p, p, nullptr, 0, 0.0, 1.0));
            wg_tri_solve<false>(Sm, m, 0, X, p, m, p);                                              // K^T
        }
        for (int i = threadIdx.x; i < m; i += DT) v3[i] = v0[i] - v2[i];                            // x_meas - yhat
        __syncthreads();
        op_gemv_t(v5, X, v3, v4, 1.0, p, m);                                                        // mu- + K (x - yhat)
        if (!sq) {
            wg_gemm(gemm_op(Vo, p, X, p, true, WS, p, false, p, p, m, Vb, p, 1.0, -1.0));           // Sigma- - K (W Sigma-)
        } else {
            wg_gemm(gemm_op(A1, p, X, p, true, WS, p, false, p, p, m, Vb, p, 1.0, -1.0));           // L- - K (W L-)
            wg_transpose(S, p, A1, p, p, p, 0);
            wg_gemm(gemm_op(S + pp, p, V, m, true, X, p, false, m, p, m, nullptr, 0, 0.0, 1.0));    // (K V^{1/2})^T
            wg_qr_r(S, p, p + m, p);
            wg_transpose(Vo, p, S, p, p, p, 1);
        }
        op_scatter(a.o_mean_filt, v5, p, n, b); op_scatter(a.o_var_filt, Vo, (int)pp, n, b);
        __syncthreads();
    };

    if (a.op == OP_PREDICT || a.op == OP_FILTER) {
        predict();
        if (a.op == OP_FILTER) update(false);
        return;
    }
    if (a.op == OP_UPDATE || a.op == OP_FORECAST) {
        op_gather(v4, a.mean_state_pred, p, n, b); op_gather(Vb, a.var_state_pred, (int)pp, n, b);
        __syncthreads();
        update(a.op == OP_FORECAST);
        return;
    }
    // ---- smoothers (standard.py:160-255, 339-371 / square_root.py:158-261, 347-385) ----
    op_gather(Q, a.wgt_state, (int)pp, n, b); op_gather(Va, a.var_state_filt, (int)pp, n, b);      // Va = filt, Vb = pred
    op_gather(Vb, a.var_state_pred, (int)pp, n, b);
    op_gather(v0, a.mean_state_filt, p, n, b); op_gather(v1, a.mean_state_pred, p, n, b);
    if (sq) op_gather(R, a.var_state, (int)pp, n, b);
    const bool want_mv = a.op == OP_SMOOTH_MV || a.op == OP_SMOOTH;
    const bool want_sim = a.op == OP_SMOOTH_SIM || a.op == OP_SMOOTH, want_cond = a.op == OP_SMOOTH_COND;
    if (want_mv) { op_gather(Vc, a.var_state_next, (int)pp, n, b); op_gather(v2, a.mean_state_next, p, n, b); }
    if (want_sim) op_gather(v3, a.x_state_next, p, n, b);
    __syncthreads();
    if (!sq) {
        wg_gemm(gemm_op(A3, p, Q, p, false, Va, p, true, p, p, p, nullptr, 0, 0.0, 1.0));           // T^T = Q Sigma_f^T
        for (int e = threadIdx.x; e < (int)pp; e += DT) { A4[e] = A3[e]; A2[e] = Vb[e]; }
        __syncthreads();
        wg_lu_solve(A2, p, A3, p, p, p, piv);                                                       // G^T (standard.py:176)
    } else {
        wg_gemm(gemm_op(A1, p, Va, p, false, Va, p, true, p, p, p, nullptr, 0, 0.0, 1.0));          // L_f L_f^T
        for (int e = threadIdx.x; e < (int)pp; e += DT) A2[e] = Q[e];
        __syncthreads();
        wg_tri_solve<true>(Vb, p, 0, A2, p, p, p);
        wg_gemm(gemm_op(A3, p, A2, p, false, A1, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
        wg_tri_solve<false>(Vb, p, 1, A3, p, p, p);                                                 // G^T (square_root.py:170-175)
        wg_gemm(gemm_op(A1, p, Q, p, true, A3, p, false, p, p, p, nullptr, 0, 0.0, -1.0));          // J^T = I - Q^T G^T
        for (int i = threadIdx.x; i < p; i += DT) A1[(size_t)i * p + i] += 1.0;
        __syncthreads();
    }
    if (want_mv) {
        for (int i = threadIdx.x; i < p; i += DT) v5[i] = v2[i] - v1[i];
        __syncthreads();
        op_gemv_t(v6, A3, v5, v0, 1.0, p, p);                                                       // mu_f + G (mu_next - mu-)
        if (!sq) {
            for (int e = threadIdx.x; e < (int)pp; e += DT) A1[e] = Vc[e] - Vb[e];
            __syncthreads();
            wg_gemm(gemm_op(A2, p, A3, p, true, A1, p, false, p, p, p, nullptr, 0, 0.0, 1.0));      // G D
            wg_gemm(gemm_op(Vo, p, A2, p, false, A3, p, false, p, p, p, Va, p, 1.0, 1.0));          // Sigma_f + (G D) G^T
        } else {
            wg_gemm(gemm_op(S, p, Vc, p, true, A3, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
            wg_gemm(gemm_op(S + pp, p, R, p, true, A3, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
            wg_gemm(gemm_op(S + 2 * pp, p, Va, p, true, A1, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
            wg_qr_r(S, p, 3 * p, p);                                                                // square_root.py:217-218
            wg_transpose(Vo, p, S, p, p, p, 1);
        }
        op_scatter(a.o_mean_smooth, v6, p, n, b); op_scatter(a.o_var_smooth, Vo, (int)pp, n, b);
        __syncthreads();
    }
    if (want_sim || want_cond) {
        if (!sq) {
            wg_gemm(gemm_op(Vo, p, A3, p, true, A4, p, false, p, p, p, Va, p, 1.0, -1.0));          // Sigma_f - G T^T
        } else {
            wg_gemm(gemm_op(S, p, R, p, true, A3, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
            wg_gemm(gemm_op(S + pp, p, Va, p, true, A1, p, false, p, p, p, nullptr, 0, 0.0, 1.0));
            wg_qr_r(S, p, 2 * p, p);                                                                // square_root.py:259-260
            wg_transpose(Vo, p, S, p, p, p, 1);
        }
        if (want_cond) {
            op_gemv_t(v6, A3, v1, v0, -1.0, p, p);                                                  // mu_f - G mu-
            wg_transpose(A2, p, A3, p, p, p, 0);                                                    // G
            op_scatter(a.o_wgt_cond, A2, (int)pp, n, b); op_scatter(a.o_mean_cond, v6, p, n, b);
            op_scatter(a.o_var_cond, Vo, (int)pp, n, b);
        } else {
            for (int i = threadIdx.x; i < p; i += DT) v5[i] = v3[i] - v1[i];
            __syncthreads();
            op_gemv_t(v7, A3, v5, v0, 1.0, p, p);                                                   // mu_f + G (x_next - mu-)
            op_scatter(a.o_mean_sim, v7, p, n, b); op_scatter(a.o_var_sim, Vo, (int)pp, n, b);
        }
    }
}

int dense_op_launch(rk_handle h, const OpArgs& a) {
    RK_REQUIRE(a.p <= 768, RK_ERR_UNSUPPORTED, "per-step operators: n_state = %d exceeds 768 (LDS staging of the dense products)", a.p);
    const size_t stride = dense_op_ws_doubles(a.p, a.m > 0 ? a.m : 1);
    const size_t need = stride * (size_t)a.n * sizeof(double);
    if (h->op_scratch_bytes < need) {                     // grow-only scratch of the handle (freed by rk_destroy)
        if (h->op_scratch) { RK_HIP(hipStreamSynchronize(h->stream)); RK_HIP(hipFree(h->op_scratch)); h->op_scratch = nullptr; h->op_scratch_bytes = 0; }
        RK_HIP(hipMalloc(&h->op_scratch, need));
        h->op_scratch_bytes = need;
    }
    hipLaunchKernelGGL(dense_op_kernel, dim3(a.n), dim3(DT), 0, h->stream, a, (double*)h->op_scratch, stride);
    RK_HIP(hipGetLastError());
    return RK_OK;
}

}  // namespace rk
