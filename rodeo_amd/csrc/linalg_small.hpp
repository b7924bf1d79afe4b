// Register-resident small dense linear algebra for one (trajectory, block) per lane.
// All loops are fully unrolled over compile-time sizes so every array lives in VGPRs (runtime-indexed arrays would
// go to scratch).  Operation order follows the reference's expressions (src/rodeo/kalmantv/standard.py) so that the
// rounding pattern stays as close to the JAX path as a different BLAS allows.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace rk {

// 1/x to ~1 ulp: v_rcp_f64 + two Newton steps (~41 cycles and 5 instructions instead of the ~12-instruction IEEE
// division sequence; measured in profiles/r01_probe2_fp64_valu_mfma_latency.log).  x = 0 / inf / NaN give NaN or inf,
// which propagate silently like the reference's singular solves.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// 1 / x from v_rcp_f64 and ONE cubic step y0 (1 + e + e^2), e = 1 - x y0: three dependent operations instead of the
// four of two Newton steps, for use on a latency-bound dependent chain.  Error <= 1 ulp like fast_rcp
// (profiles/r01_probe7_rcp_accuracy.log).
__device__ __forceinline__ double fast_rcp_cubic(double x) {
    const double y0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y0, 1.0);
    return fma(y0, fma(e, e, e), y0);
}

// sqrt(x) for x >= 0 in the normal range (variances), from v_rsq_f64 and two coupled Goldschmidt / Newton steps:
// eight dependent operations, no scaling branches (what sqrt() adds for denormal / huge arguments); <= 1 ulp.
__device__ __forceinline__ double fast_sqrt_pos(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return x > 0.0 ? g : 0.0;
}

// C = A (RxK) * B (KxC)
template <int R, int K, int C>
__device__ __forceinline__ void mm(const double (&A)[R][K], const double (&B)[K][C], double (&out)[R][C]) {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) {
            double s = A[i][0] * B[0][j];
#pragma unroll
            for (int k = 1; k < K; ++k) s = fma(A[i][k], B[k][j], s);
            out[i][j] = s;
        }
}

// C = A (RxK) * B^T, B is (CxK)
template <int R, int K, int C>
__device__ __forceinline__ void mm_nt(const double (&A)[R][K], const double (&B)[C][K], double (&out)[R][C]) {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) {
            double s = A[i][0] * B[j][0];
#pragma unroll
            for (int k = 1; k < K; ++k) s = fma(A[i][k], B[j][k], s);
            out[i][j] = s;
        }
}

// y = A (RxK) x
template <int R, int K>
__device__ __forceinline__ void mv(const double (&A)[R][K], const double (&x)[K], double (&y)[R]) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
        double s = A[i][0] * x[0];
#pragma unroll
        for (int k = 1; k < K; ++k) s = fma(A[i][k], x[k], s);
        y[i] = s;
    }
}

template <int K>
__device__ __forceinline__ double dot(const double (&a)[K], const double (&b)[K]) {
    double s = a[0] * b[0];
#pragma unroll
    for (int k = 1; k < K; ++k) s = fma(a[k], b[k], s);
    return s;
}

// X = A^{-1} B by LU with partial pivoting (what jnp.linalg.solve does, src/rodeo/utils.py:119 -> LAPACK gesv:
// first-maximum pivot search, right-looking elimination).  A (PxP) and B (PxNR) are destroyed; B <- X.
// Row swaps are predicated selects so the arrays stay in registers.  A singular pivot yields inf/NaN, silently,
// like the reference.
// The two halves are separate functions so that a kernel can run them in different pipeline stages.
template <int P, int NR>
__device__ __forceinline__ void lu_factor_fwd(double (&A)[P][P], double (&B)[P][NR], double (&rpiv)[P]) {
#pragma unroll
    for (int k = 0; k < P; ++k) {
        // pivot search in column k (first maximum of |a_ik|, i >= k)
        int piv = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < P; ++i) {
            const double v = fabs(A[i][k]);
            const bool gt = v > best;
            best = gt ? v : best;
            piv = gt ? i : piv;
        }
#pragma unroll
        for (int i = k + 1; i < P; ++i) {
            const bool sw = (piv == i);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const double t = A[k][j];
                A[k][j] = sw ? A[i][j] : t;
                A[i][j] = sw ? t : A[i][j];
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const double t = B[k][j];
                B[k][j] = sw ? B[i][j] : t;
                B[i][j] = sw ? t : B[i][j];
            }
        }
        rpiv[k] = fast_rcp(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < P; ++i) {
            const double l = A[i][k] * rpiv[k];
#pragma unroll
            for (int j = k + 1; j < P; ++j) A[i][j] = fma(-l, A[k][j], A[i][j]);
#pragma unroll
            for (int j = 0; j < NR; ++j) B[i][j] = fma(-l, B[k][j], B[i][j]);
        }
    }
}
// back substitution with the upper factor left in A by lu_factor_fwd
template <int P, int NR>
__device__ __forceinline__ void lu_back(const double (&A)[P][P], double (&B)[P][NR], const double (&rpiv)[P]) {
#pragma unroll
    for (int k = P - 1; k >= 0; --k) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            double s = B[k][j];
#pragma unroll
            for (int i = k + 1; i < P; ++i) s = fma(-A[k][i], B[i][j], s);
            B[k][j] = s * rpiv[k];
        }
    }
}
template <int P, int NR>
__device__ __forceinline__ void lu_solve(double (&A)[P][P], double (&B)[P][NR]) {
    double rpiv[P];
    lu_factor_fwd<P, NR>(A, B, rpiv);
    lu_back<P, NR>(A, B, rpiv);
}

// Lower-triangular F with F F^T = A for symmetric positive semi-definite A (reads the lower triangle); a
// non-positive pivot zeroes its column instead of producing NaN.  Mirror: oracle/interrogations.py psd_factor.
template <int P>
__device__ __forceinline__ void psd_factor(const double (&A)[P][P], double (&L)[P][P]) {
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) L[i][j] = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d = fma(-L[j][k], L[j][k], d);
        const bool ok = d > 0.0;
        const double dj = sqrt(ok ? d : 1.0);
        const double rdj = fast_rcp(dj);
        L[j][j] = ok ? dj : 0.0;
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
            double s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s = fma(-L[i][k], L[j][k], s);
            L[i][j] = ok ? s * rdj : 0.0;
        }
    }
}

// Symmetric M x M eigendecomposition by cyclic Jacobi rotations (M <= 3: eight sweeps are far past convergence):
// A is destroyed, w <- eigenvalues, V <- eigenvectors in its columns.  For the log-density rule of utils.py:60-78
// (jnp.linalg.eigh, eigenvalues with |w| <= 1e-8 dropped); the value does not depend on the order or signs returned.
template <int M>
__device__ __forceinline__ void sym_eig_jacobi(double (&A)[M][M], double (&w)[M], double (&V)[M][M]) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
        for (int p_ = 0; p_ < M - 1; ++p_)
#pragma unroll
            for (int q = p_ + 1; q < M; ++q) {
                const double apq = A[p_][q];
                const bool go = apq != 0.0;
                const double theta = go ? (A[q][q] - A[p_][p_]) / (2.0 * apq) : 0.0;
                const double t = go ? (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)) : 0.0;
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
                for (int k = 0; k < M; ++k) {                    // A <- A J
                    const double akp = A[k][p_], akq = A[k][q];
                    A[k][p_] = c * akp - sn * akq;
                    A[k][q] = sn * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < M; ++k) {                    // A <- J^T A ; V <- V J
                    const double apk = A[p_][k], aqk = A[q][k];
                    A[p_][k] = c * apk - sn * aqk;
                    A[q][k] = sn * apk + c * aqk;
                    const double vkp = V[k][p_], vkq = V[k][q];
                    V[k][p_] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) w[i] = A[i][i];
}


}  // namespace rk
