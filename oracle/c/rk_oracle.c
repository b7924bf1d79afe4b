/*
 * TEST INFRASTRUCTURE (see oracle/__init__.py) -- plain-C restatement of rodeo's solve_mv hot loop for one CPU.
 * Used only (a) as the timed `cpu_baseline` ("port") leg of bench.py and (b) cross-checked against the NumPy
 * restatement in tests/test_oracle_c.py.  Never linked into or called by the product (rodeo_amd/).
 *
 * Follows, per trajectory and per block, exactly the reference recursion:
 *   forward  src/rodeo/solve.py:57-97  with  predict  src/rodeo/kalmantv/standard.py:57-59,
 *            interrogation src/rodeo/interrogate.py:59-62 (schober) / :74-84 (kramer) / :109-115 (rodeo),
 *            update   src/rodeo/kalmantv/standard.py:93-102  (n_bmeas = 1: the LU solve of utils.py:119 is a divide)
 *   backward src/rodeo/solve.py:279-301 with smooth_mv standard.py:175-176 (LU with partial pivoting), :210-216
 * and, like the reference, STORES the predicted moments of every step (solve.py:93-96) and reads them back in the
 * backward pass -- it is the reference's algorithm, not the GPU kernel's.
 *
 * Timing hygiene (bench.py's cpu_baseline leg): `rko_solve_mv_ws` works on buffers the caller allocated ONCE
 * (outputs + one scratch area per thread for the stored predictions), which `rko_first_touch` has paged in with the
 * same static thread schedule, so that neither malloc nor first-touch page faults fall into the timed region; the
 * per-trajectory routine is force-inlined into callers with compile-time (n_block, n_bstate) for the benchmark
 * shapes so that gcc unrolls the p x p loops.
 *
 * Layout: the reference's, batch first: mean (B, N+1, d, p), var (B, N+1, d, p, p); theta (B, 3); x0 (B, d, p);
 * W (d, 1, p), Q (d, p, p), R (d, p, p) shared by all trajectories.  OpenMP over trajectories.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PMAX 8
#define DMAX 4

enum { RHS_FITZ = 1, RHS_LORENZ = 2, RHS_HIGHER = 3 };
enum { ITG_RODEO = 0, ITG_SCHOBER = 1, ITG_KRAMER = 2 };

/* f_b(X, t) and the block-diagonal Jacobian row J[b][:] = d f_b / d X[b][:] */
static inline __attribute__((always_inline)) void rhs_eval(int rhs, int d, int p, const double* X /* d x p */, double t, const double* th, double* f,
                     double* J /* d x p or NULL */) {
    if (J) memset(J, 0, sizeof(double) * d * p);
    if (rhs == RHS_FITZ) {
        const double a = th[0], b = th[1], c = th[2], V = X[0], R = X[p];
        f[0] = c * (V - V * V * V / 3 + R);
        f[1] = -1 / c * (V - a + b * R);
        if (J) { J[0] = c * (1.0 - V * V); J[p] = -b / c; }
    } else if (rhs == RHS_LORENZ) {
        const double rho = th[0], sigma = th[1], beta = th[2], x = X[0], y = X[p], z = X[2 * p];
        f[0] = -sigma * x + sigma * y;
        f[1] = rho * x - y - x * z;
        f[2] = -beta * z + x * y;
        if (J) { J[0] = -sigma; J[p] = -1.0; J[2 * p] = -beta; }
    } else {
        f[0] = sin(2 * t) - X[0];
        if (J) J[0] = -1.0;
    }
}

/* X = A^{-1} B, LU with partial pivoting; A (n x n), B (n x nr) row-major, destroyed */
static inline __attribute__((always_inline)) void lu_solve(const int n, int nr, double* A, double* B) {
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            for (int j = 0; j < nr; ++j) { double t = B[k * nr + j]; B[k * nr + j] = B[piv * nr + j]; B[piv * nr + j] = t; }
        }
        const double r = 1.0 / A[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double l = A[i * n + k] * r;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
            for (int j = 0; j < nr; ++j) B[i * nr + j] -= l * B[k * nr + j];
        }
    }
    for (int k = n - 1; k >= 0; --k)
        for (int j = 0; j < nr; ++j) {
            double s = B[k * nr + j];
            for (int i = k + 1; i < n; ++i) s -= A[k * n + i] * B[i * nr + j];
            B[k * nr + j] = s / A[k * n + k];
        }
}

static inline __attribute__((always_inline)) void solve_one(const int rhs, const int itg, const int N, const int d, const int p, double t_min, double t_max, const double* W,
                      const double* x0, const double* Q, const double* R, const double* th, double* mean,
                      double* var, double* mpred, double* vpred) {
    const int pp = p * p;
    /* index 0: (ode_init, 0) (solve.py:114-121) */
    memcpy(mean, x0, sizeof(double) * d * p);
    memcpy(mpred, x0, sizeof(double) * d * p);
    memset(var, 0, sizeof(double) * d * pp);
    memset(vpred, 0, sizeof(double) * d * pp);
    double f[DMAX], J[DMAX * PMAX], A[PMAX * PMAX];
    for (int n = 0; n < N; ++n) {
        const double* mf = mean + (size_t)n * d * p;
        const double* vf = var + (size_t)n * d * pp;
        double* mp = mpred + (size_t)(n + 1) * d * p;
        double* vp = vpred + (size_t)(n + 1) * d * pp;
        double* mo = mean + (size_t)(n + 1) * d * p;
        double* vo = var + (size_t)(n + 1) * d * pp;
        for (int b = 0; b < d; ++b) {                      /* predict: standard.py:57-59 */
            const double* Qb = Q + b * pp; const double* Rb = R + b * pp;
            for (int i = 0; i < p; ++i) {
                double s = 0;
                for (int k = 0; k < p; ++k) s += Qb[i * p + k] * mf[b * p + k];
                mp[b * p + i] = s;
            }
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) {
                    double s = 0;
                    for (int k = 0; k < p; ++k) s += Qb[i * p + k] * vf[b * pp + k * p + j];
                    A[i * p + j] = s;
                }
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) {
                    double s = 0;
                    for (int k = 0; k < p; ++k) s += A[i * p + k] * Qb[j * p + k];
                    vp[b * pp + i * p + j] = s + Rb[i * p + j];
                }
        }
        const double t = t_min + (t_max - t_min) * (double)(n + 1) / (double)N;       /* solve.py:74 */
        rhs_eval(rhs, d, p, mp, t, th, f, itg == ITG_KRAMER ? J : NULL);
        for (int b = 0; b < d; ++b) {
            double Wm[PMAX], WS[PMAX], SW[PMAX], a, V = 0.0;
            const double* Sp = vp + b * pp;
            if (itg == ITG_KRAMER) {                        /* interrogate.py:80-83 */
                double jm = 0;
                for (int j = 0; j < p; ++j) jm += J[b * p + j] * mp[b * p + j];
                a = -f[b] + jm;
                for (int j = 0; j < p; ++j) Wm[j] = W[b * p + j] + (-J[b * p + j]);   /* solve.py:79 */
            } else {
                a = -f[b];
                for (int j = 0; j < p; ++j) Wm[j] = W[b * p + j] + 0.0;
                if (itg == ITG_RODEO) {                     /* interrogate.py:110-113 */
                    for (int j = 0; j < p; ++j) {
                        double s = 0;
                        for (int i = 0; i < p; ++i) s += W[b * p + i] * Sp[i * p + j];
                        WS[j] = s;
                    }
                    for (int j = 0; j < p; ++j) V += WS[j] * W[b * p + j];
                }
            }
            /* update: standard.py:93-102 with x_meas = 0 */
            double yhat = 0, S = 0;
            for (int j = 0; j < p; ++j) yhat += Wm[j] * mp[b * p + j];
            yhat += a;
            for (int j = 0; j < p; ++j) {
                double s = 0;
                for (int i = 0; i < p; ++i) s += Wm[i] * Sp[i * p + j];
                WS[j] = s;
            }
            for (int j = 0; j < p; ++j) S += WS[j] * Wm[j];
            S += V;
            for (int i = 0; i < p; ++i) {
                double s = 0;
                for (int j = 0; j < p; ++j) s += Sp[i * p + j] * Wm[j];
                SW[i] = s;
            }
            for (int i = 0; i < p; ++i) {
                const double K = SW[i] / S;
                mo[b * p + i] = mp[b * p + i] + K * (0.0 - yhat);
                for (int j = 0; j < p; ++j) vo[b * pp + i * p + j] = Sp[i * p + j] - K * WS[j];
            }
        }
    }
    /* backward: solve.py:279-301; smoothed values overwrite the filtered ones from n = N-1 down to 1 */
    double T[PMAX * PMAX], Xr[PMAX * PMAX], G[PMAX * PMAX], D[PMAX * PMAX], GD[PMAX * PMAX], ms[PMAX], Ss[PMAX * PMAX];
    for (int b = 0; b < d; ++b) {
        const double* Qb = Q + b * pp;
        memcpy(ms, mean + ((size_t)N * d + b) * p, sizeof(double) * p);
        memcpy(Ss, var + ((size_t)N * d + b) * pp, sizeof(double) * pp);
        for (int n = N - 1; n >= 1; --n) {
            double* mf = mean + ((size_t)n * d + b) * p;
            double* vf = var + ((size_t)n * d + b) * pp;
            const double* mp = mpred + ((size_t)(n + 1) * d + b) * p;
            const double* vp = vpred + ((size_t)(n + 1) * d + b) * pp;
            for (int i = 0; i < p; ++i)                     /* T = Sigma_f Q^T : standard.py:175 */
                for (int j = 0; j < p; ++j) {
                    double s = 0;
                    for (int k = 0; k < p; ++k) s += vf[i * p + k] * Qb[j * p + k];
                    T[i * p + j] = s;
                }
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) { A[i * p + j] = vp[i * p + j]; Xr[i * p + j] = T[j * p + i]; }
            lu_solve(p, p, A, Xr);                          /* standard.py:176 */
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) G[i * p + j] = Xr[j * p + i];
            double msn[PMAX];
            for (int i = 0; i < p; ++i) {                   /* standard.py:213-214 */
                double s = 0;
                for (int k = 0; k < p; ++k) s += G[i * p + k] * (ms[k] - mp[k]);
                msn[i] = mf[i] + s;
            }
            for (int i = 0; i < pp; ++i) D[i] = Ss[i] - vp[i];
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) {
                    double s = 0;
                    for (int k = 0; k < p; ++k) s += G[i * p + k] * D[k * p + j];
                    GD[i * p + j] = s;
                }
            for (int i = 0; i < p; ++i)                     /* standard.py:215-216 */
                for (int j = 0; j < p; ++j) {
                    double s = 0;
                    for (int k = 0; k < p; ++k) s += GD[i * p + k] * G[j * p + k];
                    Ss[i * p + j] = vf[i * p + j] + s;
                }
            memcpy(ms, msn, sizeof(double) * p);
            memcpy(mf, ms, sizeof(double) * p);
            memcpy(vf, Ss, sizeof(double) * pp);
        }
    }
}

/* compile-time (n_block, n_bstate) instances for the benchmark shapes; anything else runs the runtime-size loops */
static void solve_one_dispatch(int rhs, int itg, int N, int d, int p, double t_min, double t_max, const double* W,
                               const double* x0b, const double* Q, const double* R, const double* thb, double* mean_b,
                               double* var_b, double* mpred, double* vpred) {
    if (d == 2 && p == 3) solve_one(rhs, itg, N, 2, 3, t_min, t_max, W, x0b, Q, R, thb, mean_b, var_b, mpred, vpred);
    else if (d == 3 && p == 4) solve_one(rhs, itg, N, 3, 4, t_min, t_max, W, x0b, Q, R, thb, mean_b, var_b, mpred, vpred);
    else if (d == 2 && p == 4) solve_one(rhs, itg, N, 2, 4, t_min, t_max, W, x0b, Q, R, thb, mean_b, var_b, mpred, vpred);
    else if (d == 2 && p == 5) solve_one(rhs, itg, N, 2, 5, t_min, t_max, W, x0b, Q, R, thb, mean_b, var_b, mpred, vpred);
    else solve_one(rhs, itg, N, d, p, t_min, t_max, W, x0b, Q, R, thb, mean_b, var_b, mpred, vpred);
}

static int check_shape(int rhs, int itg, int B, int N, int d, int p) {
    if (d > DMAX || p > PMAX || d < 1 || p < 2 || B < 1 || N < 1) return -1;
    if (itg < 0 || itg > ITG_KRAMER || rhs < RHS_FITZ || rhs > RHS_HIGHER) return -2;
    return 0;
}

/* doubles of per-thread scratch (the stored predictions of one trajectory, solve.py:93-96) */
size_t rko_scratch_doubles(int N, int d, int p) { return (size_t)(N + 1) * d * p * (1 + (size_t)p); }

/* page in `n_per_traj` doubles per trajectory of `buf` and the threads' scratch with the schedule the solver uses */
void rko_first_touch(double* buf, size_t n_per_traj, int B, double* scratch, size_t scratch_per_thread, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        if (scratch) memset(scratch + (size_t)tid * scratch_per_thread, 0, sizeof(double) * scratch_per_thread);
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b)
            if (buf) memset(buf + (size_t)b * n_per_traj, 0, sizeof(double) * n_per_traj);
    }
}

/*
 * `reps` passes over the B trajectories into caller-owned buffers; scratch holds nthreads x rko_scratch_doubles().
 * Returns 0 and the wall time of the passes (seconds, measured around the parallel region) in *seconds.
 */
int rko_solve_mv_ws(int rhs, int itg, int B, int N, int d, int p, double t_min, double t_max, const double* W,
                    const double* x0, const double* Q, const double* R, const double* theta, int n_theta, double* mean,
                    double* var, double* scratch, int reps, int nthreads, double* seconds) {
    const int rc = check_shape(rhs, itg, B, N, d, p);
    if (rc) return rc;
    if (!scratch || reps < 1) return -4;
    const size_t ms = (size_t)(N + 1) * d * p, vs = ms * p;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    const double t0 = omp_get_wtime();
#endif
#pragma omp parallel
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double* mpred = scratch + (size_t)tid * (ms + vs);
        double* vpred = mpred + ms;
        for (int r = 0; r < reps; ++r) {
#pragma omp for schedule(static)
            for (int b = 0; b < B; ++b)
                solve_one_dispatch(rhs, itg, N, d, p, t_min, t_max, W, x0 + (size_t)b * d * p, Q, R,
                                   theta ? theta + (size_t)b * n_theta : NULL, mean + b * ms, var + b * vs, mpred, vpred);
        }
    }
#ifdef _OPENMP
    if (seconds) *seconds = omp_get_wtime() - t0;
#else
    if (seconds) *seconds = 0.0;
#endif
    return 0;
}

/* convenience form for the tests: allocates the scratch itself.  returns 0 on success; nthreads <= 0 -> OpenMP default */
int rko_solve_mv(int rhs, int itg, int B, int N, int d, int p, double t_min, double t_max, const double* W,
                 const double* x0, const double* Q, const double* R, const double* theta, int n_theta, double* mean,
                 double* var, int nthreads) {
    const int rc = check_shape(rhs, itg, B, N, d, p);
    if (rc) return rc;
#ifdef _OPENMP
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    const int nt = 1;
#endif
    double* scratch = (double*)malloc(sizeof(double) * (size_t)nt * rko_scratch_doubles(N, d, p));
    if (!scratch) return -3;
    const int rc2 = rko_solve_mv_ws(rhs, itg, B, N, d, p, t_min, t_max, W, x0, Q, R, theta, n_theta, mean, var, scratch,
                                    1, nt, NULL);
    free(scratch);
    return rc2;
}

int rko_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
