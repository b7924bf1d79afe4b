// Interrogation of the dense ("non-block", prior.indep_init) path for right-hand sides that arrive through hiprtc: any
// traced / hand-written ode_fun(X, t) with X (1, p) -> (1, m), like the reference's non-block form takes any ode_fun
// (src/rodeo/prior/indep_init.py:8-23, examples/solve_nb.py).  The dense forward pass (solve_dense.hip) is a sequence
// of workgroup-wide GEMM / LU phases around the one place where the ODE enters -- interrogate.py:13-115 -- and a step of
// it takes ~1 ms at the sizes this path serves, so for these right-hand sides the step is cut there: the precompiled
// kernel runs predict (standard.py:57-59) up to mu-, THIS kernel (built at run time around the user's code) evaluates f
// and its dense Jacobian at mu- and leaves W~ = W - J and the offset a in the trajectory's workspace, the precompiled
// kernel continues with the update (standard.py:93-102).  One workgroup per trajectory; thread j evaluates the
// right-hand side once on forward-mode duals seeded in direction j, which gives column j of the m x p Jacobian that
// jax.jacfwd returns in interrogate.py:76 (every thread also holds f itself).  RTC-safe.
#pragma once
#include "rk_enums.hpp"
#include "solve_args.hpp"
#include "dual.hpp"

namespace rk {

struct DenseItgArgs {
    int B, N, n;                     // trajectories, steps, the step this launch interrogates (time t_{n+1})
    double t_min, t_max;
    const double* W;                 // (m, p), shared
    const double* theta;             // (n_theta[, B]) or null
    int theta_b;
    double* ws;                      // dense workspace; per trajectory ws_stride doubles
    size_t ws_stride, off_mup, off_Wt, off_am, off_x;      // off_x: where f is evaluated (mu-, or chkrebtii's draw)
};

template <class U, int P, int ITG>
__global__ void __launch_bounds__(256) dense_interrogate_kernel(DenseItgArgs a) {
    static_assert(U::D == 1, "dense path: one block holding all variables");
    constexpr int M = U::M;
    const int b = blockIdx.x;
    double* const w = a.ws + (size_t)b * a.ws_stride;
    const double* const mup = w + a.off_mup;
    const double* const xe = w + a.off_x;          // mu- (kramer, schober, rodeo) or x ~ N(mu-, Sigma-) (chkrebtii, interrogate.py:30-34)
    double* const Wt = w + a.off_Wt;
    double* const am = w + a.off_am;
    __shared__ double fv[M];
    double th[U::NTHETA];
#pragma unroll
    for (int k = 0; k < U::NTHETA; ++k) th[k] = a.theta ? ld(a.theta, k, a.theta_b, a.B, b) : 0.0;
    const double t = a.t_min + (a.t_max - a.t_min) * (double)(a.n + 1) / (double)a.N;          // solve.py:74
    for (int j = threadIdx.x; j < P; j += blockDim.x) {
        Dual<1> X[1][P], out[1][M];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            X[0][k] = Dual<1>(xe[k]);              // (entries the function does not read are dropped by the compiler)
            X[0][k].d[0] = k == j ? 1.0 : 0.0;
        }
        U::template rhs<Dual<1>, P>(X, t, th, out);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const double Jij = ITG == RK_INTERROGATE_KRAMER ? out[0][i].d[0] : 0.0;
            Wt[(size_t)i * P + j] = a.W[(size_t)i * P + j] + (-Jij);                           // W + wgt_meas (solve.py:79)
            if (j == 0) fv[i] = out[0][i].v;
        }
    }
    __syncthreads();
    // mean_meas = -f (+ J mu- for kramer, interrogate.py:81-82), J mu- = (W - W~) mu-
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        double jm = 0.0;
        if (ITG == RK_INTERROGATE_KRAMER)
            for (int j = 0; j < P; ++j) jm = fma(a.W[(size_t)i * P + j] - Wt[(size_t)i * P + j], mup[j], jm);
        am[i] = ITG == RK_INTERROGATE_KRAMER ? -fv[i] + jm : -fv[i];
    }
}

}  // namespace rk
