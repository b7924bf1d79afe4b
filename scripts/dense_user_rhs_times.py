"""The dense (non-block) path with a traced Python right-hand side at the C5 size (n_vars = 32, n_deriv = 5, 256
trajectories): the forward pass is cut at the interrogation (2 N + 1 launches, csrc/solve_dense_itg_kernels.hpp); time per
step next to the built-in linear right-hand side's single launch."""
import json, os, sys, time
import numpy as np
from scipy.linalg import block_diag
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodeo_amd as ra

n_vars, n_deriv, B, N = 32, 5, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 50
p, t_max = n_vars * n_deriv, 0.01 * N
rng = np.random.default_rng(1)


def ring(X, t, kc):
    x = X[0, ::n_deriv]
    return np.array([[-kc[0] * x[i] + kc[1] * np.sin(x[(i + 1) % n_vars]) - 0.1 * x[i] ** 3 + 0.05 * np.cos(t)
                      for i in range(n_vars)]])


Wb, _ = ra.utils.first_order_pad(lambda x, t: x, n_vars, n_deriv)
W = block_diag(*[w for w in Wb])[None]
prior = ra.indep_init(ra.ibm_init(t_max / N, n_deriv, 0.5 * np.ones(n_vars)))
X0 = np.zeros((B, 1, p))
X0[:, 0, ::n_deriv] = 0.5 * rng.standard_normal((B, n_vars))
kc = np.array([0.8, 0.6])
plan = ra.SolvePlan(ring, W, X0, 0.0, t_max, N, ra.interrogate.interrogate_rodeo, prior, kc=kc)
t0 = time.perf_counter(); plan.mv(None); plan.dev.sync(); first = time.perf_counter() - t0
plan.dev.profile_enable(True)
plan.mv(None)
k_user = dict(plan.dev.profile_last())
A = -np.eye(n_vars) + 0.1 * rng.standard_normal((n_vars, n_vars)) / np.sqrt(n_vars)
lin = ra.SolvePlan(ra.ode.linear_dense(n_vars, n_deriv), W, X0, 0.0, t_max, N, ra.interrogate.interrogate_rodeo, prior, A=A)
lin.mv(None); lin.mv(None)
k_lin = dict(lin.dev.profile_last())
print(json.dumps({"config": f"dense path, n_vars={n_vars} n_deriv={n_deriv} B={B} N={N}, solve_mv + interrogate_rodeo",
                  "traced_python_rhs_kernels_ms": k_user, "builtin_linear_rhs_kernels_ms": k_lin,
                  "first_call_with_hiprtc_s": first,
                  "forward_ms_per_step": {"traced": k_user["dense_fwd_kernel<user, stepwise>"] / N,
                                          "linear": k_lin["dense_fwd_kernel"] / N}}))
