"""
oracle/ -- CPU restatement of rodeo's Kalman time-stepping core.  TEST INFRASTRUCTURE ONLY.

This package is the *checker* for the HIP product in ``rodeo_amd/``; it is never the thing that is
shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  ``rodeo_amd`` never imports it and has no CPU fallback.

What it restates (all file:line relative to the reference tree, mlysy/rodeo v1.1.3):

* ``kalman_ops.py``   <- src/rodeo/kalmantv/standard.py:57-59,93-102,175-176,210-216,248-254,333-335,366-370
                          and src/rodeo/utils.py:105-119 (LU solve, *not* Cholesky)
* ``sqrt_ops.py``     <- src/rodeo/kalmantv/square_root.py and src/rodeo/utils.py:10-24 (add_sqrt = QR)
* ``scan.py``         <- src/rodeo/solve.py:31-122 (forward), :125-205 (solve_sim), :208-302 (solve_mv)
* ``interrogations.py`` <- src/rodeo/interrogate.py:13-47,50-62,65-84,87-115
* ``priors.py``       <- src/rodeo/prior/ibm.py:37-88, src/rodeo/prior/indep_init.py:8-23,
                          src/rodeo/utils.py:80-102 (first_order_pad)
* ``odes.py``         <- the ODE right-hand sides used by the reference's docs/tests
                          (README.md:92-99, docs/examples/lorenz.md:85-92, docs/examples/higher_order.md:47-59)
* ``joint_gaussian.py`` <- the reference's *test oracle*: tests/gauss_markov.py:30-144,
                          tests/utils.py:24-63 and src/rodeo/utils.py:27-57 (mvncond)
* ``counter_rng.py``  <- NOT from the reference: the Philox4x32-10 + Box-Muller stream the HIP kernels use
                          for solve_sim / interrogate_chkrebtii draws (the reference uses JAX threefry).
* ``c/``              <- plain-C restatement of the solve_mv / solve_sim hot loop, used as the timed
                          ``cpu_baseline`` ("port") in bench.py and cross-checked against the NumPy version.

PARITY PINNING STATUS
---------------------
The reference is pure Python on JAX; ``jax``/``jaxlib``/``blackjax`` are not installed here (ordinary
ModuleNotFoundError, no wheels, no network), so the reference itself cannot be run, and its test-suite stores
**no golden vectors** (all expected values are computed at test time from ``jax.random.PRNGKey(0)``).
The oracle is therefore pinned through the reference's own *test oracles*, restated:
K1 joint-Gaussian conditioning for every kalmantv op, K2 scan == double for-loop, K3 odeint on
FitzHugh-Nagumo, K4 analytic solution of x'' = sin 2t - x, K5 add_sqrt squares to A + B, K6 IBM closed forms.
Deterministic paths are pinned by K1-K6.  **Random draws (solve_sim, interrogate_chkrebtii) are
"parity unpinned"**: JAX's threefry bit-stream and SVD factor cannot be reproduced without JAX; they are
checked through their pre-sampling moments, injected normals and distributional tests only.
"""
