"""The plain-C restatement (oracle/c, the timed cpu_baseline 'port') agrees with the pinned NumPy restatement."""
import numpy as np
import pytest
from oracle import scan, odes, priors, c_port, interrogations as oi


@pytest.mark.parametrize("itg", ["rodeo", "schober", "kramer"])
def test_c_port_fitz(itg):
    rng = np.random.default_rng(1)
    B, N = 5, 120
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.1 * rng.standard_normal((B, 3)))
    W, init = priors.first_order_pad(odes.fitzhugh_nagumo, 2, 3)
    x0 = np.stack([init(np.array([-1., 1.]) + 0.1 * rng.standard_normal(2), 0.0, theta=theta[b]) for b in range(B)])
    prior = priors.ibm_init(6.0 / N, 3, np.array([.1, .1]))
    m, v = c_port.solve_mv("fitzhugh_nagumo", itg, W, x0, 0.0, 6.0, N, prior, theta, nthreads=2)
    mo, vo = scan.solve_mv(None, odes.fitzhugh_nagumo, W, x0, 0.0, 6.0, N, getattr(oi, "interrogate_" + itg), prior,
                           theta=theta)
    assert np.max(np.abs(m - mo)) < 1e-9
    assert np.max(np.abs(v - vo)) < 1e-9 * np.max(np.abs(vo))


def test_c_port_lorenz_and_higher():
    theta = np.array([[28., 10., 8. / 3.]])
    W, init = priors.first_order_pad(odes.lorenz63, 3, 4)
    x0 = init(np.array([-12., -5., 38.]), 0.0, theta=theta[0])[None]
    prior = priors.ibm_init(1e-3, 4, np.array([5e7] * 3))
    m, v = c_port.solve_mv("lorenz63", "kramer", W, x0, 0.0, 0.5, 500, prior, theta)
    mo, vo = scan.solve_mv(None, odes.lorenz63, W, x0[0], 0.0, 0.5, 500, oi.interrogate_kramer, prior, theta=theta[0])
    scale = np.max(np.abs(mo), axis=(0, 1))                       # per derivative order (x''' ~ 1e6)
    assert np.max(np.abs(m[0] - mo) / scale) < 1e-10
    W = np.array([[[0., 0., 1., 0.]]]); x0 = np.array([[[-1., 0., 1., 0.]]])
    prior = priors.ibm_init(0.1, 4, np.array([.001]))
    m, v = c_port.solve_mv("higher_order", "kramer", W, x0, 0.0, 10.0, 100, prior, None)
    mo, vo = scan.solve_mv(None, odes.higher_order, W, x0[0], 0.0, 10.0, 100, oi.interrogate_kramer, prior)
    assert np.max(np.abs(m[0] - mo)) < 1e-9
