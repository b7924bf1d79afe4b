"""Philox4x32-10 known-answer vectors (Random123 kat_vectors) and sanity of the Box-Muller normals."""
import numpy as np
from oracle import counter_rng as cr


def _kat(c, k):
    out = cr.philox4x32_10(*[np.uint32(x) for x in c], *[np.uint32(x) for x in k])
    return [int(x) for x in out]


def test_philox_kat():
    assert _kat([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _kat([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _kat([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_normals_shape_and_moments():
    z = cr.normals(seed=12345, traj=np.arange(20000), step=3, n_block=2, n_bstate=3, purpose=cr.PURPOSE_SMOOTH)
    assert z.shape == (20000, 2, 3)
    assert np.all(np.isfinite(z))
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    # distinct counters give distinct, uncorrelated streams
    z2 = cr.normals(12345, np.arange(20000), 4, 2, 3, cr.PURPOSE_SMOOTH)
    assert abs(np.corrcoef(z.ravel(), z2.ravel())[0, 1]) < 0.02
    z3 = cr.normals(12345, np.arange(20000), 3, 2, 3, cr.PURPOSE_INTERROGATE)
    assert abs(np.corrcoef(z.ravel(), z3.ravel())[0, 1]) < 0.02


def test_normals_keyed_by_global_index():
    """Sharding invariance: trajectories 100..149 drawn alone equal the slice of a larger draw."""
    full = cr.normals(99, np.arange(200), 7, 3, 4, cr.PURPOSE_SMOOTH)
    part = cr.normals(99, np.arange(100, 150), 7, 3, 4, cr.PURPOSE_SMOOTH)
    assert np.array_equal(full[100:150], part)
