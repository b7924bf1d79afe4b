"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  NumPy restatement of the counter-based normal stream used by
the HIP kernels (rodeo_amd/csrc/philox.hpp) for ``solve_sim`` and ``interrogate_chkrebtii`` draws.

NOT a restatement of the reference: rodeo draws from JAX threefry keys (src/rodeo/solve.py:147,179;
src/rodeo/interrogate.py:23,30-34) whose bit-stream cannot be reproduced without JAX ("parity unpinned" for
draws).  Philox4x32-10 (Salmon et al., SC'11) is pinned by the Random123 known-answer vectors in
tests/test_counter_rng.py.

Stream layout (identical on device):
    key     = (seed & 0xffffffff, seed >> 32)                         seed = 64-bit user seed
    counter = (traj, step, block | purpose << 16, chunk)              traj = GLOBAL trajectory/draw index
    one Philox call -> 4 x u32 -> 2 uniforms in (0,1) (53 bits each) -> Box-Muller -> 2 normals (cos, sin)
    normal j of a p-vector comes from chunk j // 2, element j % 2.
Because the counter holds the global trajectory index, draws do not depend on how the batch is sharded.
"""
import numpy as np

PURPOSE_INTERROGATE = 0     # chkrebtii draw made at forward step n (for time index n+1)
PURPOSE_SMOOTH = 1          # backward draw for time index n (terminal draw uses n = N)

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3, k0, k1 = [np.asarray(a, dtype=np.uint32) for a in np.broadcast_arrays(c0, c1, c2, c3, k0, k1)]
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            if r < 9:
                k0 = (k0 + _W0).astype(np.uint32)
                k1 = (k1 + _W1).astype(np.uint32)
    return c0, c1, c2, c3


def _u53(hi, lo):
    """(0,1) uniform from 64 random bits: ((hi << 21 | lo >> 11) + 0.5) * 2^-53."""
    x = (hi.astype(np.uint64) << np.uint64(21)) | (lo.astype(np.uint64) >> np.uint64(11))
    return (x.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal_pair(seed, traj, step, block, purpose, chunk):
    """Two standard normals (z0, z1) for each broadcast combination of the integer arguments."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32)
    c2 = (np.asarray(block, dtype=np.uint32) | (np.asarray(purpose, dtype=np.uint32) << np.uint32(16)))
    r0, r1, r2, r3 = philox4x32_10(np.asarray(traj, dtype=np.uint32), np.asarray(step, dtype=np.uint32),
                                   c2, np.asarray(chunk, dtype=np.uint32), k0, k1)
    u1, u2 = _u53(r0, r1), _u53(r2, r3)
    r = np.sqrt(-2.0 * np.log(u1))
    th = (2.0 * np.pi) * u2
    return r * np.cos(th), r * np.sin(th)


def normals(seed, traj, step, n_block, n_bstate, purpose):
    """Standard normals of shape (len(traj), n_block, n_bstate) for one (step, purpose)."""
    traj = np.atleast_1d(np.asarray(traj, dtype=np.uint32))
    n_chunk = (n_bstate + 1) // 2
    z0, z1 = normal_pair(seed, traj[:, None, None], step, np.arange(n_block, dtype=np.uint32)[None, :, None],
                         purpose, np.arange(n_chunk, dtype=np.uint32)[None, None, :])
    z = np.stack([z0, z1], axis=-1).reshape(len(traj), n_block, 2 * n_chunk)
    return z[..., :n_bstate]
