"""
N > 1 path on CPU: world_size-2 gloo processes shard a batch of parameter draws, each evaluates its shard (with the
oracle standing in for the device here -- there is no GPU in this container), and the all-gather of per-draw scalars
reproduces the single-process result exactly.  Draws are keyed by the global index, so sharding does not change them.
"""
import os
import socket
import numpy as np
import pytest
from rodeo_amd import shard as rs

# torch (its gloo backend) is imported only inside the tests that use it: a module-level import would load the wheel's own ROCm
# runtime into every pytest process that merely COLLECTS this file -- including the `-m gpu` run, whose kernels and
# run-time builds would then go through that runtime instead of the system's (tests/test_user_rhs.py,
# test_run_time_builds_survive_another_rocm_in_the_process).


def test_partition_covers_everything():
    for n, g in [(8192, 8), (1000, 3), (5, 8), (0, 2), (7, 1)]:
        spans = [rs.partition(n, r, g) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        rs.partition(10, 3, 3)
    x = np.arange(10)
    assert np.array_equal(np.concatenate([rs.shard(x, r, 3) for r in range(3)]), x)
    assert rs.shard(x, 1, 3, batched=False) is x


def test_padded_ragged_gather_layout():
    """The equal-width all-gather every channel uses (gloo, host TCP, RCCL): shards padded to ceil(n / G), trimmed on
    arrival.  Three ranks, ten scalars (shards 4, 3, 3): whatever sits in the pads (NaN here, stale device memory over
    RCCL) never reaches the result."""
    n, g = 10, 3
    w = rs.padded_width(n, g)
    assert w == 4 and rs.padded_width(8192, 8) == 1024 and rs.padded_width(5, 8) == 1
    want = np.arange(n, dtype=np.float64) * 1.5
    sent = []
    for r in range(g):
        lo, hi = rs.partition(n, r, g)
        buf = np.full(w, np.nan)
        buf[:hi - lo] = want[lo:hi]
        sent.append(buf)
    np.testing.assert_array_equal(rs.trim_padded(np.concatenate(sent), n, g), want)
    # the device variant packs through a staging buffer of that width and one rk_allgather_f64: same layout (fake device)

    class Arr:
        def __init__(self, a): self.a, self.shape = np.asarray(a, dtype=np.float64), np.shape(a)
        def copy_from(self, other, count): self.a[:count] = other.a[:count]
        def to_host(self): return self.a.copy()

    class Dev:
        def zeros(self, shape): return Arr(np.zeros(shape))
        def empty(self, shape): return Arr(np.full(shape, np.nan))

    class Comm:                                           # what ncclAllGather does, with the other ranks' buffers filled in
        def __init__(self, rank): self.rank = rank
        def allgather(self, send, recv, count):
            for r in range(g):
                recv.a[r * count:(r + 1) * count] = send.a if r == self.rank else sent[r]
    for r in range(g):
        lo, hi = rs.partition(n, r, g)
        full = rs.gather_scalars_device(Arr(want[lo:hi]), n, r, g, Comm(r), Dev())
        np.testing.assert_array_equal(full, want)
    with pytest.raises(ValueError):
        rs.gather_scalars_device(Arr(want[:2]), n, 0, g, Comm(0), Dev())
    with pytest.raises(ValueError):
        rs.gather_scalars(want[:2], n, 0, g, None)


def _logpost_of_shard(lo, hi, seed):
    """Per-draw scalar of a chkrebtii solve_sim for global draws [lo, hi) (oracle stands in for the GPU on CPU)."""
    import functools
    from oracle import scan, odes, priors, interrogations as oi
    rng = np.random.default_rng(20242)
    n_tot = 10
    u = np.array([np.log(.2), np.log(.2), np.log(3.), -1., 1.]) + 0.01 * rng.standard_normal((n_tot, 5))
    theta = np.exp(u[lo:hi, :3]); x0v = u[lo:hi, 3:5]
    W, init = priors.first_order_pad(odes.fitzhugh_nagumo, 2, 3)
    x0 = np.stack([init(x0v[i], 0., theta=theta[i]) for i in range(hi - lo)])
    prior = priors.ibm_init(0.1, 3, np.array([.1, .1]))
    itg = functools.partial(oi.interrogate_chkrebtii, kalman_type="standard")
    x = scan.solve_sim(seed, odes.fitzhugh_nagumo, W, x0, 0., 2., 20, itg, prior, traj_offset=lo, theta=theta)
    return x[:, -1, 0, 0] + x[:, 10, 1, 0]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rs.partition(10, rank, world)
    local = _logpost_of_shard(lo, hi, seed=7)
    full = rs.gather_scalars(local, 10, rank, world)
    if rank == 0:
        q.put(full)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single_process(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(full, _logpost_of_shard(0, 10, seed=7))


# ---- the standard-library host channel (what bench.py / scripts use instead of torch.distributed) ----------------------
def _hg_worker(rank, world, rdzv, q):
    from rodeo_amd.hostgroup import HostGroup
    g = HostGroup(rank, world, rdzv_dir=rdzv, timeout=60)
    assert g.bcast_bytes(b"uid-%d" % rank, src=0) == b"uid-0"
    assert g.allgather(rank * 10) == [r * 10 for r in range(world)]
    assert g.allreduce(float(rank), "max") == world - 1 and g.allreduce(rank + 1, "min") == 1
    g.barrier()
    lo, hi = rs.partition(10, rank, world)
    full = rs.gather_scalars(_logpost_of_shard(lo, hi, seed=7), 10, rank, world, g)
    if rank == 0:
        q.put(full)
    g.barrier()
    g.close()


@pytest.mark.parametrize("world", [2, 3])
def test_hostgroup_sharded_equals_single_process(world, tmp_path):
    import multiprocessing
    ctx = multiprocessing.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hg_worker, args=(r, world, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(full, _logpost_of_shard(0, 10, seed=7))


def test_hostgroup_missing_rank_times_out(tmp_path):
    from rodeo_amd.hostgroup import HostGroup
    with pytest.raises(TimeoutError):
        HostGroup(0, 2, rdzv_dir=str(tmp_path), timeout=0.5)          # rank 1 never shows up: error, not a hang
    with pytest.raises(TimeoutError):
        HostGroup(1, 2, rdzv_dir=str(tmp_path / "nobody"), timeout=0.5)


def test_spawn_ranks_failed_rank_ends_the_job(tmp_path):
    """One rank exits non-zero (a failed RCCL bootstrap in bench.py --require-rccl): the launcher terminates the others
    by PID and reports the failure -- no hang."""
    import sys, time
    from rodeo_amd.hostgroup import spawn_ranks
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                      "time.sleep(120)\n")
    t0 = time.monotonic()
    assert spawn_ranks([sys.executable, str(script)], 3) == 3
    assert time.monotonic() - t0 < 30
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nassert os.environ['WORLD_SIZE'] == '2' and os.path.isdir(os.environ['RK_RDZV_DIR'])\n")
    assert spawn_ranks([sys.executable, str(ok)], 2) == 0


def test_rendezvous_directory_is_private(tmp_path):
    """The rendezvous directory carries the handshake nonce: created 0700, an open one of ours is closed, a symlink or a
    directory of another owner is refused (rodeo_amd/hostgroup.py _private_dir)."""
    import stat
    from rodeo_amd import hostgroup as hg
    d = tmp_path / "rdzv"
    hg._private_dir(str(d))
    assert stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    os.chmod(d, 0o755)
    hg._private_dir(str(d))
    assert stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    target = tmp_path / "elsewhere"
    target.mkdir()
    link = tmp_path / "link"
    os.symlink(target, link)
    with pytest.raises(PermissionError):
        hg._private_dir(str(link))
    f = tmp_path / "file"
    f.write_text("x")
    with pytest.raises(PermissionError):
        hg._private_dir(str(f))
