"""
Multi-GPU: the path shards by independent trajectories / parameter draws -- one process per
GPU, a contiguous split of the batch axis, NO collective on the data path.  The only exchange is the all-gather of
per-trajectory scalars (e.g. the 8192 log-posteriors of BASELINE config 4), which runs over RCCL/xGMI inside
librodeo_kalman.so (``rk_allgather_f64``) on GPUs, or over a host channel: ``rodeo_amd.hostgroup.HostGroup`` (standard
library) or any ``torch.distributed`` backend (gloo in the CPU tests).

Random draws are keyed by the GLOBAL trajectory index (``traj_offset`` of the C ABI), so results do not depend on the
number of ranks.
"""
import ctypes as C
import numpy as np
from . import _lib


def partition(n_total, rank, nranks):
    """Contiguous split: rank r owns [lo, hi); sizes differ by at most one (first ranks get the remainder)."""
    if not (0 <= rank < nranks):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_total), int(nranks))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard(arr, rank, nranks, batched=True):
    """Slice the leading batch axis of ``arr`` for this rank (shared, un-batched inputs pass through)."""
    if not batched:
        return arr
    lo, hi = partition(np.shape(arr)[0], rank, nranks)
    return arr[lo:hi]


def padded_width(n_total, nranks):
    """Scalars per rank in the equal-width all-gather: ceil(n_total / nranks) (the largest shard of ``partition``)."""
    return -(-int(n_total) // int(nranks))


def trim_padded(flat, n_total, nranks):
    """(nranks * width,) gathered with every shard padded to ``padded_width`` -> (n_total,) in global trajectory order."""
    width = padded_width(n_total, nranks)
    flat = np.asarray(flat, dtype=np.float64).reshape(nranks, width)
    return np.concatenate([flat[r, :partition(n_total, r, nranks)[1] - partition(n_total, r, nranks)[0]]
                           for r in range(nranks)])


def gather_scalars(local, n_total, rank, nranks, dist=None):
    """
    All-gather per-trajectory scalars (n_local,) -> (n_total,) in global trajectory order over a HOST channel:
    ``dist`` is a ``rodeo_amd.hostgroup.HostGroup`` (standard-library TCP star, what bench.py and the scripts use) or a
    ``torch.distributed``-like module / process group (any backend; gloo in the CPU tests).  Ragged shards travel padded to
    the common width ``padded_width`` and are trimmed on arrival -- the same scheme as ``gather_scalars_device`` (RCCL).
    """
    local = np.ascontiguousarray(local, dtype=np.float64)
    lo, hi = partition(n_total, rank, nranks)
    if local.shape != (hi - lo,):
        raise ValueError(f"rank {rank} holds {local.shape} scalars, its shard has {hi - lo}")
    if nranks == 1:
        return local.copy()
    width = padded_width(n_total, nranks)
    buf = np.zeros(width)
    buf[:hi - lo] = local
    if hasattr(dist, "allgather_f64"):                    # HostGroup
        parts = dist.allgather_f64(buf)
        if any(np.shape(p_) != (width,) for p_ in parts):
            raise ValueError(f"a rank sent {[np.shape(p_) for p_ in parts]} scalars, the padded width is {width}")
        return trim_padded(np.concatenate(parts), n_total, nranks)
    import torch
    import torch.distributed as tdist
    dist = dist or tdist
    outs = [torch.zeros(width, dtype=torch.float64) for _ in range(nranks)]
    dist.all_gather(outs, torch.from_numpy(buf))
    return trim_padded(torch.cat(outs).numpy(), n_total, nranks)


def gather_scalars_device(local_dev, n_total, rank, nranks, comm, device):
    """
    The same all-gather over RCCL / xGMI from a DEVICE array (n_local,) of this rank's scalars (e.g.
    ``FitzLogPosterior.device``): ``rk_allgather_f64`` with every shard padded to ``padded_width`` in a device staging buffer
    (equal counts are what ncclAllGather takes; the pad of a short shard is whatever the buffer held and is trimmed away),
    one download of the result.  Ragged shards included; nothing passes through the host on the way in.
    """
    lo, hi = partition(n_total, rank, nranks)
    if tuple(local_dev.shape) != (hi - lo,):
        raise ValueError(f"rank {rank} holds {tuple(local_dev.shape)} scalars, its shard has {hi - lo}")
    width = padded_width(n_total, nranks)
    cache = device.__dict__.setdefault("_gather_bufs", {})
    if cache.get("key") != (width, nranks):
        cache["key"], cache["send"], cache["recv"] = (width, nranks), device.zeros((width,)), device.empty((nranks * width,))
    send, recv = cache["send"], cache["recv"]
    send.copy_from(local_dev, hi - lo)
    comm.allgather(send, recv, width)
    return trim_padded(recv.to_host(), n_total, nranks)


def init_rccl_or_fail(device, group, deadline=90.0):
    """
    Bring up the RCCL communicator of ``device`` for the ranks of ``group`` (a HostGroup carries the unique id).
    Returns ``(comm, must_hard_exit, why)``: ``comm`` is an ``RcclComm`` when EVERY rank got one, else None (all ranks
    agree on the outcome; ``why`` then says what went wrong on this rank).  ``rk_comm_init`` runs in a helper thread
    with a deadline; if any rank's bootstrap is stuck, ``must_hard_exit`` is True on ALL ranks: they must leave the
    process with ``os._exit`` once their work is done (a thread is still inside RCCL) -- a stuck bootstrap must not
    hang the job.
    """
    import threading
    state = {"comm": None, "err": None}
    uid = b""
    if group.rank == 0:
        buf = (C.c_char * _lib.COMM_UID_BYTES)()
        try:
            _lib.check(device.lib.rk_comm_uid(buf))
            uid = bytes(buf)
        except Exception as e:                           # noqa: BLE001
            state["err"] = e
    uid = group.bcast_bytes(uid, src=0)
    stuck = False
    if len(uid) == _lib.COMM_UID_BYTES:
        def _init():
            try:
                state["comm"] = RcclComm(device, group.rank, group.world, uid=uid)
            except Exception as e:                       # noqa: BLE001
                state["err"] = e
        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(timeout=deadline)
        stuck = th.is_alive()
    ok = state["comm"] is not None and not stuck
    all_ok = group.allreduce(1 if ok else 0, "min") == 1
    any_stuck = group.allreduce(1 if stuck else 0, "max") == 1
    if all_ok:
        return state["comm"], False, ""
    if ok:
        try:
            state["comm"].close()
        except Exception:                                # noqa: BLE001
            pass
    why = "bootstrap timed out" if stuck else (state["err"] or "another rank failed")
    return None, bool(stuck or any_stuck), str(why)


class RcclComm:
    """RCCL communicator bound to a ``Device`` handle; the 128-byte unique id travels over the host launcher."""

    def __init__(self, device, rank, nranks, uid=None, bcast=None):
        self.dev, self.rank, self.nranks = device, rank, nranks
        lib = device.lib
        buf = (C.c_char * _lib.COMM_UID_BYTES)()
        if uid is None:
            if rank == 0:
                _lib.check(lib.rk_comm_uid(buf))
            if bcast is None:
                raise ValueError("need either uid or a bcast(bytes, src=0) callable")
            uid = bcast(bytes(buf))
        buf = (C.c_char * _lib.COMM_UID_BYTES).from_buffer_copy(uid)
        _lib.check(lib.rk_comm_init(device.h, rank, nranks, buf))

    def allgather(self, send, recv, count_per_rank):
        """Device arrays: send (count,), recv (nranks * count,)."""
        _lib.check(self.dev.lib.rk_allgather_f64(self.dev.h, send.ptr, recv.ptr, count_per_rank))

    def barrier(self):
        _lib.check(self.dev.lib.rk_comm_barrier(self.dev.h))

    def close(self):
        _lib.check(self.dev.lib.rk_comm_destroy(self.dev.h))
