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


def gather_scalars(local, n_total, rank, nranks, dist=None):
    """
    All-gather per-trajectory scalars (n_local,) -> (n_total,) in global trajectory order over a HOST channel:
    ``dist`` is a ``rodeo_amd.hostgroup.HostGroup`` (standard-library TCP star, what bench.py and the scripts use) or a
    ``torch.distributed``-like module / process group (any backend; gloo in the CPU tests).  Ragged shards are fine.
    On GPUs with an RCCL communicator use ``RcclComm.allgather`` instead (device buffers, xGMI).
    """
    local = np.ascontiguousarray(local, dtype=np.float64)
    if nranks == 1:
        return local.copy()
    if hasattr(dist, "allgather_f64"):                    # HostGroup
        parts = dist.allgather_f64(local)
        for r in range(nranks):
            lo, hi = partition(n_total, r, nranks)
            if parts[r].shape[0] != hi - lo:
                raise ValueError(f"rank {r} sent {parts[r].shape[0]} scalars, its shard has {hi - lo}")
        return np.concatenate(parts)
    import torch
    import torch.distributed as tdist
    dist = dist or tdist
    width = -(-int(n_total) // nranks)
    buf = torch.zeros(width, dtype=torch.float64)
    buf[:local.shape[0]] = torch.from_numpy(local)
    outs = [torch.zeros(width, dtype=torch.float64) for _ in range(nranks)]
    dist.all_gather(outs, buf)
    parts = []
    for r in range(nranks):
        lo, hi = partition(n_total, r, nranks)
        parts.append(outs[r][:hi - lo].numpy())
    return np.concatenate(parts)


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
