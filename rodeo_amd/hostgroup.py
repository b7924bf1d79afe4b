"""
Host-side rendezvous of the one-process-per-GPU ranks, standard library only (no torch, no MPI).

The data path has no collective: ranks need a host channel only to hand round the 128-byte
``ncclUniqueId`` before ``rk_comm_init``, for the barrier / max-over-ranks around a timed region, and -- when no RCCL
communicator is available (CPU tests, rehearsals) -- to all-gather per-draw scalars.  A star over TCP through rank 0
carries that: every operation is "all ranks send a frame to rank 0, rank 0 answers every rank".

Where rank 0 listens is agreed through a small file (``<rdzv_dir>/rank0.addr``): under ``torch.distributed.run`` the
MASTER_PORT itself belongs to the launcher's own store, so rank 0 binds an ephemeral port and publishes it.  The
directory comes from RK_RDZV_DIR (set by ``bench.py`` when it spawns its own ranks) or is derived from MASTER_PORT
and the launcher's run id; a stale file of an earlier run is harmless (connection refused or a failed handshake make
the client read the file again until the deadline).

Every blocking call has a deadline and raises ``TimeoutError`` -- a missing rank must end the job, never hang it.
"""
import json
import os
import socket
import struct
import tempfile
import time

_MAGIC = b"RKHG1"


def _send_frame(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        buf += chunk
    return bytes(buf)


def _recv_frame(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > (1 << 30):
        raise ConnectionError("rendezvous frame too large")
    return _recv_exact(sock, n)


def default_rdzv_dir(env=None):
    env = os.environ if env is None else env
    if env.get("RK_RDZV_DIR"):
        return env["RK_RDZV_DIR"]
    tag = "%s_%s_%d" % (env.get("MASTER_PORT", "0"), env.get("TORCHELASTIC_RUN_ID", "none"), os.getuid())
    return os.path.join(tempfile.gettempdir(), "rk_rdzv_" + tag)


def _private_dir(path):
    """
    The rendezvous directory holds rank 0's address AND the nonce that authenticates the handshake, under a predictable
    name in /tmp when a launcher gives no RK_RDZV_DIR: it must be ours alone.  Created with mode 0700; an existing one is
    refused unless it is a real directory (no symlink) owned by this user and closed to group and others.
    """
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    except FileNotFoundError:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        try:
            os.mkdir(path, 0o700)
        except FileExistsError:
            pass
    st = os.lstat(path)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
        raise PermissionError(f"rendezvous directory {path} is not a directory owned by uid {os.getuid()}")
    if st.st_mode & 0o077:
        os.chmod(path, 0o700)                       # ours, but open: close it (a directory another rank of ours just made)


class HostGroup:
    """rank / world + bcast, allgather, allreduce, barrier over a TCP star through rank 0."""

    def __init__(self, rank, world, rdzv_dir=None, addr="127.0.0.1", timeout=120.0):
        if not (0 <= rank < world):
            raise ValueError("rank out of range")
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self._peers = []           # rank 0: sockets indexed by rank - 1
        self._sock = None          # other ranks: the connection to rank 0
        self._addr_file = None
        if world == 1:
            return
        rdzv_dir = rdzv_dir or default_rdzv_dir()
        _private_dir(rdzv_dir)
        path = os.path.join(rdzv_dir, "rank0.addr")
        deadline = time.monotonic() + self.timeout
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, 0))
            srv.listen(world)
            nonce = "%d.%d" % (os.getpid(), time.time_ns())
            tmp = path + ".%d" % os.getpid()
            with open(tmp, "w") as f:
                json.dump({"addr": addr, "port": srv.getsockname()[1], "nonce": nonce, "world": world}, f)
            os.replace(tmp, path)                        # atomic publish
            self._addr_file = path
            peers = [None] * (world - 1)
            try:
                while any(p is None for p in peers):
                    left = deadline - time.monotonic()
                    if left <= 0:
                        raise TimeoutError("rendezvous: ranks %s never connected" %
                                           [i + 1 for i, p in enumerate(peers) if p is None])
                    srv.settimeout(left)
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        continue
                    conn.settimeout(self.timeout)
                    try:
                        hello = json.loads(_recv_frame(conn))
                        ok = (hello.get("magic") == _MAGIC.decode() and hello.get("nonce") == nonce
                              and hello.get("world") == world and 1 <= hello.get("rank", 0) < world
                              and peers[hello["rank"] - 1] is None)
                    except Exception:                    # noqa: BLE001 -- a stray connection is not ours
                        ok = False
                    if not ok:
                        conn.close()
                        continue
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send_frame(conn, _MAGIC)
                    peers[hello["rank"] - 1] = conn
            finally:
                srv.close()
            self._peers = peers
        else:
            while True:
                if time.monotonic() > deadline:
                    raise TimeoutError("rendezvous: rank 0 did not appear at %s" % path)
                try:
                    with open(path) as f:
                        info = json.load(f)
                    if info.get("world") != world:
                        raise ValueError("stale rendezvous file")
                    s = socket.create_connection((info["addr"], info["port"]), timeout=5.0)
                    s.settimeout(self.timeout)
                    _send_frame(s, json.dumps({"magic": _MAGIC.decode(), "nonce": info["nonce"], "world": world,
                                               "rank": rank}).encode())
                    if _recv_frame(s) != _MAGIC:
                        raise ConnectionError("bad handshake")
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._sock = s
                    break
                except (OSError, ValueError, ConnectionError):
                    time.sleep(0.05)

    @classmethod
    def from_env(cls, timeout=120.0):
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        return cls(rank, world, timeout=timeout)

    # ---- the one primitive: everyone contributes bytes, everyone gets the list of all contributions ----
    def allgather_bytes(self, payload):
        if self.world == 1:
            return [bytes(payload)]
        if self.rank == 0:
            parts = [bytes(payload)] + [_recv_frame(p) for p in self._peers]
            blob = struct.pack("<I", len(parts)) + b"".join(struct.pack("<Q", len(x)) + x for x in parts)
            for p in self._peers:
                _send_frame(p, blob)
            return parts
        _send_frame(self._sock, bytes(payload))
        blob = _recv_frame(self._sock)
        (n,) = struct.unpack_from("<I", blob, 0)
        off, parts = 4, []
        for _ in range(n):
            (ln,) = struct.unpack_from("<Q", blob, off)
            off += 8
            parts.append(blob[off:off + ln])
            off += ln
        return parts

    def bcast_bytes(self, payload, src=0):
        return self.allgather_bytes(payload if self.rank == src else b"")[src]

    def allgather(self, obj):
        """JSON-serialisable objects (numbers, lists, strings) from every rank, in rank order."""
        return [json.loads(x) for x in self.allgather_bytes(json.dumps(obj).encode())]

    def allreduce(self, x, op="max"):
        vals = self.allgather(x)
        return {"max": max, "min": min, "sum": sum}[op](vals)

    def allgather_f64(self, arr):
        """1-D float64 arrays (ragged allowed) from every rank, as a list of NumPy arrays in rank order."""
        import numpy as np
        a = np.ascontiguousarray(arr, dtype=np.float64)
        return [np.frombuffer(x, dtype=np.float64).copy() for x in self.allgather_bytes(a.tobytes())]

    def barrier(self):
        self.allgather_bytes(b"")

    def close(self):
        for p in self._peers:
            try:
                p.close()
            except OSError:
                pass
        self._peers = []
        if self._sock is not None:
            try:
                self._sock.close()
            except OSError:
                pass
            self._sock = None
        if self._addr_file:
            try:
                os.remove(self._addr_file)
            except OSError:
                pass
            self._addr_file = None


def spawn_ranks(argv, world, env=None, timeout=None):
    """
    Start ``world`` rank processes of ``argv`` (RANK / LOCAL_RANK / WORLD_SIZE / RK_RDZV_DIR set) from a parent that has
    touched no GPU, wait for all of them, and return the largest exit code (124 when the deadline passed).  If one rank fails or the deadline passes
    the others are terminated by PID, so a broken rank never leaves the job hanging.
    """
    import shutil
    import subprocess
    rdzv = tempfile.mkdtemp(prefix="rk_rdzv_")
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(world), "RK_RDZV_DIR": rdzv, "MASTER_ADDR": "127.0.0.1"})
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    try:
        for r in range(world):
            e = dict(base)
            e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
            procs.append(subprocess.Popen(argv, env=e))
        deadline = None if timeout is None else time.monotonic() + timeout
        codes = [None] * world
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            failed = any(c not in (None, 0) for c in codes)
            late = deadline is not None and time.monotonic() > deadline
            if failed or late:
                own_failures = [abs(c) for c in codes if c not in (None, 0)]     # before anything is terminated here
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        p.terminate()
                t_kill = time.monotonic() + 10
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        try:
                            codes[i] = p.wait(timeout=max(0.1, t_kill - time.monotonic()))
                        except subprocess.TimeoutExpired:
                            p.kill()
                            codes[i] = p.wait()
                return max(own_failures) if failed else 124
            time.sleep(0.05)
        return max(abs(c) for c in codes)
    finally:
        shutil.rmtree(rdzv, ignore_errors=True)
