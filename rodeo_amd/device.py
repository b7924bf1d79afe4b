"""
Device plumbing above the C ABI: one ``Device`` = one rk_handle (one HIP device + one stream), and ``DeviceArray``
= a device buffer tagged with its logical per-trajectory shape in the batch-minor layout of
include/rodeo_kalman.h (element e of trajectory b at ``ptr[e * B + b]``).
"""
import ctypes as C
import os
import numpy as np
from . import _lib


class Device:
    def __init__(self, device_id=None):
        lib = _lib.load()
        if device_id is None:
            device_id = int(os.environ.get("LOCAL_RANK", "0"))
            n = C.c_int(0)
            _lib.check(lib.rk_device_count(C.byref(n)))
            device_id = device_id % max(n.value, 1)
        self.lib = lib
        self.device_id = device_id
        h = C.c_void_p()
        _lib.check(lib.rk_create(device_id, C.byref(h)))
        self.h = h
        self.rank, self.nranks = 0, 1

    # ---- memory ----
    def alloc(self, nbytes):
        p = C.c_void_p()
        _lib.check(self.lib.rk_alloc(self.h, nbytes, C.byref(p)))
        return p

    def free(self, ptr):
        if self.h:
            _lib.check(self.lib.rk_free(self.h, ptr))

    def sync(self):
        _lib.check(self.lib.rk_sync(self.h))

    def name(self):
        buf = C.create_string_buffer(256)
        _lib.check(self.lib.rk_device_name(self.h, buf, 256))
        return buf.value.decode()

    # ---- arrays ----
    def empty(self, shape, dtype=np.float64, pad_bytes=0):
        return DeviceArray(self, tuple(int(s) for s in shape), np.dtype(dtype), pad_bytes)

    def zeros(self, shape, dtype=np.float64):
        a = self.empty(shape, dtype)
        _lib.check(self.lib.rk_memset(self.h, a.ptr, 0, a.nbytes))
        return a

    def to_device(self, host):
        host = np.ascontiguousarray(host)
        a = self.empty(host.shape, host.dtype)
        if a.nbytes:
            _lib.check(self.lib.rk_h2d(self.h, a.ptr, host.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    # ---- timing on the handle's stream ----
    def timer_start(self):
        _lib.check(self.lib.rk_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_double(0.0)
        _lib.check(self.lib.rk_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True, keep=False):
        """HIP events around every launch; ``keep=True`` accumulates over calls (read them all with ``profile_last(cap)``)."""
        _lib.check(self.lib.rk_profile_enable(self.h, (2 if keep else 1) if on else 0))

    def profile_last(self, cap=16):
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        n = C.c_int(0)
        _lib.check(self.lib.rk_profile_last(self.h, cap, names, ms, C.byref(n)))
        return [(names[i].decode(), ms[i]) for i in range(min(n.value, cap))]

    def close(self):
        if getattr(self, "h", None):
            self.lib.rk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceArray:
    """A device buffer of a given shape/dtype (C-contiguous).  Freed when garbage-collected."""

    def __init__(self, dev, shape, dtype, pad_bytes=0):
        self.dev, self.shape, self.dtype = dev, shape, dtype
        self.nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        self.ptr = dev.alloc(self.nbytes + pad_bytes) if self.nbytes else C.c_void_p()

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            _lib.check(self.dev.lib.rk_d2h(self.dev.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def upload(self, host):
        """Overwrite the buffer with a host array of the same shape and dtype (asynchronous-safe: rk_h2d synchronises)."""
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.shape != tuple(self.shape):
            raise ValueError(f"upload: shape {host.shape} != {tuple(self.shape)}")
        if self.nbytes:
            _lib.check(self.dev.lib.rk_h2d(self.dev.h, self.ptr, host.ctypes.data_as(C.c_void_p), self.nbytes))

    def copy_from(self, other, count=None):
        """Device-to-device copy of the first ``count`` elements of ``other`` (default: all) into this buffer's start; asynchronous
        on the handle's stream (``rk_d2d``)."""
        count = int(np.prod(other.shape, dtype=np.int64)) if count is None else int(count)
        nb = count * self.dtype.itemsize
        if other.dtype != self.dtype or nb > self.nbytes or nb > other.nbytes:
            raise ValueError("copy_from: dtype / size mismatch")
        if nb:
            _lib.check(self.dev.lib.rk_d2d(self.dev.h, self.ptr, other.ptr, nb))

    def slice0_host(self, i):
        """Download only the i-th slice along the leading axis."""
        out = np.empty(self.shape[1:], self.dtype)
        nb = out.nbytes
        i = i % self.shape[0]
        if nb:
            src = C.c_void_p(self.ptr.value + i * nb)
            _lib.check(self.dev.lib.rk_d2h(self.dev.h, out.ctypes.data_as(C.c_void_p), src, nb))
        return out

    def batch_first(self):
        """Download a batch-minor array (S..., B) and return the (B, S...) strided view (no extra copy)."""
        return np.moveaxis(self.to_host(), -1, 0)

    def __del__(self):
        try:
            if self.nbytes and self.dev.h:
                self.dev.free(self.ptr)
        except Exception:
            pass


_default = None


def default_device():
    """Process-wide default handle (device LOCAL_RANK % n_devices)."""
    global _default
    if _default is None:
        _default = Device()
    return _default


def batch_minor(host, batched):
    """(B, S...) -> contiguous (S..., B) for batched inputs; shared inputs stay (S...)."""
    host = np.asarray(host, dtype=np.float64)
    if batched:
        return np.ascontiguousarray(np.moveaxis(host, 0, -1))
    return np.ascontiguousarray(host)
