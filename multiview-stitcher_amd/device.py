"""Device-resident arrays: the tile residency the reference lacks.

The reference's CuPy backend copies every chunk host->device->host
(fusion/_core.py:1584-1587, 1716-1721).  A ``DeviceArray`` keeps a tile (or a
fused result) in HBM so that registration and fusion share one upload; basic
slicing returns a zero-copy strided window, which is how chunk slabs are handed
to ``mvs_fuse_chunk``.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class DeviceArray:
    def __init__(self, buf, ptr, shape, strides, dtype, device):
        self._buf = buf            # DeviceBuffer keeping the allocation alive (or any owner object)
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.strides = tuple(int(s) for s in strides)   # in elements
        self.dtype = np.dtype(dtype)
        self.device = int(device)

    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape)))
    nbytes = property(lambda self: self.size * self.dtype.itemsize)

    @classmethod
    def from_host(cls, array, device=0):
        array = np.ascontiguousarray(array)
        if array.dtype not in _lib.DTYPE_CODES:
            raise TypeError(f"unsupported dtype {array.dtype} (uint8/uint16/float32)")
        buf = _lib.DeviceBuffer(device, max(array.nbytes, 1)).upload(array)
        strides = [s // array.itemsize for s in array.strides]
        return cls(buf, buf.ptr, array.shape, strides, array.dtype, device)

    @classmethod
    def empty(cls, shape, dtype, device=0):
        dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in shape)
        buf = _lib.DeviceBuffer(device, max(int(np.prod(shape)) * dtype.itemsize, 1))
        strides = [int(np.prod(shape[i + 1:])) for i in range(len(shape))]
        return cls(buf, buf.ptr, shape, strides, dtype, device)

    @classmethod
    def from_pointer(cls, ptr, shape, dtype, device=0, owner=None):
        """Wrap foreign device memory (e.g. ``torch_tensor.data_ptr()``); ``owner`` is kept alive."""
        shape = tuple(int(s) for s in shape)
        strides = [int(np.prod(shape[i + 1:])) for i in range(len(shape))]
        return cls(owner, ptr, shape, strides, dtype, device)

    def on_device(self, device):
        """This array on GPU ``device`` (the low byte of a context id; the high bits are a lane): itself when it already
        lives there, otherwise a peer copy (mvs_memcpy_peer) of the spanned range with the same shape and strides.
        Copies of whole library-owned allocations are cached on the owner, so a tile is fetched once per device; the
        cache entry carries the owner's write version (bumped by upload / fill_zero / copy_into / kernels writing into
        ``out=``), so a copy taken before a later write is refreshed (into the same peer allocation) instead of being
        handed out stale.  ``drop_peer_copies()`` releases them."""
        dev = int(device)
        if (self.device & 0xff) == (dev & 0xff):
            return self
        item = self.dtype.itemsize
        owner = self._buf
        lib = _lib.init(dev)
        if isinstance(owner, _lib.DeviceBuffer) and owner.ptr:
            cache = owner.__dict__.setdefault("_peer_copies", {})
            entry = cache.get(dev & 0xff)
            if entry is None or entry[1] != owner.version:
                peer = entry[0] if entry is not None else _lib.DeviceBuffer(dev, owner.nbytes)
                _lib.check(lib.mvs_memcpy_peer(dev, C.c_void_p(peer.ptr), self.device, C.c_void_p(owner.ptr), owner.nbytes), dev, "mvs_memcpy_peer")
                cache[dev & 0xff] = entry = (peer, owner.version)
            peer = entry[0]
            return DeviceArray(peer, peer.ptr + (self.ptr - owner.ptr), self.shape, self.strides, self.dtype, dev)
        span = sum((n - 1) * st for n, st in zip(self.shape, self.strides)) + 1
        buf = _lib.DeviceBuffer(dev, span * item)
        _lib.check(lib.mvs_memcpy_peer(dev, C.c_void_p(buf.ptr), self.device, C.c_void_p(self.ptr), span * item), dev, "mvs_memcpy_peer")
        return DeviceArray(buf, buf.ptr, self.shape, self.strides, self.dtype, dev)

    def mark_written(self):
        """Tell the owning allocation that its contents changed (invalidates peer copies on other GPUs)."""
        if isinstance(self._buf, _lib.DeviceBuffer):
            self._buf.mark_written()

    def drop_peer_copies(self):
        """Release the copies of this array's allocation that ``on_device`` left on other GPUs."""
        if isinstance(self._buf, _lib.DeviceBuffer):
            self._buf.__dict__.pop("_peer_copies", None)

    def fill_zero(self):
        """Stream-ordered zero fill (contiguous arrays)."""
        if not self.is_contiguous():
            raise ValueError("fill_zero needs a contiguous array")
        _lib.check(_lib.init(self.device).mvs_memset(self.device, C.c_void_p(self.ptr), 0, self.nbytes), self.device, "mvs_memset")
        self.mark_written()

    def copy_into(self, dst, offset):
        """Copy this (contiguous) array into the window of the contiguous DeviceArray ``dst`` that starts at ``offset``
        (device to device, stream-ordered: mvs_copy_into)."""
        if not (self.is_contiguous() and dst.is_contiguous()) or self.dtype != dst.dtype or self.ndim > dst.ndim or dst.ndim > 3:
            raise ValueError("copy_into needs contiguous arrays of one dtype, the source rank <= the destination rank <= 3")
        if len(offset) != dst.ndim:
            raise ValueError("one offset per destination axis")
        s3 = (1,) * (3 - self.ndim) + tuple(self.shape)          # a lower-rank source is a box with leading extent 1
        d3 = (1,) * (3 - dst.ndim) + tuple(dst.shape)
        o3 = (0,) * (3 - dst.ndim) + tuple(int(v) for v in offset)
        rc = _lib.init(dst.device).mvs_copy_into(dst.device, C.c_void_p(self.ptr), _lib.DTYPE_CODES[self.dtype], _lib.i64x3(s3),
                                                 C.c_void_p(dst.ptr), _lib.i64x3(d3), _lib.i64x3(o3))
        _lib.check(rc, dst.device, "mvs_copy_into")
        dst.mark_written()

    @property
    def __cuda_array_interface__(self):
        """Zero-copy hand-over to torch / cupy-style consumers (``torch.as_tensor(arr, device="cuda")``); the array must
        stay alive while the consumer uses the memory, and the producing call must have been synchronised."""
        item = self.dtype.itemsize
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 3,
                "strides": None if self.is_contiguous() else tuple(s * item for s in self.strides)}

    def is_contiguous(self):
        expect = [int(np.prod(self.shape[i + 1:])) for i in range(self.ndim)]
        return all(s == e or n == 1 for s, e, n in zip(self.strides, expect, self.shape))

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        key = key + (slice(None),) * (self.ndim - len(key))
        off = 0
        shape, strides = [], []
        for k, n, st in zip(key, self.shape, self.strides):
            if isinstance(k, (int, np.integer)):
                k = int(k) + (n if k < 0 else 0)
                off += k * st
            elif isinstance(k, slice):
                lo, hi, step = k.indices(n)
                if step != 1:
                    raise IndexError("DeviceArray supports unit-step slices only")
                off += lo * st
                shape.append(max(hi - lo, 0))
                strides.append(st)
            else:
                raise IndexError("DeviceArray supports ints and slices only")
        return DeviceArray(self._buf, self.ptr + off * self.dtype.itemsize, shape, strides, self.dtype, self.device)

    def get(self):
        """Copy to a numpy array."""
        if self.is_contiguous():
            out = np.empty(self.shape, dtype=self.dtype)
            _lib.check(_lib.load().mvs_memcpy_d2h(self.device, out.ctypes.data, self.ptr, out.nbytes), self.device, "d2h")
            return out
        # strided window: fetch the spanned range and view it
        span = sum((n - 1) * st for n, st in zip(self.shape, self.strides)) + 1
        flat = np.empty(span, dtype=self.dtype)
        _lib.check(_lib.load().mvs_memcpy_d2h(self.device, flat.ctypes.data, self.ptr, flat.nbytes), self.device, "d2h")
        return np.lib.stride_tricks.as_strided(
            flat, self.shape, [s * self.dtype.itemsize for s in self.strides]
        ).copy()

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a.astype(dtype) if dtype is not None else a

    def astype(self, dtype):
        if np.dtype(dtype) == self.dtype:
            return self
        raise TypeError("DeviceArray.astype: dtype conversion happens inside the kernels")

    def __repr__(self):
        return f"<DeviceArray {self.shape} {self.dtype} dev{self.device}>"


def is_device_array(x):
    return isinstance(x, DeviceArray)


def to_device(sim, device=0):
    """Return a copy of a SpatialImage whose data lives on ``device``."""
    if is_device_array(sim.data):
        return sim
    return sim.copy(data=DeviceArray.from_host(sim.data, device))
