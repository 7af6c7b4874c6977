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


_STAGED_GET_MIN_BYTES = [256 << 20]      # DeviceArray.get() of at least this many (contiguous) bytes goes through pinned staging buffers


class _Pending:
    """The upload a DeviceArray's allocation is still waiting for (shared by all windows cut from it): a ticket of mvs_copy_async
    and the pinned host array it reads, kept alive until the ticket has been waited for on the host."""

    __slots__ = ("ticket", "source")

    def __init__(self, ticket, source):
        self.ticket, self.source = int(ticket), source


def pinned_empty(shape, dtype):
    """numpy array in pinned (page-locked) host memory (mvs_host_alloc): what asynchronous uploads read and downloads write.
    The memory is released when the array and all its views are gone."""
    import weakref

    dtype = np.dtype(dtype)
    shape = tuple(int(v) for v in shape)
    nbytes = max(int(np.prod(shape)) * dtype.itemsize, 1)
    lib = _lib.load()
    p = C.c_void_p()
    rc = lib.mvs_host_alloc(nbytes, C.byref(p))
    if rc != 0 or not p.value:
        raise MemoryError(f"mvs_host_alloc({nbytes}) failed with code {rc}")
    raw = (C.c_ubyte * nbytes).from_address(p.value)
    arr = np.frombuffer(raw, dtype=np.uint8, count=int(np.prod(shape)) * dtype.itemsize).view(dtype).reshape(shape)
    weakref.finalize(raw, lib.mvs_host_free, C.c_void_p(p.value))      # (`arr.base` chain holds `raw`)
    return arr


def is_pinned(array):
    """True for arrays made by ``pinned_empty`` (and views of them)."""
    b = array
    while isinstance(b, np.ndarray) and b.base is not None:
        b = b.base
    return isinstance(b, C.Array) or type(b).__name__.startswith("c_ubyte_Array")


class DeviceArray:
    def __init__(self, buf, ptr, shape, strides, dtype, device, pending=None):
        self._buf = buf            # DeviceBuffer keeping the allocation alive (or any owner object)
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.strides = tuple(int(s) for s in strides)   # in elements
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        self._pending = pending    # _Pending of an asynchronous upload that may still be in flight (from_host_async), or None

    @property
    def ready_ticket(self):
        """Ticket of the upload this array still waits for (0: none).  Work queued on a lane after ``wait_ready(lane)`` -- or a
        pair job carrying the ticket (mvs_register_pairs) -- starts when the upload has landed; the host never waits."""
        return self._pending.ticket if self._pending is not None and self._pending.ticket else 0

    def wait_ready(self, device=None):
        """Make the stream of context ``device`` (default: the array's own device, lane 0) wait for the pending upload."""
        t = self.ready_ticket
        if t:
            dev = self.device if device is None else int(device)
            _lib.check(_lib.init(dev).mvs_event_wait(dev, t), dev, "mvs_event_wait")
        return self

    def sync_ready(self):
        """The HOST waits for the pending upload; afterwards the array is an ordinary resident one."""
        t = self.ready_ticket
        if t:
            _lib.check(_lib.load().mvs_ticket_sync(t), self.device, "mvs_ticket_sync")
            self._pending.ticket, self._pending.source = 0, None
        return self

    @classmethod
    def from_host_async(cls, array, device=0):
        """Upload ``array`` (pinned host memory: ``pinned_empty``; contiguous) on the device's copy stream without waiting: the
        returned array carries the upload's ticket (``ready_ticket``).  ``array`` must not be written until the upload has landed."""
        if not (isinstance(array, np.ndarray) and array.flags.c_contiguous and is_pinned(array)):
            raise ValueError("from_host_async needs a C-contiguous array in pinned host memory (device.pinned_empty)")
        if array.dtype not in _lib.DTYPE_CODES:
            raise TypeError(f"unsupported dtype {array.dtype} (uint8/uint16/float32)")
        buf = _lib.DeviceBuffer(device, max(array.nbytes, 1))
        t = C.c_uint64()
        lib = _lib.init(device)
        _lib.check(lib.mvs_copy_async(device, C.c_void_p(buf.ptr), C.c_void_p(array.ctypes.data), array.nbytes, 0, 0, C.byref(t)), device, "mvs_copy_async")
        buf.mark_written()
        strides = [s // array.itemsize for s in array.strides]
        return cls(buf, buf.ptr, array.shape, strides, array.dtype, device, pending=_Pending(t.value, array))

    def download_async(self, out, after=0):
        """Copy this (contiguous) array into the pinned host array ``out`` on the copy stream, after ticket ``after`` (e.g. an
        ``mvs_mark`` on the lane that produces the data); returns the download's ticket (``_lib.ticket_sync`` waits for it)."""
        if not self.is_contiguous() or not (isinstance(out, np.ndarray) and out.flags.c_contiguous and is_pinned(out)) \
                or out.nbytes != self.nbytes or out.dtype != self.dtype:
            raise ValueError("download_async needs a contiguous device array and a pinned C-contiguous host array of the same size and dtype")
        t = C.c_uint64()
        lib = _lib.init(self.device)
        _lib.check(lib.mvs_copy_async(self.device, C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr), self.nbytes, 1, int(after), C.byref(t)),
                   self.device, "mvs_copy_async")
        return int(t.value)

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
        self.sync_ready()      # (a peer copy reads the memory from another GPU's stream)
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

    def copy_box_to(self, dst):
        """Copy this array (any unit-step window of an allocation: rows contiguous) into ``dst``, a window of the same shape and dtype
        on the same device -- device to device, stream-ordered (mvs_copy_box).  <= 3 axes after leading axes of extent 1."""
        if self.dtype != dst.dtype or tuple(self.shape) != tuple(dst.shape) or (self.device & 0xff) != (dst.device & 0xff):
            raise ValueError("copy_box_to needs two windows of one shape and dtype on one device")
        lead = max(self.ndim - 3, 0)
        if any(n != 1 for n in self.shape[:lead]) or self.ndim == 0:
            raise ValueError("copy_box_to handles boxes of up to three axes")
        sh = (1,) * (3 - min(self.ndim, 3)) + tuple(self.shape[lead:])
        item = self.dtype.itemsize

        def pitches(a):
            st = (0,) * (3 - min(a.ndim, 3)) + tuple(a.strides[lead:])
            if sh[2] > 1 and st[2] != 1:
                raise ValueError("copy_box_to needs contiguous rows")
            py = st[1] * item if sh[1] > 1 else sh[2] * item
            pz = st[0] * item if sh[0] > 1 else max(py, 0) * sh[1]
            return (C.c_int64 * 2)(py, pz)

        box = (C.c_int64 * 3)(sh[0], sh[1], sh[2] * item)
        _lib.check(_lib.init(dst.device).mvs_copy_box(dst.device, C.c_void_p(self.ptr), pitches(self), C.c_void_p(dst.ptr), pitches(dst), box),
                   dst.device, "mvs_copy_box")
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
        return DeviceArray(self._buf, self.ptr + off * self.dtype.itemsize, shape, strides, self.dtype, self.device, pending=self._pending)

    def _get_staged(self, piece=128 << 20, depth=3):
        """A large contiguous array to the host through pinned staging buffers: asynchronous downloads of ``piece`` bytes on the copy
        stream (after everything queued on this array's context so far), each copied into the result by the I/O pool while the next
        ones are in flight.  One synchronous copy into pageable memory moves 17 GB/s; this way the link's 50+."""
        from . import streaming

        out = np.empty(self.shape, dtype=self.dtype)
        flat = out.reshape(-1).view(np.uint8)
        pool = streaming.shared_pinned_pool()
        after = mark(self.device)
        inflight = []

        def drain():
            a, n, raw, buf, t = inflight.pop(0)
            ticket_sync(t)
            streaming.parallel_copy(flat[a:a + n], buf, kind="write")
            pool.put(raw)

        for a in range(0, flat.size, piece):
            n = min(piece, flat.size - a)
            raw, buf = pool.get((n,), np.uint8)
            src = DeviceArray(self._buf, self.ptr + a, (n,), (1,), np.uint8, self.device)
            inflight.append((a, n, raw, buf, src.download_async(buf, after=after)))
            if len(inflight) >= depth:
                drain()
        while inflight:
            drain()
        return out

    def get(self):
        """Copy to a numpy array."""
        self.sync_ready()
        if self.is_contiguous():
            if self.nbytes >= _STAGED_GET_MIN_BYTES[0] and _lib.device_count() > 0:
                return self._get_staged()
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


def to_device_async(sims, device=0):
    """Copies of the SpatialImages ``sims`` (spatial dims only; data in pinned host memory, ``pinned_empty``) whose data is being
    uploaded to ``device`` on its copy stream, in list order, without waiting: ``register()`` starts a pair when its two tiles have
    landed, ``fuse()`` / ``fuse_np`` make their stream wait for the tiles they read (``DeviceArray.ready_ticket``).  Views that
    are already on a device, and the metadata-only placeholders of views another rank holds (``sharding.RemoteArray``), pass through."""
    from .sharding import RemoteArray

    return [s if (is_device_array(s.data) or isinstance(s.data, RemoteArray))
            else s.copy(data=DeviceArray.from_host_async(np.asarray(s.data), device)) for s in sims]


def mark(device=0):
    """A timed ticket on the stream of context ``device`` (mvs_mark): passes when the work queued so far is done."""
    t = C.c_uint64()
    _lib.check(_lib.init(device).mvs_mark(device, C.byref(t)), device, "mvs_mark")
    return int(t.value)


def ticket_sync(ticket):
    if ticket:
        _lib.check(_lib.load().mvs_ticket_sync(int(ticket)), 0, "mvs_ticket_sync")


def ticket_elapsed_ms(t0, t1):
    ms = C.c_double()
    _lib.check(_lib.load().mvs_ticket_elapsed_ms(int(t0), int(t1), C.byref(ms)), 0, "mvs_ticket_elapsed_ms")
    return float(ms.value)
