"""Read-ahead / write-behind around the launch blocks of ``fusion.fuse`` when tiles are read from Zarr stores or the result is
written to one (or to host memory).

The reference streams such a fusion through dask: chunk tasks read their slabs lazily from the input stores
(spatial_image_utils.py:712-860) and write their regions of the output store (fusion/_core.py:1068-1170, 2044-2156), many of them in
flight at once.  Here a launch block is one ``mvs_fuse_chunk`` call, and until round 6 its five steps ran one after the other:
read the slabs, upload, fuse, download, write.  ``BlockPipeline`` runs them as three stages with up to three blocks staged ahead / written behind and two blocks being read and two
being written at any time (MVS_STREAM_DEPTH / MVS_STREAM_READERS / MVS_STREAM_WRITERS):

  reader thread   the slabs of block k + 1 -> pinned host buffers (chunk files read by a small pool of threads: file reads and
                  copies release the GIL) -> asynchronous uploads on the device's copy stream (csrc/mvs_transfer.hip);
  caller          block k: ``fuse_np`` on device-resident slabs (its stream waits for their uploads' tickets), result left on
                  the device, a timed mark, an asynchronous download into a pinned buffer after the mark;
  writer thread   block k - 1: waits for the download's ticket, writes the region (chunk files by the pool), returns the
                  pinned buffers to the pool.

Same calls, same arguments, same voxels as the serial loop: only the order in which the host gets to them changes.
"""
from __future__ import annotations

import itertools
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import device as dev_mod
from . import zarr_io


class PinnedPool:
    """Pinned host buffers handed out by size.  hipHostMalloc pins pages at a few GB/s -- 100+ ms for the 1 GiB staging buffer of a
    launch block -- so buffers go back to the pool and the pool outlives the pipeline that filled it (``shared_pinned_pool``: the
    second streamed ``fuse()`` of a process finds its buffers); what the pool keeps is capped (``cap_bytes``, default 16 GiB or
    ``MVS_PINNED_POOL_MB``), buffers returned beyond the cap are released."""

    def __init__(self, cap_bytes=None):
        self._free = {}
        self._busy = {}       # id(raw) -> ticket of a transfer that was still reading the buffer when it came back
        self._limbo = []
        self._order = []      # free buffers, oldest first (what makes room when the cap is reached)
        self._lock = threading.Lock()
        self._held = 0
        self.cap_bytes = int(os.environ.get("MVS_PINNED_POOL_MB", 16 << 10)) << 20 if cap_bytes is None else int(cap_bytes)

    def get(self, shape, dtype):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        cap = max(1 << 20, 1 << int(np.ceil(np.log2(max(n, 1)))))
        with self._lock:
            lst = self._free.get(cap)
            raw = lst.pop() if lst else None
            if raw is not None:
                self._held -= raw.size
                ticket = self._busy.pop(id(raw), 0)
                self._order = [o for o in self._order if o is not raw]
        if raw is None:
            raw = dev_mod.pinned_empty((cap,), np.uint8)
        elif ticket:
            dev_mod.ticket_sync(ticket)      # (handed back while a transfer still read it: that transfer has to be through)
        return raw, raw[:n].view(dtype).reshape(shape)

    def put(self, raw, after=0):
        """Hand ``raw`` back; ``after``: ticket of a transfer that may still be reading it (waited for when the buffer is given out again).
        Over the cap the buffers that have lain in the pool longest make room (a pipeline's own sizes stay: dropping the buffer that
        comes back, as the first version did, made a pipeline whose sizes were not the previous one's re-pin every buffer on every
        call -- 100-150 ms stalls per block on the C5 leg after the host-array leg had filled the pool)."""
        with self._lock:
            if raw.size > self.cap_bytes:
                if after:
                    self._limbo = self._limbo[-63:] + [(raw, after)]      # (kept alive until the transfer is through, then dropped)
                return
            while self._held + raw.size > self.cap_bytes and self._order:
                old = self._order.pop(0)
                lst = self._free.get(old.size)
                if lst and any(o is old for o in lst):
                    lst[:] = [o for o in lst if o is not old]
                    self._held -= old.size
                    t = self._busy.pop(id(old), 0)
                    if t:
                        self._limbo = self._limbo[-63:] + [(old, t)]
            self._free.setdefault(raw.size, []).append(raw)
            self._order.append(raw)
            self._held += raw.size
            if after:
                self._busy[id(raw)] = int(after)

    def clear(self):
        with self._lock:
            busy = list(self._busy.values())
        for t in busy:
            dev_mod.ticket_sync(t)
        with self._lock:
            self._free.clear()
            self._busy.clear()
            self._limbo = []
            self._order = []
            self._held = 0


_SHARED_POOL = [None]


def shared_pinned_pool():
    """The process-wide pool the pipelines of streamed ``fuse()`` calls draw their staging buffers from."""
    with _IO_LOCK:
        if _SHARED_POOL[0] is None:
            _SHARED_POOL[0] = PinnedPool()
        return _SHARED_POOL[0]


LAST_TIMELINE = []     # (measurement) the per-block records of the most recent finished pipeline
_IO_POOL = {}
_IO_LOCK = threading.Lock()


def io_pool(workers=8, kind="read"):
    """Thread pool of the chunk-file tasks of one direction ("read" / "write"): two pools, so that the write-behind of block k - 1 does
    not queue behind the read-ahead of block k + 1 (file reads, writes and the numpy copies around them release the GIL)."""
    with _IO_LOCK:
        if _IO_POOL.get(kind) is None:
            _IO_POOL[kind] = ThreadPoolExecutor(max_workers=int(os.environ.get("MVS_IO_THREADS", workers)), thread_name_prefix="mvs-io-" + kind)
        return _IO_POOL[kind]


def parallel_copy(dst, src, kind="read"):
    """``dst[...] = src`` cut along the first axis into pieces for the I/O pool of ``kind`` (a single thread copies ~8 GB/s; the
    staging buffers of a streamed fuse() of host arrays want the link's 55)."""
    src = np.asarray(src)
    n = dst.shape[0] if dst.ndim else 1
    if dst.ndim == 0 or dst.nbytes < (8 << 20) or n < 2:
        dst[...] = src
        return
    pieces = min(n, 16)
    cuts = np.linspace(0, n, pieces + 1).astype(int)
    list(io_pool(kind=kind).map(lambda k: np.copyto(dst[cuts[k]:cuts[k + 1]], src[cuts[k]:cuts[k + 1]], casting="unsafe"), range(pieces)))


def upload_host_sims_async(sims, device, depth=4):
    """Copies of the SpatialImages ``sims`` (plain host numpy data, or windows of Zarr arrays) whose data is on its way to ``device``:
    every tile is copied -- or read, chunk file by chunk file -- into a pinned staging buffer by the I/O pool and uploaded on the copy
    stream without waiting (``DeviceArray.from_host_async``; the
    returned arrays carry the uploads' tickets, so ``register()`` starts a pair when its two tiles have landed).  ``depth`` staging
    buffers are cycled: a buffer is reused when the upload that read it is through."""
    from .device import DeviceArray

    pool = shared_pinned_pool()
    ring, out = [], []
    for s_ in sims:
        lazy = zarr_io.is_zarr_backed(s_.data)
        data = s_.data if lazy else np.asarray(s_.data)
        shape, dtype = tuple(data.shape), np.dtype(data.dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if len(ring) >= depth:
            raw, ticket = ring.pop(0)
            dev_mod.ticket_sync(ticket)
            buf = raw[:nbytes].view(dtype).reshape(shape) if raw.size >= nbytes else None
            if buf is None:
                pool.put(raw)
                raw, buf = pool.get(shape, dtype)
        else:
            raw, buf = pool.get(shape, dtype)
        if lazy:
            read_window(data, buf)
        else:
            parallel_copy(buf, data)
        d = DeviceArray.from_host_async(buf, device)
        ring.append((raw, d.ready_ticket))
        out.append(s_.copy(data=d))
    for raw, ticket in ring:
        pool.put(raw, after=ticket)
    return out


def read_window(view, out):
    """Fill ``out`` (any writable array of the window's shape) with the Zarr window ``view`` (ZarrView / ZarrArray), the chunk
    files read and copied by the I/O pool."""
    if isinstance(view, zarr_io.ZarrArray):
        view = view[...]
    arr = view.array
    win = view._window()
    starts, stops = [a for a, _ in win], [b for _, b in win]
    full = out.reshape([b - a for a, b in win])

    def one(idx):
        lo = [max(a, i * c) for a, i, c in zip(starts, idx, arr.chunks)]
        hi = [min(b, (i + 1) * c) for b, i, c in zip(stops, idx, arr.chunks)]
        dst = tuple(slice(l - a, h - a) for l, h, a in zip(lo, hi, starts))
        chunk = arr.read_chunk(idx, scratch=True)
        if chunk is None:
            full[dst] = arr.fill_value
        else:
            full[dst] = chunk[tuple(slice(l - i * c, h - i * c) for l, h, i, c in zip(lo, hi, idx, arr.chunks))]

    if full.size:
        list(io_pool().map(one, list(arr._touched(starts, stops))))
    return out


def write_region(zarr_out, starts, data):
    """``zarr_out.write(starts, data)`` with the touched chunk files written by the I/O pool (chunk-aligned cuts: every chunk is
    written by exactly one task)."""
    data = np.asarray(data)
    starts = [int(a) for a in starts]
    stops = [a + n for a, n in zip(starts, data.shape)]
    pieces = []
    for idx in zarr_out._touched(starts, stops):
        c0 = [i * c for i, c in zip(idx, zarr_out.chunks)]
        lo = [max(a, o) for a, o in zip(starts, c0)]
        hi = [min(b, o + c) for b, o, c in zip(stops, c0, zarr_out.chunks)]
        pieces.append((lo, data[tuple(slice(l - a, h - a) for l, h, a in zip(lo, hi, starts))]))
    list(io_pool(kind="write").map(lambda p: zarr_out.write(p[0], p[1]), pieces))


def write_tiles(zarr_out, indices, tiles):
    """Chunk files ``indices`` of ``zarr_out`` from ``tiles[i]`` (whole chunks, contiguous: straight out of the download buffer)."""
    list(io_pool(kind="write").map(lambda p: zarr_out.write_chunk(p[0], p[1].reshape(zarr_out.chunks)), list(zip(indices, tiles))))


class BlockPipeline:
    """Three-stage pipeline over the launch blocks of one ``fuse()`` call (see the module docstring).  ``submit(kwargs, sink)``:
    ``kwargs`` are the ``fuse_np`` arguments of a block (``sims``: slabs that may be Zarr-backed or host arrays), ``sink(chunk)``
    stores the fused block (a numpy array in pinned memory, valid during the call).  ``finish()`` waits for everything."""

    def __init__(self, fuse_np, device, depth=None):
        depth = int(os.environ.get("MVS_STREAM_DEPTH", "3")) if depth is None else depth
        self.fuse_np, self.device, self.depth = fuse_np, device, depth
        self.pool = shared_pinned_pool()
        # (blocks in flight per stage: the chunk files of a block are 20-300 tasks for the I/O pool of its direction, and the last
        # round of one block's tasks leaves threads idle unless another block's are queued behind them)
        self.reader = ThreadPoolExecutor(max_workers=int(os.environ.get("MVS_STREAM_READERS", "2")), thread_name_prefix="mvs-read")
        self.writer = ThreadPoolExecutor(max_workers=int(os.environ.get("MVS_STREAM_WRITERS", "2")), thread_name_prefix="mvs-write")
        self.staged = []       # (future of the staged block, sink)
        self.writes = []
        self.timeline = []     # per block: host-clock seconds since the pipeline was made (slabs staged, fuse queued, downloaded, written)
        self._t0 = time.perf_counter()
        # the class kernels of the blocks' launches stay on the lane's own stream while the copy stream is busy (see fusion.fuse_to_host)
        self._serial = os.environ.get("MVS_STREAM_FORK", "0") != "1"
        if self._serial:
            from . import _lib

            _lib.set_option("serial_classes", 1, device)

    # -- stage 1: slabs -> pinned -> async upload
    def _stage(self, kwargs):
        from .device import DeviceArray, is_device_array

        raws, sims = [], []
        t_in = time.perf_counter() - self._t0
        for s_ in kwargs["sims"]:
            data = s_.data
            if is_device_array(data):
                sims.append(s_)
                continue
            raw, buf = self.pool.get(data.shape, s_.dtype)
            if zarr_io.is_zarr_backed(data):
                read_window(data, buf)
            else:
                parallel_copy(buf, data)
            raws.append(raw)
            sims.append(s_.copy(data=DeviceArray.from_host_async(buf, self.device)))
        slab_mb = sum(int(np.prod(s_.data.shape)) * np.dtype(s_.dtype).itemsize for s_ in sims) / 2 ** 20
        return dict(kwargs, sims=sims), raws, (t_in, time.perf_counter() - self._t0, slab_mb)

    def submit(self, kwargs, sink, tiling=None):
        """``tiling`` = (ZarrArray, starts): the fused block goes into that array at ``starts`` (one entry per array axis).  When the
        block is made of whole chunks of the array (up to the array's border) it is re-tiled ON THE DEVICE into chunk-major order
        (mvs_copy_box), so that every chunk is one contiguous piece of the download and its file is written straight out of pinned
        memory -- no gather on the host, whose memory traffic is what bounds this stage; ``sink`` is then not called."""
        # (The same was built for the READ side -- chunk files read straight into the slots of a pinned chunk-major buffer, uploaded,
        # assembled into the slab by mvs_copy_box on a lane of its own -- and bought nothing: 0.323-0.327 s against 0.308-0.322 s on the
        # C5 z-slab, three alternating runs on one box; whole chunks are 1.25 x the window's bytes and the stage is not the long pole.)
        self.staged.append((self.reader.submit(self._stage, kwargs), sink, tiling))
        while len(self.staged) > self.depth:
            self._run_one()

    @staticmethod
    def _tile_plan(tiling, block_shape):
        """[(chunk index in the array, block-local start, extent)] + tile shape + "some chunk is cut by the array's border", or None when the
        block is not made of whole chunks / the array's fill value is not 0 / leading axes are chunked."""
        if tiling is None:
            return None
        arr, starts = tiling
        nd = len(block_shape)
        lead = arr.ndim - nd
        if lead < 0 or nd > 3 or len(starts) != arr.ndim or any(c != 1 for c in arr.chunks[:lead]):
            return None
        try:
            if arr.fill_value != 0:
                return None
        except (TypeError, ValueError):
            return None
        cs, shp, st = arr.chunks[lead:], arr.shape[lead:], [int(v) for v in starts[lead:]]
        ranges, cut = [], False
        for a, n, c, full in zip(st, block_shape, cs, shp):
            b = a + int(n)
            if a % c or (b % c and b != full) or b > full:
                return None
            cut = cut or bool(b % c)
            ranges.append(range(a // c, -(-b // c)))
        plan = []
        for idx in itertools.product(*ranges):
            lo = [i * c - a for i, c, a in zip(idx, cs, st)]
            ext = [min(c, a + int(n) - i * c) for i, c, a, n in zip(idx, cs, st, block_shape)]
            plan.append((tuple(int(v) for v in starts[:lead]) + tuple(idx), lo, ext))
        return plan, tuple(cs), cut

    # -- stage 2 (caller's thread): fuse on the device, queue the download
    def _run_one(self):
        fut, sink, tiling = self.staged.pop(0)
        kwargs, raws, (t_in, t_staged, slab_mb) = fut.result()
        rec = {"read_start": t_in, "staged": t_staged, "fuse_start": time.perf_counter() - self._t0, "slab_mb": slab_mb,
               "views": len(kwargs["sims"])}
        self.timeline.append(rec)
        # (content-based weights: the fast path's overflow flag is looked at per block, since the block leaves the device right away)
        chunk = self.fuse_np(output_on_backend=True, **dict(kwargs, _cb_check=True))
        tiles = self._tile_plan(tiling, chunk.shape)
        if tiles is not None:
            plan, tile_shape, cut = tiles
            dev_tiles = dev_mod.DeviceArray.empty((len(plan),) + tile_shape, chunk.dtype, self.device)
            if cut:
                dev_tiles.fill_zero()         # (chunks cut by the array's border are stored whole: fill value beyond it)
            for i, (_, lo, ext) in enumerate(plan):
                chunk[tuple(slice(l, l + e) for l, e in zip(lo, ext))].copy_box_to(dev_tiles[i][tuple(slice(0, e) for e in ext)])
            src, sink = dev_tiles, (lambda out, plan=plan, arr=tiling[0]: write_tiles(arr, [p[0] for p in plan], out))
            rec["tiles"] = len(plan)
        else:
            src = chunk
        mark = dev_mod.mark(self.device)
        raw_out, out = self.pool.get(src.shape, src.dtype)
        ticket = src.download_async(out, after=mark)
        rec["fuse_queued"] = time.perf_counter() - self._t0
        self.writes.append(self.writer.submit(self._write, (chunk, src), kwargs, raws, raw_out, out, ticket, sink, rec))
        while len(self.writes) > self.depth:
            self.writes.pop(0).result()

    # -- stage 3: wait for the download, store, recycle the buffers
    def _write(self, chunk, kwargs, raws, raw_out, out, ticket, sink, rec):
        dev_mod.ticket_sync(ticket)
        rec["downloaded"] = time.perf_counter() - self._t0
        sink(out)
        rec["written"] = time.perf_counter() - self._t0
        for r in raws:
            self.pool.put(r)
        self.pool.put(raw_out)
        del chunk, kwargs

    def abort(self):
        """A block failed on the caller's thread: drop what is queued, let the blocks already in a stage run out, keep no threads."""
        for fut, _, _ in self.staged:
            fut.cancel()
        self.staged, self.writes = [], []
        self.reader.shutdown(wait=False, cancel_futures=True)
        self.writer.shutdown(wait=False, cancel_futures=True)
        self._restore()

    def _restore(self):
        if self._serial:
            from . import _lib

            self._serial = False
            _lib.set_option("serial_classes", 0, self.device)

    def finish(self):
        try:
            while self.staged:
                self._run_one()
            for w in self.writes:
                w.result()
            self.writes = []
        finally:
            self.reader.shutdown(wait=True)
            self.writer.shutdown(wait=True)
            self._restore()
            LAST_TIMELINE[:] = self.timeline
