"""Host logic of the streaming layer (multiview_stitcher_amd/streaming.py) that needs no GPU: the chunk-major tile plan of a fused
block, the pinned staging pool's bookkeeping (buffers stood in for by plain arrays), the parallel copies, and the Zarr window reads /
tile writes they are built on (reference: fusion/_core.py:1123-1171, 2044-2156 write regions of a dask array chunk by chunk)."""
import numpy as np
import pytest

from multiview_stitcher_amd import streaming, zarr_io


def test_tile_plan_of_a_block_of_whole_chunks(tmp_path):
    arr = zarr_io.ZarrArray.create(str(tmp_path / "a.zarr"), [1, 1, 300, 520, 700], [1, 1, 256, 256, 256], np.uint16)
    plan, tile, cut = streaming.BlockPipeline._tile_plan((arr, [0, 0, 256, 0, 512]), (44, 512, 188))      # the array's far corner
    assert tile == (256, 256, 256) and cut
    assert [p[0] for p in plan] == [(0, 0, 1, 0, 2), (0, 0, 1, 1, 2)]
    assert [p[1] for p in plan] == [[0, 0, 0], [0, 256, 0]] and [p[2] for p in plan] == [[44, 256, 188], [44, 256, 188]]
    plan, tile, cut = streaming.BlockPipeline._tile_plan((arr, [0, 0, 0, 0, 0]), (256, 512, 512))
    assert len(plan) == 4 and not cut
    # not whole chunks / not at a chunk boundary / chunked leading axes / a fill value that is not 0: the row-major path
    assert streaming.BlockPipeline._tile_plan((arr, [0, 0, 0, 0, 0]), (256, 256, 300)) is None
    assert streaming.BlockPipeline._tile_plan((arr, [0, 0, 10, 0, 0]), (246, 256, 256)) is None
    assert streaming.BlockPipeline._tile_plan(None, (4, 4, 4)) is None
    arr2 = zarr_io.ZarrArray.create(str(tmp_path / "b.zarr"), [4, 64, 64], [2, 32, 32], np.uint16)
    assert streaming.BlockPipeline._tile_plan((arr2, [0, 0, 0]), (32, 32)) is None
    arr3 = zarr_io.ZarrArray.create(str(tmp_path / "c.zarr"), [64, 64], [32, 32], np.uint16, fill_value=7)
    assert streaming.BlockPipeline._tile_plan((arr3, [0, 0]), (32, 64)) is None


def test_write_tiles_and_window_reads_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    arr = zarr_io.ZarrArray.create(str(tmp_path / "a.zarr"), [1, 70, 90, 130], [1, 32, 32, 64], np.uint16)
    full = rng.integers(0, 60000, (1, 70, 90, 130), dtype=np.uint16)
    streaming.write_region(arr, [0, 0, 0, 0], full)                      # every chunk file, border chunks padded
    np.testing.assert_array_equal(np.asarray(arr[...]), full)
    # a block re-tiled into chunk-major order (what the device hands over): tiles of whole chunks, fill value beyond the border
    plan, tile, cut = streaming.BlockPipeline._tile_plan((arr, [0, 64, 0, 0]), (6, 90, 130))
    assert cut and tile == (32, 32, 64) and len(plan) == 3 * 3
    tiles = np.zeros((len(plan),) + tile, np.uint16)
    block = rng.integers(0, 60000, (6, 90, 130), dtype=np.uint16)
    for i, (_, lo, ext) in enumerate(plan):
        tiles[i][tuple(slice(0, e) for e in ext)] = block[tuple(slice(l, l + e) for l, e in zip(lo, ext))]
    streaming.write_tiles(arr, [p[0] for p in plan], tiles)
    full[0, 64:] = block
    np.testing.assert_array_equal(np.asarray(arr[...]), full)
    out = np.empty((1, 50, 60, 100), np.uint16)
    streaming.read_window(arr[:, 10:60, 20:80, 15:115], out)
    np.testing.assert_array_equal(out, full[:, 10:60, 20:80, 15:115])
    # a window over chunks that were never written reads the fill value
    arr2 = zarr_io.ZarrArray.create(str(tmp_path / "b.zarr"), [40, 40], [16, 16], np.float32, fill_value=0)
    buf = np.ones((40, 40), np.float32)
    streaming.read_window(arr2, buf)
    assert not buf.any()


def test_parallel_copy_equals_plain_assignment():
    rng = np.random.default_rng(1)
    src = rng.integers(0, 255, (37, 301, 257), dtype=np.uint8)            # > 8 MiB: cut into pieces for the pool
    dst = np.zeros_like(src)
    streaming.parallel_copy(dst, src)
    np.testing.assert_array_equal(dst, src)
    big = np.zeros((5, 64, 64), np.float32)
    streaming.parallel_copy(big[1:4], np.ones((3, 64, 64), np.float64), kind="write")      # small + a cast: the plain assignment
    assert big[1:4].all() and not big[0].any() and not big[4].any()


def test_pinned_pool_keeps_recent_sizes_within_its_cap(monkeypatch):
    made, synced = [], []
    monkeypatch.setattr(streaming.dev_mod, "pinned_empty", lambda shape, dtype: made.append(shape) or np.empty(shape, dtype))
    monkeypatch.setattr(streaming.dev_mod, "ticket_sync", lambda t: synced.append(t))
    pool = streaming.PinnedPool(cap_bytes=5 << 20)
    a, va = pool.get((1000, 500), np.uint16)                                # 1 MB -> the 1 MiB class
    assert a.size == 1 << 20 and va.shape == (1000, 500) and va.dtype == np.uint16
    b, _ = pool.get((2 << 20,), np.uint8)
    c, _ = pool.get((2 << 20,), np.uint8)
    pool.put(a)
    pool.put(b, after=41)                                                   # a transfer still reads b
    pool.put(c)
    assert pool._held == 5 << 20
    d, _ = pool.get((3 << 20,), np.uint8)                                   # the 4 MiB class: nothing cached
    assert d.size == 4 << 20 and len(made) == 4
    pool.put(d)                                                             # over the cap: the oldest (a, then b, then c) make room
    assert pool._held == 4 << 20 and sorted(pool._free) == [1 << 20, 2 << 20, 4 << 20] and not pool._free[1 << 20] and not pool._free[2 << 20]
    e, _ = pool.get((4 << 20,), np.uint8)
    assert e is d and len(made) == 4 and not synced
    pool.put(e, after=43)
    f, _ = pool.get((4 << 20,), np.uint8)
    assert f is e and synced == [43]                                        # handed out again only when its transfer is through
    pool.put(np.empty(8 << 20, np.uint8))                                   # larger than the cap: never kept
    assert pool._held == 0
    pool.clear()
