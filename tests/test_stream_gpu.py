"""The transfer-overlapped path of the product (csrc/mvs_transfer.hip, device.to_device_async, fusion.fuse_to_host): tiles that
start in host memory and a result that ends there (fusion/_core.py:1068-1170, 2044-2156)."""
import numpy as np
import pytest

from tests.helpers import squeeze_field

pytestmark = pytest.mark.gpu


def _mosaic(tile=256, jitter=2, seed=3):
    from multiview_stitcher_amd import sample_data

    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(tile,) * 3, tiles=(2, 2, 2), overlap=(int(tile * 0.2),) * 3, dtype=np.uint16,
                                                    max_jitter=jitter, seed=seed)
    return [squeeze_field(s) for s in sims]


def test_pinned_arrays_and_async_round_trip(hip_device):
    from multiview_stitcher_amd import device

    a = device.pinned_empty((5, 7, 9), np.uint16)
    assert device.is_pinned(a) and device.is_pinned(a[1:3]) and not device.is_pinned(np.zeros(4))
    a[:] = np.arange(a.size, dtype=np.uint16).reshape(a.shape)
    d = device.DeviceArray.from_host_async(a, 0)
    assert d.ready_ticket != 0 and d[1:3].ready_ticket == d.ready_ticket      # windows share the pending upload
    b = device.pinned_empty(a.shape, a.dtype)
    b[:] = 0
    device.ticket_sync(d.download_async(b, after=d.ready_ticket))
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(d.get(), a)          # (get() waits for the upload itself)
    assert d.ready_ticket == 0
    with pytest.raises(ValueError):
        device.DeviceArray.from_host_async(np.zeros((3, 3), np.uint16), 0)      # pageable memory: refused, not silently synchronous


def test_large_get_goes_through_pinned_staging(hip_device):
    """DeviceArray.get() of a large contiguous array: pieces downloaded asynchronously into pinned buffers and copied out by the I/O
    pool (3 x the rate of one copy into pageable memory); same bytes, piece sizes that do not divide the array, windows untouched."""
    from multiview_stitcher_amd import device

    rng = np.random.default_rng(1)
    a = rng.integers(0, 65535, (7, 333, 1001), dtype=np.uint16)
    d = device.DeviceArray.from_host(a, 0)
    old = device._STAGED_GET_MIN_BYTES[0]
    device._STAGED_GET_MIN_BYTES[0] = 1 << 20
    try:
        np.testing.assert_array_equal(d.get(), a)
        np.testing.assert_array_equal(d._get_staged(piece=(1 << 20) + 12346, depth=2), a)
        np.testing.assert_array_equal(d[2:5, 10:300, 7:900].get(), a[2:5, 10:300, 7:900])      # (a strided window: the plain path)
        np.testing.assert_array_equal(d[3].get(), a[3])                                       # (a contiguous window)
    finally:
        device._STAGED_GET_MIN_BYTES[0] = old


def test_streamed_register_and_fuse_equal_the_resident_run_and_overlap(hip_device):
    """Tiles uploaded with ``to_device_async`` while ``register()`` registers the pairs whose tiles have landed, the mosaic fused in
    slabs with every slab's download under the next slab's fuse (``fuse_to_host``): (1) parameters and fused voxels equal the run
    on resident tiles bit for bit; (2) the timeline -- timed tickets of the uploads, of every pair's last kernel, of every slab's
    fuse and download -- shows that kernels do not wait for transfers they do not depend on.  The test mosaic is small (uploads of
    5 ms, downloads of 1 ms: too short to show anything against the host's own milliseconds), so the copy stream is given a backlog
    on purpose: a 1 GiB upload queued BEFORE the last tile (the pairs of the other tiles must be done long before that tile lands)
    and a 1 GiB download queued before ``fuse_to_host`` (all slabs must be fused while it still runs, i.e. before the first slab's
    own download can even start -- with parameter blocks uploaded by a copy engine they queued behind it: round 6)."""
    from multiview_stitcher_amd import device, fusion, registration, sample_data

    key = sample_data.METADATA_TRANSFORM_KEY
    sims = _mosaic()
    res_sims = [device.to_device(s, 0) for s in sims]
    p_ref = registration.register(res_sims, transform_key=key, new_transform_key="reg", device=0)
    want = fusion.fuse(res_sims, transform_key="reg", output_on_backend=True, device=0).data.get()

    host = []
    for s in sims:
        h = device.pinned_empty(s.data.shape, s.data.dtype)
        h[:] = np.asarray(s.data)
        host.append(s.copy(data=h))
    ballast_host = device.pinned_empty((1 << 30,), np.uint8)            # ~19 ms of the host link per direction
    ballast_host[:] = 0
    ballast_dev = device.DeviceArray.empty((1 << 30,), np.uint8, 0)
    registration._pair_timeline = pairs = []
    try:
        t0 = device.mark(0)
        a_sims = device.to_device_async(host[:-1], 0)
        ballast_up = device.DeviceArray.from_host_async(ballast_host, 0)           # the copy stream keeps the order: tiles 0-6, ballast, tile 7
        a_sims += device.to_device_async(host[-1:], 0)
        uploads = [s.data.ready_ticket for s in a_sims]
        assert all(uploads)
        p_got = registration.register(a_sims, transform_key=key, new_transform_key="reg", device=0)
    finally:
        registration._pair_timeline = None
    out_host = device.pinned_empty(want.shape, want.dtype)          # (made before the clock starts: pinning 200 MB takes longer than the ballast)
    t1 = device.mark(0)
    ballast_down = ballast_dev.download_async(ballast_host, after=ballast_up.ready_ticket)
    fused, slabs = fusion.fuse_to_host(a_sims, transform_key="reg", n_slabs=4, out=out_host, device=0, return_timeline=True)
    for a, b in zip(p_got, p_ref):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    got = np.asarray(fused.data)
    assert device.is_pinned(got) and got.shape == want.shape
    np.testing.assert_array_equal(got, want)

    up_ms = [device.ticket_elapsed_ms(t0, t) for t in uploads]
    pair_ms = [device.ticket_elapsed_ms(t0, t) for _, t in pairs]
    assert len(pairs) >= 7 and all(t > 0 for t in pair_ms)
    assert up_ms == sorted(up_ms)                                   # the copy stream keeps the order of the list
    assert up_ms[-1] - up_ms[-2] > 8.0                              # (the ballast sits in front of the last tile)
    # a pair cannot end before its two tiles have landed ...
    last = len(host) - 1
    for (i, j), t in zip([e for e, _ in pairs], pair_ms):
        assert t >= max(up_ms[i], up_ms[j]) - 1e-3
    # ... and pairs that do not touch the last tile are done while it is still on its way (not necessarily all of them: the lanes'
    # streams share hardware queues, and a lane that already waits for the last tile holds up the stream behind it in its queue)
    early = [t for (i, j), t in zip([e for e, _ in pairs], pair_ms) if last not in (i, j)]
    assert len(early) >= 6 and sum(t < up_ms[-1] - 5.0 for t in early) >= len(early) // 2, (sorted(early), up_ms)
    # every slab was fused while the ballast download was still running -- before the first slab's own download could start
    down_ms = device.ticket_elapsed_ms(t1, ballast_down)
    assert down_ms > 8.0
    assert len(slabs) == 4 and all(d >= f for f, d in slabs)
    assert max(f for f, _ in slabs) < min(d for _, d in slabs), slabs
    assert slabs[0][1] - slabs[0][0] > 4.0, slabs                   # (slab 0's download waited behind the ballast; its fuse did not)


@pytest.mark.parametrize("weights", [None, "content_based"])
def test_block_pipeline_equals_the_serial_loop(hip_device, tmp_path, weights):
    """fuse() of Zarr-backed tiles into a Zarr store in several launch blocks: with read-ahead, asynchronous transfers and
    write-behind around the blocks (streaming.BlockPipeline, the default) the store holds exactly what the serial loop writes."""
    from multiview_stitcher_amd import fusion, ngff_utils, sample_data, zarr_io, spatial_image_utils as si

    key = sample_data.METADATA_TRANSFORM_KEY
    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(64, 96, 128), tiles=(2, 2, 2), overlap=(12, 20, 26), dtype=np.uint16, max_jitter=0, seed=8)
    lazy = []
    for i, s in enumerate(sims):
        z = ngff_utils.write_sim_to_ome_zarr(s, str(tmp_path / f"tile{i}.zarr"))
        si.set_sim_affine(z, si.get_affine_from_sim(s, key), key)
        lazy.append(z)
    kw = dict(transform_key=key, output_chunksize={"z": 32, "y": 64, "x": 64})
    if weights:
        kw.update(weights_func=fusion.content_based, weights_func_kwargs={"sigma_1": 1.5, "sigma_2": 3.0})
    old_budget = fusion.MAX_STREAM_BYTES
    fusion.MAX_STREAM_BYTES = 1 << 20                 # several launch blocks even for this small mosaic
    try:
        from multiview_stitcher_amd import streaming

        got = fusion.fuse(lazy, output_zarr_url=str(tmp_path / "piped.zarr"), **kw)
        # (the blocks of this store are whole chunks up to the array's border -- shape (116, 172, 230) in (32, 64, 64) chunks -- so they
        # left the device re-tiled into chunk-major order, border chunks padded with the fill value: mvs_copy_box)
        assert streaming.LAST_TIMELINE and all(b.get("tiles", 0) >= 1 for b in streaming.LAST_TIMELINE)
        fusion._STREAM_TILES[0] = False
        try:
            plain = fusion.fuse(lazy, output_zarr_url=str(tmp_path / "piped_rows.zarr"), **kw)      # row-major download + host gather
        finally:
            fusion._STREAM_TILES[0] = True
        assert all("tiles" not in b for b in streaming.LAST_TIMELINE)
        np.testing.assert_array_equal(np.asarray(plain.data), np.asarray(got.data))
        fusion._STREAM_PIPELINE[0] = False
        try:
            want = fusion.fuse(lazy, output_zarr_url=str(tmp_path / "serial.zarr"), **kw)
        finally:
            fusion._STREAM_PIPELINE[0] = True
        host = fusion.fuse(lazy, **kw)                # Zarr in, host memory out: the pipeline's other sink
    finally:
        fusion.MAX_STREAM_BYTES = old_budget
    assert zarr_io.is_zarr_backed(got.data)
    a, b = np.asarray(got.data), np.asarray(want.data)
    assert a.shape == b.shape and a.any()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(np.asarray(host.data), b)
    # the chunk FILES are the same bytes too (border chunks are stored whole, padded with the fill value, by both paths)
    za, zb = zarr_io.ZarrArray.open(got.data.array.path if hasattr(got.data, "array") else got.data.path), \
        zarr_io.ZarrArray.open(want.data.array.path if hasattr(want.data, "array") else want.data.path)
    for idx in np.ndindex(*za.grid):
        ca, cb = za.read_chunk(idx), zb.read_chunk(idx)
        assert (ca is None) == (cb is None)
        if ca is not None:
            np.testing.assert_array_equal(ca, cb)


def test_copy_box_between_pitched_windows(hip_device):
    """DeviceArray.copy_box_to / mvs_copy_box: a window of one allocation into a window of another, both pitched (rows contiguous),
    odd extents and offsets, 2 and 3 axes, leading axes of extent 1."""
    from multiview_stitcher_amd import _lib
    from multiview_stitcher_amd.device import DeviceArray

    rng = np.random.default_rng(4)
    for dtype, shape_a, shape_b, lo_a, lo_b, ext in (
            (np.uint16, (9, 37, 53), (5, 64, 64), (2, 5, 7), (1, 3, 11), (4, 29, 41)),
            (np.uint8, (1, 1, 20, 45), (1, 1, 33, 71), (0, 0, 3, 2), (0, 0, 10, 30), (1, 1, 17, 41)),
            (np.float32, (31, 17), (40, 40), (4, 0), (7, 23), (25, 17))):
        a = (rng.random(shape_a) * 200).astype(dtype)
        b = (rng.random(shape_b) * 200).astype(dtype)
        da, db = DeviceArray.from_host(a, 0), DeviceArray.from_host(b, 0)
        sa = tuple(slice(l, l + e) for l, e in zip(lo_a, ext))
        sb = tuple(slice(l, l + e) for l, e in zip(lo_b, ext))
        da[sa].copy_box_to(db[sb])
        _lib.synchronize(0)
        want = b.copy()
        want[sb] = a[sa]
        np.testing.assert_array_equal(db.get(), want)
    with pytest.raises(ValueError):
        da[:3, :3].copy_box_to(db[:3, :4])


def test_fuse_of_plain_host_arrays_takes_the_block_pipeline_and_equals_one_launch_block(hip_device, tmp_path):
    """``fuse()`` of host numpy tiles (>= 256 MiB) into a host numpy result -- the reference's own call -- runs its launch blocks through
    the block pipeline by default; same voxels as the single launch block ``mvs_fuse_chunk`` uploads, fuses and downloads itself."""
    from multiview_stitcher_amd import fusion, sample_data, streaming

    key = sample_data.METADATA_TRANSFORM_KEY
    sims = _mosaic()
    assert sum(s.data.nbytes for s in sims) >= fusion._HOST_STREAM_MIN_BYTES and all(isinstance(s.data, np.ndarray) for s in sims)
    old_budget = fusion.MAX_STREAM_BYTES
    fusion.MAX_STREAM_BYTES = 64 << 20                # several launch blocks
    try:
        streaming.LAST_TIMELINE[:] = []
        got = fusion.fuse(sims, transform_key=key, device=0)
        assert len(streaming.LAST_TIMELINE) >= 2      # (the pipeline ran)
    finally:
        fusion.MAX_STREAM_BYTES = old_budget
    fusion._HOST_STREAM[0] = False
    try:
        streaming.LAST_TIMELINE[:] = []
        want = fusion.fuse(sims, transform_key=key, device=0)
        assert not streaming.LAST_TIMELINE
    finally:
        fusion._HOST_STREAM[0] = True
    assert isinstance(got.data, np.ndarray) and got.data.any()
    np.testing.assert_array_equal(np.asarray(got.data), np.asarray(want.data))
    # tiles resident on the device, result a host array: the same pipeline (the blocks' downloads under the next blocks' launches)
    from multiview_stitcher_amd import device

    streaming.LAST_TIMELINE[:] = []
    got_d = fusion.fuse([device.to_device(s, 0) for s in sims], transform_key=key, device=0)
    assert streaming.LAST_TIMELINE and isinstance(got_d.data, np.ndarray)
    np.testing.assert_array_equal(np.asarray(got_d.data), np.asarray(want.data))
    # register() of the same host tiles: staged through pinned buffers and uploaded while the pairs start (upload_host_sims_async);
    # same parameters as with one synchronous upload per tile, and the caller's arrays are left alone
    from multiview_stitcher_amd import registration

    before = [s.data for s in sims]
    p_async = registration.register(sims, transform_key=key, new_transform_key="reg_a", device=0)
    registration._HOST_UPLOAD_ASYNC[0] = False
    try:
        p_sync = registration.register(sims, transform_key=key, new_transform_key="reg_s", device=0)
    finally:
        registration._HOST_UPLOAD_ASYNC[0] = True
    assert all(s.data is b for s, b in zip(sims, before))
    for a, b in zip(p_async, p_sync):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    # ... and of the same tiles as windows of Zarr arrays: chunk files read straight into the pinned staging buffers
    from multiview_stitcher_amd import ngff_utils, zarr_io, spatial_image_utils as si

    lazy = []
    for i, s in enumerate(sims):
        z = ngff_utils.write_sim_to_ome_zarr(s, str(tmp_path / f"tile{i}.zarr"))
        si.set_sim_affine(z, si.get_affine_from_sim(s, key), key)
        assert zarr_io.is_zarr_backed(z.data)
        lazy.append(z)
    p_zarr = registration.register(lazy, transform_key=key, new_transform_key="reg_z", device=0)
    for a, b in zip(p_zarr, p_sync):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
