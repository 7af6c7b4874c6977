"""GPU tests of the workflow API (register / fuse) incl. restated known-answer tests of the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_register_recovers_hidden_jitter_2d(hip_device):
    from multiview_stitcher_amd import msi_utils, registration, sample_data

    sims, jit, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(128, 160), tiles=(2, 3), overlap=(40, 48),
                                                      dtype=np.uint16, max_jitter=3, seed=2)
    msims = [msi_utils.get_msim_from_sim(s) for s in sims]
    params = registration.register(msims, transform_key=sample_data.METADATA_TRANSFORM_KEY, new_transform_key="reg",
                                   reg_channel_index=0)
    from multiview_stitcher_amd import param_utils
    got = np.array([param_utils.select_time(p, 0)[:-1, -1] for p in params])   # params are t-stacked like the reference's
    np.testing.assert_allclose(got - got[0], jit - jit[0], atol=0.5)      # relative to tile 0 (the resolver picks its own reference view)
    # new key = params rebased on the metadata transform (msi_utils.set_affine_transform, base_transform_key)
    t_new = msi_utils.get_transform_from_msim(msims[3], "reg")
    t_old = msi_utils.get_transform_from_msim(msims[3], sample_data.METADATA_TRANSFORM_KEY)
    np.testing.assert_allclose(t_new, params[3] @ t_old)


def test_register_3d_with_binning_and_device_tiles(hip_device):
    from multiview_stitcher_amd import device, registration, sample_data, spatial_image_utils as si

    sims, jit, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(48, 96, 96), tiles=(1, 2, 2), overlap=(0, 32, 32),
                                                      dtype=np.uint16, max_jitter=2, seed=5)
    sims = [device.to_device(s.isel({"c": 0, "t": 0}), 0) for s in sims]
    res = registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="reg",
                                registration_binning={"z": 1, "y": 2, "x": 2}, return_dict=True)
    got = np.array([p[:-1, -1] for p in res["params"]])
    want = jit - jit[0]
    np.testing.assert_allclose(got - got[0], want, atol=1.01)      # binning 2 -> half-resolution shifts
    assert all(q > 0.5 for q in res["pairwise_registration"]["metrics"]["qualities"].values())
    # every pair found its two tiles pre-binned (register() bins all tiles of a regular mosaic while it builds the graph)
    stats = res["bin_cache_stats"]
    n_pairs = len(res["pairwise_registration"]["edges"])
    # integer tiles on one pixel grid: the batched pair path takes its crops from the RAW tiles (binning inside the crop kernel), no
    # binned copy of a tile is made
    assert stats is not None and stats["pairs_with_raw_crops"] == n_pairs >= 3 and stats["misses"] == 0
    # ... and through binned copies (queued in groups, stream tickets) the pairs find every view pre-binned; same result
    registration._raw_crops_enabled[0] = False
    try:
        res2 = registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, registration_binning={"z": 1, "y": 2, "x": 2}, return_dict=True)
    finally:
        registration._raw_crops_enabled[0] = True
    stats2 = res2["bin_cache_stats"]
    assert stats2["pairs_with_raw_crops"] == 0 and stats2["misses"] == 0 and stats2["hits"] >= len(sims)
    for ra, rb in zip(res["pairwise_registration"]["results"][0], res2["pairwise_registration"]["results"][0]):
        np.testing.assert_array_equal(ra["transform"], rb["transform"])
        assert ra["quality"] == rb["quality"]


def test_peer_copy_follows_later_writes(hip_device):
    """DeviceArray.on_device caches the copy it makes for another GPU context; a later write into the source allocation
    (upload, fill_zero, a kernel writing into ``out=``) must not leave that copy stale.  With one GPU on the box the cache
    is driven directly: the entry is stamped with the owner's write version and refreshed when the version moved."""
    from multiview_stitcher_amd import _lib
    from multiview_stitcher_amd.device import DeviceArray

    a = DeviceArray.from_host(np.arange(24, dtype=np.uint16).reshape(2, 3, 4), 0)
    buf = a._buf
    v0 = buf.version
    a.fill_zero()
    assert buf.version == v0 + 1
    b = DeviceArray.from_host(np.full((2, 3, 4), 7, np.uint16), 0)
    b.copy_into(a, [0, 0, 0])
    assert buf.version == v0 + 2 and np.all(a.get() == 7)
    buf.upload(np.full((2, 3, 4), 9, np.uint16))
    assert buf.version == v0 + 3
    # a cache entry taken at an older version is not handed out: simulate the peer entry of "device 1"
    stale = _lib.DeviceBuffer(0, buf.nbytes).upload(np.zeros((2, 3, 4), np.uint16))
    buf.__dict__["_peer_copies"] = {1: (stale, buf.version - 1)}
    entry = buf.__dict__["_peer_copies"][1]
    assert entry[1] != buf.version          # on_device(1) would re-copy into entry[0] instead of returning it as is
    a.drop_peer_copies()
    assert "_peer_copies" not in buf.__dict__


def test_register_then_fuse_end_to_end(hip_device):
    """T/test_integration.py:19-89 in miniature: register -> fuse; the fused mosaic reproduces the ground truth."""
    from multiview_stitcher_amd import fusion, registration, sample_data

    sims, jit, gt = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(96, 96), tiles=(2, 2), overlap=(32, 32),
                                                       dtype=np.uint16, max_jitter=2, seed=9)
    registration.register(sims, transform_key=sample_data.METADATA_TRANSFORM_KEY, new_transform_key="reg", reg_channel_index=0,
                          groupwise_resolution_kwargs={"reference_view": 0})     # tile 0 keeps its stage position
    fused = fusion.fuse(sims, transform_key="reg", output_chunksize={"y": 64, "x": 64})
    data = np.asarray(fused.data)[0, 0]
    o = np.array([fused.coords["y"][0], fused.coords["x"][0]])
    # fused pixel (y,x) shows ground-truth pixel (y + oy + pad, x + ox + pad), pad = max_jitter + 1
    pad = 3
    ys, xs = np.meshgrid(np.arange(10, data.shape[0] - 10), np.arange(10, data.shape[1] - 10), indexing="ij")
    ref = gt[(ys + int(round(o[0])) + pad), (xs + int(round(o[1])) + pad)]
    # sub-pixel residuals of the registration (<= 0.5 px) on a smooth image: compare by correlation and mean error
    a, b = data[10:-10, 10:-10].astype(float).ravel(), ref.astype(float).ravel()
    assert np.corrcoef(a, b)[0, 1] > 0.99
    assert np.mean(np.abs(a - b)) < 0.03 * 4095
    assert data.min() >= 0 and data[10:-10, 10:-10].min() > 0      # T/test_fusion.py:899-929: no holes


# ---- restated known-answer tests of the reference's fusion suite ----
def _sim(arr, dims, scale, translation, key="k", affine=None):
    from multiview_stitcher_amd import spatial_image_utils as si

    return si.get_sim_from_array(arr, dims=dims, scale=scale, translation=translation, transform_key=key, affine=affine)


def test_kat_axis_aligned_translation_max_fusion(hip_device):
    """T/test_fusion.py:204-237: two constant tiles (1, 2), second at x=6, max fusion, 4x4 chunks."""
    from multiview_stitcher_amd import fusion

    sims = [_sim(np.ones((1, 1, 8, 8)) .astype(np.float32) * v, ["c", "t", "y", "x"], {"y": 1.0, "x": 1.0}, {"y": 0.0, "x": xo})
            for v, xo in [(1, 0.0), (2, 6.0)]]
    fused = fusion.fuse(sims, transform_key="k", fusion_func=fusion.max_fusion, output_chunksize={"y": 4, "x": 4}, merge_chunks=False)
    d = np.asarray(fused.data)
    assert d.shape == (1, 1, 8, 14)
    np.testing.assert_array_equal(d[..., :, :6], 1)
    np.testing.assert_array_equal(d[..., :, 6:], 2)


def test_kat_singleton_view_slice_preserves_spacing(hip_device):
    """T/test_fusion.py:480-530: order-0 fusion, 9 zeros then 20 ones; chunk 10 sees a single source pixel."""
    from multiview_stitcher_amd import fusion

    sim = _sim(np.ones((2, 20), dtype=np.uint16), ["y", "x"], {"y": 0.3, "x": 0.3}, {"y": 0.0, "x": 0.0})
    props = {"origin": {"y": 0.0, "x": -2.7}, "spacing": {"y": 0.3, "x": 0.3}, "shape": {"y": 2, "x": 29}}
    fused = fusion.fuse([sim], transform_key="k", fusion_func=fusion.max_fusion, interpolation_order=0,
                        output_stack_properties=props, output_chunksize={"y": 2, "x": 10}, merge_chunks=False)
    want = np.tile(np.concatenate([np.zeros(9, np.uint16), np.ones(20, np.uint16)]), (2, 1))
    np.testing.assert_array_equal(np.squeeze(np.asarray(fused.data)), want)


def test_kat_large_origin_roundoff(hip_device):
    """T/test_fusion.py:533-573: large origin, grid-aligned chunk edge tolerates coordinate round-off."""
    from multiview_stitcher_amd import fusion, spatial_image_utils as si

    origin, scale = 861.5120670572916, 0.13810709635416665
    sim = _sim(np.ones((2, 4084), dtype=np.uint16), ["y", "x"], {"y": scale, "x": scale}, {"y": 0.0, "x": origin})
    s = si.get_spacing_from_sim(sim)["x"]
    props = {"origin": {"y": 0.0, "x": origin - 9 * s}, "spacing": {"y": s, "x": s}, "shape": {"y": 2, "x": 4093}}
    fused = fusion.fuse([sim], transform_key="k", fusion_func=fusion.max_fusion, interpolation_order=0,
                        output_stack_properties=props, output_chunksize={"y": 2, "x": 4084}, merge_chunks=False)
    want = np.tile(np.concatenate([np.zeros(9, np.uint16), np.ones(4084, np.uint16)]), (2, 1))
    np.testing.assert_array_equal(np.squeeze(np.asarray(fused.data)), want)


def test_kat_fractional_translation_grid(hip_device):
    """T/test_fusion.py:756-810: four 10x10 tiles at fractional 8.5 offsets -> 18x18, max 4, min > 0."""
    from multiview_stitcher_amd import fusion, msi_utils

    a = 8.5
    msims = []
    for iv, tr in enumerate([{"y": 0, "x": 0}, {"y": a, "x": 0}, {"y": 0, "x": a}, {"y": a, "x": a}]):
        sim = _sim(np.full((2, 10, 10), iv + 1, dtype=np.uint16), ["c", "y", "x"], {"y": 1, "x": 1}, tr)
        msims.append(msi_utils.get_msim_from_sim(sim, scale_factors=[]))
    fused_msim = fusion.fuse(images=msims, transform_key="k", output_chunksize={"y": 5, "x": 5}, merge_chunks=False)
    assert msi_utils.is_msim(fused_msim)              # MultiscaleSpatialImages in, a multiscale result out (_core.py:939-1064)
    fused = msi_utils.get_sim_from_msim(fused_msim, scale="scale0")
    d = np.asarray(fused.data)
    assert fused.sizes["y"] == 18 and fused.sizes["x"] == 18
    assert d.max() == 4 and d.min() > 0


def test_kat_fused_field_slice(hip_device):
    """T/test_fusion.py:932-987: a single output plane of an anisotropic, translated view equals imval everywhere."""
    from multiview_stitcher_amd import fusion, param_utils

    imval = 1.0
    sdims = ["z", "y", "x"]
    spacing = {"z": 3.5, "y": 2.5, "x": 4.5}
    tr = {"z": 1.3, "y": 1, "x": 2}
    sim = _sim(np.full((5, 50, 100), imval, np.float32), sdims, spacing, None,
               affine=param_utils.affine_from_translation([tr[d] for d in sdims]))
    props = {"spacing": spacing, "origin": {d: t + 1 * spacing[d] for d, t in tr.items()}, "shape": {"z": 1, "y": 40, "x": 70}}
    fused = fusion.fuse([sim], transform_key="k", interpolation_order=1, output_stack_properties=props)
    assert not np.any(np.asarray(fused.data).ravel() - imval)


def test_kat_halo_chunks_match_unchunked(hip_device):
    """T/test_fusion.py:439-477 spirit: chunked fusion with a halo equals single-chunk fusion (blend weights do not
    depend on the chunking)."""
    from multiview_stitcher_amd import fusion, sample_data

    sims, _, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(40, 44), tiles=(2, 2), overlap=(10, 12), max_jitter=0)
    a = fusion.fuse(sims, transform_key=sample_data.METADATA_TRANSFORM_KEY, output_chunksize={"y": 7, "x": 7}, overlap_in_pixels=3)
    b = fusion.fuse(sims, transform_key=sample_data.METADATA_TRANSFORM_KEY, output_chunksize={"y": 1000, "x": 1000})
    np.testing.assert_array_equal(np.asarray(a.data), np.asarray(b.data))


def test_fuse_multi_device_farm_equals_single(hip_device):
    """T/test_browser.py:760-786 shape: the same inputs through the 1-device and the N-device executors give
    identical results (N fake devices = the same GPU twice when only one is present)."""
    from multiview_stitcher_amd import executors, fusion, registration, sample_data

    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(16, 40, 40), tiles=(1, 2, 2), overlap=(0, 10, 10), max_jitter=1)
    key = sample_data.METADATA_TRANSFORM_KEY
    one = fusion.fuse(sims, transform_key=key, output_chunksize={"z": 16, "y": 32, "x": 32})
    two = executors.fuse_on_devices(sims, devices=(0, 0), transform_key=key, output_chunksize={"z": 16, "y": 32, "x": 32})
    np.testing.assert_array_equal(np.asarray(one.data), np.asarray(two.data))
    p1 = registration.register(sims, transform_key=key, reg_channel_index=0)
    p2 = registration.register(sims, transform_key=key, reg_channel_index=0, pairwise_executor=executors.DevicePairExecutor((0, 0)))
    for a, b in zip(p1, p2):
        np.testing.assert_allclose(a, b, atol=1e-6)


def test_chunked_fuse_with_device_resident_mosaic(hip_device):
    """The chunked workflow with tiles and result on the device: every chunk is fused on the GPU and copied into its window
    of the mosaic device-to-device (mvs_copy_into), calls return without synchronising.  Must equal the same chunking with
    host output bit for bit, and the unchunked result up to the float32 rounding of chunk-relative coordinates (+-1)."""
    from multiview_stitcher_amd import fusion, sample_data
    from multiview_stitcher_amd.device import DeviceArray

    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(40, 48, 72), tiles=(2, 2, 2), overlap=(10, 12, 18),
                                                    dtype=np.uint16, max_jitter=0, seed=3)
    key = sample_data.METADATA_TRANSFORM_KEY
    from tests.helpers import squeeze_field

    fields = [squeeze_field(s) for s in sims]
    dsims = [f.copy(data=DeviceArray.from_host(np.ascontiguousarray(f.data), 0)) for f in fields]
    kw = dict(transform_key=key, output_chunksize={"z": 32, "y": 32, "x": 32}, merge_chunks=False)
    host = np.asarray(fusion.fuse(fields, **kw).data)
    dev = fusion.fuse(dsims, output_on_backend=True, **kw)       # slabs are strided windows of the device tiles
    np.testing.assert_array_equal(dev.data.get().reshape(host.shape), host)
    whole = np.asarray(fusion.fuse(fields, transform_key=key, output_chunksize={d: 4096 for d in "zyx"}).data)
    diff = host.astype(np.int64) - whole.astype(np.int64)
    assert np.abs(diff).max() <= 1
    assert (diff != 0).mean() < 0.02
    # merge_chunks (the default): the 32^3 request becomes one launch block = the unchunked result, host or device
    kw["merge_chunks"] = True
    np.testing.assert_array_equal(np.asarray(fusion.fuse(fields, **kw).data), whole)
    np.testing.assert_array_equal(fusion.fuse(dsims, output_on_backend=True, **kw).data.get().reshape(whole.shape), whole)


@pytest.mark.parametrize("ndim", [2, 3])
def test_lean_pair_path_equals_generic_on_device(hip_device, ndim):
    """register() through the lean per-pair host path (plain-float plans + mvs_register_views: resample both crops and register
    them in one call) and through the generic numpy path (transform_sim x 2 + mvs_register_crops): identical transforms,
    qualities and bounding boxes, with device-resident tiles, binning and non-integer stage positions."""
    from multiview_stitcher_amd import device, registration, sample_data, spatial_image_utils as si

    if ndim == 2:
        sims, _, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(160, 144), tiles=(2, 3), overlap=(40, 36), dtype=np.uint16, max_jitter=3, seed=3,
                                                        spacing=(0.7, 0.7))
        binning = {"y": 1, "x": 1}
    else:
        sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(40, 96, 80), tiles=(2, 2, 2), overlap=(12, 24, 20), dtype=np.uint16, max_jitter=2,
                                                        seed=8, spacing=(2.0, 0.5, 0.5))
        binning = {"z": 1, "y": 2, "x": 2}
    sims = [device.to_device(s.isel({d: 0 for d in si.get_nonspatial_dims_from_sim(s)}), 0) for s in sims]
    out = []
    for lean in (True, False):
        registration._lean_enabled[0] = lean
        try:
            out.append(registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, registration_binning=binning, return_dict=True,
                                             n_parallel_pairwise_regs=3))
        finally:
            registration._lean_enabled[0] = True
    a, b = out
    assert a["pairwise_registration"]["edges"] == b["pairwise_registration"]["edges"] and len(a["pairwise_registration"]["edges"]) >= 4
    for ra, rb in zip(a["pairwise_registration"]["results"][0], b["pairwise_registration"]["results"][0]):
        np.testing.assert_array_equal(ra["transform"], rb["transform"])
        np.testing.assert_array_equal(ra["bbox"], rb["bbox"])
        assert ra["quality"] == rb["quality"]
    for pa, pb in zip(a["params"], b["params"]):
        np.testing.assert_array_equal(pa, pb)


@pytest.mark.parametrize("ndim", [2, 3])
def test_batched_pairs_and_native_host_steps_equal_the_per_pair_path(hip_device, ndim):
    """register() with all pairs in one library call (mvs_plan_pairs + mvs_register_pairs on native worker threads), the overlap
    graph + pruning in one call (mvs_view_graph_prune) and the groupwise resolution in one call (mvs_resolve_translations),
    against the same run through the per-pair interpreter threads and the Python graph / resolution functions: identical edge
    lists, pairwise transforms, qualities, boxes and parameters -- with pre-binned tiles (tickets instead of a host wait),
    without binning, and with an overlap tolerance."""
    import warnings

    from multiview_stitcher_amd import device, registration, sample_data, spatial_image_utils as si

    if ndim == 2:
        sims, _, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(160, 144), tiles=(3, 3), overlap=(40, 36), dtype=np.uint16, max_jitter=3, seed=4,
                                                        spacing=(0.7, 0.7))
        variants = [dict(registration_binning={"y": 1, "x": 1}), dict(registration_binning={"y": 2, "x": 2}, overlap_tolerance=1.5)]
    else:
        sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(40, 96, 80), tiles=(2, 2, 3), overlap=(12, 24, 20), dtype=np.uint16, max_jitter=2,
                                                        seed=9, spacing=(2.0, 0.5, 0.5))
        variants = [dict(registration_binning={"z": 1, "y": 2, "x": 2}), dict(registration_binning={"z": 1, "y": 1, "x": 1}, overlap_tolerance={"z": 0.0, "y": 1.0, "x": 0.5})]
    sims = [device.to_device(s.isel({d: 0 for d in si.get_nonspatial_dims_from_sim(s)}), 0) for s in sims]
    # crops from the raw tiles (binning inside the crop kernel) == crops from binned copies, incl. bins that do not divide the tile
    # and windows that start inside a tile
    for binning in ([{"y": 2, "x": 2}, {"y": 3, "x": 2}] if ndim == 2 else [{"z": 1, "y": 2, "x": 2}, {"z": 3, "y": 2, "x": 2}, {"z": 2, "y": 3, "x": 3}]):
        got = []
        for raw in (True, False):
            registration._raw_crops_enabled[0] = raw
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    got.append(registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, return_dict=True, registration_binning=binning))
            finally:
                registration._raw_crops_enabled[0] = True
        # (the raw tiles serve when every crop is a whole-pixel translation of the BINNED grid, i.e. the bins divide the tile step;
        # otherwise the crops interpolate and the pairs go through binned copies either way)
        steps = {"y": 120, "x": 108} if ndim == 2 else {"z": 28, "y": 72, "x": 60}
        whole = all(steps[d] % b == 0 for d, b in binning.items())
        n_edges = len(got[0]["pairwise_registration"]["edges"])
        # (pairs on the crop-length knife edge take the reference's sequence one by one: registration._reference_crop_differs)
        n_ref = got[0]["bin_cache_stats"]["pairs_on_reference_sequence"]
        assert got[0]["bin_cache_stats"]["pairs_with_raw_crops"] == (n_edges - n_ref if whole else 0) and n_edges > n_ref >= 0
        assert got[1]["bin_cache_stats"]["pairs_with_raw_crops"] == 0
        for ra, rb in zip(got[0]["pairwise_registration"]["results"][0], got[1]["pairwise_registration"]["results"][0]):
            np.testing.assert_array_equal(ra["transform"], rb["transform"])
            np.testing.assert_array_equal(ra["bbox"], rb["bbox"])
            assert ra["quality"] == rb["quality"] or (np.isnan(ra["quality"]) and np.isnan(rb["quality"]))
    for kw in variants:
        out = []
        for native in (True, False):
            registration._batch_enabled[0] = registration._native_graph[0] = registration._native_resolution[0] = native
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out.append(registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, return_dict=True, n_parallel_pairwise_regs=5, **kw))
            finally:
                registration._batch_enabled[0] = registration._native_graph[0] = registration._native_resolution[0] = True
        a, b = out
        assert a["pairwise_registration"]["edges"] == b["pairwise_registration"]["edges"] and len(a["pairwise_registration"]["edges"]) >= 10
        for ra, rb in zip(a["pairwise_registration"]["results"][0], b["pairwise_registration"]["results"][0]):
            np.testing.assert_array_equal(ra["transform"], rb["transform"])
            np.testing.assert_array_equal(ra["bbox"], rb["bbox"])
            assert ra["quality"] == rb["quality"]
        for pa, pb in zip(a["params"], b["params"]):
            np.testing.assert_array_equal(pa, pb)
        ia, ib = a["groupwise_resolution"]["info"][0], b["groupwise_resolution"]["info"][0]
        assert ia["used_edges"] == ib["used_edges"]
        assert ia["metrics"][0]["max_residual"] == ib["metrics"][0]["max_residual"]


def test_batched_pairs_report_constant_overlaps_and_errors(hip_device):
    """A constant overlap gives the identity with a warning on the batched path too (registration.py:1500-1520), and a failing
    pair surfaces as an exception that names the pair."""
    from multiview_stitcher_amd import _lib, device, registration, spatial_image_utils as si

    rng = np.random.default_rng(0)
    tiles = []
    for k in range(3):
        data = rng.integers(0, 4000, (64, 80), dtype=np.uint16)
        if k == 2:
            data[:] = 7                                  # a constant tile: its overlap with tile 1 is constant
        sim = si.to_spatial_image(data, dims=["y", "x"], scale={"y": 1.0, "x": 1.0}, translation={"y": 0.0, "x": 60.0 * k})
        si.set_sim_affine(sim, np.eye(3), "stage")
        tiles.append(device.to_device(sim, 0))
    with pytest.warns(UserWarning, match="all zero or constant"):
        res = registration.compute_pairwise_registrations(tiles, [(0, 1), (1, 2)], "stage", registration_binning={"y": 1, "x": 1})
    assert np.isnan(res[1]["quality"]) and np.array_equal(res[1]["transform"], np.eye(3))
    assert np.isfinite(res[0]["quality"])
    with pytest.raises(ValueError, match="do not overlap"):
        registration.compute_pairwise_registrations(tiles, [(0, 2)], "stage", registration_binning={"y": 1, "x": 1})


def test_fuse_replay_of_a_remembered_geometry_equals_the_full_derivation(hip_device):
    """fuse() of device-resident tiles into one device-resident launch block remembers what it derived for a geometry (output
    stack, slab windows, view records) and replays it with the current tiles' pointers: same voxels as the full derivation, a hit
    only for the same geometry AND parameters, new tiles at new addresses are read (not the remembered ones)."""
    from multiview_stitcher_amd import device, fusion, sample_data, spatial_image_utils as si

    def mosaic(seed):
        sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(24, 64, 72), tiles=(1, 2, 3), overlap=(0, 16, 20), dtype=np.uint16, max_jitter=0, seed=seed)
        return [device.to_device(s.isel({d: 0 for d in si.get_nonspatial_dims_from_sim(s)}), 0) for s in sims]

    key = si.DEFAULT_TRANSFORM_KEY
    sims = mosaic(1)
    fusion._REPLAY_MEMO.clear()
    first = fusion.fuse(sims, transform_key=key, output_on_backend=True)            # derives and remembers
    assert len(fusion._REPLAY_MEMO) == 1
    again = fusion.fuse(sims, transform_key=key, output_on_backend=True)            # replayed
    fusion._REPLAY[0] = False
    try:
        full = fusion.fuse(sims, transform_key=key, output_on_backend=True)
    finally:
        fusion._REPLAY[0] = True
    np.testing.assert_array_equal(np.asarray(again.data), np.asarray(full.data))
    np.testing.assert_array_equal(np.asarray(first.data), np.asarray(full.data))
    for d in "zyx":
        np.testing.assert_array_equal(again.coords[d], full.coords[d])
    np.testing.assert_array_equal(again.attrs["transforms"][key], full.attrs["transforms"][key])
    # other voxels, same geometry: a hit that reads the NEW tiles
    other = mosaic(2)
    got = fusion.fuse(other, transform_key=key, output_on_backend=True)
    assert len(fusion._REPLAY_MEMO) == 1
    fusion._REPLAY[0] = False
    try:
        want = fusion.fuse(other, transform_key=key, output_on_backend=True)
    finally:
        fusion._REPLAY[0] = True
    np.testing.assert_array_equal(np.asarray(got.data), np.asarray(want.data))
    assert not np.array_equal(np.asarray(got.data), np.asarray(full.data))
    # other parameters: a miss, derived anew
    p = np.eye(4)
    p[:3, 3] = [0.0, 1.5, -2.0]
    si.set_sim_affine(other[1], p, key)
    moved = fusion.fuse(other, transform_key=key, output_on_backend=True)
    assert len(fusion._REPLAY_MEMO) == 2
    fusion._REPLAY[0] = False
    try:
        want = fusion.fuse(other, transform_key=key, output_on_backend=True)
    finally:
        fusion._REPLAY[0] = True
    np.testing.assert_array_equal(np.asarray(moved.data), np.asarray(want.data))
    assert moved.shape == want.shape
    # options that change the result are part of the key
    mx = fusion.fuse(other, transform_key=key, output_on_backend=True, fusion_func=fusion.max_fusion)
    assert len(fusion._REPLAY_MEMO) == 3 and not np.array_equal(np.asarray(mx.data), np.asarray(moved.data))


def test_fuse_launch_blocks_respect_budget_and_fall_back(hip_device, monkeypatch):
    """fuse(merge_chunks=True) sizes its launch blocks from the output bytes PLUS the view slabs that must be staged on
    the device, against the free device memory (mvs_mem_info), and falls back to the requested chunk grid when a merged
    block still fails to allocate.  Whatever the block size, the mosaic is the same."""
    import warnings

    from multiview_stitcher_amd import _lib, fusion, sample_data

    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(24, 48, 56), tiles=(1, 2, 2), overlap=(0, 12, 14), max_jitter=0)
    key = sample_data.METADATA_TRANSFORM_KEY
    cs = {"z": 8, "y": 16, "x": 16}
    want = np.asarray(fusion.fuse(sims, transform_key=key, output_chunksize=cs, merge_chunks=False).data)

    def same(got):
        # the tiles are cut from one volume, so every weighted mean in an overlap is a mean of EQUAL values and sits exactly on a
        # truncation boundary: block sizes that lead to different kernels (tiny chunks take the column kernel) may land on
        # either side of it in a handful of voxels, by one count
        d = np.abs(np.asarray(got).astype(np.int64) - want.astype(np.int64))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3
    shapes = []
    real = fusion.fuse_np

    def spy(*a, **k):
        shapes.append(tuple(int(v) for v in k["output_properties"]["shape"].values()))
        return real(*a, **k)

    monkeypatch.setattr(fusion, "fuse_np", spy)
    # (1) default budget: one launch block
    same(fusion.fuse(sims, transform_key=key, output_chunksize=cs).data)
    assert len(shapes) == 1
    # (2) a small output cap (MVS_MAX_LAUNCH_BYTES): several blocks of whole chunks
    shapes.clear()
    monkeypatch.setattr(fusion, "MAX_LAUNCH_BYTES", 40_000)
    same(fusion.fuse(sims, transform_key=key, output_chunksize=cs).data)
    assert len(shapes) > 1 and all(s[k] % c == 0 or True for s in shapes for k, c in enumerate(cs.values()))
    assert max(int(np.prod(s)) * 2 for s in shapes) <= 40_000
    monkeypatch.setattr(fusion, "MAX_LAUNCH_BYTES", 32 << 30)
    # (3) host-resident tiles and little free device memory: the staged slabs count against the budget
    shapes.clear()
    tile_bytes = 24 * 48 * 56 * 2
    monkeypatch.setattr(_lib, "mem_info", lambda device=0: (int((tile_bytes + 60_000) / 0.9), 1 << 34))
    same(fusion.fuse(sims, transform_key=key, output_chunksize=cs).data)
    assert len(shapes) > 1 and max(int(np.prod(s)) * 2 for s in shapes) < 60_000
    monkeypatch.undo()
    # (4) a merged block that fails to allocate: warning + the requested chunk grid
    calls = []

    def failing(*a, **k):
        shp = tuple(int(v) for v in k["output_properties"]["shape"].values())
        calls.append(shp)
        if int(np.prod(shp)) > 8 * 16 * 16:
            raise _lib.DeviceMemoryError("mvs_fuse_chunk failed (code -5): hipMalloc(123456) failed: out of memory", -5)
        return real(*a, **k)

    monkeypatch.setattr(fusion, "fuse_np", failing)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = np.asarray(fusion.fuse(sims, transform_key=key, output_chunksize=cs).data)
    same(got)
    assert any("falling back" in str(x.message) for x in w) and len(calls) > 2
    # merge_chunks passed positionally is overridden all the same (the arguments are bound to the signature); other errors
    # -- also ones that merely mention an allocation -- are not retried
    import inspect

    names = list(inspect.signature(fusion._fuse_once).parameters)
    args = [p.default for p in inspect.signature(fusion._fuse_once).parameters.values()][:names.index("merge_chunks") + 1]
    args[0], args[names.index("transform_key")], args[names.index("output_chunksize")], args[-1] = sims, key, cs, True
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        same(np.asarray(fusion.fuse(*args).data))

    def failing_other(*a, **k):
        raise RuntimeError("mvs_fuse_chunk failed (code -2): hipMalloc mentioned, but not an out-of-memory code")

    monkeypatch.setattr(fusion, "fuse_np", failing_other)
    with pytest.raises(RuntimeError, match="code -2"):
        fusion.fuse(sims, transform_key=key, output_chunksize=cs)


def test_fuse_all_fields_on_backend(hip_device):
    """output_on_backend with several (c, t) fields (_core.py:1275-1306 loops the fields): one device array, every field
    fused into its own sub-array; equals the host result."""
    from multiview_stitcher_amd import fusion, param_utils
    from multiview_stitcher_amd.device import is_device_array

    rng = np.random.default_rng(0)
    sims = []
    for iv, tr in enumerate([{"y": 0.0, "x": 0.0}, {"y": 3.0, "x": 20.5}]):
        data = rng.integers(0, 4000, (2, 3, 30, 36)).astype(np.uint16)
        sims.append(_sim(data, ["c", "t", "y", "x"], {"y": 1.0, "x": 1.0}, tr))
    host = fusion.fuse(sims, transform_key="k", output_chunksize={"y": 16, "x": 16})
    dev = fusion.fuse(sims, transform_key="k", output_chunksize={"y": 16, "x": 16}, output_on_backend=True)
    assert is_device_array(dev.data) and list(dev.dims) == ["c", "t", "y", "x"]
    np.testing.assert_array_equal(dev.data.get(), np.asarray(host.data))
    dev_c = fusion.fuse(sims, transform_key="k", output_chunksize={"y": 16, "x": 16}, output_on_backend=True, merge_chunks=False)
    np.testing.assert_array_equal(dev_c.data.get(), np.asarray(host.data))


def test_fuse_untrimmed_halo_chunks(hip_device):
    """trim_overlap=False (_core.py:1252-1254, 1687-1711): the chunks keep their halo and the result is their block
    assembly -- every block equals fuse_np on the chunk's bounding box grown by the halo."""
    from multiview_stitcher_amd import fusion, mv_graph, sample_data
    from multiview_stitcher_amd import spatial_image_utils as si

    sims, _, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(40, 44), tiles=(2, 2), overlap=(10, 12), max_jitter=0)
    key = sample_data.METADATA_TRANSFORM_KEY
    halo, cs = 3, {"y": 32, "x": 40}
    got = fusion.fuse(sims, transform_key=key, output_chunksize=cs, overlap_in_pixels=halo, trim_overlap=False)
    trimmed = fusion.fuse(sims, transform_key=key, output_chunksize=cs, overlap_in_pixels=halo)
    shape = [trimmed.sizes[d] for d in "yx"]
    ny, nx = -(-shape[0] // cs["y"]), -(-shape[1] // cs["x"])
    assert got.sizes["y"] == shape[0] + 2 * halo * ny and got.sizes["x"] == shape[1] + 2 * halo * nx
    g, t = np.asarray(got.data)[0, 0], np.asarray(trimmed.data)[0, 0]
    oy = 0
    for by in range(ny):
        hy = min(cs["y"], shape[0] - by * cs["y"])
        ox = 0
        for bx in range(nx):
            hx = min(cs["x"], shape[1] - bx * cs["x"])
            block = g[oy:oy + hy + 2 * halo, ox:ox + hx + 2 * halo]
            # the block's core is the trimmed chunk
            np.testing.assert_array_equal(block[halo:-halo, halo:-halo], t[by * cs["y"]:by * cs["y"] + hy, bx * cs["x"]:bx * cs["x"] + hx])
            ox += hx + 2 * halo
        oy += hy + 2 * halo


def test_register_on_a_pyramid_level(hip_device):
    """reg_res_level / registration_binning on multiscale images (registration.py:1639-1717): registering level 1 of the
    pyramids is registering the level-1 images themselves; a binning of 2 picks level 1 by itself (msi_utils.py:688-773);
    level + binning applies the remaining factor; the overlap graph stays on scale0."""
    from multiview_stitcher_amd import msi_utils, registration, sample_data

    sims, jit, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(32, 128, 128), tiles=(1, 2, 2), overlap=(0, 40, 40),
                                                      dtype=np.uint16, max_jitter=2, seed=4)
    key = sample_data.METADATA_TRANSFORM_KEY
    sf = [{"z": 1, "y": 2, "x": 2}]
    msims = [msi_utils.get_msim_from_sim(s, scale_factors=sf) for s in sims]
    lvl1 = [msi_utils.get_sim_from_msim(m, scale="scale1") for m in msims]
    kw = dict(transform_key=key, reg_channel_index=0, return_dict=True, groupwise_resolution_kwargs={"reference_view": 0})
    a = registration.register(msims, reg_res_level=1, **kw)
    b = registration.register(lvl1, registration_binning={"z": 1, "y": 1, "x": 1}, **kw)
    c = registration.register(msims, registration_binning={"z": 1, "y": 2, "x": 2}, **kw)       # picks scale1, remainder 1
    assert a["pairwise_registration"]["edges"] == c["pairwise_registration"]["edges"]
    for ra, rb, rc in zip(a["pairwise_registration"]["results"][0], b["pairwise_registration"]["results"][0], c["pairwise_registration"]["results"][0]):
        np.testing.assert_array_equal(ra["transform"], rb["transform"])
        np.testing.assert_array_equal(ra["transform"], rc["transform"])
        assert ra["quality"] == rb["quality"] == rc["quality"]
    from multiview_stitcher_amd import param_utils
    got = np.array([param_utils.select_time(p, 0)[:-1, -1] for p in a["params"]])
    np.testing.assert_allclose(got - got[0], jit - jit[0], atol=1.01)        # half-resolution shifts in y / x
    # level 1 + a total binning of 4 in y / x: the remaining factor 2 is applied to level 1
    d = registration.register(msims, reg_res_level=1, registration_binning={"z": 1, "y": 4, "x": 4}, **kw)
    e = registration.register(lvl1, registration_binning={"z": 1, "y": 2, "x": 2}, **kw)
    for rd, re_ in zip(d["pairwise_registration"]["results"][0], e["pairwise_registration"]["results"][0]):
        np.testing.assert_array_equal(rd["transform"], re_["transform"])
    with pytest.raises(ValueError, match="does not exist"):
        registration.register(msims, reg_res_level=3, **kw)


def test_index_frame_that_cannot_be_applied_warns(hip_device):
    """fuse_np silently dropped a frame_origin whose grid the chunk does not sit on -- and with it the
    voxel-for-voxel guarantee across chunkings.  It warns now; fuse_shard turns the warning into an error."""
    import warnings

    from multiview_stitcher_amd import fusion, sample_data
    from tests.helpers import bb_to_dicts, sim_to_view, squeeze_field, union_bb

    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(16, 40, 48), tiles=(1, 1, 2), overlap=(0, 0, 10), max_jitter=0)
    sims = [squeeze_field(s) for s in sims]
    params = [np.eye(4) for _ in sims]
    views, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(3))
    sd = ["z", "y", "x"]
    fvb = [bb_to_dicts(b, sd) for b in bbs]
    on_grid = {d: float(out_bb["origin"][k]) - 4.0 for k, d in enumerate(sd)}
    off_grid = {d: float(out_bb["origin"][k]) - 4.3 for k, d in enumerate(sd)}
    with warnings.catch_warnings():
        warnings.simplefilter("error", fusion.IndexFrameWarning)
        want = fusion.fuse_np(sims, params, bb_to_dicts(out_bb, sd), full_view_bbs=fvb, frame_origin=on_grid)
    with pytest.warns(fusion.IndexFrameWarning, match="frame_origin cannot be applied"):
        got = fusion.fuse_np(sims, params, bb_to_dicts(out_bb, sd), full_view_bbs=fvb, frame_origin=off_grid)
    np.testing.assert_array_equal(got, want)       # (integer offsets: the per-chunk fallback gives the same voxels here)


def test_fuse_of_multiscale_images_is_multiscale(hip_device, tmp_path):
    """fusion/_core.py:939-1064: every output level is FUSED from the coarsest input level that is still fine enough for it
    (centre-of-pixel origins of the coarser levels), not downsampled from the level above; mixed inputs are refused; with a Zarr
    output one level is written and a multiscale image comes back."""
    from multiview_stitcher_amd import fusion, msi_utils, sample_data
    from multiview_stitcher_amd import spatial_image_utils as si

    sims, _, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(160, 200), tiles=(2, 2), overlap=(30, 30), dtype=np.uint16,
                                                    max_jitter=0, seed=3)
    key = sample_data.METADATA_TRANSFORM_KEY
    msims = [msi_utils.get_msim_from_sim(s, scale_factors=[2]) for s in sims]
    out = fusion.fuse(msims, transform_key=key)
    assert msi_utils.is_msim(out) and msi_utils.get_sorted_scale_keys(out) == ["scale0", "scale1"]
    lvl0 = msi_utils.get_sim_from_msim(out, scale="scale0")
    lvl1 = msi_utils.get_sim_from_msim(out, scale="scale1")
    want0 = fusion.fuse(sims, transform_key=key)
    np.testing.assert_array_equal(np.asarray(lvl0.data), np.asarray(want0.data))
    sp0, o0 = si.get_spacing_from_sim(want0), si.get_origin_from_sim(want0)
    assert si.get_spacing_from_sim(lvl1) == {d: 2 * sp0[d] for d in sp0}
    assert si.get_origin_from_sim(lvl1) == {d: o0[d] + sp0[d] / 2 for d in sp0}
    assert lvl1.sizes["y"] == want0.sizes["y"] // 2 and lvl1.sizes["x"] == want0.sizes["x"] // 2
    # level 1 of the result == the level-1 input images fused onto the level-1 output grid
    props1 = {"shape": {d: int(lvl1.sizes[d]) for d in sp0}, "spacing": si.get_spacing_from_sim(lvl1), "origin": si.get_origin_from_sim(lvl1)}
    want1 = fusion.fuse([msi_utils.get_sim_from_msim(m, scale="scale1") for m in msims], transform_key=key, output_stack_properties=props1)
    np.testing.assert_array_equal(np.asarray(lvl1.data), np.asarray(want1.data))
    with pytest.raises(ValueError, match="same kind"):
        fusion.fuse([msims[0], sims[1]], transform_key=key)
    z = fusion.fuse(msims, transform_key=key, output_zarr_url=str(tmp_path / "f.zarr"), output_chunksize={"y": 128, "x": 128})
    assert msi_utils.is_msim(z) and msi_utils.get_sorted_scale_keys(z) == ["scale0"]
    np.testing.assert_array_equal(np.asarray(msi_utils.get_sim_from_msim(z).data), np.asarray(want0.data))
