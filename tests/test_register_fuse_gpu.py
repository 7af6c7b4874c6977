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
    fused = fusion.fuse(images=msims, transform_key="k", output_chunksize={"y": 5, "x": 5}, merge_chunks=False)
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
