"""GPU parity: mvs_fuse_chunk (through fusion.fuse_np) against the oracle's fuse_np."""
import numpy as np
import pytest

from oracle import fuse_oracle as fo
from tests.helpers import (assert_fused_close, bb_to_dicts, reference_noise_floor, sim_to_view, squeeze_field,
                           union_bb)

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["fast", "generic", "rows"], autouse=True)
def kernel_path(request, hip_device):
    """Every parity case runs through each kernel family: the default translation fast path (region kernels; float
    tiles: row kernels), the generic affine kernel forced, and the direct-load row kernels for every dtype -- all must match
    the oracle."""
    from multiview_stitcher_amd import _lib

    _lib.set_option("force_generic", 1 if request.param == "generic" else 0)
    _lib.set_option("rows_v1", 1 if request.param == "rows" else 0)
    yield request.param
    _lib.set_option("force_generic", 0)
    _lib.set_option("rows_v1", 0)


def _grid_case(ndim, dtype, tiles, tile_shape, overlap, frac_shift, seed=0, spacing=None):
    from multiview_stitcher_amd import sample_data, spatial_image_utils as si

    sims, jit, _ = sample_data.generate_tiled_dataset(
        ndim=ndim, tile_shape=tile_shape, tiles=tiles, overlap=overlap, dtype=dtype, seed=seed, spacing=spacing
    )
    sims = [squeeze_field(s) for s in sims]
    rng = np.random.default_rng(seed + 7)
    params = []
    for s in sims:
        p = np.eye(ndim + 1)
        if frac_shift:
            p[:ndim, ndim] = rng.uniform(-2, 2, ndim)
        params.append(p)
    return sims, params


def _run_both(sims, params, out_bb, **kw):
    from multiview_stitcher_amd import fusion, spatial_image_utils as si

    sdims = si.get_spatial_dims_from_sim(sims[0])
    views, bbs = zip(*[sim_to_view(s) for s in sims])
    okw = dict(kw)
    fusion_name = okw.pop("fusion", "weighted_average")
    want, want_f, dbg = fo.fuse_np(list(views), params, out_bb, fusion=fusion_name, full_view_bbs=list(bbs),
                                   return_debug=True, **okw)
    floor = reference_noise_floor(dbg, want_f)
    ffunc = {"weighted_average": fusion.weighted_average_fusion, "max": fusion.max_fusion,
             "simple_average": fusion.simple_average_fusion}[fusion_name]
    got = fusion.fuse_np(
        list(sims), params, bb_to_dicts(out_bb, sdims), fusion_func=ffunc,
        full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs],
        interpolation_order=kw.get("interpolation_order", 1),
        trim_overlap_in_pixels=kw.get("trim_overlap_in_pixels", 0),
        blending_widths=kw.get("blending_widths"),
    )
    return got, want, (want_f, floor)


@pytest.mark.parametrize("dtype", [np.uint16, np.float32, np.uint8])
@pytest.mark.parametrize("frac_shift", [False, True])
def test_fuse_2d_grid(hip_device, dtype, frac_shift):
    sims, params = _grid_case(2, dtype, (2, 3), (96, 80), 17, frac_shift)
    if dtype == np.uint8:
        sims = [s.copy(data=(np.asarray(s.data) >> 4).astype(np.uint8)) for s in sims]
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(2))
    got, want, want_f = _run_both(sims, params, out_bb)
    assert_fused_close(got, want, want_f[0], noise_floor=want_f[1])


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
@pytest.mark.parametrize("frac_shift", [False, True])
def test_fuse_3d_grid(hip_device, dtype, frac_shift):
    sims, params = _grid_case(3, dtype, (2, 2, 2), (24, 40, 72), (6, 9, 15), frac_shift)
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(3))
    got, want, want_f = _run_both(sims, params, out_bb)
    assert_fused_close(got, want, want_f[0], noise_floor=want_f[1])


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8, np.float32])
@pytest.mark.parametrize("frac_shift", [False, True])
def test_mixed_class_launch_equals_the_class_launches(hip_device, dtype, frac_shift, kernel_path):
    """Option "fuse_mixed" (profiles/round5_fuse_mixed.txt): the copy / one-view / two-view bricks of the region decomposition in ONE
    launch over a space-ordered, per-XCD balanced and padded list, the class chosen per wavefront -- the same bricks through the same
    code as the per-class launches: identical voxels, and the oracle's."""
    if kernel_path != "fast":
        pytest.skip("the region kernels are the fast path")
    from multiview_stitcher_amd import _lib

    sims, params = _grid_case(3, dtype, (2, 3, 3), (24, 72, 136), (6, 18, 34), frac_shift, seed=3)
    if dtype == np.uint8:
        sims = [s.copy(data=(np.asarray(s.data) >> 4).astype(np.uint8)) for s in sims]
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(3))
    outs = []
    for flag in (0, 1):
        _lib.set_option("fuse_mixed", flag)
        try:
            got, want, want_f = _run_both(sims, params, out_bb)
        finally:
            _lib.set_option("fuse_mixed", 0)
        outs.append(np.asarray(got))
        assert_fused_close(got, want, want_f[0], noise_floor=want_f[1])
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("ndim", [2, 3])
def test_float_tiles_holding_nan_are_poisoned_like_scipy(hip_device, ndim):
    """scipy's linear interpolation reads both taps of every axis even at an integer offset (weights 1 and 0), so a NaN
    right of / below / behind a voxel makes the resampled value NaN (transformation.py:136-139), the view drops out
    of that voxel (weights.py:325-345) and a voxel seen by no other view becomes 0.  At the upper border the second
    tap is the mirrored index n - 2.  Integer offsets, default options."""
    tiles, shape, ov = ((2, 2), (40, 56), 11) if ndim == 2 else ((1, 2, 2), (10, 30, 44), (0, 8, 12))
    sims, params = _grid_case(ndim, np.float32, tiles, shape, ov, False, seed=3)
    rng = np.random.default_rng(5)
    new = []
    for s in sims:
        d = np.asarray(s.data).copy()
        idx = tuple(rng.integers(0, n, 60) for n in d.shape)
        d[idx] = np.nan
        d[tuple(n - 2 for n in d.shape)] = np.nan          # mirrored second tap of the last voxel
        d[(0,) * (d.ndim - 1) + (d.shape[-1] - 2,)] = np.nan
        new.append(s.copy(data=d))
    _, bbs = zip(*[sim_to_view(s) for s in new])
    out_bb = union_bb(bbs, params, np.ones(ndim))
    got, want, want_f = _run_both(new, params, out_bb)
    assert np.isfinite(got).all()
    assert_fused_close(got, want, want_f[0], noise_floor=want_f[1])


@pytest.mark.parametrize("fusion_name", ["max", "simple_average"])
@pytest.mark.parametrize("order", [0, 1])
def test_fuse_modes_and_order(hip_device, fusion_name, order):
    sims, params = _grid_case(3, np.uint16, (1, 2, 2), (20, 33, 47), (0, 7, 11), True)
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(3))
    got, want, want_f = _run_both(sims, params, out_bb, fusion=fusion_name, interpolation_order=order)
    assert_fused_close(got, want, want_f[0], noise_floor=want_f[1])


def test_fuse_order0_weighted(hip_device):
    sims, params = _grid_case(2, np.float32, (2, 2), (50, 61), 9, True)
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(2))
    got, want, want_f = _run_both(sims, params, out_bb, interpolation_order=0)
    assert_fused_close(got, want, want_f[0], noise_floor=want_f[1])


def test_fuse_full_affine_two_views_anisotropic(hip_device):
    """Config C4 in miniature: two views, the second with a full 3x4 affine and z spacing 2."""
    from multiview_stitcher_amd import sample_data, spatial_image_utils as si

    gt = sample_data.make_ground_truth((40, 56, 64), np.float32, seed=3)
    v0 = si.get_sim_from_array(gt, dims=["z", "y", "x"], scale={"z": 1.0, "y": 1.0, "x": 1.0})
    v1 = si.get_sim_from_array(
        np.ascontiguousarray(gt[::2] * 0.9 + 0.05), dims=["z", "y", "x"], scale={"z": 2.0, "y": 1.0, "x": 1.0},
        translation={"z": 0.5, "y": -1.0, "x": 2.0},
    )
    sims = [squeeze_field(v0), squeeze_field(v1)]
    th = np.deg2rad(7.0)
    R = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
    S = np.diag([1.01, 0.99, 1.0])
    c = np.array([20.0, 28.0, 32.0])
    p1 = np.eye(4)
    p1[:3, :3] = R @ S
    p1[:3, 3] = c - p1[:3, :3] @ c + np.array([3.3, -2.1, 4.7])
    params = [np.eye(4), p1]
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.array([1.0, 1.0, 1.0]))
    got, want, want_f = _run_both(sims, params, out_bb, blending_widths={"z": 4.0, "y": 6.0, "x": 6.0})
    assert_fused_close(got, want, want_f[0], noise_floor=want_f[1])


def test_fuse_trim_and_output_spacing(hip_device):
    """Chunk with halo trimmed (trim_overlap_in_pixels) and an output spacing != input spacing."""
    sims, params = _grid_case(3, np.uint16, (1, 2, 2), (18, 40, 44), (0, 8, 8), True)
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.array([1.0, 0.75, 1.25]))
    got, want, want_f = _run_both(sims, params, out_bb, trim_overlap_in_pixels=3)
    assert got.shape == tuple(out_bb["shape"] - 6)
    assert_fused_close(got, want, want_f[0], noise_floor=want_f[1])


def test_single_view_is_exact_including_corner_quirk(hip_device):
    """One view, identity: every voxel equals the input except where the blending weight rounds
    to 0 in float32 (tile corners; weights.py:502-507) -> the reference outputs 0 there."""
    sims, params = _grid_case(3, np.uint16, (1, 1, 1), (33, 65, 130), 0, False)
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(3))
    got, want, _ = _run_both(sims, params, out_bb)
    np.testing.assert_array_equal(got, want)


def test_device_resident_slabs_and_output(hip_device):
    """Strided device windows in, device array out == host path."""
    from multiview_stitcher_amd import device, fusion, spatial_image_utils as si

    sims, params = _grid_case(3, np.uint16, (1, 2, 2), (16, 48, 52), (0, 10, 12), True)
    sdims = ["z", "y", "x"]
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(3))
    host = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims),
                          full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs])
    dsims = [device.to_device(s, 0) for s in sims]
    # crop a window of every tile: zero-copy strided slabs with shifted origins
    dsl = [s.isel({"y": slice(2, None), "x": slice(3, None)}) for s in dsims]
    hsl = [s.isel({"y": slice(2, None), "x": slice(3, None)}) for s in sims]
    a = fusion.fuse_np(hsl, params, bb_to_dicts(out_bb, sdims), full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs])
    b = fusion.fuse_np(dsl, params, bb_to_dicts(out_bb, sdims), full_view_bbs=[bb_to_dicts(b_, sdims) for b_ in bbs],
                       output_on_backend=True)
    assert device.is_device_array(b)
    np.testing.assert_array_equal(a, b.get())
    assert host.shape == a.shape


def _assert_cb_float_close(got, want):
    """north_star's bar for float32 fused voxels: 1e-4 relative (values below 1e-3 of the data range: 1e-4 of that floor)."""
    rng = float(np.nanmax(np.abs(want)) or 1.0)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = 1e-4 * np.maximum(np.abs(want), 1e-3 * rng)
    assert np.all(err <= tol), f"{int((err > tol).sum())} / {err.size} voxels beyond 1e-4; worst rel {(err / np.maximum(np.abs(want), 1e-3 * rng)).max():.2e}"


@pytest.mark.parametrize("ndim,dtype", [(3, np.uint16), (2, np.float32)])
def test_fuse_content_based_weights(hip_device, ndim, dtype, kernel_path):
    """weights_func=content_based (weights.py:22-74) with its halo (2*sigma_2) trimmed: chunk-level parity."""
    from multiview_stitcher_amd import fusion, spatial_image_utils as si

    if kernel_path != "fast":
        pytest.skip("content-based weights have a single implementation")
    if ndim == 3:
        sims, params = _grid_case(3, dtype, (1, 2, 2), (20, 40, 44), (0, 12, 14), True, seed=4)
        sig = {"sigma_1": 1.5, "sigma_2": 3.0}
    else:
        sims, params = _grid_case(2, dtype, (2, 2), (60, 70), (20, 22), True, seed=5)
        sig = {"sigma_1": 2.0, "sigma_2": 4.0}
    sdims = si.get_spatial_dims_from_sim(sims[0])
    views, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(ndim))
    halo = int(2 * sig["sigma_2"])
    want, want_f, dbg = fo.fuse_np(list(views), params, out_bb, full_view_bbs=list(bbs), weights="content_based",
                                   weights_kwargs=sig, trim_overlap_in_pixels=halo, return_debug=True)
    got = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), weights_func=fusion.content_based,
                         weights_func_kwargs=sig, full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs],
                         trim_overlap_in_pixels=halo)
    assert got.shape == want.shape
    if np.issubdtype(dtype, np.integer):
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 0.02
    else:
        _assert_cb_float_close(got, want)


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_fuse_content_based_default_sigmas_3d(hip_device, dtype, kernel_path):
    """The reference's DEFAULT content-based parameters sigma_1 = 5, sigma_2 = 11 (weights.py:26-27; Gaussian radii 20
    and 44, halo 2 * sigma_2 = 22 px: the C3 configuration) on a 3D chunk of (64 + 44)^3-class extent, so that lines
    both shorter and longer than the LDS-staged filter's tile take part.  Fractional offsets."""
    from multiview_stitcher_amd import fusion, spatial_image_utils as si

    if kernel_path != "fast":
        pytest.skip("content-based weights have a single implementation")
    sims, params = _grid_case(3, dtype, (1, 2, 2), (108, 84, 90), (0, 40, 44), True, seed=11)
    sig = {"sigma_1": 5.0, "sigma_2": 11.0}
    sdims = si.get_spatial_dims_from_sim(sims[0])
    views, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(3))
    halo = 22
    want, want_f, dbg = fo.fuse_np(list(views), params, out_bb, full_view_bbs=list(bbs), weights="content_based",
                                   weights_kwargs=sig, trim_overlap_in_pixels=halo, return_debug=True)
    got = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), weights_func=fusion.content_based,
                         weights_func_kwargs=sig, full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs],
                         trim_overlap_in_pixels=halo)
    assert got.shape == want.shape and min(got.shape) >= 64
    if np.issubdtype(dtype, np.integer):
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 0.02
        # the +-1 LSB flips must sit at truncation boundaries of the reference's own float result
        frac = want_f - np.floor(want_f)
        assert np.all(np.minimum(frac, 1 - frac)[d == 1] < 1e-4 * np.maximum(np.abs(want_f[d == 1]), 1.0))
    else:
        _assert_cb_float_close(got, want)


@pytest.mark.parametrize("ndim,dtype", [(3, np.uint16), (3, np.float32), (2, np.float32), (-3, np.uint16)])
def test_content_based_paired_passes_equal_separate_passes(hip_device, ndim, dtype, kernel_path):
    """Round 4: value and mask lines of every NaN-aware Gaussian are filtered in one launch (gauss1d_pair_kernel), the
    preparation and the two quotients fused into the first / last pass, the views' chains side by side on the context's side
    streams.  Same arithmetic per quantity -> bit for bit the result of the separate passes (option ``cb_unpaired``)."""
    from multiview_stitcher_amd import _lib, fusion, spatial_image_utils as si

    if kernel_path != "fast":
        pytest.skip("content-based weights have a single implementation")
    if ndim == -3:
        # long lines along z, boxes that start inside the chunk on every axis (a view whose first staged sample is invalid while
        # the rest of the tile is not: the constant-tile short cut must decide per quantity)
        ndim = 3
        sims, params = _grid_case(3, dtype, (2, 2, 2), (300, 60, 70), (80, 20, 24), True, seed=21)
        sig, halo = {"sigma_1": 5.0, "sigma_2": 11.0}, 22
    elif ndim == 3:
        sims, params = _grid_case(3, dtype, (2, 2, 2), (70, 60, 66), (30, 24, 26), True, seed=21)
        sig, halo = {"sigma_1": 5.0, "sigma_2": 11.0}, 22
    else:
        sims, params = _grid_case(2, dtype, (2, 2), (120, 140), (40, 44), True, seed=22)
        sig, halo = {"sigma_1": 2.0, "sigma_2": 4.0}, 8
    sdims = si.get_spatial_dims_from_sim(sims[0])
    views, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(ndim))
    kw = dict(weights_func=fusion.content_based, weights_func_kwargs=sig, full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs],
              trim_overlap_in_pixels=halo)
    _lib.set_option("cb_exact", 1)      # (round 6: the default is the fast path of mvs_gauss_fast.inc; these are the bit-faithful passes)
    try:
        got = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), **kw)
        _lib.set_option("cb_unpaired", 1)
        try:
            ref = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), **kw)
        finally:
            _lib.set_option("cb_unpaired", 0)
        np.testing.assert_array_equal(got, ref)
        assert len(sims) == 2 ** ndim and np.isfinite(got.astype(np.float64)).all()
        # ... and so are the paired y / z passes that keep both quantities in one workgroup (the default splits them: one quantity
        # per workgroup, twice the lines)
        _lib.set_option("cb_nosplit", 1)
        try:
            np.testing.assert_array_equal(fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), **kw), ref)
        finally:
            _lib.set_option("cb_nosplit", 0)
    finally:
        _lib.set_option("cb_exact", 0)


@pytest.mark.parametrize("case", ["u16_int", "u16_frac", "f32_frac", "2d", "long_z", "nan_blob"])
def test_content_based_mask_tables_equal_the_filtered_mask(hip_device, case, kernel_path):
    """Round 5: the valid mask of a view -- (view finite) & !(normalised blending weight < 1e-7) -- is checked on the device to be
    a BOX (count of valid voxels == volume of their bounding box); its z / y mask passes then come from 1-D / 2-D tables and the x
    pass stages table(z, y) mx(x) instead of a filtered mask (``cb_mask_bbox_kernel`` / ``cb_mask_table_kernel``, option
    ``cb_mask_closed_form``; default OFF: at C3's tile size the mask is a box minus a few corner voxels, see profiles/round5_cb_mask.txt).  Same sums in the same order -> the fused chunk equals the fully filtered form BIT FOR
    BIT: integer and fractional offsets, float tiles, 2D, boxes that start inside the chunk, and a float tile with a NaN blob
    inside -- whose mask is NOT a box and must keep the filtered path (decided per view by the same record)."""
    from multiview_stitcher_amd import _lib, fusion, spatial_image_utils as si

    if kernel_path != "fast":
        pytest.skip("content-based weights have a single implementation")
    sig, halo, ndim = {"sigma_1": 5.0, "sigma_2": 11.0}, 22, 3
    if case == "u16_int":
        sims, params = _grid_case(3, np.uint16, (2, 2, 2), (70, 60, 66), (30, 24, 26), False, seed=31)
    elif case == "u16_frac":
        sims, params = _grid_case(3, np.uint16, (2, 2, 2), (70, 60, 66), (30, 24, 26), True, seed=32)
    elif case == "f32_frac":
        sims, params = _grid_case(3, np.float32, (1, 2, 2), (64, 72, 80), (0, 30, 34), True, seed=33)
    elif case == "2d":
        sims, params = _grid_case(2, np.uint16, (2, 2), (120, 140), (40, 44), True, seed=34)
        sig, halo, ndim = {"sigma_1": 2.0, "sigma_2": 4.0}, 8, 2
    elif case == "long_z":
        sims, params = _grid_case(3, np.uint16, (2, 2, 2), (300, 60, 70), (80, 20, 24), True, seed=35)
    else:
        sims, params = _grid_case(3, np.float32, (1, 2, 2), (64, 72, 80), (0, 30, 34), False, seed=36)
        d = np.array(sims[1].data, dtype=np.float32, copy=True)
        d[20:30, 25:40, 30:50] = np.nan                       # a hole inside the tile: its mask is not a box
        sims[1] = sims[1].copy(data=d)
    sdims = si.get_spatial_dims_from_sim(sims[0])
    views, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(ndim))
    kw = dict(weights_func=fusion.content_based, weights_func_kwargs=sig, full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs],
              trim_overlap_in_pixels=halo)
    _lib.set_option("cb_exact", 1)      # (round 6: both sides of this comparison are forms of the bit-faithful passes)
    _lib.set_option("cb_mask_count", 1)
    _lib.set_option("cb_mask_closed_form", 1)
    for key in ("cb_mask_views", "cb_mask_boxes"):
        _lib.get_counter(key, reset=True)
    try:
        got = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), **kw)
        n_views, n_boxes = _lib.get_counter("cb_mask_views", reset=True), _lib.get_counter("cb_mask_boxes", reset=True)
        _lib.set_option("cb_mask_count", 0)
        _lib.set_option("cb_mask_closed_form", 0)
        ref = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), **kw)        # every mask through the line filters
    finally:
        _lib.set_option("cb_mask_count", 0)
        _lib.set_option("cb_mask_closed_form", 0)
        _lib.set_option("cb_exact", 0)
    np.testing.assert_array_equal(got, ref)
    assert n_views == len(sims)
    if case == "nan_blob":
        assert n_boxes == len(sims) - 1             # the tile with the hole keeps the filtered path
    elif case == "long_z":
        assert 1 <= n_boxes <= len(sims)            # (views that reach the chunk by a sliver thinner than the zero-weight rim have no valid voxel)
    else:
        assert n_boxes == len(sims)                 # every view's mask was found to be a box: the tables were used
    assert np.isfinite(np.asarray(got, dtype=np.float64)).all() or case == "nan_blob"


def _cb_case(case):
    sig, halo, ndim = {"sigma_1": 5.0, "sigma_2": 11.0}, 22, 3
    if case == "u16_int":
        sims, params = _grid_case(3, np.uint16, (2, 2, 2), (70, 60, 66), (30, 24, 26), False, seed=41)
    elif case == "u16_frac":
        sims, params = _grid_case(3, np.uint16, (2, 2, 2), (70, 60, 66), (30, 24, 26), True, seed=42)
    elif case == "f32_frac":
        sims, params = _grid_case(3, np.float32, (1, 2, 2), (64, 72, 80), (0, 30, 34), True, seed=43)
    elif case == "2d":
        sims, params = _grid_case(2, np.float32, (2, 2), (120, 140), (40, 44), True, seed=44)
        sig, halo, ndim = {"sigma_1": 2.0, "sigma_2": 4.0}, 8, 2
    elif case == "long_z":
        sims, params = _grid_case(3, np.uint16, (2, 2, 2), (300, 60, 70), (80, 20, 24), True, seed=45)
    elif case in ("nan_pinholes", "nan_blob"):
        sims, params = _grid_case(3, np.float32, (1, 2, 2), (64, 72, 80), (0, 30, 34), False, seed=46)
        d = np.array(sims[1].data, dtype=np.float32, copy=True)
        if case == "nan_blob":
            d[20:30, 25:40, 30:50] = np.nan                   # 3 000 voxels: more than a list holds
        else:
            d[20:22, 25:27, 30:33] = np.nan                   # 12 voxels in the interior ...
            d[3, 70, 5] = np.nan                              # ... and single ones next to the tile's border and the chunk's reflection
            d[60, 2, 77] = np.nan
        sims[1] = sims[1].copy(data=d)
    else:
        raise ValueError(case)
    return sims, params, sig, halo, ndim


@pytest.mark.parametrize("taps", ["f32", "f64"])
@pytest.mark.parametrize("case", ["u16_int", "u16_frac", "f32_frac", "2d", "long_z", "nan_pinholes", "nan_blob"])
def test_content_based_fast_path_against_oracle_and_exact_passes(hip_device, case, taps, kernel_path):
    """Round 6: the DEFAULT content-based path (csrc/mvs_gauss_fast.inc): the valid mask of a view is its box minus a short list of
    voxels found on the device, gaussian(mask) comes from 1-D tables minus the listed voxels' separable bumps, every line pass
    carries one quantity and all views of the chunk share one launch per pass; float64 taps (``cb_taps_f64`` = 0: float32).
    Bar: the ORACLE at north_star's tolerance (float 1e-4 relative, u16 +-1 LSB at truncation boundaries) -- and the same bar
    against the bit-faithful passes (option ``cb_exact``).  ``nan_pinholes``: NaN voxels inside a float tile go through the list
    (their bumps, incl. the images under the chunk's reflection); ``nan_blob``: a hole larger than the list raises the overflow
    flag and the chunk is redone on the exact passes -- bit for bit their result."""
    from multiview_stitcher_amd import _lib, fusion, spatial_image_utils as si

    if kernel_path != "fast":
        pytest.skip("content-based weights have a single implementation")
    sims, params, sig, halo, ndim = _cb_case(case)
    sdims = si.get_spatial_dims_from_sim(sims[0])
    views, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(ndim))
    kw = dict(weights_func=fusion.content_based, weights_func_kwargs=sig, full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs],
              trim_overlap_in_pixels=halo)
    want, want_f, dbg = fo.fuse_np(list(views), params, out_bb, full_view_bbs=list(bbs), weights="content_based",
                                   weights_kwargs=sig, trim_overlap_in_pixels=halo, return_debug=True)
    _lib.set_option("cb_exact", 1)
    try:
        exact = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), **kw)
    finally:
        _lib.set_option("cb_exact", 0)
    for key in ("cb_line_launches", "cb_overflows_redone"):
        _lib.get_counter(key, reset=True)
    _lib.set_option("cb_taps_f64", 1 if taps == "f64" else 0)
    try:
        got = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), **kw)
        got_dev = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), output_on_backend=True, **kw).get()
    finally:
        _lib.set_option("cb_taps_f64", 1)      # (the default)
    launches, redone = _lib.get_counter("cb_line_launches", reset=True), _lib.get_counter("cb_overflows_redone", reset=True)
    assert launches == 2 * (2 * ndim)                          # one launch per pass for ALL views of the chunk, two calls
    np.testing.assert_array_equal(got, got_dev)                # (deterministic: the list is sorted, the bumps are summed in list order)
    if case == "nan_blob":
        assert redone == 1                                     # the host-result call noticed the overflow and redid the chunk itself ...
        np.testing.assert_array_equal(got, exact)              # ... and so did the device-result call, through fuse_np's check
        return
    assert redone == 0
    for ref in (want, exact):
        if np.issubdtype(got.dtype, np.integer):
            d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
            assert d.max() <= 1 and (d > 0).mean() < 0.02
            if ref is want:
                frac = want_f - np.floor(want_f)
                assert np.all(np.minimum(frac, 1 - frac)[d == 1] < 1e-4 * np.maximum(np.abs(want_f[d == 1]), 1.0))
        else:
            ok = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), ok)
            _assert_cb_float_close(got[ok], ref[ok])


def test_fuse_content_based_chunked_workflow(hip_device, kernel_path):
    """fusion.fuse(weights_func=content_based): halo = 2*sigma_2 from required_overlap, chunks trimmed (T/test_fusion.py:845-896)."""
    from multiview_stitcher_amd import fusion, sample_data

    if kernel_path != "fast":
        pytest.skip("content-based weights have a single implementation")
    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(12, 40, 40), tiles=(1, 2, 2), overlap=(0, 10, 10), max_jitter=0)
    fused = fusion.fuse(sims, transform_key=sample_data.METADATA_TRANSFORM_KEY, weights_func=fusion.content_based,
                        weights_func_kwargs={"sigma_1": 1, "sigma_2": 2}, output_chunksize={"z": 12, "y": 32, "x": 32})
    d = np.asarray(fused.data)[0, 0]
    assert d.shape == (12, 70, 70) and d[:, 2:-2, 2:-2].min() > 0


def test_more_than_64_views_on_one_column_falls_back(hip_device, kernel_path):
    """The fast kernel lists at most 64 views per column; a chunk where more overlap is redone generically."""
    from multiview_stitcher_amd import fusion, spatial_image_utils as si

    rng = np.random.default_rng(0)
    sims, params = [], []
    for i in range(70):
        arr = rng.integers(100, 4000, (24, 40)).astype(np.uint16)
        s = si.get_sim_from_array(arr, dims=["y", "x"], scale={"y": 1.0, "x": 1.0}, translation={"y": 0.0, "x": 0.0})
        sims.append(squeeze_field(s))
        p = np.eye(3)
        p[:2, 2] = rng.integers(-3, 4, 2)
        params.append(p)
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(2))
    got, want, want_f = _run_both(sims, params, out_bb)
    assert_fused_close(got, want, want_f[0], noise_floor=want_f[1])


def test_user_callables_get_device_resampled_views(hip_device):
    """docs/extension_api_fusion.md: a custom fusion_func / weights_func receives the resampled float32 views, the
    normalised blending weights and its own fusion weights; written like the built-in weighted average it must give
    what mvs_fuse_chunk gives for the same chunk (the built-ins run fused, the callables on device-resampled arrays)."""
    from multiview_stitcher_amd import fusion, spatial_image_utils as si

    sims, params = _grid_case(2, np.float32, (2, 2), (50, 61), 9, True)
    sdims = si.get_spatial_dims_from_sim(sims[0])
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(2))
    seen = {}

    def my_weights(transformed_views, blending_weights, params, gain=1.0):
        seen["weights_args"] = (transformed_views.shape, blending_weights.shape, len(params), gain)
        return np.full(transformed_views.shape, gain, np.float32)

    def my_fusion(transformed_views, blending_weights, fusion_weights, output_spacing, params):
        seen["fusion_args"] = (transformed_views.dtype, sorted(output_spacing), len(params))
        additive = blending_weights * fusion_weights
        wsum = np.nansum(additive, axis=0)
        wsum[wsum == 0] = 1
        return np.nansum(transformed_views * (additive / wsum), axis=0).astype(np.float32)

    kw = dict(full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs], trim_overlap_in_pixels=2)
    got = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), fusion_func=my_fusion, weights_func=my_weights,
                         weights_func_kwargs={"gain": 2.0}, **kw)
    want = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), fusion_func=fusion.weighted_average_fusion, **kw)
    assert got.dtype == want.dtype and got.shape == want.shape
    assert seen["weights_args"][0] == (4,) + tuple(int(v) for v in out_bb["shape"]) and seen["weights_args"][2:] == (4, 2.0)
    assert seen["fusion_args"] == (np.dtype(np.float32), sorted(sdims), 4)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * float(np.max(want)))


@pytest.mark.parametrize("fusion_name", ["weighted_average_fusion", "max_fusion", "simple_average_fusion"])
def test_custom_weights_func_with_builtin_fusion_func(hip_device, fusion_name):
    """The documented extension case (_core.py:1663-1690, docs/extension_api_fusion.md): a user weights_func together with
    a BUILT-IN fusion function.  Constant fusion weights must reproduce the plain built-in result; view-selecting weights
    must reproduce that view."""
    from multiview_stitcher_amd import fusion, spatial_image_utils as si

    sims, params = _grid_case(2, np.float32, (1, 2), (40, 52), (0, 20), True, seed=2)
    sdims = si.get_spatial_dims_from_sim(sims[0])
    _, bbs = zip(*[sim_to_view(s) for s in sims])
    out_bb = union_bb(bbs, params, np.ones(2))
    ffunc = getattr(fusion, fusion_name)
    kw = dict(full_view_bbs=[bb_to_dicts(b, sdims) for b in bbs])

    def flat_weights(transformed_views):
        return np.ones(transformed_views.shape, np.float32)

    got = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), fusion_func=ffunc, weights_func=flat_weights, **kw)
    want = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), fusion_func=ffunc, **kw)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * float(np.max(want)))

    if fusion_name == "weighted_average_fusion":
        def first_view_only(transformed_views):
            w = np.zeros(transformed_views.shape, np.float32)
            w[0] = 1
            return w

        got1 = fusion.fuse_np(list(sims), params, bb_to_dicts(out_bb, sdims), fusion_func=ffunc, weights_func=first_view_only, **kw)
        only = fusion.fuse_np([sims[0]], [params[0]], bb_to_dicts(out_bb, sdims), fusion_func=ffunc, full_view_bbs=[bb_to_dicts(bbs[0], sdims)])
        np.testing.assert_allclose(got1, only, rtol=1e-4, atol=1e-4 * float(np.max(only)))
