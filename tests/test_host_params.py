"""CPU tests: the product's host-side parameter derivation against the oracle's restatement
(two independent restatements of transformation.py:37-83 and weights.py:430-470)."""
import numpy as np
import pytest

from oracle import fuse_oracle as fo
from multiview_stitcher_amd import transformation, weights


def random_affine(rng, ndim, scale=0.05):
    A = np.eye(ndim + 1)
    A[:ndim, :ndim] += rng.normal(0, scale, (ndim, ndim))
    A[:ndim, ndim] = rng.normal(0, 20, ndim)
    return A


@pytest.mark.parametrize("ndim", [2, 3])
def test_pixel_affine_matches_oracle(ndim):
    rng = np.random.default_rng(0)
    for _ in range(50):
        p = random_affine(rng, ndim)
        in_o, in_s = rng.normal(0, 100, ndim), rng.uniform(0.2, 3, ndim)
        out = fo.bb(rng.normal(0, 100, ndim), rng.uniform(0.2, 3, ndim), [5] * ndim)
        m0, o0 = fo.transform_params(p, in_o, in_s, out)
        m1, o1 = transformation.get_pixel_affine(p, in_o, in_s, out["origin"], out["spacing"])
        np.testing.assert_array_equal(m0, m1)
        np.testing.assert_array_equal(o0, o1)


def test_pixel_affine_large_origin_kat():
    """Restates the reference KAT T/test_transformation.py:41-87: a large shared origin must not
    leak round-off into the offset (to 1e-8) and a small scale must survive the 10-decimal rounding."""
    origin = 1e7 + 0.123456789
    scale = 0.13810709635416665
    p = np.eye(3)
    m, o = transformation.get_pixel_affine(p, [0.0, origin], [scale, scale], [0.0, origin - 9 * scale], [scale, scale])
    np.testing.assert_allclose(m, np.eye(2), atol=1e-10)
    np.testing.assert_allclose(o, [0.0, -9.0], atol=1e-8)
    assert o[1] == -9.0  # snapped to the integer


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("shrink", [0, 1.5])
def test_blending_support_closed_form_equals_scipy_edt(ndim, shrink):
    rng = np.random.default_rng(1)
    for _ in range(20):
        shape = rng.integers(8, 600, ndim)
        src = fo.bb(rng.normal(0, 50, ndim), rng.uniform(0.2, 3, ndim), shape)
        bw = dict(zip(["z", "y", "x"][-ndim:], rng.uniform(1, 20, ndim)))
        t0, o0, s0 = fo.edt_support(src, bw, shrink)
        o0, s0 = fo.coords_origin_spacing(o0, s0, t0.shape)
        t1, o1, s1 = weights.blending_support(src, bw, shrink)
        np.testing.assert_array_equal(t0.astype(np.float32), t1)
        np.testing.assert_array_equal(o0, o1)
        np.testing.assert_array_equal(s0, s1)


def test_embed3_2d():
    m, o = transformation.embed3(np.array([[1.0, 0.1], [0.2, 0.9]]), np.array([3.0, 4.0]))
    np.testing.assert_array_equal(m, [[1, 0, 0], [0, 1.0, 0.1], [0, 0.2, 0.9]])
    np.testing.assert_array_equal(o, [0, 3, 4])


def test_builtin_maps_reference_function_objects_by_name():
    """the reference's own function objects select the kernel mode of the same name."""
    from multiview_stitcher_amd import fusion

    def weighted_average_fusion(transformed_views, blending_weights, fusion_weights=None):   # stand-in for the reference's object
        raise AssertionError

    assert fusion.builtin(weighted_average_fusion) is fusion.weighted_average_fusion
    assert fusion.builtin("max_fusion") is fusion.max_fusion
    assert fusion.builtin(None) is None
    with pytest.raises(NotImplementedError):
        fusion.builtin(lambda x: x)


def test_merged_chunksize_is_whole_multiples_within_budget():
    """fuse(merge_chunks=True): launch blocks are multiples of the requested chunks, the whole stack when it fits."""
    from multiview_stitcher_amd import fusion

    shape = {"z": 1742, "y": 1742, "x": 1742}
    cs = {d: 256 for d in "zyx"}
    assert fusion._merged_chunksize(cs, shape, "zyx", 2) == shape                    # 10.6 GB: one block
    m = fusion._merged_chunksize(cs, shape, "zyx", 2, max_bytes=3 << 30)
    assert all(m[d] % 256 == 0 or m[d] == shape[d] for d in "zyx")
    assert m["z"] * m["y"] * m["x"] * 2 <= 3 << 30
    assert m["z"] <= m["y"] <= m["x"]                                                # z is cut first
    tiny = fusion._merged_chunksize(cs, shape, "zyx", 2, max_bytes=1)
    assert tiny == cs                                                                 # never below the request
    assert fusion._merged_chunksize({"y": 5, "x": 5}, {"y": 18, "x": 18}, "yx", 2) == {"y": 18, "x": 18}


@pytest.mark.parametrize("ndim", [2, 3])
def test_stacked_view_records_equal_the_per_view_form(ndim):
    """fuse_np builds the records of all views of a chunk from stacked arrays (get_pixel_affines, blending_supports,
    embed3_stack): bit for bit the values of the per-view functions, for pure translations and for general affines."""
    rng = np.random.default_rng(5)
    sdims = ["z", "y", "x"][-ndim:]
    n = 7
    ps, origins, spacings, shapes = [], [], [], []
    for i in range(n):
        p = np.eye(ndim + 1)
        if i % 3 == 1:       # general affine
            p[:ndim, :ndim] = np.eye(ndim) + rng.uniform(-0.2, 0.2, (ndim, ndim))
        if i % 3 == 2:       # anisotropic scaling only
            p[:ndim, :ndim] = np.diag(rng.uniform(0.5, 2.0, ndim))
        p[:ndim, ndim] = rng.uniform(-50, 50, ndim)
        ps.append(p)
        origins.append(rng.uniform(-1e4, 1e4, ndim))
        spacings.append(rng.uniform(0.2, 3.0, ndim))
        shapes.append(rng.integers(20, 600, ndim).astype(np.float64))
    out_origin, out_spacing = rng.uniform(-100, 100, ndim), rng.uniform(0.3, 2.0, ndim)
    p_inv = np.linalg.inv(np.stack(ps))
    mats, offs = transformation.get_pixel_affines(p_inv, np.stack(origins), np.stack(spacings), out_origin, out_spacing)
    for shrink in (0, 1.5):
        tables, so, ss = weights.blending_supports(np.stack(origins), np.stack(spacings), np.stack(shapes), sdims, None, shrink)
        wm, wo = transformation.get_pixel_affines(p_inv, so, ss, out_origin, out_spacing)
        for i in range(n):
            m1, o1 = transformation.get_pixel_affine(np.linalg.inv(ps[i]), origins[i], spacings[i], out_origin, out_spacing)
            np.testing.assert_array_equal(mats[i], m1)
            np.testing.assert_array_equal(offs[i], o1)
            bb = {"origin": dict(zip(sdims, origins[i])), "spacing": dict(zip(sdims, spacings[i])), "shape": dict(zip(sdims, shapes[i]))}
            t1, so1, ss1 = weights.blending_support(bb, None, shrink)
            np.testing.assert_array_equal(tables[i], t1)
            np.testing.assert_array_equal(so[i], so1)
            np.testing.assert_array_equal(ss[i], ss1)
            wm1, wo1 = transformation.get_pixel_affine(np.linalg.inv(ps[i]), so1, ss1, out_origin, out_spacing)
            np.testing.assert_array_equal(wm[i], wm1)
            np.testing.assert_array_equal(wo[i], wo1)
    m3, o3 = transformation.embed3_stack(mats, offs)
    for i in range(n):
        a, b = transformation.embed3(mats[i], offs[i])
        np.testing.assert_array_equal(m3[i], a.reshape(-1))
        np.testing.assert_array_equal(o3[i], b)


# ---- pyramid level selection of the pairwise registration (registration.py:1639-1717, msi_utils.py:688-773) -----------------
def _msim_with_levels(shape=(64, 96, 128), factors=(2, 2)):
    from multiview_stitcher_amd import msi_utils
    from multiview_stitcher_amd import spatial_image_utils as si_utils

    rng = np.random.default_rng(0)
    sim = si_utils.to_spatial_image(rng.integers(0, 1000, shape).astype(np.uint16), ["z", "y", "x"],
                                    {"z": 2.0, "y": 0.5, "x": 0.5}, {"z": 3.0, "y": -7.0, "x": 11.0})
    si_utils.set_sim_affine(sim, np.eye(4), "k")
    return msi_utils.get_msim_from_sim(sim, scale_factors=list(factors))


def test_res_level_from_binning_factors_follows_the_reference():
    from multiview_stitcher_amd import msi_utils

    m = _msim_with_levels()                # scale1 = 2x, scale2 = 4x on every axis
    f = msi_utils.get_res_level_from_binning_factors
    assert f(m, {"z": 1, "y": 1, "x": 1}) == ("scale0", {"z": 1, "y": 1, "x": 1})
    assert f(m, {"z": 2, "y": 2, "x": 2}) == ("scale1", {"z": 1, "y": 1, "x": 1})
    assert f(m, {"z": 4, "y": 8, "x": 8}) == ("scale2", {"z": 1, "y": 2, "x": 2})
    assert f(m, {"z": 2, "y": 4, "x": 4}) == ("scale1", {"z": 1, "y": 2, "x": 2})      # scale2 would over-bin z
    assert f(m, {"z": 3, "y": 3, "x": 3}) == ("scale0", {"z": 3, "y": 3, "x": 3})      # 2 does not divide 3
    assert f(m, {"z": 1, "y": 2, "x": 2}) == ("scale0", {"z": 1, "y": 2, "x": 2})      # z must stay at 1
    # a dim missing from the request places no constraint and gets no further binning (msi_utils.py:734-735, 762-763)
    assert f(m, {"y": 4, "x": 4}) == ("scale2", {"z": 1, "y": 1, "x": 1})


def test_select_registration_level_branches():
    from multiview_stitcher_amd import msi_utils, registration

    m1, m2 = _msim_with_levels(), _msim_with_levels()
    sel = registration._select_registration_level
    s1, s2, b = sel(m1, m2, None, 1)                                   # level alone: no binning
    assert s1.shape == (32, 48, 64) and s2.shape == (32, 48, 64) and b == {"z": 1, "y": 1, "x": 1}
    s1, _, b = sel(m1, m2, {"z": 2, "y": 4, "x": 4}, 1)               # level + binning: the rest is still applied
    assert s1.shape == (32, 48, 64) and b == {"z": 1, "y": 2, "x": 2}
    with pytest.raises(ValueError, match="not a divisor"):
        sel(m1, m2, {"z": 3, "y": 4, "x": 4}, 1)
    with pytest.raises(ValueError, match="does not exist"):
        sel(m1, m2, None, 5)
    s1, _, b = sel(m1, m2, {"z": 4, "y": 4, "x": 4}, None)            # binning alone: the lowest level that divides it
    assert s1.shape == (16, 24, 32) and b == {"z": 1, "y": 1, "x": 1}
    # level sims carry their own (shifted) origin and coarser spacing: scale = spacing * f, origin + (f - 1) * spacing / 2
    from multiview_stitcher_amd import spatial_image_utils as si_utils
    assert si_utils.get_spacing_from_sim(s1) == {"z": 8.0, "y": 2.0, "x": 2.0}
    assert si_utils.get_origin_from_sim(s1) == {"z": 3.0 + 3.0, "y": -7.0 + 0.75, "x": 11.0 + 0.75}
    # plain SpatialImages are scale0-only images
    a, b_ = msi_utils.get_sim_from_msim(m1), msi_utils.get_sim_from_msim(m2)
    s1, s2, b = sel(a, b_, {"z": 1, "y": 2, "x": 2}, None)
    assert s1 is a and s2 is b_ and b == {"z": 1, "y": 2, "x": 2}
    with pytest.raises(ValueError, match="does not exist"):
        sel(a, b_, None, 1)


def test_spline_orders_above_one_are_refused_at_call_time():
    """``order`` > 1 used to surface as MVS_ERR_UNSUPPORTED from inside a chunk."""
    from multiview_stitcher_amd import fusion, msi_utils

    sim = msi_utils.get_sim_from_msim(_msim_with_levels(factors=()))
    with pytest.raises(NotImplementedError, match="interpolation_order=3"):
        fusion.fuse([sim, sim], transform_key="k", interpolation_order=3)
    with pytest.raises(NotImplementedError, match="order=3"):
        transformation.transform_sim(sim, np.eye(4), output_stack_properties={"origin": {"z": 0, "y": 0, "x": 0}, "spacing": {"z": 1, "y": 1, "x": 1}, "shape": {"z": 4, "y": 4, "x": 4}}, order=3)
    with pytest.raises(NotImplementedError, match="interpolation_order=2"):
        fusion.fuse_np([sim], [np.eye(4)], {"origin": {"z": 0, "y": 0, "x": 0}, "spacing": {"z": 1, "y": 1, "x": 1}, "shape": {"z": 4, "y": 4, "x": 4}}, interpolation_order=2)
