"""GPU parity of the two standalone entry points of the fuse path against the oracle:
mvs_resample (transformation.transform_sim -> scipy affine_transform, transformation.py:15-148; also the registration
pre-transform with cval = NaN, registration.py:318-338) and mvs_blend_weights (weights.get_blending_weights,
weights.py:391-511), plus the hand-derived blending ramp of tests/test_weights_oracle.py through the HIP fuse."""
import numpy as np
import pytest

from oracle import fuse_oracle as fo
from tests.helpers import bb_to_dicts
from tests.test_weights_oracle import two_tile_case

pytestmark = pytest.mark.gpu


def _random_affine(ndim, rng, rot=0.3, scale=0.1, shift=4.0):
    a = np.eye(ndim + 1)
    m = np.eye(ndim) + rng.uniform(-rot, rot, (ndim, ndim))
    m *= 1 + rng.uniform(-scale, scale, ndim)[:, None]
    a[:ndim, :ndim] = m
    a[:ndim, ndim] = rng.uniform(-shift, shift, ndim)
    return a


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("dtype", [np.uint16, np.float32, np.uint8])
@pytest.mark.parametrize("kind", ["translation", "integer", "affine"])
def test_resample_matches_scipy_with_nan_cval(hip_device, ndim, order, dtype, kind):
    from multiview_stitcher_amd import transformation

    rng = np.random.default_rng(ndim * 10 + order)
    shape = (37, 52) if ndim == 2 else (11, 29, 33)
    data = (rng.random(shape) * 200).astype(dtype)
    if dtype == np.float32:
        data[tuple(rng.integers(0, n, 12) for n in shape)] = np.nan          # NaN taps poison like scipy
    in_o, in_s = rng.uniform(-3, 3, ndim), np.array([1.0, 0.8, 1.3][-ndim:])
    out_shape = tuple(int(n * 1.2) for n in shape)
    out_bb = fo.bb(in_o - 2.7, np.array([0.9, 1.0, 1.1][-ndim:]), out_shape)
    if kind == "affine":
        p = _random_affine(ndim, rng)
    else:
        p = np.eye(ndim + 1)
        p[:ndim, ndim] = rng.integers(-3, 4, ndim) if kind == "integer" else rng.uniform(-3, 3, ndim)
    if kind == "integer":                                                # same grid, integer pixel offsets
        in_s = np.ones(ndim)
        out_bb = fo.bb(in_o - 2.0, np.ones(ndim), out_shape)
    want = fo.transform_array(data.astype(np.float32), p, in_o, in_s, out_bb, order=order, cval=np.nan)
    m, o = transformation.get_pixel_affine(p, in_o, in_s, out_bb["origin"], out_bb["spacing"])
    got = transformation.resample_array(data, m, o, out_shape, order=order, cval=np.nan)
    assert got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))        # exact in-bounds classification
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-5, atol=1e-4 * float(np.nanmax(np.abs(want[ok])) if ok.any() else 1))


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("kind", ["identity", "translation", "affine"])
def test_blend_weights_match_oracle(hip_device, ndim, kind):
    from multiview_stitcher_amd import weights

    rng = np.random.default_rng(ndim)
    sd = ["z", "y", "x"][-ndim:]
    shape = (41, 56) if ndim == 2 else (13, 31, 38)
    src = fo.bb(rng.uniform(-2, 2, ndim), np.array([2.0, 1.0, 1.0][-ndim:]), shape)
    tgt = fo.bb(src["origin"] - 3.3, np.array([1.5, 0.9, 1.1][-ndim:]), tuple(int(n * 1.3) for n in shape))
    if kind == "identity":
        p = np.eye(ndim + 1)
    elif kind == "translation":
        p = np.eye(ndim + 1)
        p[:ndim, ndim] = rng.uniform(-4, 4, ndim)
    else:
        p = _random_affine(ndim, rng, rot=0.2)
    bw = dict(zip(sd, [3.0, 10.0, 6.0][-ndim:]))
    want = fo.get_blending_weights(tgt, src, p, blending_widths=bw)
    got = weights.get_blending_weights(bb_to_dicts(tgt, sd), bb_to_dicts(src, sd), p, blending_widths=bw)
    assert got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_array_equal(got == 0, want == 0)                  # same support
    np.testing.assert_allclose(got, want, atol=3e-6)


@pytest.mark.parametrize("dtype", [np.float32, np.uint16])
def test_two_tile_fusion_follows_the_hand_derived_ramp(hip_device, dtype):
    """Non-constant tiles: the fused overlap must follow sin^2(pi d / (2 bw)) derived in tests/test_weights_oracle.py."""
    from multiview_stitcher_amd import fusion, spatial_image_utils as si

    n, overlap, bw = 48, 16, 6.0
    views, bbs, out_bb, (a, b, off, wa, wb) = two_tile_case(n, overlap, bw, dtype=dtype)
    sims = [si.to_spatial_image(v["data"], dims=["y", "x"], scale=dict(zip("yx", v["spacing"])), translation=dict(zip("yx", v["origin"])))
            for v in views]
    got = fusion.fuse_np(sims, [np.eye(3), np.eye(3)], bb_to_dicts(out_bb, ["y", "x"]),
                         full_view_bbs=[bb_to_dicts(bbv, ["y", "x"]) for bbv in bbs], blending_widths={"y": bw, "x": bw})
    got = np.asarray(got)
    y = n // 2
    va = np.pad(a[y].astype(np.float64), (0, off))
    vb = np.pad(b[y].astype(np.float64), (off, 0))
    want = (wa * va + wb * vb) / (wa + wb)
    if dtype == np.float32:
        np.testing.assert_allclose(got[y], want, rtol=1e-5)
    else:
        assert np.abs(got[y].astype(np.float64) - np.floor(want)).max() <= 1
