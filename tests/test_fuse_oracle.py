"""CPU: the fusion oracle (oracle/fuse_oracle.py) against the known answers the reference's own fusion / weights tests
hold (constant-tile exact outputs, weight-sum property, shapes).  These pin the checker the GPU parity tests use."""
import numpy as np
import pytest

from oracle import fuse_oracle as fo


def _view(arr, origin, spacing):
    arr = np.asarray(arr)
    o, s = np.asarray(origin, float), np.asarray(spacing, float)
    return {"data": arr, "origin": o, "spacing": s}, fo.bb(o, s, arr.shape)


def test_kat_axis_aligned_translation_max_fusion():
    """T/test_fusion.py:204-237: two constant tiles (1, 2), the second at x = 6, max fusion -> columns 0-5 are 1, 6-13 are 2."""
    (v1, b1), (v2, b2) = _view(np.ones((8, 8), np.float32), (0, 0), (1, 1)), _view(2 * np.ones((8, 8), np.float32), (0, 6), (1, 1))
    out = fo.fuse_np([v1, v2], [np.eye(3), np.eye(3)], fo.bb((0.0, 0.0), (1.0, 1.0), (8, 14)), fusion="max", full_view_bbs=[b1, b2])
    assert out.shape == (8, 14) and out.dtype == np.float32
    np.testing.assert_array_equal(out[:, :6], 1)
    np.testing.assert_array_equal(out[:, 6:], 2)


def test_kat_singleton_slice_order0():
    """T/test_fusion.py:480-530: order-0 fusion of 20 ones at spacing 0.3 into 29 samples starting 9 samples to the left."""
    v, b = _view(np.ones((2, 20), np.uint16), (0.0, 0.0), (0.3, 0.3))
    out = fo.fuse_np([v], [np.eye(3)], fo.bb((0.0, -2.7), (0.3, 0.3), (2, 29)), fusion="max", interpolation_order=0, full_view_bbs=[b])
    want = np.tile(np.concatenate([np.zeros(9, np.uint16), np.ones(20, np.uint16)]), (2, 1))
    np.testing.assert_array_equal(out, want)


def test_kat_fractional_translation_grid():
    """T/test_fusion.py:756-810: four 10x10 tiles (values 1..4) at fractional 8.5 offsets -> 18x18, max 4, min > 0."""
    a = 8.5
    views, bbs = zip(*[_view(np.full((10, 10), iv + 1, np.uint16), tr, (1, 1)) for iv, tr in enumerate([(0, 0), (a, 0), (0, a), (a, a)])])
    out = fo.fuse_np(list(views), [np.eye(3)] * 4, fo.bb((0.0, 0.0), (1.0, 1.0), (18, 18)), full_view_bbs=list(bbs))
    assert out.shape == (18, 18) and out.max() == 4 and out.min() > 0


def test_kat_fused_field_slice():
    """T/test_fusion.py:932-987: one output plane of an anisotropic, translated constant view equals the constant everywhere."""
    spacing, tr = np.array([3.5, 2.5, 4.5]), np.array([1.3, 1.0, 2.0])
    v, b = _view(np.full((5, 50, 100), 1.0, np.float32), (0, 0, 0), spacing)
    p = np.eye(4)
    p[:3, 3] = tr
    out = fo.fuse_np([v], [p], fo.bb(tr + spacing, spacing, (1, 40, 70)), full_view_bbs=[b])
    assert not np.any(out.ravel() - 1.0)


@pytest.mark.parametrize("ndim", [2, 3])
def test_normalized_blending_weights_sum_to_one(ndim):
    """T/test_weights.py:125-133: wherever at least one view has a positive weight the normalised weights sum to 1."""
    shape = (24, 30) if ndim == 2 else (10, 24, 30)
    offs = [np.zeros(ndim), np.array([0] * (ndim - 1) + [21.0]), np.array([0] * (ndim - 2) + [17.0, 0.0])]
    bbs = [fo.bb(o, np.ones(ndim), shape) for o in offs]
    target = fo.bb(np.zeros(ndim), np.ones(ndim), tuple(int(s) + 21 for s in shape))
    w = np.stack([fo.get_blending_weights(target, b, np.eye(ndim + 1)) for b in bbs])
    wn = fo.normalize_weights(w)
    covered = w.sum(0) > 0
    assert covered.any()
    np.testing.assert_allclose(wn.sum(0)[covered], 1.0, atol=1e-6)
    assert np.all(wn.sum(0)[~covered] == 0)


def test_single_view_is_the_view_wherever_its_weight_is_positive():
    """weights.py:325-345: w / w == 1, so one view comes out exactly; where its blend weight rounds to 0 (the corner voxels
    of the float32 cosine, weights.py:502-507) the reference writes nan_to_num(0 / 1 * v) = 0."""
    rng = np.random.default_rng(0)
    data = rng.integers(1, 4000, (12, 40, 44)).astype(np.uint16)
    v, b = _view(data, (0, 0, 0), (1, 1, 1))
    out = fo.fuse_np([v], [np.eye(4)], b, full_view_bbs=[b])
    w = fo.get_blending_weights(b, b, np.eye(4))
    np.testing.assert_array_equal(out[w > 0], data[w > 0])
    np.testing.assert_array_equal(out[w == 0], 0)      # (empty for a tile this small; the zeros appear at the corners of large tiles)
