"""Blending weights made observable: non-constant tiles with a HAND-DERIVED expected ramp (CPU, pins
oracle/fuse_oracle.py; the same closed form is checked against the HIP path in tests/test_fuse_gpu.py).

Derivation (weights.py:430-511).  For a source stack of N pixels at spacing 1 along an axis the 5-node support grid
spans origin - 1 px ... origin + N px (weights.py:448-457), node spacing (N + 1) / 4; the mask is 1 on the inner 3^n
nodes and the Euclidean distance transform is taken with sampling = node spacing / blending width (weights.py:459-464),
so node i carries min(i, 4 - i) * (N + 1) / (4 bw) along that axis.  Linear interpolation of that table at pixel x (node
coordinate (x + 1) * 4 / (N + 1)) gives, on the first node interval and far from the borders of the other axes,
    W(x) = (x + 1) / bw              at the lower border,
    W(x) = (N - x) / bw              at the upper border,
and the weight is (cos((1 - W) pi) + 1) / 2 = sin^2(pi W / 2) for W < 1, else 1 (weights.py:502-507).  Where the pixel
is within a quarter of the tile of TWO borders the interpolated min-table is the product form W = bw_x * Wx * Wy ... --
not used here: the rows tested are deep in y.  cf. the reference's weight tests _tests/test_weights.py:43-133 (sums of
normalised weights are 0 or 1, weights > 0 wherever a view contributes)."""
import numpy as np
import pytest

from oracle import fuse_oracle as fo


def expected_ramp(dist_px, bw):
    w = np.sin(np.pi / 2 * np.minimum(dist_px / bw, 1.0)) ** 2
    return np.where(dist_px >= bw, 1.0, w)


@pytest.mark.parametrize("n,bw", [(64, 10.0), (97, 7.5), (40, 3.0)])
def test_ramp_along_x_far_from_other_borders(n, bw):
    src = fo.bb([0.0, 0.0], [1.0, 1.0], [n, n])
    w = fo.get_blending_weights(src, src, np.eye(3), blending_widths={"y": bw, "x": bw})
    assert w.dtype == np.float32 and w.shape == (n, n)
    y = n // 2                                  # (y + 1) >= (n + 1) / 4: deep in y
    x = np.arange(n)
    want = np.minimum(expected_ramp(x + 1.0, bw), expected_ramp(n - x.astype(float), bw))
    np.testing.assert_allclose(w[y], want, atol=2e-6)
    np.testing.assert_allclose(w[:, y], want, atol=2e-6)        # and along y by symmetry


def two_tile_case(n=48, overlap=16, bw=6.0, dtype=np.float32):
    """Two tiles side by side along x, non-constant content (different linear ramps), fused on their union grid."""
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    a = (100.0 + 3.0 * xx + 0.5 * yy).astype(dtype)
    b = (900.0 - 2.0 * xx + 0.25 * yy).astype(dtype)
    off = n - overlap
    views = [{"data": a, "origin": np.array([0.0, 0.0]), "spacing": np.ones(2)},
             {"data": b, "origin": np.array([0.0, float(off)]), "spacing": np.ones(2)}]
    bbs = [fo.bb(v["origin"], v["spacing"], v["data"].shape) for v in views]
    out_bb = fo.bb([0.0, 0.0], [1.0, 1.0], [n, off + n])
    # hand-derived expectation on a row deep in y
    x = np.arange(off + n, dtype=np.float64)
    wa = np.where(x < n, np.minimum(expected_ramp(x + 1.0, bw), expected_ramp(n - x, bw)), 0.0)
    xb = x - off
    wb = np.where(xb >= 0, np.minimum(expected_ramp(xb + 1.0, bw), expected_ramp(n - xb, bw)), 0.0)
    return views, bbs, out_bb, (a, b, off, wa, wb)


def test_two_tile_fusion_follows_the_hand_derived_ramp():
    n, overlap, bw = 48, 16, 6.0
    views, bbs, out_bb, (a, b, off, wa, wb) = two_tile_case(n, overlap, bw)
    params = [np.eye(3), np.eye(3)]
    got = fo.fuse_np(views, params, out_bb, full_view_bbs=bbs, blending_widths={"y": bw, "x": bw})
    y = n // 2
    va = np.where(np.arange(off + n) < n, np.pad(a[y].astype(np.float64), (0, off)), 0.0)
    vb = np.where(np.arange(off + n) >= off, np.pad(b[y].astype(np.float64), (off, 0)), 0.0)
    want = (wa * va + wb * vb) / (wa + wb)
    np.testing.assert_allclose(got[y], want, rtol=2e-6)
    # the weights are observable: in the overlap the result moves from tile a's value to tile b's value
    assert abs(got[y, off] - a[y, off]) < abs(got[y, off] - b[y, 0])
    assert abs(got[y, n - 1] - b[y, n - 1 - off]) < abs(got[y, n - 1] - a[y, n - 1])
    mid = off + overlap // 2
    assert min(a[y, mid], b[y, mid - off]) < got[y, mid] < max(a[y, mid], b[y, mid - off])
