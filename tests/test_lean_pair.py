"""CPU test of the lean per-pair host path (registration._lean_register_pair): with the two GPU calls of a pair replaced by
recording stubs, the lean path must hand the very same arguments to the resampler and return the very same result as the
generic numpy path, for random non-dyadic origins / spacings, binning, tolerances, 2D and 3D."""
import numpy as np
import pytest

from multiview_stitcher_amd import _reg_ops, registration, transformation
from multiview_stitcher_amd import spatial_image_utils as si


def _block_mean(data, bins, device=0, wait=True):
    sl = tuple(slice(0, (n // b) * b) for n, b in zip(data.shape, bins))
    shp = []
    for n, b in zip(data.shape, bins):
        shp += [n // b, b]
    return data[sl].reshape(shp).mean(axis=tuple(range(1, 2 * data.ndim, 2))).astype(data.dtype)


@pytest.mark.parametrize("ndim", [2, 3])
def test_lean_pair_equals_generic_path(monkeypatch, ndim):
    rng = np.random.default_rng(ndim)
    calls = []

    def fake_resample(data, matrix, offset, output_shape, order=1, cval=0.0, device=0, out_on_device=None):
        calls[-1].append((tuple(np.asarray(data).shape), np.asarray(data).ravel()[:3].copy(), np.array(matrix, dtype=float), np.array(offset, dtype=float),
                          tuple(int(v) for v in output_shape), order, cval))
        return np.zeros(tuple(int(v) for v in output_shape), dtype=np.float32)

    shifts = {}

    def fake_register_crops(im0, im1, uf, region_mode=None, constant_check=False, device=0):
        t = shifts["t"]
        return np.array(t, dtype=np.float64), 0.75, 0, 5

    monkeypatch.setattr(transformation, "resample_array", fake_resample)
    monkeypatch.setattr(_reg_ops, "register_crops", fake_register_crops)
    monkeypatch.setattr(_reg_ops, "bin_mean", _block_mean)
    sdims = ["z", "y", "x"][-ndim:]
    n_checked = 0
    for trial in range(40):
        shape = rng.integers(24, 60, ndim)
        spacing = rng.choice([1.0, 0.3, 0.6931, 2.5], ndim)
        sims = []
        origin = rng.normal(0, 50, ndim) if trial % 2 else np.round(rng.normal(0, 50, ndim))
        ax = int(rng.integers(ndim))
        for k in range(2):
            off = np.zeros(ndim)
            off[ax] = 0.7 * (shape * spacing)[ax] * k        # the second view is displaced along one axis
            data = rng.integers(0, 4000, shape).astype(np.uint16)
            t = origin * 0 + off + rng.normal(0, 1.5, ndim)
            sims.append(si.get_sim_from_array(data, dims=sdims, scale=dict(zip(sdims, spacing)), translation=dict(zip(sdims, origin)),
                                              transform_key="k", affine=np.block([[np.eye(ndim), t[:, None]], [np.zeros((1, ndim)), np.ones((1, 1))]])))
        sims = [s.isel({d: 0 for d in si.get_nonspatial_dims_from_sim(s)}) for s in sims]
        binning = {d: int(b) for d, b in zip(sdims, rng.choice([1, 1, 2, 3], ndim))}
        tol = {d: float(v) for d, v in zip(sdims, rng.choice([0.0, 0.0, 1.5], ndim))}
        shifts["t"] = rng.integers(-3, 4, ndim) * 0.5
        res = []
        for lean in (True, False):
            registration._lean_enabled[0] = lean
            calls.append([])
            try:
                res.append(registration.register_pair_of_msims(sims[0], sims[1], "k", registration_binning=binning, overlap_tolerance=tol,
                                                               _bin_cache=registration._BinCache()))
            except ValueError as e:
                res.append(str(e))
            finally:
                registration._lean_enabled[0] = True
        a, b = res
        if isinstance(a, str) or isinstance(b, str):
            assert a == b
            continue
        n_checked += 1
        for (s0, d0, m0, o0, os0, or0, cv0), (s1, d1, m1, o1, os1, or1, cv1) in zip(calls[-2], calls[-1]):
            assert s0 == s1 and os0 == os1 and or0 == or1
            np.testing.assert_array_equal(d0, d1)
            np.testing.assert_array_equal(m0, m1)
            np.testing.assert_array_equal(o0, o1)
            assert np.isnan(cv0) and np.isnan(cv1)
        assert len(calls[-2]) == len(calls[-1]) == 2
        np.testing.assert_array_equal(a["transform"], b["transform"])
        np.testing.assert_array_equal(a["bbox"], b["bbox"])
        assert a["quality"] == b["quality"] == 0.75
    assert n_checked >= 20


def test_lean_path_is_skipped_for_rotated_views(monkeypatch):
    data = np.zeros((32, 32), np.uint16)
    c, s_ = np.cos(0.1), np.sin(0.1)
    rot = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0, 0, 1.0]])
    sim = si.get_sim_from_array(data, dims=["y", "x"], scale={"y": 1.0, "x": 1.0}, translation={"y": 0.0, "x": 0.0}, transform_key="k", affine=rot)
    sim = sim.isel({d: 0 for d in si.get_nonspatial_dims_from_sim(sim)})
    assert registration._TileGeom(sim, "k").t is None
