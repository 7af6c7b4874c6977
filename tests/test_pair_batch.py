"""CPU tests of the batched pair path (registration._register_pairs_batched): the plans mvs_plan_pairs derives for all pairs of
a mosaic equal registration._lean_pair_plan pair by pair, bit for bit (windows, output grid, pixel affines: the arguments
of the crop kernels), and the stacked post-processing (_pair_results_from_plan) returns the very result dicts of
_lean_register_pair -- for random non-dyadic origins / spacings, binned coordinates, tolerances, 2D and 3D.  The library calls
that need a GPU are covered by tests/test_register_fuse_gpu.py (batched == per-pair results on the device)."""
import ctypes as C

import numpy as np
import pytest

from multiview_stitcher_amd import _lib, _reg_ops, registration, transformation
from multiview_stitcher_amd import spatial_image_utils as si


def _views(rng, ndim, n_views, binned):
    sdims = ["z", "y", "x"][-ndim:]
    shape = rng.integers(24, 60, ndim)
    spacing = rng.choice([1.0, 0.3, 0.6931, 2.5], ndim)
    origin = rng.normal(0, 50, ndim) if rng.random() < 0.5 else np.round(rng.normal(0, 50, ndim))
    sims = []
    for k in range(n_views):
        off = rng.integers(-1, 2, ndim) * 0.7 * shape * spacing
        t = off + rng.normal(0, 1.5, ndim) if k else np.zeros(ndim)
        data = np.zeros(tuple(shape), np.uint16)
        sim = si.get_sim_from_array(data, dims=sdims, scale=dict(zip(sdims, spacing)), translation=dict(zip(sdims, origin)), transform_key="k",
                                    affine=np.block([[np.eye(ndim), t[:, None]], [np.zeros((1, ndim)), np.ones((1, 1))]]))
        sim = sim.isel({d: 0 for d in si.get_nonspatial_dims_from_sim(sim)})
        if binned:       # coordinates as _bin_sim leaves them: means of groups of b samples
            bins = rng.choice([1, 2, 3], ndim)
            for d, b in zip(sdims, bins):
                c = sim.coords[d]
                m = len(c) // b
                sim.coords[d] = c[: m * b].reshape(m, b).mean(axis=1)
            sim.data = np.zeros(tuple(len(sim.coords[d]) for d in sdims), np.uint16)
        sims.append(sim)
    return sims, sdims


def _plan_native(geoms, pairs, tol):
    n = len(geoms[0].sdims)
    nv, ne = len(geoms), len(pairs)
    cptr = (C.c_void_p * (nv * n))(*[g.coords[k].ctypes.data for g in geoms for k in range(n)])
    clen = np.array([len(g.coords[k]) for g in geoms for k in range(n)], dtype=np.int64)
    tr = np.array([g.t for g in geoms], dtype=np.float64).reshape(nv, n)
    tolv = np.array(tol, dtype=np.float64)
    pr = np.array(pairs, dtype=np.int32).reshape(ne, 2)
    windows = np.zeros((ne, 2, 3, 2), dtype=np.int64)
    oo, osp = np.zeros((ne, 3)), np.zeros((ne, 3))
    osh = np.ones((ne, 3), dtype=np.int64)
    md, of = np.zeros((ne, 2, 3)), np.zeros((ne, 2, 3))
    st = np.zeros(ne, dtype=np.int32)
    ptr = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    rc = _lib.load().mvs_plan_pairs(n, nv, cptr, ptr(clen, C.c_int64), ptr(tr, C.c_double), ptr(tolv, C.c_double), ne, ptr(pr, C.c_int32),
                                    ptr(windows, C.c_int64), ptr(oo, C.c_double), ptr(osp, C.c_double), ptr(osh, C.c_int64), ptr(md, C.c_double),
                                    ptr(of, C.c_double), ptr(st, C.c_int32))
    assert rc == 0
    return windows, oo, osp, osh, md, of, st


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("binned", [False, True])
def test_native_plans_equal_the_python_plans(ndim, binned):
    rng = np.random.default_rng(10 * ndim + binned)
    n_ok = n_none = 0
    for trial in range(30):
        sims, sdims = _views(rng, ndim, 5, binned)
        geoms = [registration._TileGeom(s, "k") for s in sims]
        pairs = [(a, b) for a in range(5) for b in range(5) if a != b]
        tol = [float(v) for v in rng.choice([0.0, 0.0, 1.5], ndim)]
        windows, oo, osp, osh, md, of, st = _plan_native(geoms, pairs, tol)
        for p, (a, b) in enumerate(pairs):
            want = registration._lean_pair_plan(geoms[a], geoms[b], tol)
            if want is None:
                assert st[p] == 1
                n_none += 1
                continue
            n_ok += 1
            assert st[p] == 0
            for i in range(2):
                assert [tuple(w) for w in windows[p, i, :ndim].tolist()] == [tuple(w) for w in want["windows"][i]]
                assert md[p, i, :ndim].tolist() == want["matrix_diag"][i]
                assert of[p, i, :ndim].tolist() == want["offset"][i]
            assert oo[p, :ndim].tolist() == want["out_origin"] and osp[p, :ndim].tolist() == want["out_spacing"]
            assert osh[p, :ndim].tolist() == want["out_shape"]
    assert n_ok >= 100 and n_none >= 20


@pytest.mark.parametrize("ndim", [2, 3])
def test_stacked_post_processing_equals_the_per_pair_results(monkeypatch, ndim):
    rng = np.random.default_rng(5 + ndim)
    shifts = {}

    def fake_resample(data, matrix, offset, output_shape, order=1, cval=0.0, device=0, out_on_device=None):
        return np.zeros(tuple(int(v) for v in output_shape), dtype=np.float32)

    def fake_register_crops(im0, im1, uf, region_mode=None, constant_check=False, device=0):
        return np.array(shifts["t"], dtype=np.float64), shifts["q"], shifts["status"], 5

    monkeypatch.setattr(transformation, "resample_array", fake_resample)
    monkeypatch.setattr(_reg_ops, "register_crops", fake_register_crops)
    n_checked = 0
    for trial in range(25):
        sims, sdims = _views(rng, ndim, 4, True)
        geoms = [registration._TileGeom(s, "k") for s in sims]
        tol = [float(v) for v in rng.choice([0.0, 0.0, 1.5], ndim)]
        pairs = [(a, b) for a in range(4) for b in range(4) if a != b and registration._lean_pair_plan(geoms[a], geoms[b], tol) is not None]
        if not pairs:
            continue
        windows, oo, osp, osh, md, of, st = _plan_native(geoms, pairs, tol)
        ts = rng.integers(-6, 7, (len(pairs), ndim)) * 0.5
        qs = rng.uniform(0.1, 1.0, len(pairs))
        status = np.where(rng.random(len(pairs)) < 0.2, 2, 0).astype(np.int32)
        want = []
        for p, (a, b) in enumerate(pairs):
            shifts.update(t=ts[p], q=float(qs[p]), status=int(status[p]))
            with pytest.warns(UserWarning) if status[p] == 2 else _nullcontext():
                want.append(registration._lean_register_pair(geoms[a], geoms[b], geoms[a], geoms[b], sdims, tol, None, "k", 0))
        o = np.array([g.origin for g in geoms]) - np.array(tol)
        sp = np.array([g.spacing for g in geoms])
        shp = np.array([g.shape for g in geoms]) + np.ceil(2 * np.array(tol) / sp).astype(np.int64)
        tw = np.array([g.t for g in geoms])
        lo_v, hi_v = o + tw, ((shp - 1) * 1.0 * sp + o) + tw
        pr = np.array(pairs)
        lo = np.maximum(lo_v[pr[:, 0]], lo_v[pr[:, 1]])
        up = 1.0 * (np.minimum(hi_v[pr[:, 0]], hi_v[pr[:, 1]]) - lo) + lo
        fixed_aff = np.array([g.affine for g in geoms])[pr[:, 0]]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = registration._pair_results_from_plan(ts, qs, status, oo[:, :ndim], osp[:, :ndim], osh[:, :ndim], fixed_aff, lo, up)
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g["transform"], w["transform"])
            np.testing.assert_array_equal(g["bbox"], w["bbox"])
            assert g["quality"] == w["quality"] or (np.isnan(g["quality"]) and np.isnan(w["quality"]))
            n_checked += 1
    assert n_checked >= 40


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def test_post_processing_raises_like_the_per_pair_path():
    z = np.zeros((2, 2))
    one = np.ones((2, 2))
    aff = np.stack([np.eye(3)] * 2)
    with pytest.raises(RuntimeError, match="no admissible shift candidate"):
        registration._pair_results_from_plan(z, np.zeros(2), np.array([0, 1], dtype=np.int32), z, one, one.astype(np.int64) * 4, aff, z, one)
    with pytest.raises(ValueError, match="All-NaN"):
        registration._pair_results_from_plan(z, np.zeros(2), np.array([3, 1], dtype=np.int32), z, one, one.astype(np.int64) * 4, aff, z, one)
