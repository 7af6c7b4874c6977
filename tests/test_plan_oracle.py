"""The library's chunk -> view-slab plan (mvs_fuse_plan, host code) against the literal restatement of the reference's
planner (oracle/plan_oracle.py: fusion/_core.py:354-722 + the label selection of _core.py:1371-1386): same axis
classification, same chunks, same views per chunk, same index windows.  No GPU needed."""
import numpy as np
import pytest

from multiview_stitcher_amd import fusion, mv_graph, param_utils
from multiview_stitcher_amd import spatial_image_utils as si
from oracle import plan_oracle as po


def _views(ndim, grid, tile, step, spacing, rng, jitter=0.0, frac=False):
    sdims = ["z", "y", "x"][-ndim:]
    bbs, params = [], []
    for idx in np.ndindex(*grid):
        origin = {d: float(i * st * sp) for d, i, st, sp in zip(sdims, idx, step, spacing)}
        bbs.append({"origin": origin, "spacing": dict(zip(sdims, spacing)), "shape": dict(zip(sdims, tile))})
        p = np.eye(ndim + 1)
        if jitter:
            t = rng.uniform(-jitter, jitter, ndim)
            p[:ndim, ndim] = t if frac else np.round(t) * np.asarray(spacing)
        params.append(p)
    return sdims, bbs, params


def _rot(ndim, deg, rng):
    a = np.deg2rad(deg)
    p = np.eye(ndim + 1)
    c, s = np.cos(a), np.sin(a)
    p[ndim - 2:ndim, ndim - 2:ndim] = [[c, -s], [s, c]]
    p[:ndim, ndim] = rng.uniform(-3, 3, ndim)
    return p


def _compare(sdims, bbs, params, osp, cs, halo, order):
    ndim = len(sdims)
    overlap = {d: h for d, h in zip(sdims, halo)}
    chunks = {d: c for d, c in zip(sdims, cs)}
    cbb, bidx = mv_graph.get_chunk_bbs(osp, chunks)
    cbb_ov = [cb | {"origin": {d: cb["origin"][d] - overlap[d] * osp["spacing"][d] for d in sdims}}
              | {"shape": {d: cb["shape"][d] + 2 * overlap[d] for d in sdims}} for cb in cbb]
    want = po._build_spatial_fusion_plan(
        sparams=params, views_bb=bbs, output_stack_properties=osp, output_chunksize=chunks, output_chunk_bbs=cbb,
        output_chunk_bbs_with_overlap=cbb_ov, output_chunk_bbs_for_result=cbb, block_indices=bidx, overlap_in_pixels=overlap,
        trim_overlap=True, interpolation_order=order, sdims=sdims)
    by_block, info = fusion._plan_chunks(params, bbs, osp, chunks, overlap, order, sdims)
    assert info["axis_aligned_translation_dims"] == want["axis_aligned_translation_dims"]
    assert info["grid_aligned_translation_dims"] == want["grid_aligned_translation_dims"]
    coords = [{d: bb["origin"][d] + bb["spacing"][d] * np.arange(bb["shape"][d], dtype=float) for d in sdims} for bb in bbs]
    n_pairs = 0
    for entry in want["per_chunk_entries"]:
        # windows the reference's label selection picks (an empty selection drops nothing here: the planner only lists hits)
        exp = []
        for iv, obb in entry["views"]:
            lo, n = po.slab_windows(coords[iv], obb, sdims)
            exp.append((iv, lo, n))
        planewise, got = by_block.get(tuple(entry["block_index"]), (entry["fuse_planewise"], []))
        assert got == exp, (entry["block_index"], got, exp)
        if exp:
            assert planewise == entry["fuse_planewise"]
        n_pairs += len(exp)
    assert sum(len(v[1]) for v in by_block.values()) == n_pairs
    return n_pairs


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("order", [0, 1, 3])
def test_translation_grids(ndim, order):
    rng = np.random.default_rng(ndim * 10 + order)
    for case in range(6):
        spacing = [1.0] * ndim if case % 2 == 0 else list(rng.choice([0.3, 0.5, 1.0, 2.0], ndim))
        tile = [int(v) for v in rng.integers(9, 40, ndim)]
        step = [max(int(t * 0.8), 1) for t in tile]
        grid = [1] * (ndim - 2) + [2, 3] if ndim == 3 and case < 2 else [2] * ndim
        sdims, bbs, params = _views(ndim, grid, tile, step, spacing, rng, jitter=(0 if case == 0 else 3.0), frac=(case >= 3))
        osp = _union(bbs, params, sdims, spacing)
        cs = [int(v) for v in rng.integers(5, 30, ndim)]
        halo = [0] * ndim if case % 3 else [int(v) for v in rng.integers(0, 4, ndim)]
        assert _compare(sdims, bbs, params, osp, cs, halo, order) > 0


def _union(bbs, params, sdims, spacing):
    lo = np.min([[bb["origin"][d] + p[i, -1] for i, d in enumerate(sdims)] for bb, p in zip(bbs, params)], axis=0)
    hi = np.max([[bb["origin"][d] + (bb["shape"][d] - 1) * bb["spacing"][d] + p[i, -1] for i, d in enumerate(sdims)]
                 for bb, p in zip(bbs, params)], axis=0)
    shape = [int(np.floor((h - l) / s)) + 1 for l, h, s in zip(lo, hi, spacing)]
    return {"origin": dict(zip(sdims, lo)), "spacing": dict(zip(sdims, spacing)), "shape": dict(zip(sdims, shape))}


@pytest.mark.parametrize("ndim", [2, 3])
def test_affine_views(ndim):
    """Rotated / scaled views (C4-like): the windows come from the back-projected chunk corners."""
    rng = np.random.default_rng(5 + ndim)
    sdims = ["z", "y", "x"][-ndim:]
    bbs, params = [], []
    for k in range(4):
        spacing = list(rng.choice([0.5, 1.0, 1.5], ndim))
        tile = [int(v) for v in rng.integers(12, 30, ndim)]
        bbs.append({"origin": dict(zip(sdims, rng.uniform(-5, 5, ndim))), "spacing": dict(zip(sdims, spacing)), "shape": dict(zip(sdims, tile))})
        p = _rot(ndim, float(rng.uniform(-40, 40)), rng)
        p[:ndim, :ndim] *= rng.uniform(0.9, 1.1)
        params.append(p)
    params[0] = np.eye(ndim + 1)      # one untransformed view among rotated ones: still the general path for all
    osp = {"origin": dict(zip(sdims, [-20.0] * ndim)), "spacing": dict(zip(sdims, [0.8] * ndim)), "shape": dict(zip(sdims, [70] * ndim))}
    for order, halo, cs in [(1, 0, 16), (0, 2, 25), (3, 1, 70), (1, 0, 200)]:
        assert _compare(sdims, bbs, params, osp, [cs] * ndim, [halo] * ndim, order) > 0


def test_known_answer_geometries():
    """The chunk-edge cases of the reference's own tests (T/test_fusion.py:480-573): a singleton slab and a large origin
    whose pixel offset only rounds to an integer within the tolerance; and a single-plane chunk on the views' z grid."""
    sdims = ["y", "x"]
    bb = {"origin": {"y": 0.0, "x": 0.0}, "spacing": {"y": 0.3, "x": 0.3}, "shape": {"y": 2, "x": 20}}
    osp = {"origin": {"y": 0.0, "x": -2.7}, "spacing": {"y": 0.3, "x": 0.3}, "shape": {"y": 2, "x": 29}}
    _compare(sdims, [bb], [np.eye(3)], osp, [2, 10], [0, 0], 0)
    origin, scale = 861.5120670572916, 0.13810709635416665
    s = (origin + scale * 1.0) - (origin + scale * 0.0)      # spacing as read back from the coordinate array
    bb = {"origin": {"y": 0.0, "x": origin}, "spacing": {"y": s, "x": s}, "shape": {"y": 2, "x": 4084}}
    osp = {"origin": {"y": 0.0, "x": origin - 9 * s}, "spacing": {"y": s, "x": s}, "shape": {"y": 2, "x": 4093}}
    _compare(sdims, [bb], [np.eye(3)], osp, [2, 4084], [0, 0], 0)
    sd3 = ["z", "y", "x"]
    bbs = [{"origin": dict(zip(sd3, [0.0, 0.0, 8.5 * k])), "spacing": dict(zip(sd3, [2.0, 1.0, 1.0])), "shape": dict(zip(sd3, [4, 10, 10]))} for k in range(2)]
    osp = {"origin": dict(zip(sd3, [0.0, 0.0, 0.0])), "spacing": dict(zip(sd3, [2.0, 1.0, 1.0])), "shape": dict(zip(sd3, [4, 10, 18]))}
    by_block, info = fusion._plan_chunks([np.eye(4)] * 2, bbs, osp, dict(zip(sd3, [1, 10, 18])), dict(zip(sd3, [0, 0, 0])), 1, sd3)
    assert info["grid_aligned_translation_dims"] == ["z", "y"] and all(pw for pw, _ in by_block.values())
    _compare(sd3, bbs, [np.eye(4)] * 2, osp, [1, 10, 18], [0, 0, 0], 1)
