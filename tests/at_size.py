"""Oracle checks on SAMPLES of full-size runs (BASELINE.json sizes).

The reference's fusion is chunk-local (fusion/_core.py:1513-1733 takes slabs), and its pairwise registration works on
the overlap crops of one pair (registration.py:353-565): after a full-size run on the GPU, a handful of output boxes
and pairs is handed -- with exactly the slabs / crops the run used -- to the CPU oracle and compared at north_star's
bar.  The oracle tasks are farmed over the host cores with joblib (the box has far more cores than samples).
"""
import numpy as np

from oracle import fuse_oracle as fo
from oracle import reg_oracle as ro
from tests.helpers import fused_close_stats, reference_noise_floor


def _geom(sim, key):
    from multiview_stitcher_amd import param_utils
    from multiview_stitcher_amd import spatial_image_utils as si

    sdims = si.get_spatial_dims_from_sim(sim)
    o = si.get_origin_from_sim(sim, asarray=True).astype(np.float64)
    s = si.get_spacing_from_sim(sim, asarray=True).astype(np.float64)
    shape = np.array([int(sim.sizes[d]) for d in sdims])
    p = np.asarray(param_utils.select_time(si.get_affine_from_sim(sim, key), 0), dtype=np.float64)
    return sdims, o, s, shape, p


def slab_window(sim, key, box_origin, box_spacing, box_shape, margin=2):
    """Index window [lo, hi) of ``sim`` holding every tap of an order-1 resample onto the world box, or None."""
    sdims, o, s, shape, p = _geom(sim, key)
    nd = len(o)
    corners = np.array(list(np.ndindex(*([2] * nd))), dtype=np.float64) * (np.asarray(box_shape) - 1) * box_spacing + box_origin
    pinv = np.linalg.inv(p)
    idx = ((corners @ pinv[:nd, :nd].T + pinv[:nd, nd]) - o) / s
    lo = np.maximum(np.floor(idx.min(0)).astype(int) - margin, 0)
    hi = np.minimum(np.ceil(idx.max(0)).astype(int) + margin + 1, shape)
    if np.any(hi <= lo):
        return None
    return lo, hi


def fetch(data, lo, hi):
    """Host copy of the window of a numpy / DeviceArray / Zarr-backed array (spatial axes last)."""
    lead = (0,) * (len(data.shape) - len(lo))
    return np.ascontiguousarray(np.asarray(data[lead + tuple(slice(int(a), int(b)) for a, b in zip(lo, hi))]))


def fuse_box_task(sims, key, out_origin, out_spacing, lo, shape, halo=0, **oracle_kw):
    """Arguments of one oracle fuse of the output box [lo, lo + shape) (mosaic index units) + ``halo`` px."""
    nd = len(lo)
    lo, shape = np.asarray(lo), np.asarray(shape)
    box_o = np.asarray(out_origin, dtype=np.float64) + (lo - halo) * np.asarray(out_spacing, dtype=np.float64)
    box_n = shape + 2 * halo
    views, params, fvbs = [], [], []
    for sim in sims:
        sdims, o, s, n, p = _geom(sim, key)
        win = slab_window(sim, key, box_o, np.asarray(out_spacing, dtype=np.float64), box_n)
        if win is None:
            continue
        wlo, whi = win
        views.append({"data": fetch(sim.data, wlo, whi), "origin": o + wlo * s, "spacing": s})
        params.append(p)
        fvbs.append(fo.bb(o, s, n))
    out_bb = fo.bb(box_o, out_spacing, box_n)
    return dict(views=views, params=params, out_bb=out_bb, fvbs=fvbs, halo=int(halo), kw=oracle_kw, nd=nd)


def run_fuse_task(task):
    """(worker) oracle fuse of one box: the reference's result, its float32 form and its own rounding-noise floor."""
    if not task["views"]:
        return None
    want, want_f, dbg = fo.fuse_np(task["views"], task["params"], task["out_bb"], full_view_bbs=task["fvbs"],
                                   trim_overlap_in_pixels=task["halo"], return_debug=True, **task["kw"])
    floor = reference_noise_floor(dbg, want_f) if task["kw"].get("weights") is None else None
    wsum = None
    if dbg["raw_weights"] is not None:
        t = dbg["trim"]
        wsum = dbg["raw_weights"].sum(0)[tuple(slice(a, -a) if a > 0 else slice(None) for a in t)].astype(np.float32)
    return want, want_f, floor, wsum


def run_pair_task(task):
    """(worker) oracle registration of one pair of crops."""
    res = ro.phase_correlation_registration(task["fixed"], task["moving"])
    return {"affine_matrix": np.asarray(res["affine_matrix"]), "quality": float(res["quality"])}


def farm(func, tasks, n_jobs=None):
    """Run the oracle tasks on the host cores (joblib / loky, one BLAS thread per worker)."""
    import os

    if not tasks:
        return []
    try:
        from joblib import Parallel, delayed
    except ImportError:   # pragma: no cover
        return [func(t) for t in tasks]
    n = min(len(tasks), n_jobs or len(os.sched_getaffinity(0)))
    if n <= 1:
        return [func(t) for t in tasks]
    return Parallel(n_jobs=n, backend="loky")(delayed(func)(t) for t in tasks)


def _marginal(task, index):
    """Is output voxel ``index`` of the task's box numerically ON a border of one of its views?  scipy decides in-bounds by
    ``c < 0 or c > n - 1`` on a double coordinate derived from the slab's origin (transformation.py:72-83,
    ni_interpolation.c); a voxel whose exact coordinate IS 0 or n - 1 (the corner of a view that defines the corner of the
    union stack) lands inside or outside depending on the last bit of that derivation, i.e. on which slab was cut."""
    bb_ = task["out_bb"]
    world = bb_["origin"] + (np.asarray(index) + task["halo"]) * bb_["spacing"]
    nd = len(world)
    for p, fvb in zip(task["params"], task["fvbs"]):
        pinv = np.linalg.inv(p)
        pix = ((pinv[:nd, :nd] @ world + pinv[:nd, nd]) - fvb["origin"]) / fvb["spacing"]
        on = (np.abs(pix) < 1e-6) | (np.abs(pix - (fvb["shape"] - 1)) < 1e-6)
        inside = np.all((pix > -1e-6) & (pix < fvb["shape"] - 1 + 1e-6))
        if inside and on.any():
            return True
    return False


def check_boxes(fused_data, tasks, los, shapes, rtol=1e-4, int_boundary_rtol=1e-4, windows=None):
    """Compare the windows of the fused mosaic with the farmed oracle results; returns the aggregate statistics of
    ``tests.helpers.fused_close_stats`` (how many voxels needed the reference's noise floor, and how large it got) plus
    ``marginal_voxels``: voxels exactly on a view border (see ``_marginal``), taken out of the comparison."""
    results = farm(run_fuse_task, tasks)
    agg = {"voxels": 0, "beyond_plain_bar": 0, "max_floor_used": 0.0, "lsb_flips": 0, "boxes": 0, "marginal_voxels": 0}
    agg_seen = []
    for task, res, lo, shape in zip(tasks, results, los, shapes):
        got = fetch(fused_data, lo, np.asarray(lo) + np.asarray(shape))
        if res is None:
            assert not got.any(), "box without contributing views must be zero"
            agg_seen.append(0)
            continue
        want, want_f, floor, wsum = res
        if windows is not None:
            # the oracle fused a LARGER box (reach of a neighbourhood filter around the compared one): windows[k] = offset of
            # the compared box inside it
            sl = tuple(slice(int(a), int(a + m)) for a, m in zip(windows[len(agg_seen)], shape))
            want, want_f = want[sl], want_f[sl]
            floor = None if floor is None else floor[sl]
            wsum = None if wsum is None else wsum[sl]
        agg_seen.append(1)
        if np.issubdtype(got.dtype, np.integer):
            far = np.argwhere(np.abs(got.astype(np.int64) - want.astype(np.int64)) > 1)
            assert len(far) <= 64, f"{len(far)} voxels differ by more than one count"
            for idx in far:
                # (a) exactly on a view border, or (b) on the knife edge of the reference's weight quantisation: its
                # float32 cos((1 - x) pi) makes every weight a multiple of 2^-25, so where ALL weights of a voxel are a few
                # quanta (the outermost voxels of the mosaic) one rounding decides between "0 / 0 -> 0" and "w v / w = v"
                knife = wsum is not None and float(wsum[tuple(idx)]) <= 8 * 2.0 ** -25
                assert _marginal(task, idx) or knife, (lo, idx, got[tuple(idx)], want[tuple(idx)], None if wsum is None else wsum[tuple(idx)])
                got[tuple(idx)] = want[tuple(idx)]
                agg["marginal_voxels"] += 1
        st = fused_close_stats(got, want, want_f, rtol=rtol, noise_floor=floor, int_boundary_rtol=int_boundary_rtol)
        agg["voxels"] += st["voxels"]
        agg["beyond_plain_bar"] += st["beyond_plain_bar"]
        agg["lsb_flips"] += st["lsb_flips"]
        agg["max_floor_used"] = max(agg["max_floor_used"], st["max_floor_used"])
        agg["boxes"] += 1
    _record(agg)
    return agg


def _record(agg):
    """With MVS_AT_SIZE_STATS=<file> every check appends its statistics (test id + counts) as one JSON line: the numbers behind
    "how many voxels needed the noise floor" end up under profiles/ (tools/profile_round.sh)."""
    import json
    import os

    path = os.environ.get("MVS_AT_SIZE_STATS")
    if not path:
        return
    rec = dict(agg)
    rec["test"] = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    rec["beyond_plain_bar_frac"] = rec["beyond_plain_bar"] / max(rec["voxels"], 1)
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")


class CapturePairs:
    """``pairwise_reg_func`` that runs the device registration and keeps the crops of the pairs it sees, so that the very
    same crops (binned, resampled onto the fixed grid, NaN outside: registration.py:280-350) can go to the oracle."""

    def __init__(self, keep=6):
        import threading

        self.keep, self.records, self._lock = keep, [], threading.Lock()
        self.tag = None        # set by the caller before a call: stored with the record (e.g. the pair's view indices)

    def __call__(self, fixed_data, moving_data, device=0, **kw):
        from multiview_stitcher_amd import registration

        got = registration.phase_correlation_registration(fixed_data, moving_data, device=device, **kw)
        with self._lock:
            shape = tuple(np.asarray(fixed_data.data.shape if hasattr(fixed_data, "data") else fixed_data.shape))
            orient = int(np.argmin(shape))
            if sum(1 for r in self.records if r["orient"] == orient) < max(1, self.keep // 3) and not isinstance(got, list):
                a = np.asarray(fixed_data.data if hasattr(fixed_data, "data") else fixed_data)
                b = np.asarray(moving_data.data if hasattr(moving_data, "data") else moving_data)
                self.records.append({"orient": orient, "fixed": a, "moving": b, "tag": self.tag, "got": {
                    "affine_matrix": np.asarray(got["affine_matrix"]).copy(), "quality": float(got["quality"])}})
        return got

    def check(self, quality_atol=1e-5):
        wants = farm(run_pair_task, [{"fixed": r["fixed"], "moving": r["moving"]} for r in self.records])
        for r, w in zip(self.records, wants):
            assert np.array_equal(r["got"]["affine_matrix"], w["affine_matrix"]), (r["got"], w)      # selected shift: bit-exact
            assert abs(r["got"]["quality"] - w["quality"]) <= quality_atol, (r["got"]["quality"], w["quality"])
        return len(wants)


def oracle_registration_crops(tile1, tile2, origin1, origin2, spacing, affine1=None, affine2=None):
    """The two crops ``phase_correlation_registration`` receives for a pair, made by the ORACLE from the raw tiles:
    registration binning (registration.py:114-191), ``coarsen(binning).mean().astype(dtype)`` (registration.py:1732-1741),
    the overlap boxes of the binned views (registration.py:194-277) and the resample of both onto the fixed view's grid
    over its box (registration.py:280-350).  ``tile*``: host arrays (z, y, x), ``origin*`` / ``spacing``: physical, zyx."""
    nd = tile1.ndim
    spacing = np.asarray(spacing, dtype=np.float64)
    binning = ro.get_optimal_registration_binning(tile1.shape, tile2.shape, spacing, spacing)
    b = np.array([binning[d] for d in ["z", "y", "x"][-nd:]])
    views, stacks = [], []
    for tile, origin in ((tile1, origin1), (tile2, origin2)):
        n = (np.array(tile.shape) // b) * b
        t = tile[tuple(slice(0, int(v)) for v in n)]
        shp = []
        for k in range(nd):
            shp += [int(n[k] // b[k]), int(b[k])]
        binned = t.reshape(shp).mean(axis=tuple(range(1, 2 * nd, 2))).astype(tile.dtype)        # float64 mean, truncating cast
        o = np.asarray(origin, dtype=np.float64) + (b - 1) * spacing / 2
        views.append({"data": binned, "origin": o, "spacing": spacing * b})
        stacks.append({"origin": o, "spacing": spacing * b, "shape": np.array(binned.shape)})
    a1 = np.eye(nd + 1) if affine1 is None else np.asarray(affine1)
    a2 = np.eye(nd + 1) if affine2 is None else np.asarray(affine2)
    lowers, uppers, _ = ro.get_overlap_bboxes(stacks[0], a1, stacks[1], a2)
    fixed, moving, _, _ = ro.sims_to_intrinsic_coord_system(views[0], views[1], a1, a2, lowers, uppers)
    return fixed, moving, binning


def assert_crops_equal(got, want, exact=True):
    """Device crop against the oracle's.  The reference sizes the crop as floor((upper - lower) / spacing + 1) from the vertices
    Qhull returns for the intersection polytope (registration.py:241-277, 319): for views on one pixel grid that quotient is an
    integer up to Qhull's round-off (1e-13), so the reference's own crop is N or N - 1 samples long depending on the sign of that
    round-off.  The product evaluates axis-aligned pairs in closed form (the exact N); a shorter oracle crop is therefore compared
    on the common window (its samples are the same samples: the origin moves by the same 1e-13)."""
    assert got.ndim == want.ndim and all(0 <= g - w <= 1 for g, w in zip(got.shape, want.shape)), (got.shape, want.shape)
    g = got[tuple(slice(0, n) for n in want.shape)]
    assert np.array_equal(np.isnan(g), np.isnan(want))
    if exact:
        np.testing.assert_array_equal(g, want)
    else:
        np.testing.assert_allclose(g, want, rtol=1e-6, atol=0, equal_nan=True)
    return bool(np.array_equal(g, want, equal_nan=True))
