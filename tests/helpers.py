"""Shared helpers of the parity tests: build the same chunk inputs for the oracle and the HIP path."""
import numpy as np

from oracle import fuse_oracle as fo


class SignedView:
    """torch's __cuda_array_interface__ import has no uint16: hand a DeviceArray's bytes over as int16 (the test mosaics stay below 4096)."""

    def __init__(self, arr):
        self.__cuda_array_interface__ = dict(arr.__cuda_array_interface__, typestr="<i2")
        self.owner = arr


def sim_to_view(sim):
    from multiview_stitcher_amd import spatial_image_utils as si

    sdims = si.get_spatial_dims_from_sim(sim)
    o, s = si.get_origin_from_sim(sim, asarray=True), si.get_spacing_from_sim(sim, asarray=True)
    return {"data": np.asarray(sim.data), "origin": o, "spacing": s}, fo.bb(o, s, [sim.sizes[d] for d in sdims])


def squeeze_field(sim):
    """Drop singleton c/t axes of a sim built by get_sim_from_array."""
    from multiview_stitcher_amd import spatial_image_utils as si

    return sim.isel({d: 0 for d in si.get_nonspatial_dims_from_sim(sim)})


def bb_to_dicts(bb, sdims):
    return {k: dict(zip(sdims, np.asarray(v).tolist())) for k, v in bb.items()}


def union_bb(views_bbs, params, spacing):
    """Union output stack of the views (oracle data model), like calc_fusion_stack_properties."""
    ndim = len(spacing)
    lo, hi = [], []
    for vbb, p in zip(views_bbs, params):
        corners = np.array(list(np.ndindex(*([2] * ndim)))) * (vbb["shape"] - 1) * vbb["spacing"] + vbb["origin"]
        w = corners @ p[:ndim, :ndim].T + p[:ndim, ndim]
        lo.append(w.min(0))
        hi.append(w.max(0))
    lo, hi = np.min(lo, 0), np.max(hi, 0)
    shape = np.floor((hi - lo) / spacing + 1e-9).astype(int) + 1
    return fo.bb(lo, spacing, shape)


def reference_noise_floor(debug, fused_f, eps_w=1.2e-7):
    """Bound on the reference's OWN float32 rounding noise per voxel.

    The reference evaluates the cosine ramp as (cos((1-x)pi)+1)/2 in float32
    (weights.py:502-507): cos() near -1 is rounded to 2^-24, so every ramp weight
    carries an absolute error of ~6e-8 regardless of its size.  Where all
    contributing weights are tiny (tile corners facing the mosaic border) that
    noise is amplified by 1/sum(w):  |d out| <= sum_v |I_v - out| * eps_w / sum_v w_v.
    The HIP kernel evaluates the same ramp as sin^2(pi x/2) (no cancellation), so
    it can differ from the reference by up to this floor; the parity tolerance is
    rtol*|want| + this floor."""
    w, views, trim = debug["raw_weights"], debug["views"], debug["trim"]
    if w is None:
        return np.zeros_like(fused_f, dtype=np.float64)
    sl = (slice(None),) + tuple(slice(t, -t) if t > 0 else slice(None) for t in trim)
    w = w[sl].astype(np.float64)
    v = np.nan_to_num(views[sl].astype(np.float64))
    wsum = w.sum(0)
    spread = (np.abs(v - fused_f.astype(np.float64)[None]) * (w > 0)).sum(0)
    return np.where(wsum > 0, spread * eps_w / np.maximum(wsum, 1e-30), 0.0)


def fused_close_stats(got, want, want_float=None, rtol=1e-4, data_range=None, max_bad_frac=0.0, noise_floor=None,
                      int_boundary_rtol=1e-4):
    """Parity bar of north_star: float32 fused voxels within 1e-4 relative; integer outputs within
    +-1 LSB (truncating cast after float accumulate) and exact where the float value is not within
    1e-4*|value| of an integer boundary.  ``noise_floor`` (reference_noise_floor) widens the bar by the reference's
    own float32 rounding noise; the returned statistics say how often that was needed and how large it got:
    ``voxels``, ``beyond_plain_bar`` (voxels that pass only thanks to the floor), ``max_floor_used`` (the largest
    floor value among them, in output units), ``lsb_flips`` (integer outputs that differ by one count)."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype
    stats = {"voxels": int(got.size), "beyond_plain_bar": 0, "max_floor_used": 0.0, "lsb_flips": 0}
    if np.issubdtype(got.dtype, np.integer):
        diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert diff.max() <= 1, f"integer output differs by {diff.max()} LSB"
        flips = diff == 1
        stats["lsb_flips"] = int(flips.sum())
        if want_float is not None and flips.any():
            frac = want_float - np.floor(want_float)
            dist = np.minimum(frac, 1 - frac)
            plain = dist <= int_boundary_rtol * np.maximum(np.abs(want_float), 1.0)
            floor = noise_floor if noise_floor is not None else np.zeros_like(dist)
            near = dist <= int_boundary_rtol * np.maximum(np.abs(want_float), 1.0) + floor
            assert np.all(near[flips]), "1-LSB flips away from an integer boundary"
            needed = flips & ~plain
            stats["beyond_plain_bar"] = int(needed.sum())
            if needed.any():
                stats["max_floor_used"] = float(np.max(np.broadcast_to(floor, dist.shape)[needed]))
        return stats
    rng = float(data_range) if data_range is not None else float(np.nanmax(np.abs(want)) or 1.0)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    plain_tol = rtol * np.maximum(np.abs(want), 1e-3 * rng)
    tol = plain_tol + noise_floor if noise_floor is not None else plain_tol
    bad = err > tol
    assert bad.mean() <= max_bad_frac, (
        f"{bad.sum()} / {bad.size} voxels beyond rtol={rtol}; worst rel err "
        f"{(err / np.maximum(np.abs(want), 1e-30)).max():.3e}, worst abs {err.max():.3e}"
    )
    needed = (err > plain_tol) & ~bad
    stats["beyond_plain_bar"] = int(needed.sum())
    if needed.any() and noise_floor is not None:
        stats["max_floor_used"] = float(np.max(np.broadcast_to(noise_floor, err.shape)[needed]))
    return stats


def assert_fused_close(got, want, want_float=None, rtol=1e-4, data_range=None, max_bad_frac=0.0, noise_floor=None,
                       max_floor_frac=0.01):
    """fused_close_stats + the bound on how often the reference's noise floor may be needed (<= 1 % of the voxels)."""
    stats = fused_close_stats(got, want, want_float, rtol, data_range, max_bad_frac, noise_floor)
    assert stats["beyond_plain_bar"] <= max_floor_frac * stats["voxels"], stats
    return stats
