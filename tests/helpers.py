"""Shared helpers of the parity tests: build the same chunk inputs for the oracle and the HIP path."""
import numpy as np

from oracle import fuse_oracle as fo


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


def assert_fused_close(got, want, want_float=None, rtol=1e-4, data_range=None, max_bad_frac=0.0):
    """Parity bar of north_star: float32 fused voxels within 1e-4 relative; integer outputs within
    +-1 LSB (truncating cast after float accumulate) and exact where the float value is not within
    1e-4*range of an integer boundary."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype
    if np.issubdtype(got.dtype, np.integer):
        diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert diff.max() <= 1, f"integer output differs by {diff.max()} LSB"
        if want_float is not None and diff.max() == 1:
            frac = want_float - np.floor(want_float)
            near = np.minimum(frac, 1 - frac) <= 1e-4 * np.maximum(np.abs(want_float), 1.0)
            assert np.all(near[diff == 1]), "1-LSB flips away from an integer boundary"
        return
    rng = float(data_range) if data_range is not None else float(np.nanmax(np.abs(want)) or 1.0)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = rtol * np.maximum(np.abs(want), 1e-3 * rng)
    bad = err > tol
    assert bad.mean() <= max_bad_frac, (
        f"{bad.sum()} / {bad.size} voxels beyond rtol={rtol}; worst rel err "
        f"{(err / np.maximum(np.abs(want), 1e-30)).max():.3e}, worst abs {err.max():.3e}"
    )
