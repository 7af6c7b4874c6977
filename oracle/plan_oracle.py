"""CPU oracle: the chunk -> view-slab planner of fusion.fuse.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Literal restatement, dict by dict, of the reference's planner

  fusion._get_axis_aligned_translation_dims      src/multiview_stitcher/fusion/_core.py:354-400
  fusion._get_grid_aligned_translation_dims      src/multiview_stitcher/fusion/_core.py:403-459
  fusion._get_axis_aligned_translation_overlap   src/multiview_stitcher/fusion/_core.py:462-533
  fusion._build_spatial_fusion_plan              src/multiview_stitcher/fusion/_core.py:536-722
  the slab selection sims[iview].sel(...)        src/multiview_stitcher/fusion/_core.py:1371-1386

The product plans with the library (mvs_fuse_plan, csrc/mvs_plan.hip); tests compare its index windows with the windows
this restatement selects (tests/test_plan_oracle.py)."""

from itertools import product

import numpy as np

from multiview_stitcher_amd import mv_graph

def _isclose(a, b, atol):
    """np.isclose(a, b, atol=atol) (rtol 1e-5) for Python scalars -- the planner calls it thousands of times per plan
    and numpy's array machinery costs ~5 us per call."""
    a, b = float(a), float(b)
    if a == b:
        return True
    return abs(a - b) <= atol + 1e-5 * abs(b)       # False for NaN, like numpy


def _is_grid_aligned(offset, spacing, tol=1e-6):
    if spacing == 0:
        return False
    po = float(offset) / float(spacing)
    return _isclose(po, round(po), tol) if np.isfinite(po) else False


def _param_entry(param, sdims, din, dout):
    names = list(sdims) + ["1"]
    return float(param[names.index(din), names.index(dout)])


def _get_axis_aligned_translation_dims(sparams, sdims, tol=1e-6):
    """_core.py:354-400."""
    res = []
    for dim in sdims:
        others = [d for d in sdims if d != dim]
        ok = True
        for p in sparams:
            if not _isclose(_param_entry(p, sdims, dim, dim), 1, tol):
                ok = False
                break
            if any(not _isclose(_param_entry(p, sdims, dim, o), 0, tol) for o in others):
                ok = False
                break
            if any(not _isclose(_param_entry(p, sdims, o, dim), 0, tol) for o in others):
                ok = False
                break
        if ok:
            res.append(dim)
    return res


def _get_grid_aligned_translation_dims(sparams, views_bb, output_stack_properties, sdims, tol=1e-6):
    """_core.py:403-459."""
    axis_aligned = set(_get_axis_aligned_translation_dims(sparams, sdims, tol))
    res = []
    for dim in sdims:
        if dim not in axis_aligned:
            continue
        if any(not _isclose(output_stack_properties["spacing"][dim], vbb["spacing"][dim], tol) for vbb in views_bb):
            continue
        ok = True
        for iview, p in enumerate(sparams):
            translation = _param_entry(p, sdims, dim, "1")
            if not _is_grid_aligned(
                output_stack_properties["origin"][dim] - translation - views_bb[iview]["origin"][dim],
                views_bb[iview]["spacing"][dim], tol,
            ):
                ok = False
                break
        if ok:
            res.append(dim)
    return res


def _get_axis_aligned_translation_overlap(target_bb, query_bb, param, sdims, additional_extent_in_pixels=None, tol=1e-6):
    """_core.py:462-533: integer source-pixel window covering the back-projected chunk."""
    if additional_extent_in_pixels is None:
        additional_extent_in_pixels = {d: 0 for d in sdims}
    oo, osz = {}, {}
    for dim in sdims:
        qs = query_bb["spacing"][dim]
        ts = target_bb["spacing"][dim]
        translation = _param_entry(param, sdims, dim, "1")
        qmin = target_bb["origin"][dim] - translation
        qmax = target_bb["origin"][dim] + (int(target_bb["shape"][dim]) - 1) * ts - translation
        qmin, qmax = sorted((qmin, qmax))
        extra = additional_extent_in_pixels[dim] * qs
        start_f = (qmin - extra - query_bb["origin"][dim]) / qs
        stop_f = (qmax + extra - query_bb["origin"][dim]) / qs
        start = int(np.floor(start_f + tol))
        stop = int(np.ceil(stop_f - tol)) + 1
        lo = max(start, 0)
        hi = min(stop, int(query_bb["shape"][dim]))
        if lo >= hi:
            return None
        oo[dim] = query_bb["origin"][dim] + lo * qs
        osz[dim] = hi - lo
    return {"origin": oo, "shape": osz, "spacing": query_bb["spacing"]}


def _build_spatial_fusion_plan(
    *, sparams, views_bb, output_stack_properties, output_chunksize, output_chunk_bbs,
    output_chunk_bbs_with_overlap, output_chunk_bbs_for_result, block_indices, overlap_in_pixels,
    trim_overlap, interpolation_order, sdims,
):
    """_core.py:536-722: which views / which slab of each view feed which output chunk."""
    axis_aligned = _get_axis_aligned_translation_dims(sparams, sdims)
    grid_aligned = _get_grid_aligned_translation_dims(sparams, views_bb, output_stack_properties, sdims)
    use_axis_aligned = set(axis_aligned) == set(sdims)
    inv_sparams = None if use_axis_aligned else [np.linalg.inv(sp) for sp in sparams]

    norm_chunks = mv_graph.normalize_chunks(
        [output_chunksize[d] for d in sdims], [output_stack_properties["shape"][d] for d in sdims]
    )
    n_blocks = [len(c) for c in norm_chunks]
    uniform_cs = [c[0] for c in norm_chunks]
    osp_origin = np.array([output_stack_properties["origin"][d] for d in sdims])
    osp_spacing = np.array([output_stack_properties["spacing"][d] for d in sdims])
    overlap_pad = np.array([overlap_in_pixels[d] for d in sdims]) * osp_spacing

    from multiview_stitcher_amd.transformation import transform_pts

    chunk_to_tiles = {}
    for iview in range(len(sparams)):
        interp_pad = np.array(
            [0.0 if d in grid_aligned else float(interpolation_order) * views_bb[iview]["spacing"][d] for d in sdims]
        )
        pad = interp_pad + overlap_pad
        corners = transform_pts(mv_graph.get_vertices_from_stack_props(views_bb[iview]), sparams[iview])
        aabb_min = np.min(corners, axis=0) - pad
        aabb_max = np.max(corners, axis=0) + pad
        ranges = []
        skip = False
        for idim in range(len(sdims)):
            cs_phys = uniform_cs[idim] * osp_spacing[idim]
            i_first = max(0, int(np.floor((aabb_min[idim] - osp_origin[idim]) / cs_phys)))
            i_last = min(n_blocks[idim] - 1, int(np.floor((aabb_max[idim] - osp_origin[idim]) / cs_phys)))
            if i_first > i_last:
                skip = True
                break
            ranges.append(range(i_first, i_last + 1))
        if skip:
            continue
        for chunk_idx in product(*ranges):
            chunk_to_tiles.setdefault(chunk_idx, []).append(iview)

    additional_extent = {d: (0 if d in grid_aligned else int(interpolation_order)) for d in sdims}
    entries = []
    for cbb, cbb_ov, cbb_res, block_index in zip(
        output_chunk_bbs, output_chunk_bbs_with_overlap, output_chunk_bbs_for_result, block_indices
    ):
        chunk_views = []
        for iview in chunk_to_tiles.get(tuple(block_index), []):
            if use_axis_aligned:
                overlap = _get_axis_aligned_translation_overlap(cbb_ov, views_bb[iview], sparams[iview], sdims, additional_extent)
            else:
                overlap = mv_graph.get_overlap_for_bbs(
                    cbb_ov, [views_bb[iview]], inv_sparams[iview], additional_extent, param_is_inverse=True
                )[0]
            if overlap is not None:
                chunk_views.append((iview, overlap))
        fuse_planewise = "z" in grid_aligned and cbb_ov["shape"].get("z", 2) == 1
        entries.append(
            {"views": chunk_views, "output_bb": cbb, "output_bb_overlap": cbb_ov, "output_bb_result": cbb_res,
             "fuse_planewise": fuse_planewise, "block_index": tuple(block_index)}
        )
    return {
        "sparams": sparams, "fix_dims": grid_aligned, "axis_aligned_translation_dims": axis_aligned,
        "grid_aligned_translation_dims": grid_aligned, "per_chunk_entries": entries,
        "uses_axis_aligned_translation": use_axis_aligned,
    }


def _select_slab(sim, tile_overlap_bb, sdims, tol=1e-6):
    """``sims[iview].sel({dim: slice(origin - tol, last + tol)})`` of _core.py:1371-1386."""
    return sim.sel(
        {
            d: slice(
                tile_overlap_bb["origin"][d] - tol,
                tile_overlap_bb["origin"][d] + (tile_overlap_bb["shape"][d] - 1) * tile_overlap_bb["spacing"][d] + tol,
            )
            for d in sdims
        }
    )




def slab_windows(sim_coords, tile_overlap_bb, sdims, tol=1e-6):
    """Index window (lo, n) per axis that ``_select_slab`` picks out of a view with coordinate arrays ``sim_coords``."""
    lo, n = [], []
    for d in sdims:
        c = np.asarray(sim_coords[d])
        a = tile_overlap_bb["origin"][d] - tol
        b = tile_overlap_bb["origin"][d] + (tile_overlap_bb["shape"][d] - 1) * tile_overlap_bb["spacing"][d] + tol
        i0 = int(np.searchsorted(c, a, side="left"))
        i1 = int(np.searchsorted(c, b, side="right"))
        lo.append(i0)
        n.append(i1 - i0)
    return tuple(lo), tuple(n)
