"""Geometry helpers of the fuse/registration planners (host side, tiny data).

Mirror of the parts of the reference's ``mv_graph`` that feed kernel arguments:
``get_chunk_bbs`` (mv_graph.py:934-986), ``get_vertices_from_stack_props``
(mv_graph.py:423-444), ``get_overlap_for_bbs`` (mv_graph.py:989-1117),
``project_bb_along_dim`` (mv_graph.py:1120-1145), plus a closed-form AABB
neighbour graph for axis-aligned tile grids standing in for the Qhull-based
``build_view_adjacency_graph_from_msims`` (mv_graph.py:35-180).

Bounding boxes are the reference's dict-of-dicts:
``{"origin": {dim: float}, "spacing": {dim: float}, "shape": {dim: int}}``.
"""

from __future__ import annotations

from itertools import product

import numpy as np

from . import transformation

SPATIAL_DIMS = ["z", "y", "x"]


def normalize_chunks(chunksizes, shape):
    """dask.array.core.normalize_chunks for int-or-sequence chunk specs."""
    out = []
    for cs, n in zip(chunksizes, shape):
        n = int(n)
        if isinstance(cs, (list, tuple, np.ndarray)):
            out.append(tuple(int(c) for c in cs))
            continue
        cs = int(cs)
        if n == 0:
            out.append((0,))
            continue
        nfull, rem = divmod(n, cs)
        out.append((cs,) * nfull + ((rem,) if rem else ()))
    return tuple(out)


def get_chunk_bbs(array_bb, chunksizes):
    """mv_graph.get_chunk_bbs (mv_graph.py:934-986): (chunk_bbs, block_indices)."""
    spatial_dims = sorted(array_bb["origin"].keys())[::-1]
    chunksizes = [chunksizes[dim] for dim in spatial_dims]
    array_shape = [array_bb["shape"][dim] for dim in spatial_dims]
    array_origin = [array_bb["origin"][dim] for dim in spatial_dims]
    normalized_chunks = normalize_chunks(chunksizes, array_shape)
    block_indices = list(product(*(range(len(bds)) for bds in normalized_chunks)))
    block_offsets = [np.cumsum((0,) + bds[:-1]) for bds in normalized_chunks]
    chunk_bbs = [
        {
            "origin": {
                dim: array_origin[idim] + array_bb["spacing"][dim] * block_offsets[idim][block_ind[idim]]
                for idim, dim in enumerate(spatial_dims)
            },
            "shape": {dim: normalized_chunks[idim][block_ind[idim]] for idim, dim in enumerate(spatial_dims)},
            "spacing": array_bb["spacing"],
        }
        for block_ind in block_indices
    ]
    return chunk_bbs, block_indices


_UNIT_CORNERS = {n: np.array(list(np.ndindex(tuple([2] * n)))) for n in (2, 3)}   # corners of the unit cube, last axis fastest


def get_vertices_from_stack_props(stack_props):
    """mv_graph.get_vertices_from_stack_props (mv_graph.py:423-444)."""
    ndim = len(stack_props["origin"])
    sdims = SPATIAL_DIMS[-ndim:]
    gv = _UNIT_CORNERS[ndim]
    vertices = gv * (np.array([stack_props["shape"][d] for d in sdims]) - 1) * np.array(
        [stack_props["spacing"][d] for d in sdims]
    ) + np.array([stack_props["origin"][d] for d in sdims])
    if "transform" in stack_props:
        affine = np.asarray(stack_props["transform"])
        if affine.ndim == 3:
            affine = affine[0]
        vertices = transformation.transform_pts(vertices, affine)
    return vertices


def project_bb_along_dim(bb, dim):
    """mv_graph.project_bb_along_dim (mv_graph.py:1120-1145)."""
    return {key: {d: bb[key][d] for d in bb[key] if d != dim} for key in bb}


def get_overlap_for_bbs(target_bb, query_bbs, param, additional_extent_in_pixels=None, tol=1e-6, param_is_inverse=False):
    """mv_graph.get_overlap_for_bbs (mv_graph.py:989-1117): slab of each query bb that the
    back-projected target bb touches (generic affine case)."""
    if additional_extent_in_pixels is None:
        additional_extent_in_pixels = {"z": 0, "y": 0, "x": 0}
    ndim = len(target_bb["origin"])
    sdims = SPATIAL_DIMS[-ndim:]
    corners_target = get_vertices_from_stack_props(target_bb)
    inv_param = np.asarray(param) if param_is_inverse else np.linalg.inv(np.asarray(param))
    corners_query = transformation.transform_pts(corners_target, inv_param)
    cmin = np.min(corners_query, axis=0)
    cmax = np.max(corners_query, axis=0)
    overlap_bbs = []
    for qbb in query_bbs:
        bo = {d: cmin[i] - additional_extent_in_pixels[d] * qbb["spacing"][d] for i, d in enumerate(sdims)}
        bs = {
            d: int(np.ceil((cmax[i] - cmin[i]) / qbb["spacing"][d])) + 1 + 2 * additional_extent_in_pixels[d]
            for i, d in enumerate(sdims)
        }
        q_last = {d: qbb["origin"][d] + (qbb["shape"][d] - 1) * qbb["spacing"][d] for d in sdims}
        if any(bo[d] - tol > q_last[d] for d in sdims):
            overlap_bbs.append(None)
            continue
        if any(bo[d] + (bs[d] - 1) * qbb["spacing"][d] < qbb["origin"][d] - tol for d in sdims):
            overlap_bbs.append(None)
            continue
        oo = {d: max(bo[d], qbb["origin"][d]) for d in sdims}
        osz = {
            d: int(np.ceil((min(bo[d] + (bs[d] - 1) * qbb["spacing"][d], q_last[d]) - oo[d]) / qbb["spacing"][d])) + 1
            for d in sdims
        }
        if any(osz[d] < 1 for d in sdims):
            overlap_bbs.append(None)
            continue
        overlap_bbs.append({"origin": oo, "shape": osz, "spacing": qbb["spacing"]})
    return overlap_bbs


def world_aabb(stack_props, affine):
    """Axis-aligned bounding box (min, max) of a view in world coordinates."""
    v = get_vertices_from_stack_props(stack_props)
    v = transformation.transform_pts(v, affine)
    return v.min(axis=0), v.max(axis=0)


def get_overlap_aabb(sp1, affine1, sp2, affine2):
    """Intersection (lower, upper) of two views' world AABBs, or None.

    For axis-aligned views this equals the halfspace-intersection polytope the
    reference computes with Qhull (mv_graph.py:301-338, registration.py:194-277)."""
    lo1, hi1 = world_aabb(sp1, affine1)
    lo2, hi2 = world_aabb(sp2, affine2)
    lo, hi = np.maximum(lo1, lo2), np.minimum(hi1, hi2)
    if np.any(hi < lo):
        return None
    return lo, hi


def build_view_adjacency_pairs(stack_props_list, affines, min_overlap_voxels=1.0):
    """Pairs (i, j, overlap_volume) of views whose world AABBs overlap.

    Stand-in for build_view_adjacency_graph_from_msims (mv_graph.py:35-180) on
    axis-aligned grids: O(N^2) AABB tests, overlap 'area' = AABB volume."""
    n = len(stack_props_list)
    boxes = [world_aabb(sp, a) for sp, a in zip(stack_props_list, affines)]
    pairs = []
    for i in range(n):
        for j in range(i + 1, n):
            lo = np.maximum(boxes[i][0], boxes[j][0])
            hi = np.minimum(boxes[i][1], boxes[j][1])
            if np.any(hi <= lo):
                continue
            pairs.append((i, j, float(np.prod(hi - lo))))
    return pairs


def prune_to_axis_aligned(pairs, stack_props_list, affines):
    """Keep face-sharing neighbours only (the reference's 'keep_axis_aligned' pruning,
    mv_graph.py:1148-1196): the two view centres differ along exactly one axis by more
    than half a tile extent."""
    kept = []
    for i, j, vol in pairs:
        ci = np.mean(world_aabb(stack_props_list[i], affines[i]), axis=0)
        cj = np.mean(world_aabb(stack_props_list[j], affines[j]), axis=0)
        ext = np.array(world_aabb(stack_props_list[i], affines[i])[1]) - np.array(world_aabb(stack_props_list[i], affines[i])[0])
        far = np.abs(ci - cj) > 0.5 * np.maximum(ext, 1e-12)
        if np.sum(far) == 1:
            kept.append((i, j, vol))
    return kept
