"""Geometry helpers of the fuse/registration planners (host side, tiny data).

Mirror of the parts of the reference's ``mv_graph`` that feed kernel arguments:
``get_chunk_bbs`` (mv_graph.py:934-986), ``get_vertices_from_stack_props``
(mv_graph.py:423-444), ``get_overlap_for_bbs`` (mv_graph.py:989-1117),
``project_bb_along_dim`` (mv_graph.py:1120-1145), and the view adjacency graph
with its pruning methods (mv_graph.py:35-338, 636-881, 1148-1196; second half
of this file).

Bounding boxes are the reference's dict-of-dicts:
``{"origin": {dim: float}, "spacing": {dim: float}, "shape": {dim: int}}``.
"""

from __future__ import annotations

from itertools import product

import numpy as np

from . import param_utils, transformation

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


# ---- view adjacency graph and its pruning ---------------------------------------------------------------
# Restatement of mv_graph.py:35-180 (graph construction), :183-338 (overlap of two views as the volume of the
# intersection of their halfspaces), :636-881 and :1148-1196 (pruning methods) without networkx / dask / skimage.
# The geometric kernels are the reference's own scipy call sites (cKDTree, linprog, HalfspaceIntersection, ConvexHull:
# scipy is present here and on the GPU box); axis-aligned pairs -- the regular-grid case -- take a closed form instead
# of the three Qhull / LP calls per pair.
class NotEnoughOverlapError(Exception):
    pass


class Graph:
    """Undirected graph with networkx's iteration orders (nodes in insertion order, a node's neighbours in the order
    its edges were added, ``edges()`` node by node with every edge once): the reference's tie breaks depend on them."""

    def __init__(self, nodes=()):
        self.adj = {}
        self.node_attrs = {}
        for n in nodes:
            self.add_node(n)

    nodes = property(lambda self: list(self.adj))

    def add_node(self, n, **attrs):
        if n not in self.adj:
            self.adj[n] = {}
            self.node_attrs[n] = {}
        if attrs:
            self.node_attrs[n].update(attrs)

    def add_edge(self, u, v, **attrs):
        adj = self.adj
        if u not in adj:
            adj[u] = {}
            self.node_attrs[u] = {}
        if v not in adj:
            adj[v] = {}
            self.node_attrs[v] = {}
        d = adj[u].get(v)
        if d is None:
            d = {}
        d.update(attrs)
        adj[u][v] = d
        adj[v][u] = d

    def remove_edge(self, u, v):
        del self.adj[u][v]
        if u != v:
            del self.adj[v][u]

    def remove_node(self, n):
        for m in list(self.adj[n]):
            if m != n:
                del self.adj[m][n]
        del self.adj[n]
        self.node_attrs.pop(n, None)

    def has_edge(self, u, v):
        return v in self.adj.get(u, {})

    def edges(self, data=False):
        # every edge once, reported from its endpoint that comes first in node order, neighbours in adjacency order (networkx's
        # order): a neighbour counts as "seen" when its node comes strictly earlier
        pos = {n: i for i, n in enumerate(self.adj)}
        if data:
            return [(n, m, d) for n, nbrs in self.adj.items() for m, d in nbrs.items() if pos[m] >= pos[n]]
        return [(n, m) for n, nbrs in self.adj.items() for m in nbrs if pos[m] >= pos[n]]

    def degree(self, n):
        return len(self.adj[n])

    def copy(self):
        g = Graph()
        g.node_attrs = {n: dict(a) for n, a in self.node_attrs.items()}
        # one attribute dict per edge, shared by both directions like in the original; node and neighbour orders are kept
        fresh = {}
        adj = {}
        for n, nbrs in self.adj.items():
            row = {}
            for m, d in nbrs.items():
                k = id(d)
                c = fresh.get(k)
                if c is None:
                    c = fresh[k] = dict(d)
                row[m] = c
            adj[n] = row
        g.adj = adj
        return g

    def connected_components(self):
        seen, comps = set(), []
        for s in self.adj:
            if s in seen:
                continue
            comp, queue = {s}, [s]
            while queue:
                v = queue.pop()
                for w in self.adj[v]:
                    if w not in comp:
                        comp.add(w)
                        queue.append(w)
            seen |= comp
            comps.append(comp)
        return comps


def relabel_nodes_inplace(g, mapping):
    """``networkx.relabel_nodes(G, mapping, copy=False)`` on a ``Graph``, including the ORDER it leaves behind (node order
    and every node's neighbour order decide the reference's tie breaks, global_optimization.py:231-246).  networkx renames
    node by node -- add the new node at the end, re-add the old node's edges, drop the old node; when old and new labels
    overlap the nodes are visited in reversed topological order of the mapping's digraph (generations of Kahn's
    algorithm, nodes in order of first appearance in ``mapping.items()``), otherwise in the graph's node order."""
    keys, vals = list(mapping.keys()), list(mapping.values())
    if set(keys) & set(vals):
        dnodes, succ, indeg = [], {}, {}
        for k, v in mapping.items():
            for n in (k, v):
                if n not in succ:
                    succ[n] = []
                    indeg[n] = 0
                    dnodes.append(n)
            if k != v:
                succ[k].append(v)
                indeg[v] += 1
        order, gen = [], [n for n in dnodes if indeg[n] == 0]
        left = dict(indeg)
        while gen:
            nxt = []
            for n in gen:
                order.append(n)
                for c in succ[n]:
                    left[c] -= 1
                    if left[c] == 0:
                        nxt.append(c)
            gen = nxt
        if len(order) != len(dnodes):
            raise ValueError("The node label sets are overlapping and no ordering can resolve the mapping. Use copy=True.")
        visit = order[::-1]
    else:
        visit = [n for n in g.nodes if n in mapping]
    for old in visit:
        if old not in mapping or old not in g.adj:
            continue
        new = mapping[old]
        g.add_node(new, **g.node_attrs.get(old, {}))
        if new == old:
            continue
        new_edges = [(new, new if old == t else t, d) for t, d in g.adj[old].items()]
        g.remove_node(old)
        for a, b, d in new_edges:
            g.add_edge(a, b, **d)
    return g


def get_faces_from_stack_props(stack_props):
    """Corner points of the 2 * ndim faces in world coordinates (mv_graph.py:386-420)."""
    sdims = [d for d in ["z", "y", "x"] if d in stack_props["spacing"]]
    ndim = len(sdims)
    gv = np.array(list(np.ndindex(*([2] * ndim))))
    faces = np.array([gv[gv[:, ax] == side] for ax in range(ndim) for side in (0, 1)], dtype=np.float64)
    faces = faces * (np.array([stack_props["shape"][d] for d in sdims]) - 1) * np.array([stack_props["spacing"][d] for d in sdims]) \
        + np.array([stack_props["origin"][d] for d in sdims])
    if "transform" in stack_props:
        shp = faces.shape
        faces = transformation.transform_pts(faces.reshape(-1, ndim), param_utils.select_time(np.asarray(stack_props["transform"]), 0)).reshape(shp)
    return faces


def get_center_from_stack_props(stack_props):
    """World coordinates of the stack centre (mv_graph.py:475-493)."""
    sdims = [d for d in SPATIAL_DIMS if d in stack_props["origin"]]
    c = np.array([stack_props["origin"][d] + stack_props["spacing"][d] * (stack_props["shape"][d] - 1) / 2 for d in sdims])
    if "transform" in stack_props:
        c = transformation.transform_pts(c[None], param_utils.select_time(np.asarray(stack_props["transform"]), 0))[0]
    return c


def _halfspace_equations_generic(stack_props):
    """The reference's face-by-face construction (mv_graph.py:183-218): any transform."""
    faces = get_faces_from_stack_props(stack_props)
    ndim = faces.shape[-1]
    center = get_center_from_stack_props(stack_props)
    eqs = []
    for face in faces:
        if ndim == 2:
            normal = np.array([-(face[1][1] - face[0][1]), face[1][0] - face[0][0]])
        else:
            normal = np.cross(face[1] - face[0], face[2] - face[0])
        normal = normal / np.linalg.norm(normal)
        if np.dot(normal, center) - np.dot(normal, face[0]) > 0:
            normal = -normal
        eqs.append(np.concatenate([normal, [-np.dot(normal, face[0])]]))
    return np.array(eqs)


def _is_pure_translation(stack_props, ndim):
    if "transform" not in stack_props:
        return True
    a = param_utils.select_time(np.asarray(stack_props["transform"], dtype=np.float64), 0)
    return a.shape == (ndim + 1, ndim + 1) and np.array_equal(a[:ndim, :ndim], np.eye(ndim))


def get_halfspace_equations_from_stack_props(stack_props):
    """Rows [n, c] with n . x + c <= 0 inside the stack (mv_graph.py:183-218).

    Views whose transform is a pure translation -- every tile of a stage-positioned mosaic -- take all faces in ONE set of array
    operations with the face-by-face form's arithmetic per element: the edge vectors of a face of such a box have one non-zero
    component each, so the cross product has one (a product of two extents; the others are exact zeros), its norm is that
    component's magnitude exactly (sqrt(x * x) == |x| in IEEE arithmetic) and every dot product has one non-zero term -- no sum in the
    chain depends on the order of its terms, and the rows come out bit for bit those of the loop (signed zeros of the normals
    included; tests/test_mv_graph_host.py).  The default crop sizing asks this once per pair geometry (registration.
    _reference_crop_differs): 144 pairs x 2 views x 6 faces of np.cross / norm / dot calls were a third of a process's first
    register()."""
    sdims = [d for d in ["z", "y", "x"] if d in stack_props["spacing"]]
    ndim = len(sdims)
    if ndim != 3 or not _is_pure_translation(stack_props, ndim):
        return _halfspace_equations_generic(stack_props)
    faces = get_faces_from_stack_props(stack_props)
    center = get_center_from_stack_props(stack_props)
    f0 = faces[:, 0]
    a, b = faces[:, 1] - f0, faces[:, 2] - f0
    normal = np.empty_like(f0)
    normal[:, 0] = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]      # (np.cross's own expressions)
    normal[:, 1] = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    normal[:, 2] = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    normal = normal / np.sqrt((normal * normal).sum(axis=1))[:, None]
    flip = ((normal * center).sum(axis=1) - (normal * f0).sum(axis=1)) > 0
    normal[flip] = -normal[flip]
    return np.concatenate([normal, -(normal * f0).sum(axis=1)[:, None]], axis=1)


def _axis_aligned_box(stack_props, tol=1e-12):
    """(lo, hi) if the stack's world frame is axis-aligned without permutation (transform = diag scale + shift), else None."""
    sdims = [d for d in ["z", "y", "x"] if d in stack_props["spacing"]]
    n = len(sdims)
    lo = np.array([stack_props["origin"][d] for d in sdims], dtype=np.float64)
    hi = lo + (np.array([stack_props["shape"][d] for d in sdims]) - 1) * np.array([stack_props["spacing"][d] for d in sdims])
    if "transform" in stack_props:
        a = param_utils.select_time(np.asarray(stack_props["transform"], dtype=np.float64), 0)
        lin = a[:n, :n]
        dg = lin.diagonal()
        off = lin.copy()
        off[np.arange(n), np.arange(n)] = 0.0
        if np.any(np.abs(off) > tol) or np.any(dg <= 0):
            return None
        lo, hi = dg * lo + a[:n, n], dg * hi + a[:n, n]
    return lo, hi


def _axis_aligned_boxes(sps, tol=1e-12):
    """[_axis_aligned_box(sp) for sp in sps]: all views of a regular mosaic (same axes, a static transform each) in one set of
    array operations -- the same element-wise arithmetic, so the same boxes; anything else view by view."""
    try:
        sdims = [d for d in ["z", "y", "x"] if d in sps[0]["spacing"]]
        n = len(sdims)
        if not all("transform" in sp and list(sp["spacing"]) == list(sps[0]["spacing"]) and np.shape(sp["transform"]) == (n + 1, n + 1) for sp in sps):
            raise ValueError
        lo = np.array([[sp["origin"][d] for d in sdims] for sp in sps], dtype=np.float64)
        hi = lo + (np.array([[sp["shape"][d] for d in sdims] for sp in sps]) - 1) * np.array([[sp["spacing"][d] for d in sdims] for sp in sps])
        a = np.array([np.asarray(sp["transform"], dtype=np.float64) for sp in sps])
        lin = a[:, :n, :n]
        dg = np.diagonal(lin, axis1=1, axis2=2)
        off = lin.copy()
        off[:, np.arange(n), np.arange(n)] = 0.0
        bad = np.any(np.abs(off) > tol, axis=(1, 2)) | np.any(dg <= 0, axis=1)
        lo2, hi2 = dg * lo + a[:, :n, n], dg * hi + a[:, :n, n]
        return [None if bad[i] else (lo2[i], hi2[i]) for i in range(len(sps))]
    except (ValueError, KeyError, TypeError):
        return [_axis_aligned_box(sp, tol) for sp in sps]


def get_overlap_between_pair_of_stack_props(stack_props1, stack_props2, closed_form=True, need_volume=True):
    """Volume (area in 2D) of the intersection of two views in world coordinates (mv_graph.py:301-338); -1 when the
    intersection is degenerate / empty.  Axis-aligned pairs: product of the interval overlaps, the value Qhull returns
    for a box; general pairs: the reference's linprog + HalfspaceIntersection + ConvexHull sequence."""
    b1, b2 = (_axis_aligned_box(stack_props1), _axis_aligned_box(stack_props2)) if closed_form else (None, None)
    if b1 is not None and b2 is not None:
        ext = np.minimum(b1[1], b2[1]) - np.maximum(b1[0], b2[0])
        return (float(np.prod(ext)), None) if np.all(ext > 0) else (-1, None)
    from scipy.optimize import linprog
    from scipy.spatial import ConvexHull, HalfspaceIntersection
    from scipy.spatial import QhullError

    eqs = np.concatenate([get_halfspace_equations_from_stack_props(stack_props1), get_halfspace_equations_from_stack_props(stack_props2)])
    norm = np.linalg.norm(eqs[:, :-1], axis=1).reshape(-1, 1)
    c = np.zeros(eqs.shape[1])
    c[-1] = -1
    res = linprog(c, A_ub=np.hstack((eqs[:, :-1], norm)), b_ub=-eqs[:, -1:], bounds=(None, None))
    if res.x is None:
        return -1, None
    try:
        hs = HalfspaceIntersection(eqs, res.x[:-1])
        return (float(ConvexHull(hs.intersections).volume) if need_volume else None), hs      # (callers that only want the vertices)
    except QhullError:
        return -1, None


def extend_stack_props(stack_props, extend_by):
    """spatial_image_utils.extend_stack_props (spatial_image_utils.py:889-913); returns a new dict."""
    sp = {k: (dict(v) if isinstance(v, dict) else v) for k, v in stack_props.items()}
    if not isinstance(extend_by, dict):
        extend_by = {d: extend_by for d in sp["spacing"]}
    for d, val in extend_by.items():
        sp["shape"][d] = sp["shape"][d] + int(np.ceil(2 * val / sp["spacing"][d]))
        sp["origin"][d] = sp["origin"][d] - val
    return sp


def build_view_adjacency_graph(stack_props_list, overlap_tolerance=None, pairs=None):
    """Views as nodes (attribute ``stack_props``, incl. ``transform``), an edge with ``overlap`` = intersection volume
    between every pair of views that overlap (mv_graph.py:35-180).  Candidate pairs: views whose centres are closer than
    the largest view diagonal + 1 (cKDTree ball query, as in the reference), unless ``pairs`` is given."""
    from scipy.spatial import cKDTree

    sps = [extend_stack_props(sp, overlap_tolerance) if overlap_tolerance is not None else sp for sp in stack_props_list]
    g = Graph(range(len(sps)))
    for i, sp in enumerate(sps):
        g.node_attrs[i]["stack_props"] = sp
    if pairs is None:
        centers = np.array([get_center_from_stack_props(sp) for sp in sps])
        diam = max(np.linalg.norm([sp["shape"][d] * sp["spacing"][d] for d in sp["spacing"]]) for sp in sps)
        tree = cKDTree(centers)
        pairs = [(i, j) for i, nbrs in enumerate(tree.query_ball_point(centers, diam + 1)) for j in nbrs if i != j]   # (one query for all views)
    # overlap volume per unordered pair (the reference evaluates (i, j) and (j, i); the volume is symmetric): all pairs of
    # axis-aligned views in one vectorised closed form, the others one by one through the halfspace intersection
    pair_keys = [(i, j) if i < j else (j, i) for i, j in pairs]
    keys = list(dict.fromkeys(pair_keys))
    boxes = _axis_aligned_boxes(sps)
    cache = {}
    aa = [k for k in keys if boxes[k[0]] is not None and boxes[k[1]] is not None]
    if aa:
        lo = np.array([b[0] if b is not None else np.zeros(len(sps[0]["spacing"])) for b in boxes])
        hi = np.array([b[1] if b is not None else np.zeros(len(sps[0]["spacing"])) for b in boxes])
        ia, ib = np.array([k[0] for k in aa]), np.array([k[1] for k in aa])
        ext = np.minimum(hi[ia], hi[ib]) - np.maximum(lo[ia], lo[ib])
        vol = np.where(np.all(ext > 0, axis=1), np.prod(ext, axis=1), -1.0)
        cache.update({k: float(v) for k, v in zip(aa, vol)})
    for k in keys:
        if k not in cache:
            cache[k] = get_overlap_between_pair_of_stack_props(sps[k[0]], sps[k[1]])[0]
    added = set()
    for (i, j), k in zip(pairs, pair_keys):
        ov = cache[k]
        if ov > 0 and k not in added:        # "overlap 0 means one pixel overlap" is not an edge; (j, i) after (i, j) changes nothing
            added.add(k)
            g.add_edge(i, j, overlap=ov)
    return g


def registration_edges_native(stack_props_list, overlap_tolerance=None, pairs=None, method="alternating_pattern",
                              pruning_method_kwargs=None):
    """The edge list ``prune_view_adjacency_graph(build_view_adjacency_graph(...), method, kwargs).edges()`` -- sorted tuples in
    networkx's edges() order, i.e. registration.register's work list -- for the common case in ONE library call
    (``mvs_view_graph_prune``, host code): every view an axis-aligned box in world coordinates (transform = positive diagonal +
    shift, off-diagonal terms exactly zero, one static transform per view) and ``method`` None or "alternating_pattern".
    Returns None when the case is not covered (the caller then takes the generic functions, which also raise the
    reference's errors); the candidate pairs still come from scipy's cKDTree ball query, whose traversal order decides the
    neighbour order of the graph (mv_graph.py:104-133).  tests/test_graph_native.py compares the two paths."""
    import ctypes as C
    from itertools import chain

    from . import _lib

    kw = dict(pruning_method_kwargs or {})
    if method not in (None, "alternating_pattern") or set(kw) - {"n_colors"} or not stack_props_list:
        return None
    sps = stack_props_list
    try:
        sdims = [d for d in ("z", "y", "x") if d in sps[0]["spacing"]]
        n = len(sdims)
        keys = list(sps[0]["spacing"])
        if not all("transform" in sp and list(sp["spacing"]) == keys and list(sp["origin"]) == list(sps[0]["origin"]) for sp in sps):
            return None
        origin = np.array([[sp["origin"][d] for d in sdims] for sp in sps], dtype=np.float64)
        spacing = np.array([[sp["spacing"][d] for d in sdims] for sp in sps], dtype=np.float64)
        shape = np.array([[sp["shape"][d] for d in sdims] for sp in sps])
        a = np.array([np.asarray(sp["transform"], dtype=np.float64) for sp in sps])
    except (KeyError, TypeError, ValueError):
        return None
    if a.shape != (len(sps), n + 1, n + 1) or shape.dtype.kind not in "iu":
        return None
    if overlap_tolerance is not None:       # extend_stack_props (spatial_image_utils.py:889-913), all views at once
        tol = overlap_tolerance if isinstance(overlap_tolerance, dict) else {d: overlap_tolerance for d in keys}
        if set(tol) - set(sdims):
            return None
        tv = np.array([float(tol.get(d, 0.0)) for d in sdims])
        shape = shape + np.ceil(2 * tv / spacing).astype(np.int64)
        origin = origin - tv
    lin = a[:, :n, :n]
    dg = np.diagonal(lin, axis1=1, axis2=2)
    off = lin.copy()
    off[:, np.arange(n), np.arange(n)] = 0.0
    last = np.zeros(n + 1)
    last[n] = 1.0
    if np.any(off != 0.0) or np.any(dg <= 0) or np.any(a[:, n, :] != last):
        return None
    t = a[:, :n, n]
    lo = origin
    hi = lo + (shape - 1) * spacing
    lo, hi = dg * lo + t, dg * hi + t
    if not (np.all(np.isfinite(lo)) and np.all(np.isfinite(hi))):
        return None
    if pairs is None:
        from scipy.spatial import cKDTree

        centers = dg * (origin + spacing * (shape - 1) / 2) + t
        diam = float(np.max(np.sqrt(np.sum((shape * spacing) ** 2, axis=1))))
        near = cKDTree(centers).query_ball_point(centers, diam + 1)
        lens = [len(x) for x in near]
        second = np.fromiter(chain.from_iterable(near), dtype=np.int32, count=sum(lens))
        first = np.repeat(np.arange(len(sps), dtype=np.int32), lens)
        keep = first != second
        pr = np.ascontiguousarray(np.stack([first[keep], second[keep]], axis=1), dtype=np.int32)
    else:
        pr = np.ascontiguousarray(np.asarray(pairs, dtype=np.int64).reshape(-1, 2), dtype=np.int32)
        if pr.size and (pr.min() < 0 or pr.max() >= len(sps) or np.any(pr[:, 0] == pr[:, 1])):
            return None
    npairs = len(pr)
    lo, hi = np.ascontiguousarray(lo), np.ascontiguousarray(hi)
    edges = np.empty((max(npairs, 1), 2), dtype=np.int32)
    ovl = np.empty(max(npairs, 1), dtype=np.float64)
    ne = C.c_int32(0)
    ptr = lambda arr, ty: arr.ctypes.data_as(C.POINTER(ty))
    rc = _lib.load().mvs_view_graph_prune(n, len(sps), ptr(lo, C.c_double), ptr(hi, C.c_double), npairs, ptr(pr, C.c_int32),
                                          0 if method is None else 1, int(kw.get("n_colors", 2)), ptr(edges, C.c_int32),
                                          ptr(ovl, C.c_double), C.byref(ne), None)
    if rc != 0 or ne.value == 0:      # not covered / no overlap at all: the generic path decides (and raises NotEnoughOverlapError)
        return None
    return [(i, j) for i, j in edges[: ne.value].tolist()]


def edge_betweenness_centrality(g):
    """networkx.edge_betweenness_centrality(g) with its defaults (Brandes, unweighted, normalised by n (n - 1)): host code in
    the library (``mvs_edge_betweenness``: 10 ms -> 0.2 ms for the 64-view mosaic), same traversal and accumulation order as
    the Python form below, which the tests pin against networkx itself."""
    import ctypes as C

    from . import _lib

    nodes = list(g.nodes)
    edges = list(g.edges())
    if not nodes or not edges:
        return {e: 0.0 for e in edges}
    index = {v: i for i, v in enumerate(nodes)}
    eid = {}
    for k, (a, b) in enumerate(edges):
        eid[(a, b)] = k
        eid[(b, a)] = k
    offsets, adj_nodes, adj_edge = [0], [], []
    for v in nodes:
        for w in g.adj[v]:
            adj_nodes.append(index[w])
            adj_edge.append(eid[(v, w)])
        offsets.append(len(adj_nodes))
    off = np.asarray(offsets, dtype=np.int32)
    an = np.asarray(adj_nodes, dtype=np.int32)
    ae = np.asarray(adj_edge, dtype=np.int32)
    bet = np.zeros(len(edges), dtype=np.float64)
    ptr = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    rc = _lib.load().mvs_edge_betweenness(len(nodes), len(edges), ptr(off, C.c_int32), ptr(an, C.c_int32), ptr(ae, C.c_int32),
                                          ptr(bet, C.c_double))
    if rc != 0:
        raise RuntimeError(f"mvs_edge_betweenness failed (code {rc})")
    return {e: float(bet[k]) for k, e in enumerate(edges)}


def _edge_betweenness_centrality_python(g):
    """Brandes' algorithm for unweighted graphs, normalised by n (n - 1) -- networkx.edge_betweenness_centrality with its
    defaults, same traversal and accumulation order (the values are compared with <=, so rounding matters)."""
    nodes = g.nodes
    bet = {e: 0.0 for e in g.edges()}
    for s in nodes:
        S, P, sigma, D = [], {v: [] for v in nodes}, dict.fromkeys(nodes, 0.0), {s: 0}
        sigma[s] = 1.0
        queue, head = [s], 0
        while head < len(queue):
            v = queue[head]
            head += 1
            S.append(v)
            for w in g.adj[v]:
                if w not in D:
                    queue.append(w)
                    D[w] = D[v] + 1
                if D[w] == D[v] + 1:
                    sigma[w] += sigma[v]
                    P[w].append(v)
        delta = dict.fromkeys(S, 0)
        while S:
            w = S.pop()
            coeff = (1 + delta[w]) / sigma[w]
            for v in P[w]:
                c = sigma[v] * coeff
                bet[(v, w) if (v, w) in bet else (w, v)] += c
                delta[v] += c
    n = len(nodes)
    if n > 1:
        for e in bet:
            bet[e] *= 1 / (n * (n - 1))
    return bet


def greedy_color(g, limit=None):
    """networkx.coloring.greedy_color, strategy largest_first: nodes by descending degree (stable), smallest free colour.
    ``limit``: return None as soon as a node needs colour ``limit`` (the caller only asks whether ``limit`` colours suffice)."""
    colors = {}
    adj = g.adj
    get = colors.get
    for u in sorted(adj, key=lambda n: len(adj[n]), reverse=True):
        used = {get(v) for v in adj[u]}          # (None for neighbours without a colour yet)
        c = 0
        while c in used:
            c += 1
        if limit is not None and c >= limit:
            return None
        colors[u] = c
    return colors


def prune_graph_to_alternating_colors(g, n_colors=2, return_colors=True):
    """Remove the weakest edges (overlap + a betweenness bonus of at most half the smallest overlap) in rising order until
    the graph can be coloured greedily with ``n_colors`` colours (mv_graph.py:664-741); edges whose removal would isolate
    a view stay.  On a regular grid this removes the diagonal neighbours."""
    if not g.edges():
        return (g, {n: 0 for n in g.nodes}) if return_colors else g
    gp = g.copy()
    cent = edge_betweenness_centrality(g)
    cmax, cmin = max(cent.values()), min(cent.values())
    min_overlap = min(d["overlap"] for _, _, d in gp.edges(data=True))
    if cmax > cmin:
        cent = {e: (c - cmin) / (cmax - cmin) * 0.5 * min_overlap for e, c in cent.items()}
    vals = {(a, b): cent[(a, b)] + d["overlap"] for a, b, d in gp.edges(data=True)}
    levels = sorted(np.unique(list(vals.values())))
    # the edges in rising order of their value: a level looks at the ones it newly reaches and at those an earlier level had
    # to keep (an end point of degree 1) -- the same set as a scan of all remaining edges, without the scan
    rising = sorted(vals, key=vals.get)
    nxt, kept = 0, []
    k = 0
    colors, changed = None, True
    while True:
        if changed:       # (a level that removes nothing leaves the colouring as it was)
            colors = greedy_color(gp, limit=n_colors)
            if colors is not None:      # (every colour used is below n_colors)
                break
        adj, lev = gp.adj, levels[k]
        while nxt < len(rising) and vals[rising[nxt]] <= lev:
            kept.append(rising[nxt])
            nxt += 1
        # (degrees as they are BEFORE this level removes anything, like the list the full scan builds first)
        drop = [(a, b) for a, b in kept if len(adj[a]) > 1 and len(adj[b]) > 1]
        if drop:
            gone = set(drop)
            kept = [e for e in kept if e not in gone]
            for a, b in drop:
                gp.remove_edge(a, b)
        changed = bool(drop)
        k += 1
    return (gp, colors) if return_colors else gp


def _bidirectional_dijkstra_path(g, source, target, weight):
    """The path ``networkx.shortest_path(G, source, target, weight)`` returns (the reference's call, mv_graph.py:786-791):
    networkx answers a weighted source-target query with a bidirectional Dijkstra search -- the two fringes are advanced
    alternately, starting with the source side; heap entries are (distance, push counter, node); whenever a relaxed node
    has been seen from both sides the concatenated path replaces the best one if it is strictly shorter; the search stops
    when a node settled on one side is popped settled on the other.  Among equal-length paths (every regular tile grid has
    many) the winner therefore depends on this exact order, which is why it is restated step by step."""
    import heapq

    if source == target:
        return [source]
    dists = ({}, {})
    paths = ({source: [source]}, {target: [target]})
    fringe = ([], [])
    seen = ({source: 0}, {target: 0})
    count = 0
    heapq.heappush(fringe[0], (0, count, source))
    count += 1
    heapq.heappush(fringe[1], (0, count, target))
    count += 1
    final_dist, final_path = None, []
    side = 1
    while fringe[0] and fringe[1]:
        side = 1 - side
        dist, _, v = heapq.heappop(fringe[side])
        if v in dists[side]:
            continue
        dists[side][v] = dist
        if v in dists[1 - side]:
            return final_path
        for w, attrs in g.adj[v].items():
            vw = dists[side][v] + attrs[weight]
            if w in dists[side]:
                if vw < dists[side][w]:
                    raise ValueError("Contradictory paths found: negative weights?")
            elif w not in seen[side] or vw < seen[side][w]:
                seen[side][w] = vw
                heapq.heappush(fringe[side], (vw, count, w))
                count += 1
                paths[side][w] = paths[side][v] + [w]
                if w in seen[0] and w in seen[1]:
                    total = seen[0][w] + seen[1][w]
                    if not final_path or final_dist > total:
                        final_dist = total
                        final_path = paths[0][w] + paths[1][w][::-1][1:]
    raise NotEnoughOverlapError(f"No path between {source} and {target}.")


def prune_to_shortest_weighted_paths(g):
    """Keep the edges on the overlap-weighted shortest paths from each component's best-connected view
    (mv_graph.py:744-805)."""
    import warnings

    ccs = g.connected_components()
    if max(len(c) for c in ccs) < 2:
        raise NotEnoughOverlapError("No overlap between views/tiles.")
    lonely = [n for c in ccs if len(c) == 1 for n in c]
    if lonely:
        warnings.warn("The following views/tiles have no links with other views:\n%s" % lonely, UserWarning, stacklevel=1)
    out = Graph()
    for n in g.nodes:
        out.add_node(n, **g.node_attrs[n])
    for a, b, d in g.edges(data=True):
        d["overlap_inv"] = 1 / (d["overlap"] + 1)
    for cc in ccs:
        totals = {n: sum(d["overlap"] for d in g.adj[n].values()) for n in g.nodes if n in cc}
        ref = max(totals, key=totals.get)
        for n in cc:
            path = _bidirectional_dijkstra_path(g, ref, n, "overlap_inv")
            for a, b in zip(path[:-1], path[1:]):
                out.add_edge(a, b, overlap=g.adj[a][b]["overlap"])
    return out


def prune_to_axis_aligned_edges(g, max_angle=0.05):
    """Keep the edges whose centre-to-centre direction lies within ``max_angle`` rad of one of the first view's axes
    (mv_graph.py:808-855).  Evaluated for all edges at once."""
    out = Graph()
    for n in g.nodes:
        out.add_node(n, **g.node_attrs[n])
    edges = g.edges(data=True)
    if not edges:
        return out
    idx = {n: k for k, n in enumerate(g.nodes)}
    verts = np.array([get_vertices_from_stack_props(g.node_attrs[n]["stack_props"]) for n in g.nodes])    # (N, 2^n, n)
    ndim = verts.shape[2]
    centers = verts.mean(axis=1)
    grid = np.array(list(np.ndindex(*([2] * ndim))))
    ax = verts[:, np.sum(grid, axis=1) == 1, :] - verts[:, :1, :]                                              # (N, n, n) axis vectors
    ax = ax / np.linalg.norm(ax, axis=2, keepdims=True)
    ia, ib = np.array([idx[a] for a, _, _ in edges]), np.array([idx[b] for _, b, _ in edges])
    vec = centers[ib] - centers[ia]
    vec = vec / np.linalg.norm(vec, axis=1, keepdims=True)
    with np.errstate(invalid="ignore"):
        angle = np.arccos(np.abs(np.einsum("ed,ekd->ek", vec, ax[ia])))
    keep = np.any(angle < max_angle, axis=1)
    for (a, b, d), k in zip(edges, keep):
        if k:
            out.add_edge(a, b, **d)
    return out


def threshold_otsu(values, nbins=256):
    """skimage.filters.threshold_otsu on a 1-D sample (the call of mv_graph.py:858-881): 256-bin histogram over the value
    range, threshold = centre of the bin that maximises the between-class variance."""
    values = np.asarray(values, dtype=np.float64).ravel()
    if values.min() == values.max():
        return values[0]
    counts, edges = np.histogram(values, bins=nbins, range=(values.min(), values.max()))
    centers = (edges[:-1] + edges[1:]) / 2.0
    counts = counts.astype(np.float64)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / w1
    m2 = (np.cumsum((counts * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return centers[np.argmax(var12)]


def filter_edges(g, weight_key="overlap", threshold=None):
    """Drop the edges whose weight lies below the (Otsu) threshold (mv_graph.py:858-881)."""
    edges = g.edges(data=True)
    if not edges:
        return g
    if threshold is None:
        threshold = threshold_otsu([d[weight_key] for _, _, d in edges])
    out = g.copy()
    for a, b, d in edges:
        if np.min(d[weight_key]) < threshold:
            out.remove_edge(a, b)
    return out


def prune_view_adjacency_graph(g, method=None, pruning_method_kwargs=None):
    """Dispatcher of mv_graph.py:1148-1196."""
    if not g.edges():
        raise NotEnoughOverlapError("Not enough overlap between views for stitching.")
    kw = dict(pruning_method_kwargs or {})
    if method is None:
        return g
    if method == "alternating_pattern":
        return prune_graph_to_alternating_colors(g, return_colors=False, **kw)
    if method == "shortest_paths_overlap_weighted":
        return prune_to_shortest_weighted_paths(g, **kw)
    if method == "otsu_threshold_on_overlap":
        return filter_edges(g, **kw)
    if method == "keep_axis_aligned":
        return prune_to_axis_aligned_edges(g, **kw)
    raise ValueError(f"Unknown graph pruning method: {method}")
