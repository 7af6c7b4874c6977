"""CPU oracle: groupwise parameter resolution and view-graph pruning.

TEST INFRASTRUCTURE (see oracle/__init__.py): imported by tests only, never by the product.

Plain numpy + networkx restatement -- the reference's own data structure here is an ``nx.Graph`` and networkx 3.4 is
installed in this image, so the graph calls are the reference's calls -- of

  param_resolution.groupwise_resolution                       src/multiview_stitcher/param_resolution/__init__.py:44-150
  groupwise_resolution_global_optimization                    .../param_resolution/global_optimization.py:16-166
  optimize_bead_subgraph                                      .../param_resolution/global_optimization.py:169-511
  get_beads_graph_from_reg_graph / compute_edge_residuals     .../param_resolution/utils.py:42-101
  transforms.TranslationTransform / Affine_Fit                src/multiview_stitcher/transforms.py:45-168
  mv_graph.get_node_with_maximal_edge_weight_sum_from_graph   src/multiview_stitcher/mv_graph.py:341-352
  mv_graph.prune_graph_to_alternating_colors                  src/multiview_stitcher/mv_graph.py:664-741
  mv_graph.prune_to_shortest_weighted_paths                   src/multiview_stitcher/mv_graph.py:744-803
  mv_graph.prune_to_axis_aligned_edges                        src/multiview_stitcher/mv_graph.py:806-855
  mv_graph.filter_edges                                       src/multiview_stitcher/mv_graph.py:858-881
  mv_graph.prune_view_adjacency_graph                         src/multiview_stitcher/mv_graph.py:1148-1196

Third-party arithmetic that is NOT installed here (scikit-image 0.26: ``EuclideanTransform`` / ``SimilarityTransform``
estimators = Umeyama's closed form, ``threshold_otsu``): restated from the published algorithms and pinned by vectors
produced with scikit-image 0.18.3 (tests/golden/skimage018_transforms.npz, tests/golden/make_skimage018_transform_fixture.py).
Data model: an edge of the registration graph carries ``transform`` ((n+1, n+1) float64, fixed -> moving in world
units), ``quality``, ``overlap`` and ``bbox`` ((2, n): lower / upper corner of the overlap in the fixed view's frame);
a node carries ``stack_props`` = {"spacing": {dim: float}, ...} -- plain arrays instead of the reference's xarray objects
(one time point; the reference loops over t around everything restated here, __init__.py:73-80).
"""

from __future__ import annotations

import copy

import networkx as nx
import numpy as np


# ---- small pieces --------------------------------------------------------------------------------------------------
def transform_pts(pts, affine):
    """transformation.transform_pts (transformation.py:151-161): one homogeneous np.dot per point."""
    pts = np.array(pts, dtype=np.float64)
    pts = np.concatenate([pts, np.ones((pts.shape[0], 1))], axis=1)
    pts_t = np.array([np.dot(np.array(affine), pt) for pt in pts])
    return pts_t[:, :-1]


def get_node_with_maximal_edge_weight_sum_from_graph(g, weight_key):
    """mv_graph.py:341-352."""
    total = {node: np.sum([g.edges[e][weight_key] for e in g.edges if node in e]) for node in g.nodes}
    return max(total, key=total.get)


def get_beads_graph_from_reg_graph(g_reg, ndim):
    """utils.py:42-78: per edge the corners of the overlap box (fixed frame) and their images under the pairwise transform."""
    g_beads = nx.Graph()
    g_beads.add_nodes_from(g_reg.nodes)
    for e in g_reg.edges:
        sorted_e = tuple(sorted(e))
        bbox_lower, bbox_upper = np.asarray(g_reg.edges[e]["bbox"], dtype=np.float64)
        gv = np.array(list(np.ndindex(tuple([2] * len(bbox_lower)))))
        bbox_vertices = gv * (bbox_upper - bbox_lower) + bbox_lower
        affine = np.asarray(g_reg.edges[e]["transform"], dtype=np.float64)
        g_beads.add_edge(
            sorted_e[0], sorted_e[1],
            beads={sorted_e[0]: bbox_vertices, sorted_e[1]: transform_pts(bbox_vertices, affine)},
            quality=g_reg.edges[e].get("quality", 1.0), overlap=g_reg.edges[e].get("overlap", 1.0),
        )
    for node in g_reg.nodes:
        g_beads.nodes[node]["affine"] = np.eye(ndim + 1)
    return g_beads


def compute_edge_residuals(g_reg, params, ndim):
    """utils.py:81-101: RMS bead residual per edge."""
    g_beads = get_beads_graph_from_reg_graph(g_reg, ndim)
    out = {}
    for e in g_beads.edges:
        n1, n2 = e
        p1 = transform_pts(g_beads.edges[e]["beads"][n1], params[n1])
        p2 = transform_pts(g_beads.edges[e]["beads"][n2], params[n2])
        out[tuple(sorted(e))] = float(np.sqrt(np.mean(np.sum((p1 - p2) ** 2, axis=1))))
    return out


# ---- point-set estimators (global_optimization.py:248-259) ------------------------------------------------------------
def estimate_translation(src, dst):
    """transforms.TranslationTransform.estimate (transforms.py:45-53)."""
    n = src.shape[1]
    p = np.eye(n + 1)
    p[:n, n] = np.mean(dst - src, 0)
    return p


def estimate_affine(src, dst):
    """transforms.AffineTransform.estimate = Affine_Fit (transforms.py:56-168): least squares ``dst ~ A src + t`` through the
    normal equations ``Q a = c`` (Q = sum q~ q~^T, c = sum q~ p^T with q~ = (q, 1)), solved by Gauss-Jordan elimination with
    partial pivoting; singular (pivot <= 1e-10) raises like the reference."""
    q, p = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    dim = q.shape[1]
    if len(q) != len(p) or len(q) < 1:
        raise ValueError("from_pts and to_pts must be of same size.")
    if len(q) < dim:
        raise ValueError("Too few points => under-determined system.")
    qt = np.concatenate([q, np.ones((len(q), 1))], axis=1)
    c = [[0.0] * dim for _ in range(dim + 1)]
    for j in range(dim):
        for k in range(dim + 1):
            for i in range(len(q)):
                c[k][j] += qt[i][k] * p[i][j]
    Q = [[0.0] * (dim + 1) for _ in range(dim + 1)]
    for row in qt:
        for i in range(dim + 1):
            for j in range(dim + 1):
                Q[i][j] += row[i] * row[j]
    M = [Q[i] + c[i] for i in range(dim + 1)]
    h, w = len(M), len(M[0])
    for y in range(h):
        maxrow = y
        for y2 in range(y + 1, h):
            if abs(M[y2][y]) > abs(M[maxrow][y]):
                maxrow = y2
        M[y], M[maxrow] = M[maxrow], M[y]
        if abs(M[y][y]) <= 1e-10:
            raise ValueError("Error: singular matrix. Points are probably coplanar.")
        for y2 in range(y + 1, h):
            f = M[y2][y] / M[y][y]
            for x in range(y, w):
                M[y2][x] -= M[y][x] * f
    for y in range(h - 1, -1, -1):
        f = M[y][y]
        for y2 in range(y):
            for x in range(w - 1, y - 1, -1):
                M[y2][x] -= M[y][x] * M[y2][y] / f
        M[y][y] /= f
        for x in range(h, w):
            M[y][x] /= f
    out = np.eye(dim + 1)
    for j in range(dim):
        for i in range(dim):
            out[j, i] = M[i][j + dim + 1]
        out[j, dim] = M[dim][j + dim + 1]
    return out


def estimate_umeyama(src, dst, estimate_scale):
    """skimage.transform.EuclideanTransform / SimilarityTransform.estimate: Umeyama, "Least-squares estimation of
    transformation parameters between two point patterns", PAMI 1991, eq. 34-43, as skimage's ``_umeyama`` spells it."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    num, dim = src.shape
    src_mean, dst_mean = src.mean(axis=0), dst.mean(axis=0)
    src_demean, dst_demean = src - src_mean, dst - dst_mean
    A = dst_demean.T @ src_demean / num
    d = np.ones((dim,), dtype=np.float64)
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1, dtype=np.float64)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.nan * T
    if rank == dim - 1:
        if np.linalg.det(U) * np.linalg.det(V) > 0:
            T[:dim, :dim] = U @ V
        else:
            s = d[dim - 1]
            d[dim - 1] = -1
            T[:dim, :dim] = U @ np.diag(d) @ V
            d[dim - 1] = s
    else:
        T[:dim, :dim] = U @ np.diag(d) @ V
    scale = 1.0 / src_demean.var(axis=0).sum() * (S @ d) if estimate_scale else 1.0
    T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
    T[:dim, :dim] *= scale
    return T


ESTIMATORS = {
    "translation": estimate_translation,
    "rigid": lambda s, d: estimate_umeyama(s, d, False),
    "similarity": lambda s, d: estimate_umeyama(s, d, True),
    "affine": estimate_affine,
}


# ---- global optimisation -----------------------------------------------------------------------------------------------
def optimize_bead_subgraph(g_beads, transform, ref_node, max_iter, rel_tol, abs_tol):
    """global_optimization.py:169-511, statement by statement (relabelling, sweep order by degree centrality, Gauss-Seidel
    sweeps with the estimator applied on top of the current affine, residual history, relative-change stop from the 7th
    sweep, edge-removal criterion and its connectivity guard)."""
    g_beads = copy.deepcopy(g_beads)
    mapping = {n: i for i, n in enumerate(g_beads.nodes)}
    inverse_mapping = dict(enumerate(g_beads.nodes))
    nx.relabel_nodes(g_beads, mapping, copy=False)
    for e in g_beads.edges:
        g_beads.edges[e]["beads"] = {mapping[k]: v for k, v in g_beads.edges[e]["beads"].items()}
    centralities = nx.degree_centrality(g_beads)
    sorted_nodes = sorted(centralities, key=centralities.get, reverse=True)
    ndim = g_beads.nodes[list(g_beads.nodes)[0]]["affine"].shape[-1] - 1
    if transform.lower() not in ESTIMATORS:
        raise ValueError(f"Unknown transformation type in parameter resolution: {transform}")
    estimate = ESTIMATORS[transform.lower()]
    all_nodes = list(mapping.values())
    new_affines = np.array([np.matmul(np.eye(ndim + 1), g_beads.nodes[n]["affine"]) for n in all_nodes])
    mean_residuals, max_residuals = [], []
    total_iterations = 0
    edge_residuals = {}
    while True:
        iter_all_residuals = []
        edges = list(g_beads.edges)
        if not len(edges):
            break
        node_edges = [list(g_beads.edges(n)) for n in all_nodes]
        node_beads = [np.concatenate([g_beads.edges[e]["beads"][n] for e in node_edges[n]], axis=0) if node_edges[n]
                      else np.zeros((0, ndim)) for n in all_nodes]
        node_beads = [np.concatenate([nb, np.ones((len(nb), 1))], axis=1) for nb in node_beads]
        adj_nodes = [[n for e in node_edges[cur] for n in e if n != cur] for cur in all_nodes]
        adj_beads = [[g_beads.edges[e]["beads"][n] for e in node_edges[cur] for n in e if n != cur] for cur in all_nodes]
        adj_beads = [[np.concatenate([abb, np.ones((len(abb), 1))], axis=1) for abb in ab] for ab in adj_beads]
        for iteration in range(max_iter):
            for cur in sorted_nodes:
                if not len(node_edges[cur]):
                    continue
                node_pts = np.dot(new_affines[cur], node_beads[cur].T).T[:, :-1]
                adj_pts = np.concatenate([np.dot(new_affines[an], adj_beads[cur][ian].T).T
                                          for ian, an in enumerate(adj_nodes[cur])], axis=0)[:, :-1]
                if cur != ref_node:
                    new_affines[cur] = np.matmul(estimate(node_pts, adj_pts), new_affines[cur])
                total_iterations += 1
            edge_residuals = {}
            for e in g_beads.edges:
                n1, n2 = e
                edge_residuals[e] = np.linalg.norm(transform_pts(g_beads.edges[e]["beads"][n1], new_affines[n1])
                                                   - transform_pts(g_beads.edges[e]["beads"][n2], new_affines[n2]), axis=1)
            mean_residuals.append(np.mean([np.mean(edge_residuals[e]) for e in g_beads.edges]))
            max_residuals.append(np.max([np.max(edge_residuals[e]) for e in g_beads.edges]))
            iter_all_residuals.append(edge_residuals)
            if iteration > 5:
                max_rel_change = np.max([
                    np.abs((iter_all_residuals[-1][e] - iter_all_residuals[-2][e]) / max_residuals[-1] if max_residuals[-1] > 0 else 0)
                    for e in g_beads.edges])
                if max_rel_change < rel_tol:
                    break
        if len(list(g_beads.edges)) < 2:
            break
        edges = list(g_beads.edges)
        if max_residuals[-1] < abs_tol:
            edge_to_remove = None
        else:
            vals = [(1 - float(g_beads.edges[e]["quality"])) ** 2 * np.sqrt(np.max(edge_residuals[e]))
                    * np.log10(np.max([len(list(g_beads.neighbors(n))) for n in e])) for e in edges]
            order = np.argsort(vals)[::-1]
            cand, found = 0, False
            while True:
                edge_to_remove = edges[order[cand]]
                nodes = list(edge_to_remove)
                tmp = copy.deepcopy(g_beads)
                tmp.remove_edge(*edge_to_remove)
                ccs = list(nx.connected_components(tmp))
                cc1 = [i for i, cc in enumerate(ccs) if nodes[0] in cc][0]
                if nodes[1] in ccs[cc1]:
                    found = True
                    break
                if cand == len(order) - 1:
                    break
                cand += 1
            if not found:
                edge_to_remove = None
        if edge_to_remove is not None:
            g_beads.remove_edge(*edge_to_remove)
        else:
            break
    if total_iterations:
        for n in all_nodes:
            g_beads.nodes[n]["affine"] = new_affines[n]
        for e, r in edge_residuals.items():
            if g_beads.has_edge(*e):
                g_beads.edges[e]["residual"] = np.mean(r)
    nx.relabel_nodes(g_beads, inverse_mapping, copy=False)
    params = {node: np.asarray(g_beads.nodes[node]["affine"]) for node in g_beads.nodes}
    history = {"mean_residual": list(mean_residuals), "max_residual": list(max_residuals)}
    return params, history, g_beads


def groupwise_resolution_global_optimization(g_reg, reference_view=None, transform="translation", max_iter=None,
                                             rel_tol=None, abs_tol=None):
    """global_optimization.py:16-166 for one connected component."""
    ndim_of = lambda g: np.asarray(g.edges[list(g.edges)[0]]["transform"]).shape[-1] - 1
    if not g_reg.number_of_edges():
        nd = len(g_reg.nodes[next(iter(g_reg.nodes))]["stack_props"]["spacing"])
        return {n: np.eye(nd + 1) for n in g_reg.nodes}, {"metrics": None, "used_edges": []}
    max_iter = 500 if max_iter is None else max_iter
    rel_tol = 1e-4 if rel_tol is None else rel_tol
    ndim = ndim_of(g_reg)
    if abs_tol is None:
        abs_tol = np.max([1.0 * np.sum([v ** 2 for v in g_reg.nodes[n]["stack_props"]["spacing"].values()]) ** 0.5
                          for n in g_reg.nodes])
    if reference_view is not None and reference_view in g_reg.nodes:
        ref_node = reference_view
    else:
        ref_node = get_node_with_maximal_edge_weight_sum_from_graph(g_reg, weight_key="quality")
    g_beads = get_beads_graph_from_reg_graph(g_reg, ndim=ndim)
    params, history, g_opt = optimize_bead_subgraph(g_beads, transform, ref_node, max_iter, rel_tol, abs_tol)
    return dict(params), {"metrics": history, "used_edges": [tuple(sorted(e)) for e in g_opt.edges]}


def groupwise_resolution(g_reg, resolver=groupwise_resolution_global_optimization, **kwargs):
    """__init__.py:44-150 (one time point): per connected component, two-view convention for the reference view."""
    if not len(g_reg.edges):
        raise ValueError("Not enough overlap between views for stitching.")
    if "reference_view" not in kwargs and len(g_reg.nodes) == 2:
        kwargs["reference_view"] = min(list(g_reg.nodes))
    params, used = {}, set()
    ndim = np.asarray(g_reg.edges[list(g_reg.edges)[0]]["transform"]).shape[-1] - 1
    for cc in nx.connected_components(g_reg):
        sub = g_reg.subgraph(list(cc))
        if not sub.number_of_edges():
            cc_params, info = {n: np.eye(ndim + 1) for n in cc}, None
        else:
            cc_params, info = resolver(sub, **kwargs)
        for n in cc:
            params[n] = cc_params[n]
        if info is not None:
            used.update(tuple(sorted(e)) for e in info.get("used_edges") or [])
    return params, {"edge_residuals": compute_edge_residuals(g_reg, params, ndim), "used_edges": sorted(used)}


# ---- pruning of the view adjacency graph ---------------------------------------------------------------------------------
def prune_graph_to_alternating_colors(g, n_colors=2, return_colors=True):
    """mv_graph.py:664-741."""
    if not len(g.edges):
        return (g, {n: 0 for n in g.nodes}) if return_colors else g
    g_pruned = copy.deepcopy(g)
    centrality = nx.edge_betweenness_centrality(g)
    max_c, min_c = max(centrality.values()), min(centrality.values())
    edges = list(g_pruned.edges(data=True))
    min_overlap = min([e[2]["overlap"] for e in edges])
    if max_c > min_c:
        centrality = {e: (centrality[e] - min_c) / (max_c - min_c) * 0.5 * min_overlap for e in centrality}
    edge_vals = {tuple(e[:2]): centrality[tuple(e[:2])] + e[2]["overlap"] for e in edges}
    sorted_unique_vals = sorted(np.unique(list(edge_vals.values())))
    thresh_ind = 0
    while 1:
        colors = nx.coloring.greedy_color(g_pruned)
        if len(set(colors.values())) <= n_colors:
            break
        g_pruned.remove_edges_from([(a, b) for a, b, attrs in g_pruned.edges(data=True)
                                    if edge_vals[(a, b)] <= sorted_unique_vals[thresh_ind]
                                    and min([len(g_pruned.edges(n)) for n in (a, b)]) > 1])
        thresh_ind += 1
    return (g_pruned, colors) if return_colors else g_pruned


def prune_to_shortest_weighted_paths(g):
    """mv_graph.py:744-803."""
    g_reg = copy.deepcopy(g)
    g_reg.remove_edges_from(list(g_reg.edges))
    ccs = list(nx.connected_components(g))
    if np.max([len(cc) for cc in ccs]) < 2:
        raise ValueError("No overlap between views/tiles.")
    for cc in ccs:
        sub = g.subgraph(list(cc))
        ref_node = get_node_with_maximal_edge_weight_sum_from_graph(sub, weight_key="overlap")
        for e in g.edges:
            g.edges[e]["overlap_inv"] = 1 / (g.edges[e]["overlap"] + 1)
        paths = {n: nx.shortest_path(g, target=n, source=ref_node, weight="overlap_inv") for n in cc}
        for _, sp in paths.items():
            if len(sp) < 2:
                continue
            for i in range(len(sp) - 1):
                g_reg.add_edge(sp[i], sp[i + 1], overlap=g[sp[i]][sp[i + 1]]["overlap"])
    return g_reg


def get_vertices_from_stack_props(stack_props):
    """mv_graph.py:423-444: the 2^n corners of a view in world coordinates (its ``transform`` applied when present)."""
    ndim = len(stack_props["origin"])
    sdims = ["z", "y", "x"][-ndim:]
    gv = np.array(list(np.ndindex(tuple([2] * ndim))))
    shape = np.array([stack_props["shape"][d] for d in sdims])
    spacing = np.array([stack_props["spacing"][d] for d in sdims], dtype=np.float64)
    origin = np.array([stack_props["origin"][d] for d in sdims], dtype=np.float64)
    verts = gv * (shape - 1) * spacing + origin
    if "transform" in stack_props and stack_props["transform"] is not None:
        verts = transform_pts(verts, np.asarray(stack_props["transform"], dtype=np.float64))
    return verts


def prune_to_axis_aligned_edges(g, max_angle=0.05):
    """mv_graph.py:806-855."""
    keep = []
    for edge in g.edges:
        verts1 = get_vertices_from_stack_props(g.nodes[edge[0]]["stack_props"])
        verts2 = get_vertices_from_stack_props(g.nodes[edge[1]]["stack_props"])
        ndim = len(verts1[0])
        edge_vec = np.mean(verts2, 0) - np.mean(verts1, 0)
        edge_vec = edge_vec / np.linalg.norm(edge_vec)
        grid = np.array(list(np.ndindex(tuple([2] * ndim))))
        ax_vecs = []
        for ind in range(len(grid)):
            if np.sum(grid[ind]) != 1:
                continue
            ax = verts1[ind] - verts1[0]
            ax_vecs.append(ax / np.linalg.norm(ax))
        for ax in ax_vecs:
            if np.arccos(np.abs(np.dot(edge_vec, ax))) < max_angle:
                keep.append(edge)
                break
    g_pruned = nx.Graph(g.edge_subgraph(keep))
    for node in g.nodes:
        if node not in g_pruned.nodes:
            g_pruned.add_node(node, **g.nodes[node])
    return g_pruned


def threshold_otsu(values, nbins=256):
    """skimage.filters.threshold_otsu on a 1-D sample (Otsu 1979): histogram of ``nbins`` bins, the bin centre maximising
    the between-class variance w1 w2 (m1 - m2)^2; a constant sample returns its value."""
    values = np.asarray(values, dtype=np.float64).ravel()
    first = values[0]
    if np.all(values == first):
        return first
    counts, edges = np.histogram(values, bins=nbins)
    centers = (edges[:-1] + edges[1:]) / 2.0
    counts = counts.astype(np.float64)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / w1
    m2 = (np.cumsum((counts * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return centers[np.argmax(var12)]


def filter_edges(g, weight_key="overlap", threshold=None):
    """mv_graph.py:858-881: drop the edges whose weight lies below the Otsu threshold of all weights."""
    edges = list(g.edges(data=True))
    if not edges:
        return g
    w = np.array([e[2][weight_key] for e in edges], dtype=np.float64)
    if threshold is None:
        threshold = threshold_otsu(w)
    g_filtered = g.copy()
    g_filtered.remove_edges_from([(a, b) for (a, b, _), wi in zip(edges, w) if wi < threshold])
    return g_filtered


def prune_view_adjacency_graph(g, method=None, pruning_method_kwargs=None):
    """mv_graph.py:1148-1196."""
    if not len(g.edges):
        raise ValueError("Not enough overlap between views for stitching.")
    kw = pruning_method_kwargs or {}
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
