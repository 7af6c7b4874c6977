"""Groupwise resolution of pairwise registrations into one transform per view, host side, numpy only.

Restatement of src/multiview_stitcher/param_resolution (``__init__.py:44-150``, ``global_optimization.py:16-511``,
``shortest_paths.py:9-99``, ``utils.py:42-101``) without networkx / pandas / xarray / skimage, none of which exist on
the MI355X box.  O(#views) work that sits between the two GPU stages of the hot path: ``registration.register``
hands the pairwise results over as a ``RegGraph`` and gets per-view affines back.

Data model: a ``RegGraph`` is a list of nodes and a dict ``{(i, j) sorted: {"transform": (n+1, n+1) array mapping
coordinates of view i's frame onto view j's (the pairwise result), "quality": float, "overlap": float, "bbox":
(2, n) lower / upper corner of the overlap in view i's frame}}`` plus per-node ``stack_props`` (for the default
``abs_tol``).  Insertion order of nodes and edges is significant exactly where networkx's is (tie breaks).
"""

from __future__ import annotations

import heapq

import numpy as np

from . import param_utils


class NotEnoughOverlapError(Exception):
    pass


class RegGraph:
    def __init__(self, nodes, stack_props=None):
        self.nodes = list(nodes)
        self.edges = {}
        self.stack_props = dict(stack_props or {})

    def add_edge(self, i, j, transform, quality=1.0, overlap=1.0, bbox=None):
        if i > j:   # stored in sorted orientation like get_beads_graph_from_reg_graph (utils.py:50-52)
            i, j, transform = j, i, np.linalg.inv(np.asarray(transform, dtype=np.float64))
        self.edges[(i, j)] = {"transform": np.asarray(transform, dtype=np.float64), "quality": float(quality),
                              "overlap": float(overlap), "bbox": None if bbox is None else np.asarray(bbox, dtype=np.float64)}
        self._beads = None          # (memo of _beads_of)

    @property
    def ndim(self):
        if self.edges:
            return next(iter(self.edges.values()))["transform"].shape[-1] - 1
        if self.stack_props:
            return len(next(iter(self.stack_props.values()))["spacing"])
        raise ValueError("Cannot determine dimensionality from graph.")

    def neighbors(self, n, edges=None):
        out = []
        for a, b in (self.edges if edges is None else edges):
            if a == n:
                out.append(b)
            elif b == n:
                out.append(a)
        return out

    def connected_components(self, edges=None):
        """Components in order of their first node (nx.connected_components)."""
        edges = list(self.edges if edges is None else edges)
        adj = {}                                  # neighbours in edge order (what neighbors() returns), built in one pass
        for a, b in edges:
            adj.setdefault(a, []).append(b)
            adj.setdefault(b, []).append(a)
        seen, comps = set(), []
        for start in self.nodes:
            if start in seen:
                continue
            comp, stack = {start}, [start]
            while stack:
                v = stack.pop()
                for w in adj.get(v, ()):
                    if w not in comp:
                        comp.add(w)
                        stack.append(w)
            seen |= comp
            comps.append(comp)
        return comps

    def subgraph(self, nodes):
        nodes = set(nodes)
        g = RegGraph([n for n in self.nodes if n in nodes], {n: self.stack_props[n] for n in nodes if n in self.stack_props})
        g.edges = {e: dict(v) for e, v in self.edges.items() if e[0] in nodes and e[1] in nodes}
        if len(g.edges) == len(self.edges):
            g._beads = getattr(self, "_beads", None)      # the same edges: the same virtual beads
        return g


def _nx_edge_order(nodes, inserted_edges):
    """Order in which networkx iterates the edges of an undirected graph whose edges were added in ``inserted_edges``
    order: node by node, each node's neighbours in the order their edges were added, every edge once.  The reference's
    tie breaks (argsort over equal scores, e.g. all qualities 1) follow this order, not insertion order."""
    adj = {n: [] for n in nodes}
    for a, b in inserted_edges:
        adj[a].append(b)
        adj[b].append(a)
    keys = set(inserted_edges)
    seen, out = set(), []
    for n in nodes:
        for m in adj[n]:
            if m not in seen:
                out.append((n, m) if (n, m) in keys else (m, n))
        seen.add(n)
    return out


def transform_pts(pts, affine):
    """transformation.transform_pts: homogeneous multiply."""
    pts = np.asarray(pts, dtype=np.float64)
    affine = np.asarray(affine, dtype=np.float64)
    return pts @ affine[:-1, :-1].T + affine[:-1, -1]


def _beads_of(g):
    """Virtual beads per edge (utils.py:42-78): the corners of the overlap box in the fixed view's frame and their images
    under the pairwise transform in the moving view's frame.  All edges at once: (E, 2^n, n) arrays."""
    keys = list(g.edges)
    if not keys:
        return {}
    # memo per graph object, valid while the edges still hold the very arrays it was made from
    stamp = tuple((k, id(e["transform"]), id(e["bbox"])) for k, e in g.edges.items())
    memo = getattr(g, "_beads", None)
    if memo is not None and memo[0] == stamp:
        return memo[1]
    n = g.ndim
    lo = np.array([g.edges[k]["bbox"][0] for k in keys], dtype=np.float64)
    hi = np.array([g.edges[k]["bbox"][1] for k in keys], dtype=np.float64)
    T = np.array([g.edges[k]["transform"] for k in keys], dtype=np.float64)
    gv = np.array(list(np.ndindex(*([2] * n))), dtype=np.float64)
    verts = gv[None] * (hi - lo)[:, None, :] + lo[:, None, :]
    moved = np.einsum("eij,ebj->ebi", T[:, :n, :n], verts) + T[:, None, :n, n]
    out = {k: {k[0]: verts[i], k[1]: moved[i]} for i, k in enumerate(keys)}
    try:
        g._beads = (stamp, out)
    except AttributeError:
        pass
    return out


def compute_edge_residuals(g, params):
    """RMS bead residual per edge in physical units (utils.py:81-101)."""
    beads = _beads_of(g)
    if not beads:
        return {}
    n = g.ndim
    keys = list(beads)
    A = np.array([beads[k][k[0]] for k in keys])
    B = np.array([beads[k][k[1]] for k in keys])
    Pa = np.array([params[k[0]] for k in keys], dtype=np.float64)
    Pb = np.array([params[k[1]] for k in keys], dtype=np.float64)
    d = (np.einsum("eij,ebj->ebi", Pa[:, :n, :n], A) + Pa[:, None, :n, n]) - (np.einsum("eij,ebj->ebi", Pb[:, :n, :n], B) + Pb[:, None, :n, n])
    r = np.sqrt(np.mean(np.sum(d ** 2, axis=2), axis=1))
    return {k: float(r[i]) for i, k in enumerate(keys)}


def _sum_like_numpy(values):
    """np.sum of a short list without the ufunc machinery: below 8 elements numpy's reduction is the plain left-to-right loop
    (its pairwise / unrolled summation starts at 8), so this is bitwise np.sum there; longer lists go to numpy."""
    if len(values) >= 8:
        return float(np.sum(values))
    t = 0.0
    for i, v in enumerate(values):
        t = float(v) if i == 0 else t + float(v)
    return t


def get_node_with_maximal_edge_weight_sum_from_graph(g, weight_key="quality"):
    """mv_graph.py:341-352 (first maximum in node order)."""
    per_node = {n: [] for n in g.nodes}           # the weights of a node's edges in edge order, gathered in one pass
    for k, e in g.edges.items():
        per_node[k[0]].append(e[weight_key])
        if k[1] != k[0]:
            per_node[k[1]].append(e[weight_key])
    totals = {n: _sum_like_numpy(w) for n, w in per_node.items()}
    return max(totals, key=totals.get)


# ---- point-set transform estimators (the classes optimize_bead_subgraph instantiates, global_optimization.py:248-259) ----
def _estimate_translation(src, dst):
    """transforms.py:45-53."""
    n = src.shape[1]
    m = np.eye(n + 1)
    m[:n, n] = np.mean(dst - src, axis=0)
    return m


def _umeyama(src, dst, estimate_scale):
    """Least-squares rigid / similarity fit (Umeyama 1991, the algorithm behind skimage's EuclideanTransform /
    SimilarityTransform.estimate, which the reference instantiates)."""
    num, dim = src.shape
    src_mean, dst_mean = src.mean(axis=0), dst.mean(axis=0)
    src_d, dst_d = src - src_mean, dst - dst_mean
    A = dst_d.T @ src_d / num
    d = np.ones(dim)
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.full((dim + 1, dim + 1), np.nan)
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
    scale = 1.0 / src_d.var(axis=0).sum() * (S @ d) if estimate_scale else 1.0
    T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
    T[:dim, :dim] *= scale
    return T


def _estimate_affine(src, dst):
    """Least-squares affine fit dst ~ A src + t (transforms.py:56-66; normal equations of Spaeth 2003 solved by lstsq)."""
    n = src.shape[1]
    X = np.concatenate([src, np.ones((len(src), 1))], axis=1)
    sol, *_ = np.linalg.lstsq(X, dst, rcond=None)
    m = np.eye(n + 1)
    m[:n, :n] = sol[:n].T
    m[:n, n] = sol[n]
    return m


_FORCE_NUMPY = False   # tests: run the translation model through the generic numpy sweeps as well


def _sweeps_numpy(estimate, all_nodes, edges, beads, sorted_nodes, ref, max_iter, rel_tol, new_affines, node_edges=None):
    """Inner loop of optimize_bead_subgraph (global_optimization.py:264-417) for any transform model: sweeps over the
    nodes (most connected first), each re-fitted to the current positions of the beads it shares with its neighbours."""
    hom = lambda p: np.concatenate([p, np.ones((len(p), 1))], axis=1)
    if node_edges is None:
        node_edges = [[e for e in edges if n in e] for n in all_nodes]
    node_beads = [hom(np.concatenate([beads[e][n] for e in node_edges[n]], axis=0)) if node_edges[n] else None for n in all_nodes]
    adj_nodes = [[m for e in node_edges[n] for m in e if m != n] for n in all_nodes]
    adj_beads = [[hom(beads[e][m]) for e in node_edges[n] for m in e if m != n] for n in all_nodes]
    iter_all, mean_res, max_res = [], [], []
    edge_residuals = {}
    for iteration in range(max_iter):
        for cur in sorted_nodes:
            if not node_edges[cur]:
                continue
            node_pts = (new_affines[cur] @ node_beads[cur].T).T[:, :-1]
            adj_pts = np.concatenate([(new_affines[an] @ adj_beads[cur][k].T).T for k, an in enumerate(adj_nodes[cur])], axis=0)[:, :-1]
            if cur != ref:
                new_affines[cur] = estimate(node_pts, adj_pts) @ new_affines[cur]
        edge_residuals = {
            e: np.linalg.norm(transform_pts(beads[e][e[0]], new_affines[e[0]]) - transform_pts(beads[e][e[1]], new_affines[e[1]]), axis=1)
            for e in edges
        }
        mean_res.append(np.mean([np.mean(edge_residuals[e]) for e in edges]))
        max_res.append(np.max([np.max(edge_residuals[e]) for e in edges]))
        iter_all.append(edge_residuals)
        if iteration > 5:
            rel = np.max([np.abs((iter_all[-1][e] - iter_all[-2][e]) / max_res[-1] if max_res[-1] > 0 else 0) for e in edges])
            if rel < rel_tol:
                break
    return edge_residuals, mean_res, max_res, len(mean_res)


def _translation_sweeps_native(ndim, all_nodes, edges, beads, sorted_nodes, ref, max_iter, rel_tol, new_affines):
    """The same loop for the translation model in the library (``mvs_beads_translation_sweeps``, host code): a 64-view
    mosaic needs the full 500 sweeps on consistent data (the residuals decay geometrically, their relative change
    never drops below rel_tol), which costs seconds in numpy and milliseconds natively."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    ne, nb = len(edges), 2 ** ndim
    en = np.ascontiguousarray([[a, b] for a, b in edges], dtype=np.int32).reshape(ne, 2)
    ba = np.ascontiguousarray([beads[e][e[0]] for e in edges], dtype=np.float64).reshape(ne, nb, ndim)
    bb = np.ascontiguousarray([beads[e][e[1]] for e in edges], dtype=np.float64).reshape(ne, nb, ndim)
    order = np.ascontiguousarray(sorted_nodes, dtype=np.int32)
    t = np.ascontiguousarray(new_affines[:, :ndim, ndim], dtype=np.float64)
    res = np.zeros((ne, nb))
    mh, xh = np.zeros(max(max_iter, 1)), np.zeros(max(max_iter, 1))
    nit = C.c_int32(0)
    ptr = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    rc = lib.mvs_beads_translation_sweeps(ndim, len(all_nodes), ne, ptr(en, C.c_int32), ptr(ba, C.c_double), ptr(bb, C.c_double), nb,
                                          ptr(order, C.c_int32), int(ref), int(max_iter), float(rel_tol), ptr(t, C.c_double),
                                          ptr(res, C.c_double), ptr(mh, C.c_double), ptr(xh, C.c_double), C.byref(nit))
    if rc != 0:
        raise RuntimeError(f"mvs_beads_translation_sweeps failed (code {rc})")
    new_affines[:, :ndim, ndim] = t
    n = int(nit.value)
    return {e: res[k] for k, e in enumerate(edges)}, list(mh[:n]), list(xh[:n]), n


def _edge_means(nodes, edge_residuals):
    """{(node a, node b): mean bead residual} -- one reduction over the (edges, beads) array instead of a numpy call per edge."""
    keys = list(edge_residuals)
    if not keys:
        return {}
    rows = [np.asarray(edge_residuals[k], dtype=np.float64).ravel() for k in keys]
    if len({len(r) for r in rows}) == 1:
        means = np.stack(rows).mean(axis=1).tolist()
    else:
        means = [float(np.mean(r)) for r in rows]
    return {(nodes[a], nodes[b]): m for (a, b), m in zip(keys, means)}


_ESTIMATORS = {
    "translation": _estimate_translation,
    "rigid": lambda s, d: _umeyama(s, d, False),
    "similarity": lambda s, d: _umeyama(s, d, True),
    "affine": _estimate_affine,
}


def groupwise_resolution_global_optimization(g, reference_view=None, transform="translation", max_iter=None, rel_tol=None,
                                             abs_tol=None):
    """Iterative virtual-bead optimisation with removal of inconsistent edges (global_optimization.py:16-170, 173-511) for
    one connected component.  Returns ``(params {node: (n+1, n+1)}, info)`` with ``info["metrics"]`` = dict of per-iteration
    mean / max residuals and ``info["used_edges"]`` = the edges that survived."""
    ndim = g.ndim
    if not g.edges:
        return {n: param_utils.identity_transform(ndim) for n in g.nodes}, {"metrics": None, "used_edges": []}
    max_iter = 500 if max_iter is None else max_iter
    rel_tol = 1e-4 if rel_tol is None else rel_tol
    if abs_tol is None:   # voxel diagonal, max over tiles (global_optimization.py:104-121)
        abs_tol = max(_sum_like_numpy([v ** 2 for v in g.stack_props[n]["spacing"].values()]) ** 0.5 for n in g.nodes)
    if transform.lower() not in _ESTIMATORS:
        raise ValueError(f"Unknown transformation type in parameter resolution: {transform}")
    estimate = _ESTIMATORS[transform.lower()]
    ref_node = reference_view if (reference_view is not None and reference_view in g.nodes) else \
        get_node_with_maximal_edge_weight_sum_from_graph(g, "quality")

    from . import mv_graph

    nodes = list(g.nodes)
    idx = {n: i for i, n in enumerate(nodes)}
    ekey = lambda a, b: (a, b) if a <= b else (b, a)
    raw_beads = _beads_of(g)
    beads = {ekey(idx[a], idx[b]): {idx[a]: v[a], idx[b]: v[b]} for (a, b), v in raw_beads.items()}
    quality = {ekey(idx[a], idx[b]): e["quality"] for (a, b), e in g.edges.items()}
    all_nodes = list(range(len(nodes)))              # list(mapping.values()) (global_optimization.py:261)
    # The bead graph with networkx's orders: nodes in the registration graph's node order, edges added in its edge iteration
    # order with sorted end points (utils.py:48-52), deep-copied (order preserving), then relabelled IN PLACE to 0..n-1
    # (global_optimization.py:222-229) -- which reorders nodes and neighbour lists whenever labels and indices differ.
    bead_graph = mv_graph.Graph(nodes)
    for a, b in _nx_edge_order(nodes, list(g.edges)):
        bead_graph.add_edge(*sorted((a, b)))
    mv_graph.relabel_nodes_inplace(bead_graph, idx)
    edges_of = lambda: [ekey(a, b) for a, b in bead_graph.edges()]
    edges = edges_of()
    # Quirk restated literally: the reference compares the RELABELLED node index with the ORIGINAL label of the reference
    # view (`curr_node != ref_node`, global_optimization.py:334; ref_node is chosen before the relabelling, :126-133), so the
    # view that keeps its transform is the one whose index equals that label -- the reference view itself only when the
    # component's labels are 0..n-1 in order (a connected mosaic), and none at all when the label is >= n.
    ref = ref_node if (isinstance(ref_node, (int, np.integer)) and not isinstance(ref_node, bool) and 0 <= int(ref_node) < len(nodes)) else -1
    ref = int(ref)
    # order of the node sweeps: descending degree centrality, stable over the relabelled graph's node order
    node_order = list(bead_graph.nodes)
    degree0 = {n: bead_graph.degree(n) for n in node_order}
    sorted_nodes = sorted(node_order, key=lambda n: -degree0[n])
    new_affines = np.array([np.eye(ndim + 1) for _ in all_nodes])
    mean_res, max_res = [], []
    total_iterations = 0
    edge_residuals = {}
    while True:
        if not edges:
            break
        # a node's own edges in ITS neighbour order (G.edges(n), global_optimization.py:277)
        node_edges = [[ekey(n, m) for m in bead_graph.adj[n]] for n in all_nodes]
        if transform.lower() == "translation" and not _FORCE_NUMPY:
            edge_residuals, it_mean, it_max, n_it = _translation_sweeps_native(
                ndim, all_nodes, edges, beads, sorted_nodes, ref, max_iter, rel_tol, new_affines)
        else:
            edge_residuals, it_mean, it_max, n_it = _sweeps_numpy(
                estimate, all_nodes, edges, beads, sorted_nodes, ref, max_iter, rel_tol, new_affines, node_edges)
        mean_res += it_mean
        max_res += it_max
        total_iterations += n_it
        if len(edges) < 2:
            break
        edge_to_remove = None
        if not max_res[-1] < abs_tol:
            with np.errstate(divide="ignore", invalid="ignore"):
                score = [
                    (1 - float(quality[e])) ** 2 * np.sqrt(np.max(edge_residuals[e]))
                    * np.log10(np.max([bead_graph.degree(n) for n in e]))
                    for e in edges
                ]
            order = np.argsort(score)[::-1]
            for cand in order:   # first candidate whose removal keeps its two views connected
                e = edges[cand]
                rest = [f for f in edges if f != e]
                comp, stack = {e[0]}, [e[0]]
                while stack:
                    v = stack.pop()
                    for a, b in rest:
                        w = b if a == v else (a if b == v else None)
                        if w is not None and w not in comp:
                            comp.add(w)
                            stack.append(w)
                if e[1] in comp:
                    edge_to_remove = e
                    break
        if edge_to_remove is None:
            break
        bead_graph.remove_edge(*edge_to_remove)
        edges = edges_of()
    params = {nodes[n]: (new_affines[n] if total_iterations else np.eye(ndim + 1)) for n in all_nodes}
    info = {
        "metrics": {"mean_residual": list(mean_res), "max_residual": list(max_res), "iteration": list(range(len(mean_res)))},
        "used_edges": [tuple(sorted((nodes[a], nodes[b]))) for a, b in edges],
        "edge_residuals_final": _edge_means(nodes, edge_residuals),
    }
    return params, info


def groupwise_resolution_shortest_paths(g, reference_view=None):
    """Concatenate pairwise transforms along the quality-weighted shortest paths to the reference view
    (shortest_paths.py:9-99) for one connected component."""
    ndim = g.ndim
    if not g.edges:
        return {n: param_utils.identity_transform(ndim) for n in g.nodes}, {"metrics": None, "used_edges": [], "edge_residuals": {}}
    qmin = np.min([e["quality"] for e in g.edges.values()])
    weight = {k: 1.0 / ((e["quality"] - qmin) + 0.5) for k, e in g.edges.items()}
    ref = reference_view if (reference_view is not None and reference_view in g.nodes) else \
        get_node_with_maximal_edge_weight_sum_from_graph(g, "quality")
    # Dijkstra from the reference view
    dist, prev = {ref: 0.0}, {}
    heap, count = [(0.0, 0, ref)], 1
    done = set()
    while heap:
        d, _, v = heapq.heappop(heap)
        if v in done:
            continue
        done.add(v)
        for (a, b), w in weight.items():
            u = b if a == v else (a if b == v else None)
            if u is None or u in done:
                continue
            if d + w < dist.get(u, np.inf):
                dist[u], prev[u] = d + w, v
                heapq.heappush(heap, (d + w, count, u))
                count += 1
    used, params = set(), {}
    for n in g.nodes:
        path = [n]
        while path[-1] != ref:
            path.append(prev[path[-1]])
        path = path[::-1]
        p = np.eye(ndim + 1)
        for a, b in zip(path[:-1], path[1:]):
            used.add(tuple(sorted((a, b))))
            t = g.edges[(a, b)]["transform"] if (a, b) in g.edges else np.linalg.inv(g.edges[(b, a)]["transform"])
            p = t @ p                      # rebase_affine(edge transform, path so far)
        params[n] = np.linalg.inv(p)
    return params, {"metrics": None, "used_edges": list(used)}


# ---- linear two-pass resolver (param_resolution/linear_two_pass.py:216-544) ------------------------------------------
def _polar_rotation(linear):
    """Closest rotation of a linear map (polar decomposition through the SVD, determinant forced to +1)."""
    u, _, vt = np.linalg.svd(linear)
    if np.linalg.det(u @ vt) < 0:
        u = u.copy()
        u[:, -1] *= -1
    return u @ vt


def _rotvec(rmat, ndim):
    if ndim == 2:
        return np.array([np.arctan2(rmat[1, 0], rmat[0, 0])])
    from scipy.spatial.transform import Rotation

    return Rotation.from_matrix(rmat).as_rotvec()


def _rotmat(vec, ndim):
    if ndim == 2:
        c, s_ = np.cos(float(vec[0])), np.sin(float(vec[0]))
        return np.array([[c, -s_], [s_, c]])
    from scipy.spatial.transform import Rotation

    return Rotation.from_rotvec(vec).as_matrix()


def _difference_lsq(edge_list, values, weights, nodes, ref, dim, prior_lambda, lsqr_kwargs):
    """Weighted least squares of x_u - x_v = value over the edges with x_ref = 0 (a graph-Laplacian system, solved with
    scipy's LSQR on the sparse incidence matrix like the reference), optional Tikhonov rows sqrt(lambda) x = 0."""
    from scipy import sparse
    from scipy.sparse.linalg import lsqr

    free = [n for n in nodes if n != ref]
    col = {n: i * dim for i, n in enumerate(free)}
    npar = len(free) * dim
    rows, cols, data, rhs = [], [], [], []
    r = 0
    for (u, v), val, w in zip(edge_list, values, weights):
        sw = np.sqrt(w)
        for k in range(dim):
            rhs.append(sw * val[k])
            if u != ref:
                rows.append(r); cols.append(col[u] + k); data.append(sw)
            if v != ref:
                rows.append(r); cols.append(col[v] + k); data.append(-sw)
            r += 1
    if prior_lambda > 0 and npar > 0:
        sl = float(np.sqrt(prior_lambda))
        for n in free:
            for k in range(dim):
                rhs.append(0.0)
                rows.append(r); cols.append(col[n] + k); data.append(sl)
                r += 1
    out = {n: np.zeros(dim) for n in nodes}
    if r == 0 or npar == 0:
        return out
    mat = sparse.coo_matrix((data, (rows, cols)), shape=(r, npar)).tocsr()
    sol = lsqr(mat, np.asarray(rhs, dtype=np.float64), **lsqr_kwargs)[0]
    for n in free:
        out[n] = sol[col[n]:col[n] + dim]
    return out


def _kruskal_mst(nodes, edge_list, weights):
    """Minimum spanning forest, edges taken in ascending weight with ties in the given order (Kruskal with a stable sort,
    what networkx.minimum_spanning_tree does)."""
    parent = {n: n for n in nodes}

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    keep = set()
    for k in sorted(range(len(edge_list)), key=lambda i: weights[i]):
        a, b = find(edge_list[k][0]), find(edge_list[k][1])
        if a != b:
            parent[a] = b
            keep.add(tuple(sorted(edge_list[k])))
    return keep


def groupwise_resolution_linear_two_pass(g, reference_view=None, transform="rigid", residual_threshold=None, mad_k=2.0,
                                         keep_mst=True, weight_mode="quality_overlap", prior_lambda=0.0, **kwargs):
    """One global linearisation with two-pass outlier pruning (linear_two_pass.py:216-544).

    Every pairwise transform A_uv is reduced to a rotation R_uv (polar part, "rigid" only) and a displacement measured at
    the centre p of the pair's overlap box, d = A p - p ("translation") or A p - R p ("rigid").  Rotation vectors then
    translations are solved as weighted difference systems x_u - x_v = measurement with the reference view pinned
    (translations of "rigid": t_u - t_v = R_v d).  Pass 1 uses all edges; edges whose RMS bead residual exceeds
    ``residual_threshold`` (default: median + mad_k * MAD) are dropped unless a minimum spanning tree over the residuals
    needs them (``keep_mst``); pass 2 solves again on the kept edges.  Weights: quality * overlap | quality | overlap |
    uniform."""
    if "mode" in kwargs:
        transform = kwargs.pop("mode")
    if "prune_quantile" in kwargs:
        raise TypeError("prune_quantile is not supported; use residual_threshold or mad_k.")
    ndim = g.ndim
    if not g.edges:
        return {n: param_utils.identity_transform(ndim) for n in g.nodes}, {"metrics": None, "used_edges": []}
    if transform not in ("translation", "rigid"):
        raise ValueError(f"Unknown transform: {transform}")
    if ndim not in (2, 3):
        raise ValueError("Only 2D and 3D supported.")
    rigid = transform == "rigid"
    rdim = 1 if ndim == 2 else 3
    ref = reference_view if (reference_view is not None and reference_view in g.nodes) else \
        get_node_with_maximal_edge_weight_sum_from_graph(g, "quality")
    nodes = list(g.nodes)
    lsqr_kwargs = {k: v for k, v in kwargs.items() if k in {"damp", "atol", "btol", "conlim", "iter_lim", "show", "calc_var"}}

    def weight_of(e):
        q, o = float(np.mean(e.get("quality", 1.0))), float(np.mean(e.get("overlap", 1.0)))
        w = {"quality_overlap": q * o, "quality": q, "overlap": o, "uniform": 1.0}.get(weight_mode)
        if w is None:
            raise ValueError(f"Unknown weight_mode: {weight_mode}")
        return w if (np.isfinite(w) and w >= 0) else 0.0

    order = _nx_edge_order(nodes, list(g.edges))          # the order networkx iterates g.edges in
    keys, rots, trans, wts = [], [], [], []
    for key in order:
        e = g.edges[key]
        A = np.asarray(e["transform"], dtype=np.float64)
        lin, t = A[:ndim, :ndim], A[:ndim, ndim]
        center = np.zeros(ndim) if e.get("bbox") is None else np.mean(np.asarray(e["bbox"], dtype=np.float64)[:2], axis=0)
        if rigid:
            R = _polar_rotation(lin)
            rots.append(_rotvec(R, ndim))
            trans.append((lin @ center + t) - R @ center)
        else:
            rots.append(None)
            trans.append((lin @ center + t) - center)
        keys.append(key)
        wts.append(weight_of(e))

    def solve(idx):
        ek = [keys[i] for i in idx]
        w = [wts[i] for i in idx]
        if rigid:
            rv = _difference_lsq(ek, [rots[i] for i in idx], w, nodes, ref, rdim, prior_lambda, lsqr_kwargs)
            rhs = [_rotmat(rv[keys[i][1]], ndim) @ trans[i] for i in idx]
        else:
            rv = {n: np.zeros(rdim) for n in nodes}
            rhs = [trans[i] for i in idx]
        tv = _difference_lsq(ek, rhs, w, nodes, ref, ndim, prior_lambda, lsqr_kwargs)
        params = {}
        for n in nodes:
            M = np.eye(ndim + 1)
            if rigid:
                M[:ndim, :ndim] = _rotmat(rv[n], ndim)
            M[:ndim, ndim] = tv[n]
            params[n] = M
        return params

    all_idx = list(range(len(keys)))
    p1 = solve(all_idx)
    res_by_edge = compute_edge_residuals(g, p1)
    residuals = np.array([res_by_edge.get(tuple(sorted(k)), np.nan) for k in keys], dtype=np.float64)
    finite = residuals[np.isfinite(residuals)]
    if residual_threshold is not None:
        thr = float(residual_threshold)
    elif finite.size:
        med = float(np.median(finite))
        thr = med + float(mad_k) * float(np.median(np.abs(finite - med)))
    else:
        thr = np.inf
    r_keep = np.where(np.isfinite(residuals), residuals, np.inf)
    keep = r_keep <= thr
    forced = _kruskal_mst(nodes, keys, r_keep.tolist()) if keep_mst else set()
    metrics, final = [], []
    for i, k in enumerate(keys):
        kept = bool(keep[i]) or tuple(sorted(k)) in forced
        metrics.append({"u": k[0], "v": k[1], "weight": wts[i], "residual": float(residuals[i]), "kept_pass2": kept})
        if kept:
            final.append(i)
    if not final:
        final = all_idx
        for m in metrics:
            m["kept_pass2"] = True
    params = solve(final)
    return params, {"metrics": {"edges": metrics}, "used_edges": [tuple(sorted(keys[i])) for i in final]}



_METHODS = {
    "global_optimization": groupwise_resolution_global_optimization,
    "shortest_paths": groupwise_resolution_shortest_paths,
    "linear_two_pass": groupwise_resolution_linear_two_pass,
}


def resolve_translations_native(n_views, edges, results, spacings, reference_view=None, transform="translation", max_iter=None,
                                rel_tol=None, abs_tol=None):
    """``groupwise_resolution(g, "global_optimization", ...)`` for registration.register's common case in ONE library call
    (``mvs_resolve_translations``, host code): views 0 .. n_views - 1 in one connected component, every pairwise result a pure
    translation, the translation model, and sweeps that end below ``abs_tol`` (no edge is removed).  ``edges``: sorted view
    pairs in the order they would be added to the RegGraph; ``results``: their {"transform", "quality", "bbox"}; ``spacings``:
    (n_views, ndim).  Returns ``(params list, info)`` exactly as groupwise_resolution reports them, or None when the case is
    not covered -- the caller then builds the RegGraph and takes the Python form (tests/test_resolve_native.py compares)."""
    import ctypes as C

    from . import _lib

    ne = len(edges)
    if ne < 1 or str(transform).lower() != "translation" or n_views < 2:
        return None
    if reference_view is None and n_views == 2:
        reference_view = 0                                    # min(g.nodes) (param_resolution/__init__.py:97-98)
    if reference_view is not None:
        if isinstance(reference_view, bool) or not isinstance(reference_view, (int, np.integer)):
            return None
        if not 0 <= int(reference_view) < n_views:
            reference_view = None                             # not a node: the maximal-quality view is taken
    try:
        T = np.array([r["transform"] for r in results], dtype=np.float64)
        q = np.array([r["quality"] for r in results], dtype=np.float64)
        bb = np.array([r["bbox"] for r in results], dtype=np.float64)
        en = np.ascontiguousarray(edges, dtype=np.int32)
    except (TypeError, ValueError, KeyError):
        return None
    sp = np.ascontiguousarray(spacings, dtype=np.float64)
    nd = sp.shape[1] if sp.ndim == 2 else 0
    if T.shape != (ne, nd + 1, nd + 1) or bb.shape != (ne, 2, nd) or q.shape != (ne,) or en.shape != (ne, 2) or sp.shape != (n_views, nd):
        return None
    if np.any(T[:, :nd, :nd] != np.eye(nd)) or np.any(T[:, nd, :nd] != 0.0) or np.any(T[:, nd, nd] != 1.0):
        return None
    max_iter = 500 if max_iter is None else int(max_iter)
    rel_tol = 1e-4 if rel_tol is None else float(rel_tol)
    if max_iter < 1:
        return None
    pt = np.ascontiguousarray(T[:, :nd, nd])
    lo, hi = np.ascontiguousarray(bb[:, 0]), np.ascontiguousarray(bb[:, 1])
    t_out = np.zeros((n_views, nd))
    rms = np.zeros(ne)
    mh, xh = np.zeros(max_iter), np.zeros(max_iter)
    nit, ref = C.c_int32(0), C.c_int32(-1)
    ptr = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    rc = _lib.load().mvs_resolve_translations(
        nd, n_views, ne, ptr(en, C.c_int32), ptr(pt, C.c_double), ptr(q, C.c_double), ptr(lo, C.c_double), ptr(hi, C.c_double),
        ptr(sp, C.c_double), -1 if reference_view is None else int(reference_view), max_iter, rel_tol,
        -1.0 if abs_tol is None else float(abs_tol), ptr(t_out, C.c_double), ptr(rms, C.c_double), ptr(mh, C.c_double),
        ptr(xh, C.c_double), C.byref(nit), C.byref(ref))
    if rc != 0:
        return None
    n = int(nit.value)
    params = np.zeros((n_views, nd + 1, nd + 1))
    params[:, np.arange(nd + 1), np.arange(nd + 1)] = 1.0
    params[:, :nd, nd] = t_out
    keys = [(int(a), int(b)) for a, b in en.tolist()]
    info = {"metrics": [{"mean_residual": mh[:n].tolist(), "max_residual": xh[:n].tolist(), "iteration": list(range(n)), "icc": 0}],
            "edge_residuals": {0: dict(zip(keys, rms.tolist()))}, "used_edges": {0: sorted(keys)}}
    return list(params), info


def register_groupwise_resolution_method(name, resolver):
    """``resolver(RegGraph of one connected component, **kwargs) -> (params, info)`` (__init__.py:23-33)."""
    if not callable(resolver):
        raise TypeError("Resolver must be callable.")
    _METHODS[name] = resolver


def groupwise_resolution(g, method="global_optimization", **kwargs):
    """Run the method per connected component and collect parameters, residuals and used edges (__init__.py:44-150;
    a single time point -- ``registration.register`` loops over t)."""
    if callable(method):
        resolver = method
    elif method in _METHODS:
        resolver = _METHODS[method]
    else:
        raise ValueError(f"Unknown groupwise optimization method: {method}")
    if not g.edges:
        raise NotEnoughOverlapError("Not enough overlap between views for stitching.")
    if "reference_view" not in kwargs and len(g.nodes) == 2:
        kwargs["reference_view"] = min(g.nodes)
    params, metrics, used = {}, [], set()
    _beads_of(g)          # (memoised: a component that keeps every edge inherits them, compute_edge_residuals below reuses them)
    for icc, cc in enumerate(g.connected_components()):
        sub = g.subgraph(cc)
        if not sub.edges:
            cc_params, cc_info = {n: param_utils.identity_transform(g.ndim) for n in cc}, None
        else:
            cc_params, cc_info = resolver(sub, **kwargs)
        params.update({n: np.asarray(cc_params[n], dtype=np.float64) for n in cc})
        if cc_info is not None:
            if cc_info.get("metrics") is not None:
                metrics.append(dict(cc_info["metrics"], icc=icc))
            used.update(tuple(sorted(e)) for e in cc_info.get("used_edges") or [])
    info = {"metrics": metrics or None, "edge_residuals": {0: compute_edge_residuals(g, params)}, "used_edges": {0: sorted(used)}}
    return params, info
