"""Product (multiview_stitcher_amd.param_resolution / mv_graph) against the oracle restatement of the reference's
groupwise resolution and graph pruning (oracle/resolve_oracle.py <- param_resolution/global_optimization.py:16-511,
param_resolution/utils.py:42-101, mv_graph.py:664-881, 1148-1196) on the north-star graph (4 x 4 x 4 tiles), on noisy
graphs with outlier edges (the edge-removal loop), on disconnected graphs and for every transform model.  CPU only."""
import numpy as np
import pytest

nx = pytest.importorskip("networkx")

from multiview_stitcher_amd import mv_graph, param_utils
from multiview_stitcher_amd import param_resolution as pr
from oracle import resolve_oracle as oracle


def _grid_stack_props(grid, tile, overlap_frac=0.2, seed=1, jitter=3):
    grid, tile = np.asarray(grid), np.asarray(tile)
    nd = len(grid)
    sd = ["z", "y", "x"][-nd:]
    step = tile - np.round(tile * overlap_frac).astype(int)
    rng = np.random.default_rng(seed)
    sps, jit = [], []
    for idx in np.ndindex(*grid):
        j = rng.integers(-jitter, jitter + 1, nd) * (1 if any(idx) else 0)
        sps.append({"origin": dict(zip(sd, (np.array(idx) * step).astype(float))), "spacing": dict(zip(sd, [1.0] * nd)),
                    "shape": dict(zip(sd, [int(v) for v in tile])), "transform": np.eye(nd + 1)})
        jit.append(j)
    return sps, np.array(jit, dtype=float)


def _to_nx(g):
    """The same graph (node order, edge insertion order, attributes) as an nx.Graph."""
    h = nx.Graph()
    for n in g.nodes:
        h.add_node(n, **g.node_attrs[n])
    for a, b, d in g.edges(data=True):
        h.add_edge(a, b, **dict(d))
    # networkx keeps per-node neighbour order = order in which the node's edges were added; replay the product's order
    h2 = nx.Graph()
    for n in g.nodes:
        h2.add_node(n, **g.node_attrs[n])
    done = set()
    order = []
    for n in g.nodes:
        for m in g.adj[n]:
            if (m, n) not in done and (n, m) not in done:
                order.append((n, m))
                done.add((n, m))
    for a, b in order:
        h2.add_edge(a, b, **dict(g.adj[a][b]))
    assert [list(h2.adj[n]) for n in h2.nodes] == [list(g.adj[n]) for n in g.nodes]
    return h2


def _norm_edges(edges):
    return sorted(tuple(sorted(e)) for e in edges)


@pytest.mark.parametrize("method", ["alternating_pattern", "shortest_paths_overlap_weighted", "otsu_threshold_on_overlap",
                                    "keep_axis_aligned", None])
@pytest.mark.parametrize("grid,tile", [((4, 4, 4), (512, 512, 512)), ((3, 5), (300, 200)), ((2, 3, 2), (64, 100, 80))])
def test_pruning_matches_oracle(method, grid, tile):
    sps, _ = _grid_stack_props(grid, tile)
    g = mv_graph.build_view_adjacency_graph(sps)
    h = _to_nx(g)
    assert g.edges() == list(h.edges())
    got = mv_graph.prune_view_adjacency_graph(g, method)
    want = oracle.prune_view_adjacency_graph(h, method)
    assert _norm_edges(got.edges()) == _norm_edges(want.edges())
    assert sorted(got.nodes) == sorted(want.nodes)
    if method == "alternating_pattern":
        assert got.edges() == list(want.edges())          # same iteration order: the registration work list
        if tuple(grid) == (4, 4, 4):
            assert len(got.edges()) == 144                # the bench's pairs per step


def _reg_graphs(sps, edges, jit, noise, seed, quality=None, outliers=()):
    """Product RegGraph and the oracle's nx.Graph with the same pairwise results: transform of edge (a, b) = translation
    jit[a] - jit[b] + noise (the pairwise result that P_v = translate(jit_v) resolves), bbox = the overlap box."""
    rng = np.random.default_rng(seed)
    nd = len(sps[0]["origin"])
    sd = ["z", "y", "x"][-nd:]
    nodes = list(range(len(sps)))
    g = pr.RegGraph(nodes, {v: {"spacing": sps[v]["spacing"]} for v in nodes})
    h = nx.Graph()
    for v in nodes:
        h.add_node(v, stack_props={"spacing": sps[v]["spacing"]})
    for k, (a, b) in enumerate(edges):
        lo = np.maximum([sps[a]["origin"][d] for d in sd], [sps[b]["origin"][d] for d in sd])
        hi = np.minimum([sps[a]["origin"][d] + sps[a]["shape"][d] - 1 for d in sd], [sps[b]["origin"][d] + sps[b]["shape"][d] - 1 for d in sd])
        t = jit[a] - jit[b] + rng.normal(0, noise, nd)
        if k in outliers:
            t = t + 40.0
        q = float(rng.uniform(0.5, 1.0)) if quality is None else quality
        if k in outliers:
            q = 0.2
        T = param_utils.affine_from_translation(t)
        g.add_edge(a, b, T, quality=q, overlap=1.0, bbox=[lo, hi])
        h.add_edge(a, b, transform=T, quality=q, overlap=1.0, bbox=np.array([lo, hi]))
    return g, h


def _assert_same_resolution(g, h, **kw):
    got_p, got_info = pr.groupwise_resolution(g, method="global_optimization", **kw)
    want_p, want_info = oracle.groupwise_resolution(h, **kw)
    for v in want_p:
        np.testing.assert_allclose(got_p[v], want_p[v], rtol=0, atol=1e-9)
    assert sorted(got_info["used_edges"][0]) == sorted(want_info["used_edges"])
    for e, r in want_info["edge_residuals"].items():
        assert got_info["edge_residuals"][0][e] == pytest.approx(r, abs=1e-9)
    return got_p, want_info


def test_north_star_resolution_matches_oracle():
    """The graph register() resolves on the bench mosaic: 64 views, the 144 pairs alternating_pattern keeps, pairwise
    translations = hidden jitter differences with 0.2 px of noise, qualities in [0.5, 1]."""
    sps, jit = _grid_stack_props((4, 4, 4), (512, 512, 512))
    edges = mv_graph.prune_view_adjacency_graph(mv_graph.build_view_adjacency_graph(sps), "alternating_pattern").edges()
    g, h = _reg_graphs(sps, edges, jit, noise=0.2, seed=3)
    params, info = _assert_same_resolution(g, h)
    assert len(info["used_edges"]) == 144
    ref = np.array([params[v][:3, 3] for v in range(64)])
    assert np.abs((ref - ref[0]) - (jit - jit[0])).max() < 1.0


def test_consistent_north_star_graph_runs_all_sweeps_like_oracle():
    """Noise-free pairwise results: the relative change never falls below rel_tol while the residual decays, so the
    reference runs its max_iter sweeps (here a reduced max_iter keeps the Python oracle fast)."""
    sps, jit = _grid_stack_props((4, 4, 4), (512, 512, 512), seed=5)
    edges = mv_graph.prune_view_adjacency_graph(mv_graph.build_view_adjacency_graph(sps), "alternating_pattern").edges()
    g, h = _reg_graphs(sps, edges, jit, noise=0.0, seed=0, quality=1.0)
    _assert_same_resolution(g, h, max_iter=60)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_outlier_edges_are_removed_like_oracle(seed):
    sps, jit = _grid_stack_props((4, 4), (256, 256), seed=seed)
    edges = mv_graph.build_view_adjacency_graph(sps).edges()
    edges = [e for e in edges if abs(e[0] - e[1]) in (1, 4)]
    g, h = _reg_graphs(sps, edges, jit, noise=0.05, seed=seed, outliers=(3, 11))
    _, info = _assert_same_resolution(g, h)
    assert len(info["used_edges"]) < len(edges)


def test_disconnected_components_and_two_views_match_oracle():
    sps, jit = _grid_stack_props((2, 4), (128, 128), seed=9)
    edges = [(0, 1), (1, 5), (4, 5), (0, 4), (2, 3), (3, 7), (6, 7)]       # two components
    g, h = _reg_graphs(sps, edges, jit, noise=0.1, seed=4)
    _assert_same_resolution(g, h)
    g2, h2 = _reg_graphs(sps[:2], [(0, 1)], jit[:2], noise=0.3, seed=1)
    params, _ = _assert_same_resolution(g2, h2)
    np.testing.assert_array_equal(params[0], np.eye(3))                   # [fixed, moving] convention (__init__.py:68-71)


@pytest.mark.parametrize("transform", ["translation", "rigid", "similarity", "affine"])
@pytest.mark.parametrize("nd", [2, 3])
def test_transform_models_match_oracle(transform, nd):
    grid, tile = ((3, 3), (200, 220)) if nd == 2 else ((2, 2, 2), (90, 100, 110))
    sps, jit = _grid_stack_props(grid, tile, seed=2)
    g0 = mv_graph.build_view_adjacency_graph(sps)
    edges = mv_graph.prune_view_adjacency_graph(g0, "keep_axis_aligned").edges()
    g, h = _reg_graphs(sps, edges, jit, noise=0.1, seed=6)
    # give the pairwise transforms a small rotation / scale so that the non-translation models have something to fit
    rng = np.random.default_rng(7)
    for (a, b) in list(g.edges):
        th = rng.normal(0, 0.01)
        L = np.eye(nd + 1)
        c, s = np.cos(th), np.sin(th)
        L[nd - 2:nd, nd - 2:nd] = np.array([[c, -s], [s, c]]) * (1 + rng.normal(0, 0.002))
        T = L @ g.edges[(a, b)]["transform"]
        g.edges[(a, b)]["transform"] = T
        h.edges[(a, b)]["transform"] = T
    _assert_same_resolution(g, h, transform=transform, max_iter=40)


def test_short_sums_equal_numpy_bit_for_bit():
    """param_resolution._sum_like_numpy replaces np.sum on the short lists of the resolution (a node's edge weights, the squared
    spacings): below 8 elements numpy's reduction is the plain left-to-right loop, which the helper restates; from 8 on it hands
    over to numpy (pairwise / unrolled summation).  Bitwise equality on random lists of every length, cancellation included."""
    from multiview_stitcher_amd.param_resolution import _sum_like_numpy

    rng = np.random.default_rng(0)
    for n in range(0, 14):
        for _ in range(200):
            w = list((rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)).astype(np.float64))
            a, b = _sum_like_numpy(w), float(np.sum(w))
            assert a == b or (np.isnan(a) and np.isnan(b)), (n, w)
    assert _sum_like_numpy([np.float64(0.1), 0.2, np.float32(0.3)]) == float(np.sum([np.float64(0.1), 0.2, np.float32(0.3)]))
