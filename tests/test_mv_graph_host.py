"""CPU tests of the view adjacency graph and its pruning: graph algorithms against networkx itself,
Otsu against scikit-image 0.18.3 vectors, overlap volumes against closed forms and the reference's own assertions
(T/test_mv_graph.py:16-104)."""
import os

import numpy as np
import pytest

from multiview_stitcher_amd import mv_graph, param_utils

GOLD = os.path.join(os.path.dirname(__file__), "golden", "skimage018_transforms.npz")


def _random_graphs(n_graphs=25, seed=0):
    rng = np.random.default_rng(seed)
    for _ in range(n_graphs):
        n = int(rng.integers(3, 14))
        nodes = [int(v) for v in rng.permutation(n)]
        pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
        ins = [pairs[i] if rng.random() < 0.5 else pairs[i][::-1] for i in rng.permutation(len(pairs))[: int(rng.integers(2, len(pairs) + 1))]]
        yield nodes, ins, rng


def test_graph_orders_and_algorithms_match_networkx():
    nx = pytest.importorskip("networkx")
    for nodes, ins, rng in _random_graphs():
        g, h = mv_graph.Graph(nodes), nx.Graph()
        h.add_nodes_from(nodes)
        for a, b in ins:
            w = float(rng.integers(1, 5))
            g.add_edge(a, b, overlap=w)
            h.add_edge(a, b, overlap=w)
        assert g.edges() == list(h.edges())
        assert g.connected_components() == list(nx.connected_components(h))
        want = nx.edge_betweenness_centrality(h)
        got = mv_graph.edge_betweenness_centrality(g)
        assert list(got) == list(want) and all(got[e] == want[e] for e in want)       # bit for bit: the values feed <= tests
        assert mv_graph._edge_betweenness_centrality_python(g) == got                  # library (host C++) == Python form
        assert mv_graph.greedy_color(g) == nx.coloring.greedy_color(h)
        # removal keeps the orders; copy keeps them too
        a, b = ins[0]
        import copy
        g2, h2 = g.copy(), copy.deepcopy(h)       # the reference prunes a deepcopy (mv_graph.py:691), which keeps every order
        g2.remove_edge(a, b)
        h2.remove_edge(a, b)
        assert g2.edges() == list(h2.edges()) and g.edges() == list(h.edges())
        assert [list(g2.adj[n]) for n in g2.nodes] == [list(h2.adj[n]) for n in h2.nodes]


def test_alternating_pattern_matches_networkx_restatement():
    """prune_graph_to_alternating_colors against the same procedure spelled with networkx calls (mv_graph.py:664-741)."""
    nx = pytest.importorskip("networkx")
    import copy

    for nodes, ins, rng in _random_graphs(20, seed=3):
        g, h = mv_graph.Graph(nodes), nx.Graph()
        h.add_nodes_from(nodes)
        for a, b in ins:
            w = float(rng.choice([10.0, 10.0, 3.0, 1.0]))
            g.add_edge(a, b, overlap=w)
            h.add_edge(a, b, overlap=w)
        hp = copy.deepcopy(h)
        cent = nx.edge_betweenness_centrality(h)
        cmax, cmin = max(cent.values()), min(cent.values())
        edges = list(hp.edges(data=True))
        mo = min(e[2]["overlap"] for e in edges)
        if cmax > cmin:
            cent = {e: (cent[e] - cmin) / (cmax - cmin) * 0.5 * mo for e in cent}
        vals = {tuple(e[:2]): cent[tuple(e[:2])] + e[2]["overlap"] for e in edges}
        levels = sorted(np.unique(list(vals.values())))
        k, failed = 0, False
        while True:
            colors = nx.coloring.greedy_color(hp)
            if len(set(colors.values())) <= 2:
                break
            if k >= len(levels):
                failed = True
                break
            hp.remove_edges_from([(a, b) for a, b, _ in hp.edges(data=True) if vals[(a, b)] <= levels[k] and min(len(hp.edges(n)) for n in (a, b)) > 1])
            k += 1
        if failed:
            with pytest.raises(IndexError):
                mv_graph.prune_graph_to_alternating_colors(g)
            continue
        gp, col = mv_graph.prune_graph_to_alternating_colors(g)
        assert gp.edges() == list(hp.edges()) and col == colors


def _sp(origin, shape, spacing, transform=None):
    d = "zyx"[-len(origin):]
    sp = {"origin": dict(zip(d, map(float, origin))), "shape": dict(zip(d, map(int, shape))), "spacing": dict(zip(d, map(float, spacing)))}
    if transform is not None:
        sp["transform"] = np.asarray(transform, dtype=float)
    return sp


def _grid(ndim, tiles, tile=15, overlap=3, spacing=None):
    spacing = np.ones(ndim) if spacing is None else np.asarray(spacing, float)
    sps = []
    for idx in np.ndindex(*tiles):
        o = np.asarray(idx) * (tile - overlap) * spacing
        sps.append(_sp(np.zeros(ndim), [tile] * ndim, spacing, param_utils.affine_from_translation(o)))
    return sps


@pytest.mark.parametrize("ndim,overlap", [(n, o) for n in (2, 3) for o in (0, 1, 3)])
def test_overlap_counts_like_reference_test(ndim, overlap):
    """T/test_mv_graph.py:16-104: number of distinct overlap volumes on a 3 x 2 (x 2) grid with anisotropic spacing."""
    sps = _grid(ndim, (2, 2, 3)[-ndim:], overlap=overlap, spacing=(2, 0.5, 0.5)[-ndim:])
    areas = np.array([[mv_graph.get_overlap_between_pair_of_stack_props(a, b)[0] for b in sps] for a in sps])
    uniq = np.unique(np.round(areas, 6))
    if overlap == 0:
        assert len(uniq) == 2 and areas[0][1] == -1
    else:
        assert len(uniq) == ({1: 2, 3: 4} if ndim == 2 else {1: 2, 3: 5})[overlap]
        assert areas.min() == -1 and areas.max() > 0


def test_closed_form_equals_qhull_sequence_and_rotated_squares(monkeypatch):
    rng = np.random.default_rng(2)
    for ndim in (2, 3):
        for _ in range(10):
            a = _sp(rng.normal(0, 5, ndim), rng.integers(5, 30, ndim), rng.uniform(0.3, 2, ndim), param_utils.affine_from_translation(rng.normal(0, 3, ndim)))
            b = _sp(rng.normal(0, 5, ndim), rng.integers(5, 30, ndim), rng.uniform(0.3, 2, ndim), param_utils.affine_from_translation(rng.normal(0, 3, ndim)))
            fast = mv_graph.get_overlap_between_pair_of_stack_props(a, b)[0]
            monkeypatch.setattr(mv_graph, "_axis_aligned_box", lambda sp: None)     # force linprog + HalfspaceIntersection + ConvexHull
            slow = mv_graph.get_overlap_between_pair_of_stack_props(a, b)[0]
            monkeypatch.undo()
            assert (fast == -1 and slow == -1) or fast == pytest.approx(slow, rel=1e-9)
    # two 10 x 10 squares about the same centre, one turned by 45 degrees: regular octagon, area 8 (sqrt2 - 1) r^2, r = 5 ... in
    # closed form 2 (sqrt2 - 1) s^2
    c, s_ = np.cos(np.pi / 4), np.sin(np.pi / 4)
    rot = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
    ctr = param_utils.affine_from_translation([5.0, 5.0])
    a = _sp([0, 0], [11, 11], [1, 1])
    b = _sp([0, 0], [11, 11], [1, 1], ctr @ rot @ np.linalg.inv(ctr))
    vol, hs = mv_graph.get_overlap_between_pair_of_stack_props(a, b)
    assert vol == pytest.approx(2 * (np.sqrt(2) - 1) * 100, rel=1e-9) and hs is not None


def test_grid_graph_and_pruning_methods():
    sps = _grid(2, (3, 3), tile=64, overlap=12)
    g = mv_graph.build_view_adjacency_graph(sps)
    e = {tuple(sorted(x)) for x in g.edges()}
    face = {(a, b) for a in range(9) for b in range(9) if a < b and ((b - a == 1 and a % 3 != 2) or b - a == 3)}
    diag = {(a, b) for a in range(9) for b in range(9) if a < b and a // 3 + 1 == b // 3 and abs(a % 3 - b % 3) == 1}
    assert e == face | diag
    assert g.adj[0][1]["overlap"] == pytest.approx(11 * 63) and g.adj[0][4]["overlap"] == pytest.approx(11 * 11)
    for method in ("alternating_pattern", "otsu_threshold_on_overlap", "keep_axis_aligned"):
        p = mv_graph.prune_view_adjacency_graph(g, method)
        assert {tuple(sorted(x)) for x in p.edges()} == face, method
    sp_tree = mv_graph.prune_view_adjacency_graph(g, "shortest_paths_overlap_weighted")
    assert len(sp_tree.edges()) == 8 and len(sp_tree.connected_components()) == 1        # a spanning tree
    assert {tuple(sorted(x)) for x in sp_tree.edges()} <= face
    assert mv_graph.prune_view_adjacency_graph(g, None) is g
    with pytest.raises(ValueError):
        mv_graph.prune_view_adjacency_graph(g, "nope")
    with pytest.raises(mv_graph.NotEnoughOverlapError):
        mv_graph.prune_view_adjacency_graph(mv_graph.build_view_adjacency_graph(_grid(2, (1, 2), overlap=0)), "alternating_pattern")
    # overlap_tolerance makes touching tiles neighbours; explicit pairs restrict the candidates
    g0 = mv_graph.build_view_adjacency_graph(_grid(2, (1, 2), tile=16, overlap=1), overlap_tolerance={"y": 0.0, "x": 2.0})
    assert g0.edges() == [(0, 1)]
    g1 = mv_graph.build_view_adjacency_graph(sps, pairs=[(4, 0), (0, 8)])
    assert g1.edges() == [(0, 4)] or g1.edges() == [(4, 0)]


def test_otsu_matches_skimage_golden():
    z = np.load(GOLD)
    for k in range(int(z["n_otsu"])):
        assert mv_graph.threshold_otsu(z[f"otsu{k}_vals"]) == pytest.approx(float(z[f"otsu{k}_thr"]), rel=1e-12)


def test_halfspace_equations_of_translated_views_equal_the_face_by_face_form_bit_for_bit():
    """mv_graph.get_halfspace_equations_from_stack_props: the one-shot array form taken for pure translations against the reference's
    loop over faces (np.cross / np.linalg.norm / np.dot per face) -- normals with their signed zeros and offsets, byte for byte; an
    offset that is exactly zero may differ in the sign of its zero (a face through the origin: nothing downstream can see it)."""
    import numpy as np

    from multiview_stitcher_amd import mv_graph

    rng = np.random.default_rng(3)
    checked = 0
    for trial in range(400):
        spacing = rng.choice([1.0, 2.0, 0.5, 0.23, 1.7, 3.0], size=3)
        shape = rng.integers(2, 700, size=3)
        origin = rng.choice([0.0, 410.0, -820.5, 1230.25, 0.1, 1e-3, 12345.678], size=3) * rng.choice([1.0, 1.0, -1.0, 0.37], size=3)
        t = np.eye(4)
        kind = trial % 4
        if kind == 1:
            t[:3, 3] = rng.integers(-5, 6, size=3)
        elif kind == 2:
            t[:3, 3] = rng.normal(size=3) * 100
        sp = {"origin": dict(zip("zyx", origin)), "spacing": dict(zip("zyx", spacing)), "shape": dict(zip("zyx", shape))}
        if kind != 3:
            sp["transform"] = t
        want = mv_graph._halfspace_equations_generic(sp)
        got = mv_graph.get_halfspace_equations_from_stack_props(sp)
        assert got.shape == want.shape == (6, 4)
        assert got[:, :3].tobytes() == want[:, :3].tobytes()
        nz = want[:, 3] != 0
        assert got[nz, 3].tobytes() == want[nz, 3].tobytes() and np.all(got[~nz, 3] == 0)
        checked += 1
    assert checked == 400
    # a rotated view keeps the loop
    r = np.eye(4)
    r[:3, :3] = [[0, 1, 0], [-1, 0, 0], [0, 0, 1]]
    sp = {"origin": dict(zip("zyx", [1.0, 2.0, 3.0])), "spacing": dict(zip("zyx", [1.0, 1.0, 1.0])), "shape": dict(zip("zyx", [5, 6, 7])), "transform": r}
    assert mv_graph.get_halfspace_equations_from_stack_props(sp).tobytes() == mv_graph._halfspace_equations_generic(sp).tobytes()
