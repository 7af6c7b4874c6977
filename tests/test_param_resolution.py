"""CPU tests of the groupwise resolution restatement.  The reference's package cannot be imported here
(xarray / pandas / dask absent), so the restatement is pinned by: skimage-0.18.3 golden vectors for the Umeyama fit,
networkx itself for the iteration orders the tie breaks depend on, closed-form answers on consistent graphs, and the
properties the reference's own tests assert (T/test_param_resolution.py:329-352, 360-406, 417-464)."""
import os

import numpy as np
import pytest

from multiview_stitcher_amd import param_resolution as pr
from multiview_stitcher_amd import param_utils

GOLD = os.path.join(os.path.dirname(__file__), "golden", "skimage018_transforms.npz")


def test_umeyama_matches_skimage_golden():
    z = np.load(GOLD)
    for k in range(int(z["n_cases"])):
        src, dst = z[f"c{k}_src"], z[f"c{k}_dst"]
        np.testing.assert_allclose(pr._umeyama(src, dst, False), z[f"c{k}_rigid"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(pr._umeyama(src, dst, True), z[f"c{k}_similarity"], rtol=1e-12, atol=1e-12)


def test_edge_and_component_order_match_networkx():
    nx = pytest.importorskip("networkx")
    rng = np.random.default_rng(0)
    for _ in range(30):
        n = int(rng.integers(3, 12))
        nodes = list(rng.permutation(n))
        pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
        ins = [pairs[i] for i in rng.permutation(len(pairs))[: int(rng.integers(1, len(pairs) + 1))]]
        g = nx.Graph()
        g.add_nodes_from(nodes)
        g.add_edges_from(ins)
        want = [tuple(sorted(e)) for e in g.edges]
        assert pr._nx_edge_order(nodes, ins) == want
        rg = pr.RegGraph(nodes)
        for a, b in ins:
            rg.add_edge(a, b, np.eye(3), bbox=[[0, 0], [1, 1]])
        assert rg.connected_components() == list(nx.connected_components(g))


def _grid_graph(nx_, ny_, ndim=2, noise=0.0, seed=0, quality=1.0):
    """Views on a grid with hidden true offsets tau; edge (i, j) transform = translate(tau_i - tau_j) (+ noise), i.e. the
    pairwise result that P = translate(tau) resolves exactly."""
    rng = np.random.default_rng(seed)
    n = nx_ * ny_
    tau = rng.normal(0, 3, (n, ndim))
    g = pr.RegGraph(range(n), {v: {"spacing": dict(zip("zyx"[-ndim:], [1.0] * ndim))} for v in range(n)})
    for a in range(n):
        ya, xa = divmod(a, nx_)
        for b in [a + 1 if xa + 1 < nx_ else None, a + nx_ if ya + 1 < ny_ else None]:
            if b is None:
                continue
            t = tau[a] - tau[b] + (rng.normal(0, noise, ndim) if noise else 0)
            lo = rng.normal(0, 50, ndim)
            g.add_edge(a, b, param_utils.affine_from_translation(t), quality=quality, bbox=[lo, lo + rng.uniform(20, 60, ndim)])
    return g, tau


@pytest.mark.parametrize("transform", ["translation", "rigid", "similarity", "affine"])
@pytest.mark.parametrize("ndim", [2, 3])
def test_consistent_graph_is_resolved_exactly(transform, ndim):
    g, tau = _grid_graph(3, 3, ndim)
    params, info = pr.groupwise_resolution(g, "global_optimization", transform=transform, reference_view=0)
    for v in range(9):
        want = param_utils.affine_from_translation(tau[v] - tau[0])
        # the sweeps stop at a relative change of 1e-4 (rel_tol): models with a linear part keep a ~1e-4 residue there, which
        # the lever arm of the beads (|x| ~ 100) turns into a few 1e-2 of translation
        np.testing.assert_allclose(params[v][:ndim, :ndim], want[:ndim, :ndim], atol=1e-3)
        np.testing.assert_allclose(params[v][:ndim, ndim], want[:ndim, ndim], atol=1e-2 if transform == "translation" else 0.15)
        if transform == "translation":      # T/test_param_resolution.py:349: the linear part stays the identity
            np.testing.assert_array_equal(params[v][:ndim, :ndim], np.eye(ndim))
    assert sorted(info["used_edges"][0]) == sorted(g.edges)
    assert max(info["edge_residuals"][0].values()) < 0.05
    m = info["metrics"][0]
    assert m["max_residual"][-1] <= m["max_residual"][0] and len(m["iteration"]) == len(m["mean_residual"])


def test_noisy_graph_close_to_least_squares():
    from multiview_stitcher_amd import registration

    g, tau = _grid_graph(4, 3, 2, noise=0.2, seed=3)
    params, info = pr.groupwise_resolution(g, "global_optimization", reference_view=0, abs_tol=100.0)
    edges = list(g.edges)
    ls = registration.resolve_translations(12, edges, [{"transform": g.edges[e]["transform"]} for e in edges])
    got = np.array([params[v][:2, 2] for v in range(12)])
    want = np.array([p[:2, 2] for p in ls])
    # the bead iteration is a Jacobi / Gauss-Seidel sweep of the same quadratic problem: same fixed point up to rel_tol
    np.testing.assert_allclose(got, want, atol=0.05)
    assert all(r > 0 for r in info["edge_residuals"][0].values())      # T/test_param_resolution.py:406


@pytest.mark.parametrize("method", ["global_optimization", "shortest_paths"])
def test_bad_edge_is_not_used(method):
    """T/test_param_resolution.py:417-464: one grossly wrong, low-quality edge on a cycle must not survive."""
    g, tau = _grid_graph(3, 3, 2, noise=0.05, seed=1)
    bad = (4, 5)
    g.edges[bad]["quality"] = 0.01
    g.edges[bad]["transform"] = param_utils.affine_from_translation([100.0, 100.0])
    params, info = pr.groupwise_resolution(g, method, reference_view=0)
    assert bad not in info["used_edges"][0]
    got = np.array([params[v][:2, 2] for v in range(9)])
    np.testing.assert_allclose(got, tau - tau[0], atol=0.5)
    assert info["edge_residuals"][0][bad] > 50


def test_shortest_paths_residuals():
    """T/test_param_resolution.py:360-404: ~0 on used edges, > 0 on the others."""
    g, _ = _grid_graph(3, 3, 2, noise=0.3, seed=5)
    _, info = pr.groupwise_resolution(g, "shortest_paths", reference_view=0)
    used = set(info["used_edges"][0])
    assert len(used) == 8
    for e, r in info["edge_residuals"][0].items():
        assert (r < 1e-6) if e in used else (r > 1e-5)


def test_components_two_views_and_errors():
    g, tau = _grid_graph(2, 1, 2)
    g.nodes += [7, 8, 9]
    g.add_edge(8, 7, param_utils.affine_from_translation([1.0, 2.0]), bbox=[[0, 0], [5, 5]])     # stored as (7, 8), inverted
    g.stack_props.update({v: {"spacing": {"y": 1.0, "x": 1.0}} for v in (7, 8, 9)})
    params, info = pr.groupwise_resolution(g, "global_optimization")
    assert set(params) == {0, 1, 7, 8, 9}
    np.testing.assert_array_equal(params[9], np.eye(3))            # isolated view
    np.testing.assert_array_equal(params[0], np.eye(3))            # default reference of a 2-view graph is min(nodes) ...
    np.testing.assert_allclose(params[1][:2, 2], tau[1] - tau[0], atol=1e-9)
    np.testing.assert_allclose(params[8][:2, 2] - params[7][:2, 2], [1.0, 2.0], atol=1e-9)
    with pytest.raises(pr.NotEnoughOverlapError):
        pr.groupwise_resolution(pr.RegGraph([0, 1]), "global_optimization")
    with pytest.raises(ValueError):
        pr.groupwise_resolution(g, "no_such_method")
    with pytest.raises(ValueError):
        pr.groupwise_resolution(g, "global_optimization", transform="projective")
    pr.register_groupwise_resolution_method("ident", lambda sub, **kw: ({n: np.eye(3) for n in sub.nodes}, {"metrics": None, "used_edges": []}))
    p2, _ = pr.groupwise_resolution(g, "ident")
    assert all(np.array_equal(v, np.eye(3)) for v in p2.values())


@pytest.mark.parametrize("ndim", [2, 3])
def test_native_translation_sweeps_equal_numpy_sweeps(ndim, monkeypatch):
    """mvs_beads_translation_sweeps (host C++ in libmvs_hip.so) against the generic numpy loop on a graph with a bad edge
    (several outer iterations, edge removal): same parameters, same per-sweep history, same surviving edges."""
    g, _ = _grid_graph(4, 3, ndim, noise=0.3, seed=11, quality=0.8)
    g.edges[(5, 6)]["transform"] = param_utils.affine_from_translation([40.0] * ndim)
    g.edges[(5, 6)]["quality"] = 0.1
    p_nat, i_nat = pr.groupwise_resolution(g, "global_optimization", reference_view=2)
    monkeypatch.setattr(pr, "_FORCE_NUMPY", True)
    p_np, i_np = pr.groupwise_resolution(g, "global_optimization", reference_view=2)
    assert i_nat["used_edges"] == i_np["used_edges"] and (5, 6) not in i_nat["used_edges"][0]
    for v in p_np:
        np.testing.assert_allclose(p_nat[v], p_np[v], rtol=0, atol=1e-10)
    m_nat, m_np = i_nat["metrics"][0], i_np["metrics"][0]
    assert m_nat["iteration"] == m_np["iteration"]
    np.testing.assert_allclose(m_nat["max_residual"], m_np["max_residual"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(m_nat["mean_residual"], m_np["mean_residual"], rtol=1e-9, atol=1e-12)


def test_native_and_numpy_sweeps_agree_on_a_near_tie_of_the_edge_removal(monkeypatch):
    """The native sweeps add in another order than the numpy form (1e-10 contract, include/mvs_hip.h), and the edge removed per
    outer iteration is an argmax over scores built from the residuals: two bad edges of mirror-image placement whose scores
    differ by ~1e-6 relative -- far above the 1e-10 contract, far below anything a user would set apart -- must be removed in
    the same order by both paths, with the same surviving edges and parameters."""
    g, _ = _grid_graph(4, 4, 2, noise=0.0, seed=3, quality=0.8)
    for e, shift in (((1, 2), 40.0), ((13, 14), 40.0 * (1 + 1e-6))):       # mirror images in the 4 x 4 grid
        g.edges[e]["transform"] = param_utils.affine_from_translation([0.0, shift]) @ g.edges[e]["transform"]
        g.edges[e]["quality"] = 0.3
    out = []
    for force in (False, True):
        monkeypatch.setattr(pr, "_FORCE_NUMPY", force)
        out.append(pr.groupwise_resolution(g, "global_optimization", reference_view=0))
    (p_nat, i_nat), (p_np, i_np) = out
    assert i_nat["used_edges"] == i_np["used_edges"]
    assert (1, 2) not in i_nat["used_edges"][0] and (13, 14) not in i_nat["used_edges"][0]
    assert i_nat["metrics"][0]["iteration"] == i_np["metrics"][0]["iteration"]          # same sweeps per outer iteration
    for v in p_np:
        np.testing.assert_allclose(p_nat[v], p_np[v], rtol=0, atol=1e-10)


def test_vector_residual_pass_is_bitwise_the_scalar_one():
    """The AVX2 form of the sweeps' residual pass (3D, 8 beads per edge) against the scalar loops (MVS_RESOLVE_SCALAR=1, read once
    per process, hence two subprocesses): parameters and the whole per-sweep history bit for bit."""
    import os, subprocess, sys
    script = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from test_param_resolution import _grid_graph\n"
        "from multiview_stitcher_amd import param_resolution as pr\n"
        "g, _ = _grid_graph(4, 3, 3, noise=0.3, seed=5, quality=0.8)\n"
        "p, i = pr.groupwise_resolution(g, 'global_optimization', reference_view=2)\n"
        "m = i['metrics'][0]\n"
        "print(json.dumps([[float(x).hex() for x in np.ravel(p[v])] for v in sorted(p)] + [[float(x).hex() for x in m['mean_residual']], [float(x).hex() for x in m['max_residual']]]))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for scalar in (False, True):
        env = dict(os.environ)
        env.pop("MVS_RESOLVE_SCALAR", None)
        if scalar:
            env["MVS_RESOLVE_SCALAR"] = "1"
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


# ---- linear two-pass resolver (param_resolution/linear_two_pass.py) ------------------------------------------------------
def test_linear_two_pass_translation_solves_a_consistent_grid_exactly():
    g, tau = _grid_graph(4, 3, ndim=3, seed=3)
    params, info = pr.groupwise_resolution(g, "linear_two_pass", transform="translation", reference_view=0)
    got = np.array([params[v][:3, 3] for v in range(12)])
    np.testing.assert_allclose(got - got[0], tau - tau[0], atol=1e-6)
    for v in range(12):
        np.testing.assert_array_equal(params[v][:3, :3], np.eye(3))          # identity linear part (T/test_param_resolution.py)
    # (residuals of a consistent graph are solver noise, so the median + 2 MAD rule may still drop a few edges: the
    # spanning tree always stays)
    assert 11 <= len(info["used_edges"][0]) <= len(g.edges)
    assert max(info["edge_residuals"][0].values()) < 1e-5


def test_linear_two_pass_prunes_a_bad_edge_but_keeps_the_spanning_tree():
    g, tau = _grid_graph(4, 4, ndim=2, noise=0.01, seed=5)
    bad = (5, 6)
    g.edges[bad]["transform"] = param_utils.affine_from_translation(tau[5] - tau[6] + np.array([25.0, -30.0]))
    params, info = pr.groupwise_resolution(g, "linear_two_pass", transform="translation", reference_view=0)
    assert bad not in info["used_edges"][0]                                   # outlier removed in pass 2
    got = np.array([params[v][:2, 2] for v in range(16)])
    np.testing.assert_allclose(got - got[0], tau - tau[0], atol=0.1)
    # a bridge survives pruning even with a large residual (keep_mst)
    g2, tau2 = _grid_graph(3, 1, ndim=2, seed=1)
    g2.edges[(1, 2)]["transform"] = param_utils.affine_from_translation(tau2[1] - tau2[2] + 40.0)
    _, info2 = pr.groupwise_resolution(g2, "linear_two_pass", transform="translation", reference_view=0)
    assert (1, 2) in info2["used_edges"][0]


def test_linear_two_pass_rigid_recovers_small_rotations():
    rng = np.random.default_rng(2)
    n = 6
    ang = rng.normal(0, 0.02, n)
    ang[0] = 0
    tt = rng.normal(0, 2, (n, 2))
    tt[0] = 0

    def C(i):
        c, s = np.cos(ang[i]), np.sin(ang[i])
        M = np.eye(3)
        M[:2, :2] = [[c, -s], [s, c]]
        M[:2, 2] = tt[i]
        return M

    g = pr.RegGraph(range(n), {v: {"spacing": {"y": 1.0, "x": 1.0}} for v in range(n)})
    for a in range(n):
        for b in (a + 1, a + 2):
            if b < n:
                lo = rng.normal(0, 5, 2)
                g.add_edge(a, b, np.linalg.inv(C(b)) @ C(a), bbox=[lo, lo + 30.0])    # A_uv = C_v^-1 C_u
    params, info = pr.groupwise_resolution(g, "linear_two_pass", transform="rigid", reference_view=0)
    for v in range(n):
        got_ang = np.arctan2(params[v][1, 0], params[v][0, 0])
        assert abs(got_ang - ang[v]) < 2e-3                                   # first-order linearisation: small angles
        np.testing.assert_allclose(params[v][:2, 2], tt[v], atol=0.2)
    with pytest.raises(ValueError):
        pr.groupwise_resolution(g, "linear_two_pass", transform="affine")
    with pytest.raises(TypeError):
        pr.groupwise_resolution(g, "linear_two_pass", prune_quantile=0.9)


def test_kruskal_matches_networkx_minimum_spanning_tree():
    nx = pytest.importorskip("networkx")
    rng = np.random.default_rng(4)
    for _ in range(20):
        n = int(rng.integers(4, 10))
        pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
        ins = [pairs[i] for i in rng.permutation(len(pairs))[: int(rng.integers(n, len(pairs) + 1))]]
        w = rng.integers(0, 4, len(ins)).astype(float).tolist()              # many ties
        G = nx.Graph()
        for (a, b), ww in zip(ins, w):
            G.add_edge(a, b, weight=ww)
        want = {tuple(sorted(e)) for e in nx.minimum_spanning_tree(G, weight="weight").edges}
        order = pr._nx_edge_order(list(G.nodes), ins)
        wmap = {tuple(sorted(e)): ww for e, ww in zip(ins, w)}
        got = pr._kruskal_mst(list(G.nodes), order, [wmap[tuple(sorted(e))] for e in order])
        assert sum(wmap[e] for e in got) == sum(wmap[e] for e in want) and len(got) == len(want)
