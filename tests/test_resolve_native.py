"""Host-only: the one-call groupwise resolution of the library (mvs_resolve_translations, through
param_resolution.resolve_translations_native) against param_resolution.groupwise_resolution -- the Python restatement of
param_resolution/__init__.py:44-150 + global_optimization.py:16-511, which tests/test_resolve_oracle.py and
tests/test_param_resolution.py pin against the networkx-based oracle: same parameters (bit for bit: both run the same sweeps),
same metrics, residuals and used edges; None whenever the Python form would do something the call does not cover."""
import numpy as np
import pytest

from multiview_stitcher_amd import param_resolution as pr
from multiview_stitcher_amd import param_utils


def _mosaic(ndim, tiles, tile, overlap, seed, noise=0.01, drop=0.0, spacing=None, outlier=None, nan_quality=False):
    """Pairwise results of a jittered regular mosaic: neighbours along the axes (a fraction ``drop`` of them missing), the true
    relative jitter plus noise as translation, overlap boxes in the fixed view's frame."""
    rng = np.random.default_rng(seed)
    tiles, tile, overlap = np.asarray(tiles), np.asarray(tile, float), np.asarray(overlap, float)
    spacing = np.ones(ndim) if spacing is None else np.asarray(spacing, float)
    idxs = list(np.ndindex(*tiles))
    pos = {idx: k for k, idx in enumerate(idxs)}
    origins = np.array([np.asarray(i) * (tile - overlap) * spacing for i in idxs])
    jit = rng.integers(-3, 4, size=(len(idxs), ndim)).astype(float) * spacing
    edges, results = [], []
    for idx in idxs:
        for ax in range(ndim):
            nb = list(idx)
            nb[ax] += 1
            if tuple(nb) not in pos or rng.random() < drop:
                continue
            i, j = pos[idx], pos[tuple(nb)]
            lo = np.maximum(origins[i], origins[j])
            hi = np.minimum(origins[i], origins[j]) + (tile - 1) * spacing
            t = jit[j] - jit[i] + rng.normal(0, noise, ndim)
            edges.append((i, j))
            results.append({"transform": param_utils.affine_from_translation(t), "quality": float(rng.uniform(0.5, 1.0)),
                            "bbox": np.array([lo, hi])})
    if outlier is not None and results:
        results[outlier % len(results)]["transform"] = param_utils.affine_from_translation(np.full(ndim, 25.0))
        results[outlier % len(results)]["quality"] = 0.3
    if nan_quality and results:
        results[1]["quality"] = np.nan
    sps = {v: {"spacing": dict(zip("zyx"[-ndim:], spacing))} for v in range(len(idxs))}
    return len(idxs), edges, results, sps, np.tile(spacing, (len(idxs), 1))


def _generic(n, edges, results, sps, **kw):
    g = pr.RegGraph(range(n), sps)
    for (a, b), r in zip(edges, results):
        g.add_edge(a, b, r["transform"], quality=r["quality"], bbox=r["bbox"])
    return pr.groupwise_resolution(g, "global_optimization", **kw)


CASES = [
    dict(ndim=3, tiles=(4, 4, 4), tile=(512,) * 3, overlap=(102,) * 3, seed=0),
    dict(ndim=3, tiles=(4, 4, 4), tile=(512,) * 3, overlap=(102,) * 3, seed=1, noise=0.0),
    dict(ndim=3, tiles=(2, 4, 4), tile=(256, 512, 512), overlap=(51, 102, 102), seed=2, spacing=(2.0, 0.5, 0.5)),
    dict(ndim=2, tiles=(3, 3), tile=(2048, 2048), overlap=(410, 410), seed=3),
    dict(ndim=2, tiles=(5, 6), tile=(64, 64), overlap=(12, 12), seed=4, drop=0.15),
    dict(ndim=3, tiles=(3, 3, 3), tile=(64,) * 3, overlap=(12,) * 3, seed=5, drop=0.2, nan_quality=True),
    dict(ndim=2, tiles=(1, 2), tile=(512, 512), overlap=(0, 102), seed=6),
    dict(ndim=3, tiles=(1, 1, 3), tile=(32,) * 3, overlap=(8,) * 3, seed=7),
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
@pytest.mark.parametrize("kw", [{}, {"reference_view": 2}, {"max_iter": 20, "rel_tol": 1e-3}, {"abs_tol": 5.0, "reference_view": 0}])
def test_native_resolution_equals_python_form(case, kw):
    n, edges, results, sps, spacings = _mosaic(**case)
    # (dropped edges can split the mosaic: then the call declines and the Python form runs per component)
    want_p, want_info = _generic(n, edges, results, sps, **kw)
    got = pr.resolve_translations_native(n, edges, results, spacings, **kw)
    g = pr.RegGraph(range(n), sps)
    for (a, b), r in zip(edges, results):
        g.add_edge(a, b, r["transform"], quality=r["quality"], bbox=r["bbox"])
    one_component = len(g.connected_components()) == 1
    if not one_component:
        assert got is None
        return
    if got is None:      # the sweeps ended at or above abs_tol: the Python form went on to remove edges
        assert len(want_info["used_edges"][0]) < len(edges) or want_info["metrics"][0]["max_residual"][-1] >= (kw.get("abs_tol") or 0)
        return
    got_p, got_info = got
    for v in range(n):
        np.testing.assert_array_equal(got_p[v], want_p[v])
    assert got_info["used_edges"] == want_info["used_edges"]
    assert len(got_info["metrics"]) == 1 and got_info["metrics"][0]["icc"] == 0
    for key in ("mean_residual", "max_residual", "iteration"):
        assert got_info["metrics"][0][key] == want_info["metrics"][0][key], key
    assert list(got_info["edge_residuals"][0]) == list(want_info["edge_residuals"][0])
    for e, r in want_info["edge_residuals"][0].items():
        assert got_info["edge_residuals"][0][e] == pytest.approx(r, rel=1e-12, abs=1e-15)


def test_native_resolution_declines_edge_removal_and_other_models():
    n, edges, results, sps, spacings = _mosaic(ndim=3, tiles=(3, 3, 3), tile=(64,) * 3, overlap=(12,) * 3, seed=9, outlier=4)
    want_p, want_info = _generic(n, edges, results, sps)
    assert len(want_info["used_edges"][0]) < len(edges)                       # the Python form removed the outlier
    assert pr.resolve_translations_native(n, edges, results, spacings) is None
    n, edges, results, sps, spacings = _mosaic(ndim=2, tiles=(3, 3), tile=(64, 64), overlap=(12, 12), seed=1)
    assert pr.resolve_translations_native(n, edges, results, spacings, transform="rigid") is None
    rot = [dict(r) for r in results]
    rot[0]["transform"] = rot[0]["transform"].copy()
    rot[0]["transform"][0, 1] = 1e-3
    assert pr.resolve_translations_native(n, edges, rot, spacings) is None     # a pair result with a linear part
    assert pr.resolve_translations_native(n, edges + [edges[0]], results + [results[0]], spacings) is None      # duplicate pair
    assert pr.resolve_translations_native(n + 1, edges, results, np.ones((n + 1, 2))) is None                  # an unconnected view
    assert pr.resolve_translations_native(n, edges, results, spacings, reference_view="a") is None
    got = pr.resolve_translations_native(n, edges, results, spacings, reference_view=99)      # not a node: maximal-quality view
    want_p, _ = _generic(n, edges, results, sps, reference_view=99)
    for v in range(n):
        np.testing.assert_array_equal(got[0][v], want_p[v])


def test_reference_view_of_many_neighbours_uses_numpy_summation_order():
    """A view with >= 8 pairs: numpy's pairwise summation decides which view has the largest quality sum."""
    rng = np.random.default_rng(3)
    n = 12
    edges = [(0, j) for j in range(1, n)] + [(1, j) for j in range(2, n)]
    results = [{"transform": param_utils.affine_from_translation(rng.normal(0, 0.01, 2)), "quality": float(rng.uniform(0.1, 1.0)),
                "bbox": np.array([[0.0, 0.0], [10.0, 12.0]])} for _ in edges]
    sps = {v: {"spacing": {"y": 1.0, "x": 1.0}} for v in range(n)}
    want_p, want_info = _generic(n, edges, results, sps)
    got = pr.resolve_translations_native(n, edges, results, np.ones((n, 2)))
    assert got is not None
    for v in range(n):
        np.testing.assert_array_equal(got[0][v], want_p[v])
