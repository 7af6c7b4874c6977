"""Host-only: the one-call overlap graph + alternating-pattern pruning of the library (mvs_view_graph_prune, through
mv_graph.registration_edges_native) against the generic functions of multiview_stitcher_amd.mv_graph (which
tests/test_mv_graph_host.py pins against networkx): same edges in the same order -- the order is registration.register's
work list and decides the tie breaks of the groupwise resolution (mv_graph.py:35-180, 664-741 of the reference)."""
import numpy as np
import pytest

from multiview_stitcher_amd import mv_graph, param_utils


def _sp(origin, shape, spacing, transform):
    d = "zyx"[-len(origin):]
    return {"origin": dict(zip(d, map(float, origin))), "shape": dict(zip(d, map(int, shape))), "spacing": dict(zip(d, map(float, spacing))),
            "transform": np.asarray(transform, dtype=float)}


def _mosaic(ndim, tiles, tile, overlap, spacing, jitter, seed, in_origin=False, scale=None):
    """Regular grid with per-tile jitter of the stage positions; positions in the transform or in the stack origin."""
    rng = np.random.default_rng(seed)
    spacing = np.asarray(spacing, float)
    sps = []
    for idx in np.ndindex(*tiles):
        o = (np.asarray(idx) * (np.asarray(tile) - np.asarray(overlap)) + rng.uniform(-jitter, jitter, ndim)) * spacing
        a = param_utils.affine_from_translation(np.zeros(ndim) if in_origin else o)
        if scale is not None:
            a[:ndim, :ndim] = np.diag(scale)
        sps.append(_sp(o if in_origin else np.zeros(ndim), tile, spacing, a))
    return sps


def _generic(sps, tol, pairs, method, kw=None):
    g = mv_graph.build_view_adjacency_graph(sps, overlap_tolerance=tol, pairs=pairs)
    g = mv_graph.prune_view_adjacency_graph(g, method, kw)
    return [tuple(sorted(e)) for e in g.edges()]


CASES = [
    dict(ndim=3, tiles=(4, 4, 4), tile=(512, 512, 512), overlap=(102, 102, 102), spacing=(1, 1, 1), jitter=0.0, seed=0),       # the north star
    dict(ndim=3, tiles=(2, 4, 4), tile=(256, 512, 512), overlap=(51, 102, 102), spacing=(1, 1, 1), jitter=3.0, seed=1),        # C3
    dict(ndim=3, tiles=(3, 3, 2), tile=(40, 64, 80), overlap=(8, 10, 30), spacing=(2.0, 0.5, 0.5), jitter=2.5, seed=2, in_origin=True),
    dict(ndim=2, tiles=(3, 3), tile=(2048, 2048), overlap=(410, 410), spacing=(1, 1), jitter=0.0, seed=3),                   # C2
    dict(ndim=2, tiles=(5, 4), tile=(64, 48), overlap=(12, 20), spacing=(0.3, 0.7), jitter=4.0, seed=4),
    dict(ndim=2, tiles=(1, 2), tile=(512, 512), overlap=(0, 102), spacing=(1, 1), jitter=0.0, seed=5),                       # C1
    dict(ndim=3, tiles=(2, 2, 3), tile=(30, 30, 30), overlap=(6, 6, 6), spacing=(1, 1, 1), jitter=1.0, seed=6, scale=(1.0, 2.0, 0.5)),
    dict(ndim=2, tiles=(6, 6), tile=(32, 32), overlap=(16, 16), spacing=(1, 1), jitter=6.0, seed=7),                         # dense: many neighbours
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
@pytest.mark.parametrize("method", ["alternating_pattern", None])
@pytest.mark.parametrize("tol", [None, 0.0, 1.5])
def test_native_edges_equal_generic_path(case, method, tol):
    sps = _mosaic(**case)
    dims = list(sps[0]["spacing"])
    tol_d = None if tol is None else {d: float(tol) for d in dims}
    want = _generic(sps, tol_d, None, method)
    got = mv_graph.registration_edges_native(sps, tol_d, None, method)
    assert got is not None and got == want


def test_native_edges_with_given_pairs_and_colour_count():
    sps = _mosaic(ndim=3, tiles=(3, 3, 3), tile=(20, 20, 20), overlap=(5, 5, 5), spacing=(1, 1, 1), jitter=1.0, seed=11)
    pairs = [(i, j) for i in range(27) for j in range(27) if i != j and (i + j) % 3]
    for method, kw in (("alternating_pattern", None), ("alternating_pattern", {"n_colors": 3}), (None, None)):
        want = _generic(sps, None, pairs, method, kw)
        assert mv_graph.registration_edges_native(sps, None, pairs, method, kw) == want
    # scalar tolerance = the same value on every axis
    assert mv_graph.registration_edges_native(sps, 2.0, None) == _generic(sps, {d: 2.0 for d in "zyx"}, None, "alternating_pattern")


def test_native_path_declines_what_it_does_not_cover():
    sps = _mosaic(ndim=2, tiles=(2, 2), tile=(32, 32), overlap=(8, 8), spacing=(1, 1), jitter=0.0, seed=0)
    assert mv_graph.registration_edges_native(sps, None, None, "keep_axis_aligned") is None
    assert mv_graph.registration_edges_native(sps, None, None, "alternating_pattern", {"unknown": 1}) is None
    c, s = np.cos(0.1), np.sin(0.1)
    rot = [dict(sp) for sp in sps]
    rot[1]["transform"] = rot[1]["transform"] @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    assert mv_graph.registration_edges_native(rot, None, None) is None                       # a rotated view: halfspace path
    far = _mosaic(ndim=2, tiles=(1, 2), tile=(16, 16), overlap=(0, -40), spacing=(1, 1), jitter=0.0, seed=0)
    assert mv_graph.registration_edges_native(far, None, None) is None                       # no overlap: the generic path raises
    tstack = [dict(sp, transform=np.stack([sp["transform"]] * 2)) for sp in sps]
    assert mv_graph.registration_edges_native(tstack, None, None) is None                    # t-stacked transforms


def test_register_uses_the_same_edges_on_both_paths():
    """registration.register's graph step through the library call and through the generic functions (no GPU: the pairwise
    step is a stub executor that records the edge list)."""
    from multiview_stitcher_amd import registration
    from multiview_stitcher_amd import spatial_image_utils as si

    rng = np.random.default_rng(0)
    sims = []
    for idx in np.ndindex(2, 3, 3):
        sim = si.to_spatial_image(np.zeros((8, 8, 8), np.uint16), dims=["z", "y", "x"], scale={"z": 2.0, "y": 1.0, "x": 1.0},
                                  translation=dict(zip("zyx", np.asarray(idx) * np.array([12.0, 6.0, 6.0]) + rng.uniform(-0.5, 0.5, 3))))
        si.set_sim_affine(sim, np.eye(4), "stage")
        sims.append(sim)
    seen = {}

    def stub(msims, edges, kwargs):
        seen["edges"] = list(edges)
        return [{"transform": np.eye(4), "quality": 1.0, "bbox": np.array([[0.0] * 3, [1.0] * 3])} for _ in edges]

    out = {}
    for flag in (True, False):
        registration._native_graph[0] = flag
        try:
            registration.register(sims, transform_key="stage", pairwise_executor=stub, groupwise_resolution_method="linear")
        finally:
            registration._native_graph[0] = True
        out[flag] = seen["edges"]
    assert out[True] == out[False] and len(out[True]) >= 17
