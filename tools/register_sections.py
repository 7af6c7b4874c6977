"""Timing checkpoints inside registration.register (patched copy of its source, north-star mosaic), including the teardown of
its locals when it returns (the pair geometries held by the bin cache cost 2.5 ms there while they were lists of floats)."""
import inspect, re, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from multiview_stitcher_amd import _lib, registration
from multiview_stitcher_amd import spatial_image_utils as si
src = inspect.getsource(registration.register)
marks = [
    ("    nt = sims_reg[0].sizes.get", "sims_reg"),
    ("    # (1) graph", "prebin"),
    ("    tol = overlap_tolerance", "sps_affs"),
    ("    # (2) pairwise registrations per time point", "graph_prune_sync"),
    ("        results = compute_pairwise_registrations(", "fields"),
    ("        keep = list(range(len(edges)))", "pairs"),
    ("            p_nodes, info = param_resolution.groupwise_resolution(", "reggraph_build"),
    ("            params_t.append([p_nodes[v] for v in range(len(sims))])", "resolution"),
    ("    params = [np.stack(", "loop_end"),
    ("    # (4) write back", "params_stack"),
    ("    if return_dict:", "write_back"),
]
for anchor, name in marks:
    assert anchor in src, anchor
    ind = re.match(r"\s*", anchor).group(0)
    src = src.replace(anchor, f"{ind}_T.append(({name!r}, _time.perf_counter()))\n{anchor}", 1)
src = src.replace('    from . import msi_utils, mv_graph, param_resolution\n    from . import spatial_image_utils as si_utils\n', '    from multiview_stitcher_amd import msi_utils, mv_graph, param_resolution\n    from multiview_stitcher_amd import spatial_image_utils as si_utils\n    _T.append(("start", _time.perf_counter()))\n', 1)
tail = """
    _T.append(("pre_del", _time.perf_counter()))
    _r = params
    del fields; _T.append(("del_fields", _time.perf_counter()))
    del results, all_results; _T.append(("del_results", _time.perf_counter()))
    del g, p_nodes, info, resolution_info; _T.append(("del_resolve", _time.perf_counter()))
    del g_views, edges, sps, affs; _T.append(("del_graph", _time.perf_counter()))
    _f0 = _torch.cuda.mem_get_info()[0]
    _t0 = _time.perf_counter(); _vals = [s_["value"] for s_ in bin_cache._items.values()]; _keeps = [s_["keep"] for s_ in bin_cache._items.values()]
    del bin_cache; _T.append(("del_bincache_shell", _time.perf_counter()))
    del _keeps; _T.append(("del_keeps", _time.perf_counter()))
    import collections as _c
    _groups = _c.defaultdict(list)
    for _v in _vals: _groups[type(_v).__name__].append(_v)
    del _v, _vals
    for _k in list(_groups):
        _n = len(_groups[_k]); _ta = _time.perf_counter(); del _groups[_k]; print("del %d x %s: %.3f ms" % (_n, _k, (_time.perf_counter() - _ta) * 1e3))
    _vals = None
    del _vals; _T.append(("del_values", _time.perf_counter()))
    _T.append(("freed_MB_%d" % ((_torch.cuda.mem_get_info()[0] - _f0) >> 20), _time.perf_counter()))
    del prebin, sims_reg, sims, params_t; _T.append(("del_rest", _time.perf_counter()))
    return _r
"""
i = src.rindex("    return params")
src = src[:i] + tail
ns = dict(vars(registration)); ns["_T"] = []; ns["_time"] = time; ns["_torch"] = torch
exec(src, ns)
reg = ns["register"]
dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, np.round(tile * 0.2).astype(int), seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
for rep in range(6):
    ns["_T"].clear()
    t0 = time.perf_counter()
    reg(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="reg", device=0)
    t1 = time.perf_counter()
    T = ns["_T"]
    if rep >= 3:
        print("total %.1f ms: " % ((t1 - t0) * 1e3) + "entry %.2f " % ((T[0][1] - t0) * 1e3) + " ".join("%s %.2f" % (T[i][0], (T[i][1] - T[i - 1][1]) * 1e3) for i in range(1, len(T))) + " exit %.2f" % ((t1 - T[-1][1]) * 1e3))
