"""CPU tests of the multi-GPU farm's host logic: unit sharding across ranks, incl. a world_size-2 gloo run
(the N>1 launch path of bench.py: one process per GPU, no data-path collective)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from multiview_stitcher_amd import executors, fusion, mv_graph, registration, sample_data
from multiview_stitcher_amd import spatial_image_utils as si

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_units_partition_is_exact():
    for n, w in [(343, 8), (10, 3), (5, 8), (98, 2)]:
        owned = [executors.shard_units(n, w, r) for r in range(w)]
        flat = sorted(u for o in owned for u in o)
        assert flat == list(range(n))
        assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1


def test_shard_units_weighted_balances_load():
    rng = np.random.default_rng(0)
    wts = rng.integers(1, 100, 50)
    owned = [executors.shard_units(50, 4, r, weights=wts) for r in range(4)]
    assert sorted(u for o in owned for u in o) == list(range(50))
    loads = [wts[o].sum() for o in owned]
    assert max(loads) - min(loads) <= wts.max()


def test_gloo_two_ranks_shard_and_reduce(tmp_path):
    """Two processes, gloo: every rank takes its shard of pair/chunk units, a max-reduce over ranks gives the job
    time and an all-gather of the unit lists shows the shards are disjoint and complete."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from multiview_stitcher_amd import executors
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        units = executors.shard_units(37, w, r)
        t = torch.tensor([float(len(units)) * 0.01 + r], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        got = [None] * w
        dist.all_gather_object(got, units)
        if r == 0:
            flat = sorted(u for g in got for u in g)
            assert flat == list(range(37)), flat
            assert abs(t.item() - (len(got[1]) * 0.01 + 1)) < 1e-9, t.item()
            print("OK", [len(g) for g in got])
        dist.destroy_process_group()
    """))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script)],
        capture_output=True, text=True, timeout=300,
    )
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK [19, 18]" in out.stdout or "OK [18, 19]" in out.stdout, out.stdout


def test_planner_axis_aligned_slabs_and_graph():
    """Host planner (no GPU): chunk -> view slabs for a translation grid, and the face-neighbour graph."""
    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(16, 40, 40), tiles=(1, 2, 3), overlap=(0, 10, 10), max_jitter=0)
    sims = [s.isel({"c": 0, "t": 0}) for s in sims]
    key = sample_data.METADATA_TRANSFORM_KEY
    sps = [si.get_stack_properties_from_sim(s) for s in sims]
    affs = [si.get_affine_from_sim(s, key) for s in sims]
    pairs = mv_graph.prune_to_axis_aligned(mv_graph.build_view_adjacency_pairs(sps, affs), sps, affs)
    assert sorted((i, j) for i, j, _ in pairs) == [(0, 1), (0, 3), (1, 2), (1, 4), (2, 5), (3, 4), (4, 5)]
    osp = fusion.calc_fusion_stack_properties(sims, affs, {"z": 1.0, "y": 1.0, "x": 1.0})
    assert osp["shape"] == {"z": 16, "y": 70, "x": 100}
    cs = {"z": 16, "y": 32, "x": 32}
    cbb, bidx = mv_graph.get_chunk_bbs(osp, cs)
    plan = fusion._build_spatial_fusion_plan(
        sparams=affs, views_bb=sps, output_stack_properties=osp, output_chunksize=cs, output_chunk_bbs=cbb,
        output_chunk_bbs_with_overlap=cbb, output_chunk_bbs_for_result=cbb, block_indices=bidx,
        overlap_in_pixels={d: 0 for d in "zyx"}, trim_overlap=True, interpolation_order=1, sdims=["z", "y", "x"],
    )
    assert plan["uses_axis_aligned_translation"] and plan["grid_aligned_translation_dims"] == ["z", "y", "x"]
    first = plan["per_chunk_entries"][0]
    assert [iv for iv, _ in first["views"]] == [0, 1, 3, 4]      # chunk (0,0,0) = y 0..31, x 0..31 touches the 2x2 corner tiles
    assert first["views"][0][1]["shape"] == {"z": 16, "y": 32, "x": 32}
    assert first["views"][1][1]["shape"]["x"] == 2                # tile 1 starts at x=30: two columns reach into the chunk
    # translation least squares: a consistent loop is solved exactly
    edges = [(0, 1), (1, 2), (0, 2)]
    d = {(0, 1): [1.0, -2.0], (1, 2): [0.5, 0.5], (0, 2): [1.5, -1.5]}
    res = [{"transform": np.block([[np.eye(2), np.array(d[e])[:, None]], [np.zeros((1, 2)), np.ones((1, 1))]])} for e in edges]
    p = registration.resolve_translations(3, edges, res)
    np.testing.assert_allclose(p[1][:2, 2], [-1.0, 2.0], atol=1e-9)
    np.testing.assert_allclose(p[2][:2, 2], [-1.5, 1.5], atol=1e-9)
