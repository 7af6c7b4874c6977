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
    by_block, info = fusion._plan_chunks(affs, sps, osp, cs, {d: 0 for d in "zyx"}, 1, ["z", "y", "x"])
    assert info["axis_aligned_translation_dims"] == ["z", "y", "x"] and info["grid_aligned_translation_dims"] == ["z", "y", "x"]
    planewise, first = by_block[(0, 0, 0)]
    assert not planewise
    assert [iv for iv, _, _ in first] == [0, 1, 3, 4]      # chunk (0,0,0) = y 0..31, x 0..31 touches the 2x2 corner tiles
    assert first[0][1:] == ((0, 0, 0), (16, 32, 32))
    assert first[1][1][2] == 0 and first[1][2][2] == 2           # tile 1 starts at x=30: two columns reach into the chunk
    # translation least squares: a consistent loop is solved exactly
    edges = [(0, 1), (1, 2), (0, 2)]
    d = {(0, 1): [1.0, -2.0], (1, 2): [0.5, 0.5], (0, 2): [1.5, -1.5]}
    res = [{"transform": np.block([[np.eye(2), np.array(d[e])[:, None]], [np.zeros((1, 2)), np.ones((1, 1))]])} for e in edges]
    p = registration.resolve_translations(3, edges, res)
    np.testing.assert_allclose(p[1][:2, 2], [-1.0, 2.0], atol=1e-9)
    np.testing.assert_allclose(p[2][:2, 2], [-1.5, 1.5], atol=1e-9)


# ---- one mosaic over the GPUs of a node (sharding.py) -------------------------------------------------------
def _grid_meta(grid=(4, 4, 4), tile=512, ov=102):
    from multiview_stitcher_amd import sharding  # noqa: F401

    sps, affs = [], []
    for idx in np.ndindex(*grid):
        o = np.asarray(idx) * (tile - ov)
        sps.append({"origin": dict(zip("zyx", o.astype(float))), "spacing": dict(zip("zyx", [1.0] * 3)),
                    "shape": dict(zip("zyx", [tile] * 3))})
        affs.append(np.eye(4))
    n = (tile - ov) * (np.asarray(grid) - 1) + tile
    osp = {"origin": dict(zip("zyx", [0.0] * 3)), "spacing": dict(zip("zyx", [1.0] * 3)), "shape": dict(zip("zyx", n.tolist()))}
    edges = []
    lin = np.arange(np.prod(grid)).reshape(grid)
    for idx in np.ndindex(*grid):
        for ax in range(3):
            if idx[ax] + 1 < grid[ax]:
                j = list(idx)
                j[ax] += 1
                edges.append((int(lin[idx]), int(lin[tuple(j)])))
    return sps, affs, osp, edges


def test_brick_partition_of_the_north_star_grid():
    from multiview_stitcher_amd import sharding

    sps, affs, osp, edges = _grid_meta()
    assert len(edges) == 144
    for world, want_counts in [(1, [1, 1, 1]), (2, [2, 1, 1]), (4, [2, 2, 1]), (8, [2, 2, 2])]:
        boxes, counts = sharding.output_subboxes(osp, world)
        assert counts == want_counts
        # the sub-boxes tile the output stack exactly
        vox = sum(int(np.prod([b["shape"][d] for d in "zyx"])) for b in boxes)
        assert vox == int(np.prod([osp["shape"][d] for d in "zyx"]))
        owners = sharding.tile_owners(sps, affs, boxes)
        per_rank = np.bincount(owners, minlength=world)
        assert per_rank.tolist() == [64 // world] * world                    # 8 ranks: a 2 x 2 x 2 brick each
        eo = sharding.edge_owners(edges, owners)
        assert sorted(np.bincount(eo, minlength=world).tolist())[0] >= 144 // world - 144 // (2 * world) - 6
        for r in range(world):
            need = sharding.rank_tiles(sps, affs, boxes, edges, owners, r)
            own = [v for v, o in enumerate(owners) if o == r]
            assert set(own) <= set(need)
            # own brick + a one-tile halo: never more than the (brick + 1 layer on the inner sides) box
            if world == 8:
                assert len(need) == 27, len(need)      # 2x2x2 brick + halo towards the inner neighbours = 3x3x3
            for (i, j), o in zip(edges, eo):
                if o == r:
                    assert i in need and j in need


def test_remote_array_is_metadata_only():
    from multiview_stitcher_amd import sharding

    ra = sharding.RemoteArray((4, 5, 6), np.uint16, owner=3)
    assert ra[1:3, :, 2:].shape == (2, 5, 4) and ra.dtype == np.uint16
    import pytest

    with pytest.raises(RuntimeError, match="rank 3"):
        np.asarray(ra)


def test_gloo_two_ranks_sharded_pair_executor(tmp_path):
    """world_size 2, gloo: every rank registers the pairs whose fixed view it owns (stub registration), the results are
    all-gathered and both ranks end up with the complete, identical list in edge order."""
    script = tmp_path / "w2.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        import torch.distributed as dist
        from multiview_stitcher_amd import sharding
        sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        from test_farm_cpu import _grid_meta
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        sps, affs, osp, edges = _grid_meta((2, 2, 4), 64, 12)
        boxes, counts = sharding.output_subboxes(osp, w)
        owners = sharding.tile_owners(sps, affs, boxes)
        calls = []
        def fake_register(a, b, **kw):
            calls.append((a, b))
            return {{"transform": np.eye(4) * (a + 1) - 0.0, "quality": float("nan") if (a + b) % 5 == 0 else float(b) / 3,
                    "bbox": np.array([[a, b, -0.0], [a + 0.1, b + 1e-17, 7.0]])}}
        ex = sharding.ShardedPairExecutor(r, w, owners, register_fn=fake_register)
        # results of the standard form travel as ONE fixed-size float64 tensor (no pickling)
        _orig = dist.all_gather_object
        def _forbidden(*a, **k):
            raise AssertionError("the pair results went through all_gather_object")
        dist.all_gather_object = _forbidden
        res = ex(list(range(len(sps))), edges, {{}})
        dist.all_gather_object = _orig
        assert len(res) == len(edges)
        for (i, j), q in zip(edges, res):
            want = fake_register(i, j)
            assert np.array_equal(q["transform"], want["transform"]) and q["transform"].shape == (4, 4)
            assert q["bbox"].tobytes() == want["bbox"].tobytes()               # bit for bit, the sign of a zero included
            assert (np.isnan(q["quality"]) and np.isnan(want["quality"])) or q["quality"] == want["quality"]
        calls[:] = [c for c in calls[: len(calls) - len(edges)]]
        mine = [e for e, o in zip(edges, sharding.edge_owners(edges, owners)) if o == r]
        assert calls == mine and 0 < len(mine) < len(edges)
        got = [None] * w
        dist.all_gather_object(got, len(mine))
        if r == 0:
            assert sum(got) == len(edges)
            print("OK", got)
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
    assert out.returncode == 0, out.stderr[-3000:]
    assert "OK [" in out.stdout, out.stdout


def test_gloo_eight_ranks_halo_exchange_and_sharded_pairs(tmp_path):
    """The north star's partition launched with EIGHT processes on the CPU (gloo): 4 x 4 x 4 tiles, one 2 x 2 x 2 brick per rank.
    The point-to-point halo exchange (``sharding.exchange_halo``: batched isend / irecv, seven possible peers per rank) delivers
    exactly the tiles ``rank_tiles`` lists, with the owner's content; the sharded pair executor splits the 144 pairs by the owner
    of the fixed view and every rank ends with all 144 results in edge order.  (The same launch on the GPU box, with the real
    registration and fusion: tests/test_bench_launch_gpu.py.)"""
    script = tmp_path / "w8.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        import torch, torch.distributed as dist
        from multiview_stitcher_amd import sharding
        sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        from test_farm_cpu import _grid_meta
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        sps, affs, osp, edges = _grid_meta((4, 4, 4), 64, 12)
        boxes, counts = sharding.output_subboxes(osp, w)
        assert counts == [2, 2, 2]
        owners = sharding.tile_owners(sps, affs, boxes)
        assert sorted(np.bincount(owners).tolist()) == [8] * 8
        needs = [sharding.rank_tiles(sps, affs, boxes, edges, owners, q, margin=8.0) for q in range(w)]
        tiles = [torch.full((3, 4, 5), v, dtype=torch.int16).view(torch.uint16) if owners[v] == r else None for v in range(len(sps))]
        got = sharding.exchange_halo(torch, dist, tiles, owners, needs, r, w, "cpu", via_host=True)
        held = [v for v, t in enumerate(got) if t is not None]
        assert held == needs[r], (r, held, needs[r])
        assert 8 < len(held) <= 27 and all(int(got[v].view(torch.int16)[0, 0, 0]) == v and got[v].dtype == torch.uint16 for v in held)
        ex = sharding.ShardedPairExecutor(r, w, owners, register_fn=lambda a, b, **kw: {{"transform": np.eye(4) * (a + 1), "quality": float(b),
                                                                                     "bbox": np.zeros((2, 3)), "rank": r}})
        res = ex(list(range(len(sps))), edges, {{}})
        assert len(res) == 144 and all(q["quality"] == float(j) and q["transform"][0, 0] == i + 1 for (i, j), q in zip(edges, res))
        by_rank = np.bincount([q["rank"] for q in res], minlength=w)
        eo = sharding.edge_owners(edges, owners)
        assert by_rank.sum() == 144 and all(q["rank"] == o for o, q in zip(eo, res))
        # pairs inside a brick stay with its rank, pairs across two bricks go to the emptier of the two: 17-19 per rank (always the
        # fixed view's owner: 12-24), and a rank only ever registers pairs of which it owns a view
        assert by_rank.min() >= 17 and by_rank.max() <= 19 and all(o in (owners[i], owners[j]) for o, (i, j) in zip(eo, edges))
        peers = sorted({{owners[v] for v in held}} - {{r}})
        counts_all = [None] * w
        dist.all_gather_object(counts_all, (len(held), len(peers), int(by_rank[r])))
        if r == 0:
            print("OK", counts_all)
        dist.destroy_process_group()
    """))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script)],
        capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"),
    )
    assert out.returncode == 0, out.stderr[-3000:]
    assert "OK [" in out.stdout, out.stdout
    # every rank exchanged with all seven others (a 2 x 2 x 2 brick of this grid touches every other brick)
    import ast
    counts_all = ast.literal_eval(out.stdout[out.stdout.index("OK [") + 3:].splitlines()[0])
    assert all(c[1] == 7 for c in counts_all), counts_all
