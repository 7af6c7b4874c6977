"""The driver's N > 1 launch of bench.py exercised on the one GPU of the test box: two ranks under
``python -m torch.distributed.run`` share device 0 (gloo for the control plane, since RCCL needs one GPU per rank), ONE mosaic
sharded over them (``--mode shard``: tile bricks + halo exchange, pairs by owner of the fixed view, output sub-boxes).
Checked: the JSON line of the contract, the registration result, and the union of the ranks' fused sub-boxes == the
single-rank mosaic, voxel for voxel.  (reference: registration.py:2622-2694 one task per pair, fusion/_core.py:1133-1141
one task per block, browser/executors.py:166-194 the farm.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env, timeout=900):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert [ln for ln in p.stdout.splitlines() if ln.strip()] == lines, p.stdout[-2000:]     # rank 0's stdout is the JSON line alone
    return json.loads(lines[0])


def test_two_ranks_one_mosaic_equals_single_rank(hip_device, tmp_path):
    # (the PCIe-inclusive leg stays on: at N = 2 every rank runs its own upload -> register -> fuse -> download pipeline)
    common = ["--grid", "2,2,2", "--tile", "256,256,256", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-c3", "--no-c5"]
    d1, d2 = tmp_path / "n1", tmp_path / "n2"
    d1.mkdir()
    d2.mkdir()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common, dict(env, MVS_BENCH_DUMP=str(d1)))
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + common,
               dict(env, MVS_BENCH_DUMP=str(d2), MVS_BENCH_BACKEND="gloo"))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["mode"] == "shard" and two["scaling"] == "strong"
    # the sharded registration (pairs by owner, results all-gathered, replicated resolution) ends with the parameters of the
    # single-rank run; on 256^3 tiles (51 px of overlap, no binning) the hidden integer jitters are recovered exactly
    assert two["config"]["registration_max_abs_error_px"] == pytest.approx(one["config"]["registration_max_abs_error_px"], abs=1e-9)
    assert one["config"]["registration_max_abs_error_px"] < 1e-6
    for line in (one, two):
        assert line["steps"] == 1 and line["warmup"] == 0 and line["unit"] == "Mvoxels/s" and line["value"] > 0
    # both ranks registered a share of the pairs and fused a share of the mosaic
    assert 0 < two["config"]["pairs_per_step_rank0"] < one["config"]["pairs_per_step_rank0"]
    # PCIe-inclusive leg: defined at N = 2 as well -- both ranks uploaded their tiles (brick + halo: together at least the mosaic's
    # 8 tiles), downloaded disjoint sub-boxes that add up to the mosaic, and the rate covers the whole mosaic
    p1, p2 = one["value_incl_pcie"], two["value_incl_pcie"]
    assert "error" not in p1 and p1["value"] > 0 and p2["value"] > 0
    assert len(p2["by_rank"]["own_ms"]) == 2 and all(v > 0 for v in p2["by_rank"]["own_ms"])
    assert p2["d2h_gb"] == pytest.approx(p1["d2h_gb"], rel=1e-9)
    assert p2["h2d_gb"] >= p1["h2d_gb"] - 1e-9 and all(v > 0 for v in p2["by_rank"]["h2d_gb"])
    full = np.load(d1 / "fused_rank0of1.npy")
    got = np.zeros_like(full)
    covered = np.zeros(full.shape, dtype=bool)
    for r in range(2):
        part = np.load(d2 / f"fused_rank{r}of2.npy")
        meta = json.load(open(d2 / f"fused_rank{r}of2.json"))
        sl = tuple(slice(o, o + n) for o, n in zip(meta["index_offset"], part.shape))
        assert not covered[sl].any()
        got[sl] = part
        covered[sl] = True
    assert covered.all()
    # every rank fuses its sub-box in the index frame of the whole mosaic (sharding.fuse_shard -> fuse(frame_origin=...)):
    # the union equals the single-rank mosaic voxel for voxel
    np.testing.assert_array_equal(got, full)


def test_eight_ranks_one_mosaic_equals_single_rank(hip_device, tmp_path):
    """The north star's partition -- 8 ranks, a 2 x 2 x 2 brick of the 4 x 4 x 4 tile grid each -- launched for real: eight
    processes under ``torch.distributed.run`` share the box's one GPU (gloo control plane), exchange their one-tile halos with
    the isend / irecv schedule of ``sharding.exchange_halo`` (seven peers per rank instead of the one the 2-rank test has),
    register the pairs whose fixed view they own, all-gather the results, resolve replicated and fuse their sub-box.  Small
    tiles (96^3) keep eight contexts on one GPU cheap; the partition logic does not depend on the tile size."""
    common = ["--grid", "4,4,4", "--tile", "96,96,96", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-pcie", "--no-c3", "--no-c5"]
    d1, d8 = tmp_path / "n1", tmp_path / "n8"
    d1.mkdir()
    d8.mkdir()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common, dict(env, MVS_BENCH_DUMP=str(d1)))
    eight = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port()), "bench.py", "--gpus", "8"] + common,
                 dict(env, MVS_BENCH_DUMP=str(d8), MVS_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4"), timeout=1500)
    assert eight["n_gpus"] == 8 and eight["config"]["mode"] == "shard" and eight["scaling"] == "strong"
    assert eight["config"]["registration_max_abs_error_px"] == pytest.approx(one["config"]["registration_max_abs_error_px"], abs=1e-9)
    full = np.load(d1 / "fused_rank0of1.npy")
    got = np.zeros_like(full)
    covered = np.zeros(full.shape, dtype=bool)
    pairs, held = [], []
    for r in range(8):
        part = np.load(d8 / f"fused_rank{r}of8.npy")
        meta = json.load(open(d8 / f"fused_rank{r}of8.json"))
        assert meta["mode"] == "shard" and part.size > 0 and meta["pairs_registered"] > 0      # every rank registered and fused
        assert meta["halo_exchange_ms"] is not None                                            # ... after a completed exchange
        pairs.append(meta["pairs_registered"])
        held.append(meta["tiles_held"])
        sl = tuple(slice(o, o + n) for o, n in zip(meta["index_offset"], part.shape))
        assert not covered[sl].any()                                                           # sub-boxes are disjoint
        got[sl] = part
        covered[sl] = True
    assert covered.all()
    # the 144 pairs of the pruned overlap graph are split by the owner of the fixed view; a rank holds its 2 x 2 x 2 brick plus a
    # one-tile halo (27 tiles at a corner brick of this grid), never the whole mosaic
    assert sum(pairs) == json.load(open(d1 / "fused_rank0of1.json"))["pairs_registered"] == 144
    # the line itself carries the per-rank split: pairs per step by rank sum to the mosaic's 144, every
    # rank reports the serial host share of its own step (graph + pruning + job assembly + resolution + fuse host path)
    cfg = eight["config"]
    assert len(cfg["pairs_per_step_by_rank"]) == 8 and sum(cfg["pairs_per_step_by_rank"]) == pytest.approx(144.0)
    assert cfg["pairs_per_step_by_rank"] == pytest.approx([float(p) for p in pairs])
    assert len(cfg["serial_host_ms_by_rank"]) == 8 and all(np.isfinite(v) for v in cfg["serial_host_ms_by_rank"])
    assert one["config"]["pairs_per_step_by_rank"] == [144.0] and np.isfinite(one["config"]["serial_host_ms"])
    assert one["roofline"]["frac_of_copy_ceiling"] == pytest.approx(one["roofline"]["achieved"] / 6290.0)
    assert all(8 <= h <= 27 for h in held), held
    np.testing.assert_array_equal(got, full)
