#!/usr/bin/env python
"""bench.py -- register+fuse throughput of the MI355X hot path on a 3D tile grid.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on
rank 0.  For N>1 it is launched by ``python -m torch.distributed.run`` with one rank per GPU.

Workload (BASELINE.json north_star): a 4x4x4 grid of 512^3 uint16 tiles, 20 % overlap, per-tile
integer jitter unknown to the stage metadata; one "step" = pairwise phase-correlation registration
of all face-neighbour pairs + fusion of the whole mosaic (cosine-blend weighted average), tiles
resident in HBM.  Each rank owns one such mosaic (independent positions of a multi-position
acquisition): units shard with no data-path collective -> "scaling": "weak".

value   = fused output Mvoxels/s over all ranks (register + fuse time, max over ranks)
roofline = algorithmic HBM bytes of the dominant kernel (fuse: every input voxel once + every
           output voxel once) / its HIP-event duration, vs 8 TB/s
cpu_baseline = the numpy/scipy oracle timed on this box on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=str, default="4,4,4", help="tiles in z,y,x")
    ap.add_argument("--tile", type=str, default="512,512,512", help="tile shape z,y,x")
    ap.add_argument("--overlap-frac", type=float, default=0.2)
    ap.add_argument("--cpu-tile", type=int, default=192, help="edge of the small tiles of the cpu_baseline sample")
    ap.add_argument("--reg-threads", type=int, default=None, help="host threads / context lanes for the pairwise registrations (library default: 8)")
    ap.add_argument("--no-register", action="store_true", help="time fusion only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def make_mosaic_on_device(torch, dev, grid, tile, overlap, seed, max_jitter=3, return_ground_truth=False):
    """Seeded synthetic mosaic generated in HBM: smoothed uniform noise ground truth (uint16, 0..4095)
    cut into overlapping tiles with an integer jitter the metadata does not know."""
    grid, tile, overlap = np.asarray(grid), np.asarray(tile), np.asarray(overlap)
    step = tile - overlap
    pad = max_jitter + 1
    gt_shape = step * (grid - 1) + tile + 2 * pad
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    F = torch.nn.functional
    gt = torch.empty(tuple(int(s) for s in gt_shape), dtype=torch.uint16, device=dev)
    zslab = 64
    halo = 4
    for z0 in range(0, int(gt_shape[0]), zslab):
        z1 = min(z0 + zslab, int(gt_shape[0]))
        # smooth per slab with its own halo so slabs are independent (content only has to be
        # spatially correlated and identical wherever tiles overlap, which the cut below guarantees)
        gs = torch.Generator(device=dev)
        gs.manual_seed(seed * 7919 + z0)
        noise = torch.rand((1, 1, z1 - z0 + 2 * halo, int(gt_shape[1]), int(gt_shape[2])), generator=gs, device=dev)
        for _ in range(2):
            noise = F.avg_pool3d(noise, 5, stride=1, padding=2, count_include_pad=False)
        sm = noise[0, 0, halo:halo + (z1 - z0)]
        lo, hi = 0.35, 0.65
        sm = ((sm - lo) / (hi - lo)).clamp_(0, 1) * 4095.0
        gt[z0:z1] = sm.to(torch.int32).to(torch.uint16)
        del noise, sm
    rng = np.random.default_rng(seed + 1)
    tiles, jitters, origins = [], [], []
    for idx in np.ndindex(*grid):
        idx = np.asarray(idx)
        jit = rng.integers(-max_jitter, max_jitter + 1, size=3)
        if not idx.any():
            jit[:] = 0
        start = idx * step + pad + jit
        sl = tuple(slice(int(s), int(s + n)) for s, n in zip(start, tile))
        tiles.append(gt[sl].contiguous())
        jitters.append(jit)
        origins.append((idx * step).astype(float))
    if return_ground_truth:      # (tests) ground-truth volume; world coordinate w of the mosaic sits at index w + pad
        return tiles, np.array(jitters), np.array(origins), gt, pad
    del gt
    return tiles, np.array(jitters), np.array(origins)


def build_sims(tiles, origins, dev_index):
    from multiview_stitcher_amd import spatial_image_utils as si
    from multiview_stitcher_amd.device import DeviceArray

    sims = []
    for t, o in zip(tiles, origins):
        da = DeviceArray.from_pointer(t.data_ptr(), tuple(t.shape), np.uint16, dev_index, owner=t)
        sim = si.to_spatial_image(da, dims=["z", "y", "x"], scale={"z": 1.0, "y": 1.0, "x": 1.0},
                                  translation=dict(zip("zyx", o)))
        si.set_sim_affine(sim, np.eye(4), si.DEFAULT_TRANSFORM_KEY)
        sims.append(sim)
    return sims


def cpu_baseline(args, grid, tile, overlap):
    """Time the numpy/scipy oracle (the reference's own scipy calls, 1 thread) on a bounded sample of the same
    workload in the metric's unit: a 2x2x2 mosaic of small tiles with the same overlap fraction, fused whole, plus
    its 12 face-neighbour registrations (one pair per axis orientation is timed, x4)."""
    from multiview_stitcher_amd import sample_data
    from oracle import fuse_oracle as fo
    from oracle import reg_oracle as ro
    from tests.helpers import sim_to_view, squeeze_field, union_bb

    ts = np.minimum(tile, int(args.cpu_tile))       # small tiles with the same overlap fraction
    ov = np.maximum((ts * args.overlap_frac).astype(int), 1)
    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=ts, tiles=(2, 2, 2), overlap=ov, dtype=np.uint16, seed=5)
    sims = [squeeze_field(s) for s in sims]
    views, bbs = zip(*[sim_to_view(s) for s in sims])
    params = [np.eye(4) for _ in sims]
    out_bb = union_bb(bbs, params, np.ones(3))
    t0 = time.perf_counter()
    fused = fo.fuse_np(list(views), params, out_bb, full_view_bbs=list(bbs))
    t_fuse = time.perf_counter() - t0
    vox = float(np.prod(np.asarray(fused[0] if isinstance(fused, tuple) else fused).shape))
    t_pairs = []
    for axis in range(3):                           # overlap crops of a z-, y- and x-face pair
        a, b = ro.make_pair_for_bench(ts, ov, seed=3 + axis)
        a, b = np.ascontiguousarray(np.moveaxis(a, -1, axis)), np.ascontiguousarray(np.moveaxis(b, -1, axis))
        t0 = time.perf_counter()
        ro.phase_correlation_registration(a, b)
        t_pairs.append(time.perf_counter() - t0)
    t_reg = 4.0 * float(np.sum(t_pairs))            # 12 face pairs in a 2x2x2 grid, 4 per orientation
    return {"value": vox / (t_reg + t_fuse) / 1e6, "unit": "Mvoxels/s", "cores": 1, "kind": "port",
            "sample": f"oracle (numpy + scipy 1.15: the reference's own affine_transform / fft / uniform_filter / spearmanr calls), "
                      f"1 thread, on a 2x2x2 mosaic of uint16 tiles {ts.tolist()}, overlap {ov.tolist()}: fuse of the whole "
                      f"{int(vox)}-voxel mosaic {t_fuse:.1f} s + 12 pair registrations {t_reg:.1f} s (3 timed, one per axis "
                      f"orientation, {np.round(t_pairs, 2).tolist()} s, x4)",
            "fuse_only_mvoxels_s": vox / t_fuse / 1e6, "register_pair_s": float(np.mean(t_pairs))}


def fuse_traffic_bytes(grid, tile):
    """HBM bytes per fuse launch from the committed PMC pass (FETCH_SIZE x2 per the calibration + WRITE_SIZE,
    profiles/round1_summary.md); only valid for the workload it was measured on."""
    try:
        with open(os.path.join(ROOT, "profiles", "round1_fuse_traffic.json")) as f:
            t = json.load(f)
        if list(grid) == [4, 4, 4] and list(tile) == [512, 512, 512]:
            return t["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; "nccl" is RCCL.  (MVS_BENCH_BACKEND=gloo and more ranks than GPUs are only for exercising
    # this launch path on a box with fewer GPUs: ranks then share devices.)
    backend = os.environ.get("MVS_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if world > n_dev and backend == "nccl":
        raise SystemExit(f"{world} ranks but {n_dev} GPUs visible")
    local_rank = local_rank % max(n_dev, 1)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from multiview_stitcher_amd import _lib, fusion
    from multiview_stitcher_amd import spatial_image_utils as si

    _lib.init(local_rank)
    grid = np.array([int(v) for v in args.grid.split(",")])
    tile = np.array([int(v) for v in args.tile.split(",")])
    overlap = np.round(tile * args.overlap_frac).astype(int)

    tiles, jitters, origins = make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=1000 + rank)
    sims = build_sims(tiles, origins, local_rank)
    torch.cuda.synchronize()

    do_register = not args.no_register
    registration = None
    if do_register:
        try:
            from multiview_stitcher_amd import registration  # noqa: F811
            if not hasattr(registration, "register"):
                raise ImportError("register() not available")
        except ImportError:
            registration, do_register = None, False

    key_in, key_out = si.DEFAULT_TRANSFORM_KEY, "registered"
    out_holder = {}
    kernel_ms = []
    reg_ms = []
    fuse_ms = []

    def step():
        # the previous step's mosaic is released before the next one is fused (as a consumer would), so the 10 GB
        # output buffer is recycled by the library's pool instead of being hipMalloc'ed anew every other step
        out_holder.clear()
        t_reg0 = time.perf_counter()
        key = key_in
        if do_register:
            registration.register(sims, transform_key=key_in, new_transform_key=key_out, device=local_rank,
                                  pre_registration_pruning_method="keep_axis_aligned",
                                  n_parallel_pairwise_regs=args.reg_threads)
            key = key_out
        t_reg1 = time.perf_counter()
        fused = fusion.fuse(sims, transform_key=key, output_chunksize={d: 1 << 30 for d in "zyx"},
                            output_on_backend=True, device=local_rank)
        out_holder["fused"] = fused
        kernel_ms.append(_lib.last_kernel_ms(local_rank))   # blocks until the fuse kernels are done
        reg_ms.append((t_reg1 - t_reg0) * 1e3)
        fuse_ms.append((time.perf_counter() - t_reg1) * 1e3)
        return fused

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    reg_ms.clear()
    fuse_ms.clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    fused = out_holder["fused"]
    out_shape = fused.shape
    out_vox = float(np.prod(out_shape))
    in_vox = float(len(tiles) * np.prod(tile))
    es = 2
    alg_bytes = in_vox * es + out_vox * es
    k_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    value = out_vox * world / (elapsed / args.steps) / 1e6

    reg_err = None
    if do_register:
        from multiview_stitcher_amd import param_utils
        rec = np.array([param_utils.select_time(si.get_affine_from_sim(s, key_out), 0)[:3, 3] for s in sims])
        reg_err = float(np.max(np.abs((rec - rec[0]) - (jitters - jitters[0]))))   # relative to tile 0: the resolver fixes its own reference view
    if rank == 0:
        result = {
            "metric": "Mvoxels/s register+fuse, 3D tile grid" if do_register else "Mvoxels/s fuse only, 3D tile grid",
            "value": value,
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16 in/out, f32 accumulate, f64 coordinates",
            "data": "synthetic (seeded smoothed noise mosaic generated in HBM, integer jitter unknown to metadata)",
            "config": {
                "workload": f"{'x'.join(map(str, grid))} grid (z,y,x) of {'x'.join(map(str, tile))} uint16 tiles, "
                            f"{int(args.overlap_frac * 100)}% overlap, "
                            + ("phase-correlation register of face-neighbour pairs + " if do_register else "")
                            + "cosine-blend weighted-average fuse; one mosaic per GPU",
                "output_shape": [int(s) for s in out_shape],
                "tiles_per_gpu": len(tiles),
                "register_ms_per_step": float(np.mean(reg_ms)) if do_register else None,
                "fuse_ms_per_step": float(np.mean(fuse_ms)),
                "fuse_kernel_ms": k_ms,
                "registration_max_abs_error_px": reg_err,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "fuse launch = copy_region_kernel + fuse_region_kernel<1|2|4|8> (u16) side by side on forked streams, timed as one unit (first start to last end)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": fuse_traffic_bytes(grid, tile),
                "algorithmic_bytes_per_launch": alg_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, grid, tile, overlap)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
