#!/usr/bin/env python
"""bench.py -- register+fuse throughput of the MI355X hot path on a 3D tile grid.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on
rank 0.  For N>1 it is launched by ``python -m torch.distributed.run`` with one rank per GPU.

Workload (BASELINE.json north_star): a 4x4x4 grid of 512^3 uint16 tiles, 20 % overlap, per-tile
integer jitter unknown to the stage metadata; one "step" = registration.register (overlap graph,
pruning, pairwise phase-correlation registration of the kept pairs, groupwise resolution) + fusion of
the whole mosaic (cosine-blend weighted average), tiles resident in HBM.

N > 1, --mode shard (default): ONE mosaic over the N GPUs -- every rank owns a brick of tiles (+ a
one-tile halo fetched once by point-to-point sends, RCCL), registers the pairs whose fixed view it
owns, the pairwise results (kB) are all-gathered, the resolution runs replicated and every rank
fuses its sub-box of the output: no data-path collective, "scaling": "strong".
--mode replica: one mosaic per rank (independent positions), "scaling": "weak".

value   = fused output Mvoxels/s of the whole job (register + fuse time, max over ranks)
roofline = algorithmic HBM bytes of the dominant kernel (fuse: every input voxel once + every
           output voxel once) / its HIP-event duration, vs 8 TB/s
roofline_register = algorithmic bytes of the pairwise registrations / their wall time
value_incl_pcie = the same mosaic with tiles starting and the result ending in pinned host memory
           (uploads overlapped with registration, downloads with fusion); N > 1 (shard): every rank its own pipeline
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
HBM_COPY_CEILING_GBS = 6290.0      # measured float4-copy ceiling of the chip (MI355X_MICROARCH.md, chip-level parameters; SURVEY 8d: report both)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3, help="untimed steps (the first handful of steps of a process run 5-8 ms slower)")
    ap.add_argument("--grid", type=str, default="4,4,4", help="tiles in z,y,x")
    ap.add_argument("--tile", type=str, default="512,512,512", help="tile shape z,y,x")
    ap.add_argument("--overlap-frac", type=float, default=0.2)
    ap.add_argument("--cpu-tile", type=int, default=192, help="edge of the small tiles of the cpu_baseline sample")
    ap.add_argument("--reg-threads", type=int, default=None, help="host threads / context lanes for the pairwise registrations (library default: 16)")
    ap.add_argument("--no-register", action="store_true", help="time fusion only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["shard", "replica"], default=None,
                    help="N > 1: 'shard' (default) = ONE mosaic, tiles / pairs / output sub-boxes partitioned over the ranks "
                         "(strong scaling); 'replica' = one mosaic per rank (weak scaling)")
    ap.add_argument("--pruning", default="alternating_pattern",
                    help="pre_registration_pruning_method (reference default: alternating_pattern)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive pipeline leg (N = 1 and --mode shard)")
    ap.add_argument("--no-c3", action="store_true", help="skip the content-based leg (BASELINE config C3; N = 1 only)")
    ap.add_argument("--no-c5", action="store_true", help="skip the Zarr-streamed leg (BASELINE config C5, a z-slab of its grid; N = 1 only)")
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


def build_sims(tiles, origins, dev_index, tile_shape=None):
    """SpatialImages over the device tiles; a tile this rank does not hold (None) becomes a metadata-only RemoteArray."""
    from multiview_stitcher_amd import spatial_image_utils as si
    from multiview_stitcher_amd.device import DeviceArray
    from multiview_stitcher_amd.sharding import RemoteArray

    sims = []
    for t, o in zip(tiles, origins):
        if t is None:
            da = RemoteArray(tuple(int(v) for v in tile_shape), np.uint16)
        else:
            da = DeviceArray.from_pointer(t.data_ptr(), tuple(t.shape), np.uint16, dev_index, owner=t)
        sim = si.to_spatial_image(da, dims=["z", "y", "x"], scale={"z": 1.0, "y": 1.0, "x": 1.0},
                                  translation=dict(zip("zyx", o)))
        si.set_sim_affine(sim, np.eye(4), si.DEFAULT_TRANSFORM_KEY)
        sims.append(sim)
    return sims


def _cpu_fuse_task(views, params, bbs, sub_bb):
    from oracle import fuse_oracle as fo

    fo.fuse_np(list(views), params, sub_bb, full_view_bbs=list(bbs))
    return float(np.prod(sub_bb["shape"]))


def _slabs_for_chunk(views, params, bbs, sub_bb, margin=2):
    """The reference hands every chunk task only the SLAB of each view that reaches into the chunk (fusion/_core.py:1371-1386);
    so does this farm: (views, params, full-view boxes) restricted to the windows an order-1 resample onto ``sub_bb`` needs
    (identity / translation parameters: the mosaic of the baseline sample)."""
    out_v, out_p, out_b = [], [], []
    lo_w = sub_bb["origin"]
    hi_w = sub_bb["origin"] + (sub_bb["shape"] - 1) * sub_bb["spacing"]
    for v, p, b in zip(views, params, bbs):
        t = np.asarray(p)[:-1, -1]
        a = np.floor((lo_w - t - v["origin"]) / v["spacing"]).astype(int) - margin
        e = np.ceil((hi_w - t - v["origin"]) / v["spacing"]).astype(int) + margin + 1
        a, e = np.maximum(a, 0), np.minimum(e, np.asarray(v["data"].shape))
        if np.any(e <= a):
            continue
        sl = tuple(slice(int(x), int(y)) for x, y in zip(a, e))
        out_v.append({"data": np.ascontiguousarray(v["data"][sl]), "origin": v["origin"] + a * v["spacing"], "spacing": v["spacing"]})
        out_p.append(p)
        out_b.append(b)
    return out_v, out_p, out_b


_CPU_FARM = {}
_ORIG_AFFINITY = set()       # CPU affinity of the process before main() pins it (the CPU baseline legs restore it)


def _cpu_ranges(cpus):
    """[0, 1, 2, 5] -> '0-2,5'"""
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def _cpu_quota_cores():
    """CPU quota of this container in cores (cgroup v2 cpu.max / v1 cfs quota), None when unlimited or unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def _cpu_farm_task(shm_dir, kind, index):
    """(worker) one task of the all-core baseline: kind 0 = oracle fuse of output chunk ``index``, 1 = oracle registration
    of pair ``index``, -1 = warm-up (imports + memory maps only).  Inputs are memory-mapped from ``shm_dir`` by the worker
    itself; returns the task's own seconds."""
    import pickle

    from oracle import fuse_oracle as fo
    from oracle import reg_oracle as ro

    st = _CPU_FARM.get(shm_dir)
    if st is None:
        with open(os.path.join(shm_dir, "meta.pkl"), "rb") as f:
            meta = pickle.load(f)
        views = [dict(v, data=np.load(os.path.join(shm_dir, f"view{i}.npy"), mmap_mode="r")) for i, v in enumerate(meta["views"])]
        pairs = [(np.load(os.path.join(shm_dir, f"pair{k}a.npy"), mmap_mode="r"), np.load(os.path.join(shm_dir, f"pair{k}b.npy"), mmap_mode="r"))
                 for k in range(meta["n_pairs"])]
        st = _CPU_FARM[shm_dir] = (views, meta, pairs)
    views, meta, pairs = st
    t0, c0 = time.perf_counter(), time.process_time()
    if kind == 1:
        a, b = pairs[index]
        ro.phase_correlation_registration(np.array(a), np.array(b))
    elif kind == 0:
        sb = meta["subs"][index]
        v, p, b = _slabs_for_chunk(views, meta["params"], meta["bbs"], sb)
        fo.fuse_np(list(v), p, sb, full_view_bbs=list(b))
    return time.perf_counter() - t0, time.process_time() - c0      # (wall, CPU) seconds of this task


def _cpu_pair_task(a, b):
    from oracle import reg_oracle as ro

    ro.phase_correlation_registration(a, b)
    return 0.0


def cpu_baseline(args, grid, tile, overlap):
    """Time the numpy/scipy oracle (the reference's own scipy calls) on a bounded sample of the same workload in the
    metric's unit: a 2x2x2 mosaic of small tiles with the same overlap fraction, fused whole, plus its 12
    face-neighbour registrations (one pair per axis orientation is timed, x4) -- once on ONE core and once farmed over
    all host cores with joblib/loky, the reference's own recipe for parallel fusion (misc_utils.py:184-209,
    docs/fusion_overview.md:185-203).  The all-core leg runs R independent replicas of the sample job, each cut into 64
    output-chunk tasks + 12 pair tasks, with R chosen so that there are at least twice as many tasks as cores (a farm of
    20 tasks on 256 cores would measure the task count, not the host).  One pair with the NORTH-STAR crop size (binned
    overlap of two 512^3 tiles: 51 x 256 x 256) is timed on one core as well."""
    from multiview_stitcher_amd import sample_data
    from oracle import fuse_oracle as fo
    from oracle import reg_oracle as ro
    from tests.helpers import sim_to_view, squeeze_field, union_bb

    # the CPU legs run with the affinity the process was STARTED with: the block of CPUs main() pinned the GPU legs to would take
    # cache away from the farm (16 workers inside two CCDs instead of spread over the host: 16.1 -> 12.9-13.8 Mvoxels/s measured)
    if _ORIG_AFFINITY and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, _ORIG_AFFINITY)
        except OSError:
            pass

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
    t_pairs, pairs = [], []
    for axis in range(3):                           # overlap crops of a z-, y- and x-face pair
        a, b = ro.make_pair_for_bench(ts, ov, seed=3 + axis)
        a, b = np.ascontiguousarray(np.moveaxis(a, -1, axis)), np.ascontiguousarray(np.moveaxis(b, -1, axis))
        pairs.append((a, b))
        t0 = time.perf_counter()
        ro.phase_correlation_registration(a, b)
        t_pairs.append(time.perf_counter() - t0)
    t_reg = 4.0 * float(np.sum(t_pairs))            # 12 face pairs in a 2x2x2 grid, 4 per orientation
    # one pair at the north-star crop size: tiles binned by the reference's heuristic, overlap along the last axis
    bins = ro.get_optimal_registration_binning(tile, tile, np.ones(3), np.ones(3))
    bt = np.array([int(n) // int(bins[d]) for n, d in zip(tile, "zyx")])
    bo = np.array([int(o) // int(bins[d]) for o, d in zip(overlap, "zyx")])
    a, b = ro.make_pair_for_bench(bt, bo, seed=9)
    t0 = time.perf_counter()
    ro.phase_correlation_registration(a, b)
    t_ns_pair = time.perf_counter() - t0
    out = {"value": vox / (t_reg + t_fuse) / 1e6, "unit": "Mvoxels/s", "cores": 1, "kind": "port",
           "sample": f"oracle (numpy + scipy 1.15: the reference's own affine_transform / fft / uniform_filter / spearmanr calls), "
                     f"1 thread, on a 2x2x2 mosaic of uint16 tiles {ts.tolist()}, overlap {ov.tolist()}: fuse of the whole "
                     f"{int(vox)}-voxel mosaic {t_fuse:.1f} s + 12 pair registrations {t_reg:.1f} s (3 timed, one per axis "
                     f"orientation, {np.round(t_pairs, 2).tolist()} s, x4)",
           "fuse_only_mvoxels_s": vox / t_fuse / 1e6, "register_pair_s": float(np.mean(t_pairs)),
           "register_pair_north_star_s": t_ns_pair, "register_pair_north_star_crop": [int(v) for v in a.shape]}
    # ---- all cores: chunk / pair farm with joblib (loky processes, BLAS / pocketfft threads pinned to 1 per worker) ----
    # Workers must not wait on the parent (256 workers fed pickled slabs by one dispatcher reached an effective
    # parallelism of 12): the sample's tiles and crops are written ONCE to shared memory (/dev/shm) and every task carries three
    # small integers; the worker memory-maps the files and cuts its own slabs, like a chunk task of the reference reads its
    # slabs from the Zarr store.  Workers = physical cores; every task returns its own seconds, so that
    # parallel_efficiency = sum(task seconds) / wall says how many cores the farm really kept busy.
    shm_dir = None
    try:
        import pickle
        import shutil
        import tempfile

        from joblib import Parallel, delayed

        ncores = len(os.sched_getaffinity(0))
        try:
            import psutil

            nphys = min(ncores, psutil.cpu_count(logical=False) or ncores)
        except Exception:      # noqa: BLE001
            nphys = ncores
        shp = np.asarray(out_bb["shape"])
        ncut = 4
        cuts = [np.linspace(0, n, ncut + 1).astype(int) for n in shp]          # 4 x 4 x 4 output chunks per replica
        subs = []
        for idx in np.ndindex(ncut, ncut, ncut):
            lo = np.array([cuts[k][i] for k, i in enumerate(idx)])
            hi = np.array([cuts[k][i + 1] for k, i in enumerate(idx)])
            subs.append(fo.bb(out_bb["origin"] + lo * out_bb["spacing"], out_bb["spacing"], hi - lo))
        subs = [sb for sb in subs if _slabs_for_chunk(views, params, bbs, sb)[0]]
        shm_dir = tempfile.mkdtemp(prefix="mvs_cpu_baseline_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        for i, v in enumerate(views):
            np.save(os.path.join(shm_dir, f"view{i}.npy"), np.ascontiguousarray(v["data"]))
        for k, (a_, b_) in enumerate(pairs):
            np.save(os.path.join(shm_dir, f"pair{k}a.npy"), a_)
            np.save(os.path.join(shm_dir, f"pair{k}b.npy"), b_)
        with open(os.path.join(shm_dir, "meta.pkl"), "wb") as f:
            pickle.dump({"views": [{"origin": v["origin"], "spacing": v["spacing"]} for v in views], "params": params,
                         "bbs": list(bbs), "subs": subs, "n_pairs": len(pairs)}, f)
        per_replica = len(subs) + 12
        quota0 = _cpu_quota_cores()
        n_target = nphys if quota0 is None else max(1, min(nphys, int(np.ceil(quota0))))
        replicas = max(1, -(-8 * n_target // per_replica))       # >= 8 tasks per worker: the tail of 2.6-s pair tasks stays short
        tasks = []
        for _ in range(replicas):
            tasks += [delayed(_cpu_farm_task)(shm_dir, 1, k % 3) for k in range(12)]      # (the long tasks first)
        for _ in range(replicas):
            tasks += [delayed(_cpu_farm_task)(shm_dir, 0, k) for k in range(len(subs))]
        env = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
        for k in env:
            os.environ[k] = "1"
        try:
            quota = _cpu_quota_cores()
            nw_default = nphys if quota is None else max(1, min(nphys, int(np.ceil(quota))))
            nw = min(int(os.environ.get("MVS_CPU_WORKERS", nw_default)), len(tasks))
            with Parallel(n_jobs=nw, backend="loky", batch_size=1, pre_dispatch="all") as par:
                par([delayed(_cpu_farm_task)(shm_dir, -1, 0)] * (2 * nw))     # spawn the workers, import scipy, map the files: untimed
                t0 = time.perf_counter()
                both = par(tasks)
                t_all = time.perf_counter() - t0
                secs, cpu_secs = [b_[0] for b_ in both], [b_[1] for b_ in both]
        finally:
            for k, v in env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        out["all_cores"] = {"value": replicas * vox / t_all / 1e6, "unit": "Mvoxels/s", "cores": ncores, "physical_cores": nphys,
                            "workers": nw, "tasks": len(tasks), "replicas": replicas, "wall_s": t_all,
                            "task_seconds_sum": float(np.sum(secs)), "task_seconds_max": float(np.max(secs)),
                            "task_cpu_seconds_sum": float(np.sum(cpu_secs)),
                            # cores the farm really had: CPU seconds its tasks consumed per second of wall time (a container
                            # with a CPU quota below its visible core count shows up here: wall-based occupancy stays high,
                            # this figure does not)
                            "parallel_efficiency": float(np.sum(cpu_secs)) / t_all,
                            "worker_occupancy": float(np.sum(secs)) / t_all, "cpu_quota_cores": _cpu_quota_cores(),
                            "sample": f"{replicas} replicas of the same mosaic job, each {len(subs)} output-chunk tasks (every task cuts "
                                      f"the slabs of the views reaching into its chunk out of memory-mapped tiles, as the reference's "
                                      f"chunk tasks read theirs from Zarr) + 12 pair tasks, joblib loky, one worker per physical "
                                      f"core, one thread per worker; parallel_efficiency = sum of task seconds / wall"}
    except Exception as e:      # noqa: BLE001 - the baseline must not take the bench line down
        out["all_cores"] = {"error": repr(e)[:200]}
    finally:
        if shm_dir:
            shutil.rmtree(shm_dir, ignore_errors=True)
    return out


def csrc_digest():
    """sha256 over the library's sources: ties a committed PMC measurement to the kernels it was taken with (the GPU box
    has no .git, so the source text itself is the identity)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "multiview-stitcher_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "multiview-stitcher_amd", "csrc", "*.h"))
                    + glob.glob(os.path.join(ROOT, "multiview-stitcher_amd", "csrc", "*.inc"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def fuse_traffic_bytes(grid, tile):
    """(HBM bytes per fuse launch, where the figure comes from).  The bytes are PMC counters (FETCH_SIZE x2 per the
    calibration + WRITE_SIZE, separate rocprofv3 --pmc passes: tools/profile_round.sh) committed under profiles/ together
    with the digest of the kernel sources they were measured with; when the sources have changed since, or the workload is
    another one, the figure would be stale and None is returned instead."""
    if not (list(grid) == [4, 4, 4] and list(tile) == [512, 512, 512]):
        return None, "no PMC pass for this workload"
    for name in ("round6_fuse_traffic.json", "round5_fuse_traffic.json", "round4_fuse_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            src = f"profiles/{name} (git {t.get('git_hash', '?')}, csrc digest {t.get('csrc_digest', '?')})"
            if t.get("csrc_digest") != csrc_digest():
                return None, src + ": kernel sources changed since (current digest " + csrc_digest() + "), figure withheld"
            return t["hbm_bytes_per_launch"], src
        except (OSError, KeyError, ValueError):
            continue
    return None, "no PMC pass committed for the current kernels"


FUSE_CLASSES = ((4, "copy"), (0, "NV1"), (1, "NV2"), (2, "NV4"), (3, "NV8"))


def fuse_by_class(_lib, fusion, sims, key, device, es_in=2, es_out=2):
    """Per class of the fuse launch (copy = one view, weight > 0 everywhere: value passes through; NV1 = one-view rim boxes;
    NV2 / NV4 / NV8 = boxes seen by <= 2 / 4 / 8 views): algorithmic bytes (voxels x views read + voxels written, counted by
    the planner), the class kernel's duration ALONE (one extra launch with option serial_classes: the kernels run one after the
    other between HIP events) and the rate that gives -- so that the class furthest from the 8 TB/s ceiling is named in every line.
    In the timed loop the five kernels run side by side on forked streams (roofline.achieved is that launch)."""
    _lib.set_option("serial_classes", 1, device)
    try:
        for _ in range(2):
            out = fusion.fuse(sims, transform_key=key, output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=device)
            total_ms = _lib.last_kernel_ms(device)
            del out
        res = {}
        for k, name in FUSE_CLASSES:
            in_vox = _lib.get_counter(f"fuse_class_in_vox_{k}", device)
            out_vox = _lib.get_counter(f"fuse_class_out_vox_{k}", device)
            ms = _lib.get_counter(f"fuse_class_ms_{k}", device)
            if out_vox <= 0:
                continue
            byt = in_vox * es_in + out_vox * es_out
            gbs = byt / (ms * 1e-3) / 1e9 if ms > 0 else None
            res[name] = {"output_voxels": out_vox, "algorithmic_bytes": byt, "ms_alone": ms if ms > 0 else None, "GB/s": gbs,
                         "frac": gbs / HBM_PEAK_GBS if gbs else None}
        res["serial_launch_ms"] = total_ms
        return res
    finally:
        _lib.set_option("serial_classes", 0, device)


def c3_content_based_leg(torch, dev, local_rank, args):
    """BASELINE.json config C3 -- 4 x 4 x 2 (x, y, z) grid of 256 x 512 x 512 uint16 tiles, content-based weights (weights.py:22-74,
    sigma 5 / 11, halo 22) in the reference's default 256^3 chunks -- as its own leg of the bench line: fuse() of the whole mosaic
    with the tiles resident in HBM, timed after one warm call; algorithmic bytes per SURVEY 8d (every input voxel once + every output
    voxel once + 16 B per voxel of every (halo chunk, view) box: two filters x (read + write) x 4 B at minimum); and the oracle's
    (scipy's) time for one halo chunk of a small mosaic with 8 views beside it, on one core."""
    from multiview_stitcher_amd import _lib, fusion
    from multiview_stitcher_amd import spatial_image_utils as si

    grid, tile = np.array([2, 4, 4]), np.array([256, 512, 512])
    overlap = np.round(tile * args.overlap_frac).astype(int)
    tiles, _, org = make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=7, max_jitter=0)
    sims = build_sims(tiles, org, local_rank)
    torch.cuda.synchronize()
    cs, halo = 256, 22
    for key in ("cb_line_launches", "cb_overflows_redone"):
        _lib.get_counter(key, local_rank, reset=True)
    ms = []
    for rep in range(4):
        _lib.synchronize(local_rank)
        t0 = time.perf_counter()
        out = fusion.fuse(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, weights_func=fusion.content_based,
                          output_chunksize={d: cs for d in "zyx"}, output_on_backend=True, device=local_rank)
        _lib.synchronize(local_rank)
        ms.append((time.perf_counter() - t0) * 1e3)
        shape = np.array(out.shape)
        del out
    launches = _lib.get_counter("cb_line_launches", local_rank, reset=True) / len(ms)
    redone = _lib.get_counter("cb_overflows_redone", local_rank, reset=True)
    t_ms = float(np.mean(ms[1:]))
    fo_ = np.min(np.array(org), axis=0)
    box_vox, nchunks = 0.0, 0
    for cz in range(0, int(shape[0]), cs):
        for cy in range(0, int(shape[1]), cs):
            for cx in range(0, int(shape[2]), cs):
                c0 = np.array([cz, cy, cx])
                c1 = np.minimum(c0 + cs, shape)
                nchunks += 1
                for o in org:
                    lo = np.maximum(np.round(np.array(o) - fo_).astype(int), c0 - halo)
                    hi = np.minimum(np.round(np.array(o) - fo_).astype(int) + tile, c1 + halo)
                    if np.all(hi > lo):
                        box_vox += float(np.prod(hi - lo))
    out_vox = float(np.prod(shape))
    alg = float(len(tiles)) * float(np.prod(tile)) * 2 + out_vox * 2 + 16.0 * box_vox
    achieved = alg / (t_ms * 1e-3) / 1e9
    leg = {
        "workload": "C3: 4x4x2 (x,y,z) grid of 256x512x512 uint16 tiles, 20% overlap, weights_func=content_based (sigma 5 / 11), "
                    "256^3 output chunks + 22 px halo, tiles and result resident in HBM",
        "output_shape": [int(v) for v in shape], "chunks": nchunks, "ms": t_ms, "first_call_ms": float(ms[0]),
        "mvoxels_s": out_vox / (t_ms * 1e-3) / 1e6, "line_launches_per_fuse": launches, "chunks_redone_on_exact_passes": redone,
        "roofline": {"bound": "hbm", "kernel": "content-based fuse() call: crop / blend / normalise, mask scan, 6 line passes, final sum, per chunk",
                     "algorithmic_bytes": alg, "model": "SURVEY 8d: inputs once + output once + 16 B x voxels of every (halo chunk, view) box",
                     "box_voxel_views": box_vox, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS},
    }
    del tiles, sims
    if not args.no_cpu_baseline:
        try:
            leg["cpu_baseline"] = _cpu_content_based()
        except Exception as e:   # noqa: BLE001
            leg["cpu_baseline"] = {"error": repr(e)[:200]}
    return leg


def _cpu_content_based(ts=104):
    """The oracle (scipy's gaussian_filter / affine_transform, one core) on ONE halo chunk seen by 8 views: 2x2x2 tiles of ts^3."""
    from multiview_stitcher_amd import sample_data
    from oracle import fuse_oracle as fo
    from tests.helpers import sim_to_view, squeeze_field, union_bb

    ov = max(int(ts * 0.2), 1)
    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(ts,) * 3, tiles=(2, 2, 2), overlap=(ov,) * 3, dtype=np.uint16, seed=5)
    sims = [squeeze_field(s) for s in sims]
    views, bbs = zip(*[sim_to_view(s) for s in sims])
    params = [np.eye(4) for _ in sims]
    out_bb = union_bb(bbs, params, np.ones(3))
    t0 = time.perf_counter()
    fused = fo.fuse_np(list(views), params, out_bb, full_view_bbs=list(bbs), weights="content_based",
                       weights_kwargs={"sigma_1": 5, "sigma_2": 11}, trim_overlap_in_pixels=22)
    dt = time.perf_counter() - t0
    fused = fused[0] if isinstance(fused, tuple) else fused
    vox = float(np.prod(np.asarray(fused).shape))
    return {"value": vox / dt / 1e6, "unit": "Mvoxels/s", "cores": 1, "kind": "port", "seconds": dt,
            "sample": f"oracle fuse_np(weights='content_based', sigma 5 / 11) of one chunk of {[int(v) for v in out_bb['shape']]} voxels incl. the 22 px halo, "
                      f"8 views (2x2x2 uint16 tiles of {ts}^3), {int(vox)} voxels after the trim"}


class _original_affinity:
    """The legs whose work is done by host I/O threads (chunk files, staging copies) make those threads with the CPU affinity the process
    was STARTED with: the block of 16 CPUs main() pinned the registration loop to is two CCDs of the host, whose links to memory carry
    ~2/3 of what sixteen copying threads ask for (C5 leg, same box: 0.46-0.52 s inside the block, 0.39-0.42 s outside).  An application
    that streams data does not pin itself to two CCDs."""

    def __enter__(self):
        self.pinned = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set()
        if _ORIG_AFFINITY and hasattr(os, "sched_setaffinity"):
            try:
                os.sched_setaffinity(0, _ORIG_AFFINITY)
            except OSError:
                pass
        return self

    def __exit__(self, *exc):
        if self.pinned and hasattr(os, "sched_setaffinity"):
            try:
                os.sched_setaffinity(0, self.pinned)
            except OSError:
                pass
        return False


def host_arrays_leg(sims, key, device, key_reg_in=None):
    """``fusion.fuse`` the way a user of the reference calls it: the tiles are plain (pageable) numpy arrays in host memory, the result is
    one.  Both transfers are inside the figure; the launch blocks go through the block pipeline (pinned staging by the I/O pool,
    asynchronous transfers under the blocks)."""
    from multiview_stitcher_amd import fusion, registration

    host_sims = [s_.copy(data=np.array(s_.data.get())) for s_ in sims]
    reg_ms = []
    for _ in range(2):      # register() of the same host tiles (staged through pinned buffers, uploaded while the pairs start)
        t0 = time.perf_counter()
        registration.register(host_sims, transform_key=key_reg_in, new_transform_key="registered_host", device=device)
        reg_ms.append((time.perf_counter() - t0) * 1e3)
    ms = []
    for _ in range(2):
        t0 = time.perf_counter()
        out = fusion.fuse(host_sims, transform_key=key, device=device)
        a = np.asarray(out.data)
        ms.append((time.perf_counter() - t0) * 1e3)
        vox, out_gb = float(a.size), a.nbytes / 1e9
        del out, a
    in_gb = sum(s_.data.nbytes for s_ in host_sims) / 1e9
    return {"workload": "fusion.fuse(sims, transform_key=...) of the registered north-star mosaic with the 64 tiles as pageable numpy arrays, result a numpy array",
            "ms": ms[-1], "first_call_ms": ms[0], "mvoxels_s": vox / (ms[-1] * 1e-3) / 1e6, "in_gb": in_gb, "out_gb": out_gb,
            "gb_per_s": (in_gb + out_gb) / (ms[-1] * 1e-3), "floor_ms_h2d_at_55_gb_s": in_gb / 55.0 * 1e3,
            "register_ms": reg_ms[-1], "register_first_call_ms": reg_ms[0],
            "note": "register() and fuse() each upload the tiles (no device copy is kept between two calls on host arrays: device.to_device / "
                    "to_device_async make resident tiles both calls share)"}


def c5_stream_leg(torch, dev, local_rank, args):
    """BASELINE.json config C5 (exaSPIM-style 512 x 1024 x 1024 uint16 tiles streamed from Zarr, chunked fuse) on a z-slab of its
    grid: 1 x 2 x 3 tiles written as Zarr arrays (128^3 chunks, uncompressed) on local disk, ``fusion.fuse`` from the Zarr-backed
    sims into an output Zarr store in the reference's default 256^3 chunks.  fuse() streams the launch blocks through
    streaming.BlockPipeline (reader thread: chunk files -> pinned -> async upload; fuse on resident slabs; writer thread: async
    download -> chunk files).  The rate is (tile bytes + output bytes) / wall; the ceiling is the SLOWEST stage run alone on the same
    files with the same I/O pool: reading every tile once into pinned memory, writing the result once, and the two PCIe
    directions at the rate the PCIe leg measured -- a perfectly overlapped pipeline takes max(stage), so frac = max(stage) / wall."""
    import shutil
    import tempfile

    from multiview_stitcher_amd import _lib, fusion, ngff_utils, streaming, zarr_io
    from multiview_stitcher_amd import device as dv
    from multiview_stitcher_amd import spatial_image_utils as si

    grid, tile = np.array([1, 2, 3]), np.array([512, 1024, 1024])
    overlap = np.round(tile * args.overlap_frac).astype(int)
    need = int(np.prod(tile)) * 2 * int(np.prod(grid)) * 2.2
    base = os.environ.get("MVS_BENCH_TMP")
    if base is None:
        for cand in (tempfile.gettempdir(), "/dev/shm"):
            try:
                if shutil.disk_usage(cand).free > need * 1.3:
                    base = cand
                    break
            except OSError:
                continue
    if base is None:
        return {"skipped": "no directory with %.0f GB free" % (need * 1.3 / 1e9)}
    tmp = tempfile.mkdtemp(prefix="mvs_c5_", dir=base)
    try:
        tiles, _, org = make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=11, max_jitter=0)
        lazy = []
        t_w0 = time.perf_counter()
        for i, (t, o) in enumerate(zip(tiles, org)):
            host = t.view(torch.int16).cpu().numpy().view(np.uint16)
            s_ = si.to_spatial_image(host, dims=["z", "y", "x"], scale={d: 1.0 for d in "zyx"}, translation=dict(zip("zyx", o)))
            z = ngff_utils.write_sim_to_ome_zarr(s_, os.path.join(tmp, f"tile{i}.zarr"), zarr_array_creation_kwargs={"chunks": (128, 128, 128)})
            si.set_sim_affine(z, np.eye(4), si.DEFAULT_TRANSFORM_KEY)
            lazy.append(z)
            del host
        setup_s = time.perf_counter() - t_w0
        del tiles
        torch.cuda.empty_cache()
        in_bytes = float(np.prod(tile)) * 2 * len(lazy)
        walls = []
        for rep in range(2):                 # (the first call sizes the pinned pool and the device blocks)
            out_url = os.path.join(tmp, f"fused{rep}.zarr")
            _lib.synchronize(local_rank)
            t0 = time.perf_counter()
            fused = fusion.fuse(lazy, transform_key=si.DEFAULT_TRANSFORM_KEY, output_chunksize={d: 256 for d in "zyx"},
                                output_zarr_url=out_url, zarr_options={"ome_zarr": False}, device=local_rank)
            _lib.synchronize(local_rank)
            walls.append(time.perf_counter() - t0)
            blocks = [{k: round(v, 3) for k, v in b.items()} for b in streaming.LAST_TIMELINE]
            shape = [int(v) for v in fused.data.shape[-3:]]
            if rep == 0:
                shutil.rmtree(out_url, ignore_errors=True)
        wall = walls[-1]
        out_bytes = float(np.prod(shape)) * 2
        # the stages alone, same files, same pool
        raw, buf = streaming.PinnedPool().get(tuple(int(v) for v in tile), np.uint16)
        t0 = time.perf_counter()
        for z in lazy:
            streaming.read_window(z.data, buf)
        read_s = time.perf_counter() - t0
        res_host = np.asarray(fused.data[...]) if not isinstance(fused.data, np.ndarray) else fused.data
        res_host = res_host.reshape(shape)
        arr2 = zarr_io.ZarrArray.create(os.path.join(tmp, "again.zarr"), shape, [256] * 3, np.uint16)
        t0 = time.perf_counter()
        streaming.write_region(arr2, [0, 0, 0], res_host)
        write_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        d_ = dv.DeviceArray.from_host_async(buf, local_rank)
        dv.ticket_sync(d_.ready_ticket)
        h2d_rate = float(np.prod(tile)) * 2 / (time.perf_counter() - t0)
        del d_, raw, buf
        stages = {"read_tiles_s": read_s, "write_result_s": write_s, "h2d_s": in_bytes / h2d_rate, "d2h_s": out_bytes / h2d_rate}
        floor = max(stages.values())
        return {"workload": "C5 z-slab: 1x2x3 (z,y,x) grid of 512x1024x1024 uint16 tiles, 20% overlap, Zarr in (128^3 chunks, uncompressed) -> "
                            "fuse (256^3 output chunks merged into launch blocks) -> Zarr out, on " + base,
                "output_shape": shape, "wall_s": wall, "first_call_s": walls[0], "tile_store_setup_s": setup_s,
                "gb_per_s": (in_bytes + out_bytes) / wall / 1e9, "mvoxels_s": float(np.prod(shape)) / wall / 1e6,
                "in_gb": in_bytes / 1e9, "out_gb": out_bytes / 1e9, "stages_alone": stages,
                "slowest_stage": max(stages, key=stages.get), "ceiling_gb_per_s": (in_bytes + out_bytes) / floor / 1e9,
                "frac_of_slowest_stage": floor / wall, "h2d_gb_per_s_one_tile": h2d_rate / 1e9,
                "launch_blocks": len(blocks), "block_timeline_s": blocks,
                "note": "files were written just before: reads come from the page cache (a cold disk would lower read_tiles_s' rate and the ceiling with it)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pcie_pipeline(torch, dev, local_rank, sims, tiles, args, key_in, key_out, shard=None):
    """PCIe-inclusive run of one mosaic through the PUBLIC API only: the tiles start in pinned host memory
    (device.pinned_empty) and the fused mosaic ends there.  ``device.to_device_async`` queues the uploads in tile order on the
    device's copy stream (csrc/mvs_transfer.hip), ``registration.register`` registers every pair when its two tiles have landed
    (the pair jobs carry the uploads' tickets), ``fusion.fuse_to_host`` fuses the mosaic in z slabs and downloads every finished
    slab while the next one is fused.  N > 1 (``shard`` = {rank, world, executor, dist, backend}): every rank runs this pipeline on
    ITS tiles (brick + halo, from its own pinned host memory over its own link) and ITS output sub-box
    (sharding.fuse_shard_to_host); the time is the max over ranks between two barriers, the voxels the sum."""
    from multiview_stitcher_amd import device as dv
    from multiview_stitcher_amd import _lib, fusion, registration, sharding

    host_sims = []
    for s_ in sims:
        if not dv.is_device_array(s_.data):       # (N > 1: a view another rank holds -- metadata only)
            host_sims.append(s_)
            continue
        h = dv.pinned_empty(s_.data.shape, s_.data.dtype)
        h[...] = s_.data.get()
        host_sims.append(s_.copy(data=h))
    n_slabs = 8 if shard is None else max(2, 8 // shard["world"])
    trace = {}
    out_host = [None]
    executor = shard["executor"] if shard else None

    def sync_ranks():
        if shard:
            shard["dist"].barrier()

    def run():
        registration._pair_timeline = pairs = []
        try:
            _lib.synchronize(local_rank)
            sync_ranks()
            t0 = time.perf_counter()
            m0 = dv.mark(local_rank)
            a_sims = dv.to_device_async(host_sims, local_rank)
            uploads = [a.data.ready_ticket for a in a_sims if dv.is_device_array(a.data)]
            registration.register(a_sims, transform_key=key_in, new_transform_key=key_out, device=local_rank,
                                  pre_registration_pruning_method=args.pruning, pairwise_executor=executor)
        finally:
            registration._pair_timeline = None
        t_reg = time.perf_counter()
        if shard:
            (fused, slabs), _ = sharding.fuse_shard_to_host(a_sims, shard["rank"], shard["world"], key_out, n_slabs=n_slabs, out=out_host[0],
                                                             device=local_rank, return_timeline=True)
        else:
            fused, slabs = fusion.fuse_to_host(a_sims, transform_key=key_out, n_slabs=n_slabs, out=out_host[0], device=local_rank, return_timeline=True)
        t_own = time.perf_counter()
        sync_ranks()
        t1 = time.perf_counter()
        out_host[0] = np.asarray(fused.data)
        up_ms = [dv.ticket_elapsed_ms(m0, t) for t in uploads]
        pair_ms = sorted(dv.ticket_elapsed_ms(m0, t) for _, t in pairs)
        trace.update(upload_done_ms=max(up_ms), first_pair_done_ms=pair_ms[0] if pair_ms else None, last_pair_done_ms=pair_ms[-1] if pair_ms else None,
                     pairs_done_before_last_upload=int(sum(t < max(up_ms) for t in pair_ms)), pairs=len(pair_ms),
                     slab_fused_ms=[round(f, 1) for f, _ in slabs], slab_downloaded_ms=[round(d, 1) for _, d in slabs])
        return t1 - t0, t_reg - t0, float(np.prod(out_host[0].shape)), t_own - t0

    run()                                  # warm-up (pinned result buffer, device blocks of the uploads, plans)
    for key in ("pool_misses", "pool_miss_bytes", "pool_releases"):
        _lib.get_counter(key, local_rank, reset=True)
    if os.environ.get("MVS_PCIE_PROFILE"):      # where the interpreter (and the library calls under it) spend the timed run
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        total, t_reg, vox, own = run()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
    else:
        total, t_reg, vox, own = run()
    trace["pool_misses_in_timed_run"] = _lib.get_counter("pool_misses", local_rank)
    trace["pool_miss_gb_in_timed_run"] = _lib.get_counter("pool_miss_bytes", local_rank) / 1e9
    trace["pool_releases_in_timed_run"] = _lib.get_counter("pool_releases", local_rank)
    trace["slab_fuse_call_host_ms"] = [[round(a, 1), round(b, 1)] for a, b in getattr(fusion.fuse_to_host, "last_host_ms", [])]
    h2d_gb = sum(int(np.prod(h.data.shape)) * 2 for h in host_sims if isinstance(h.data, np.ndarray)) / 1e9
    d2h_gb = vox * 2 / 1e9
    by_rank = None
    if shard:
        dist, cpu = shard["dist"], shard["backend"] != "nccl"
        t = torch.tensor([total, vox, h2d_gb, d2h_gb, own], dtype=torch.float64, device="cpu" if cpu else dev)
        g = [torch.zeros_like(t) for _ in range(shard["world"])]
        dist.all_gather(g, t)
        total = max(float(x[0]) for x in g)
        vox = sum(float(x[1]) for x in g)
        by_rank = {"own_ms": [float(x[4]) * 1e3 for x in g], "h2d_gb": [float(x[2]) for x in g], "d2h_gb": [float(x[3]) for x in g]}
        h2d_gb_all, d2h_gb_all = sum(by_rank["h2d_gb"]), sum(by_rank["d2h_gb"])
    else:
        h2d_gb_all, d2h_gb_all = h2d_gb, d2h_gb
    return {"value": vox / total / 1e6, "unit": "Mvoxels/s", "ms": total * 1e3, "register_phase_ms": t_reg * 1e3,
            "h2d_gb": h2d_gb_all, "d2h_gb": d2h_gb_all, "fuse_slabs": n_slabs, "by_rank": by_rank,
            "upload_done_ms": trace.get("upload_done_ms"), "h2d_gb_per_s": h2d_gb / (trace["upload_done_ms"] * 1e-3) if trace.get("upload_done_ms") else None,
            "d2h_gb_per_s": d2h_gb / max(own - t_reg, 1e-9), "timeline": trace,
            "api": "device.pinned_empty, device.to_device_async, registration.register, fusion.fuse_to_host"
                   + (" / sharding.fuse_shard_to_host" if shard else "") + " (no torch streams; the one private hook, "
                   "registration._pair_timeline, only collects the pairs' tickets for the timeline below)",
            "note": "tiles in pinned host memory -> uploads in tile order on the device's copy stream, overlapped with the registration of "
                    "the pairs whose tiles have arrived (pair jobs wait for their tiles' tickets on their lanes) -> resolution -> fuse in z "
                    "slabs, each slab's download overlapped with the next slab's fuse; timeline (rank 0): timed tickets, ms since the first "
                    "upload was queued" + ("; N > 1: every rank uploads its brick + halo (halo tiles cross the host link once per rank "
                                           "that needs them: h2d_gb is their sum) and downloads its sub-box" if shard else "")}


def main():
    args = parse_args()
    # one compact block of CPUs per process, before torch / HIP start their threads (executors.pin_process_to_compact_cpus: the
    # 16 pair-worker threads of a step lose ~10 % when the scheduler spreads them over both sockets of the host)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from multiview_stitcher_amd.executors import pin_process_to_compact_cpus

    # (MVS_PIN_PROCESS=0: only the library's own pair workers keep to the block -- what a caller of register() gets without asking)
    n_cpus_before = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0
    if hasattr(os, "sched_getaffinity"):
        _ORIG_AFFINITY.update(os.sched_getaffinity(0))
    pinned_cpus = pin_process_to_compact_cpus(slot=int(os.environ.get("LOCAL_RANK", "0"))) if os.environ.get("MVS_PIN_PROCESS", "1") != "0" else None
    if pinned_cpus is not None and len(pinned_cpus) >= n_cpus_before:
        pinned_cpus = None          # (left alone: already narrow, or switched off)
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
    json_fd = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            # gloo's transport prints "[Gloo] Rank ... is connected" lines on fd 1; the contract is ONE JSON line on stdout
            sys.stdout.flush()
            json_fd = os.dup(1)
            os.dup2(2, 1)
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    mode = args.mode or "shard"
    shard = world > 1 and mode == "shard"

    from multiview_stitcher_amd import _lib, fusion, mv_graph, sharding
    from multiview_stitcher_amd import spatial_image_utils as si
    from multiview_stitcher_amd.device import DeviceArray

    _lib.init(local_rank)
    grid = np.array([int(v) for v in args.grid.split(",")])
    tile = np.array([int(v) for v in args.tile.split(",")])
    overlap = np.round(tile * args.overlap_frac).astype(int)

    # shard: every rank cuts the SAME mosaic (same seed) and keeps only the tiles it owns; replica: one mosaic per rank
    tiles, jitters, origins = make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=1000 + (0 if shard else rank))
    key_in, key_out = si.DEFAULT_TRANSFORM_KEY, "registered"
    executor, halo_ms, n_held = None, None, len(tiles)
    if shard:
        sps = [{"origin": dict(zip("zyx", o)), "spacing": dict(zip("zyx", [1.0] * 3)), "shape": dict(zip("zyx", [int(v) for v in tile]))} for o in origins]
        affs = [np.eye(4) for _ in origins]
        osp0 = _union_stack(sps)
        boxes, counts = sharding.output_subboxes(osp0, world)
        owners = sharding.tile_owners(sps, affs, boxes)
        g = mv_graph.build_view_adjacency_graph([dict(sp, transform=a) for sp, a in zip(sps, affs)], overlap_tolerance=None)
        g = mv_graph.prune_view_adjacency_graph(g, args.pruning, None)
        edges = [tuple(sorted(e)) for e in g.edges()]
        needs = [sharding.rank_tiles(sps, affs, boxes, edges, owners, r, margin=8.0) for r in range(world)]
        held = [t if owners[v] == rank else None for v, t in enumerate(tiles)]
        del tiles
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        held = sharding.exchange_halo(torch, dist, held, owners, needs, rank, world, dev, via_host=(backend != "nccl"))
        torch.cuda.synchronize()
        halo_ms = (time.perf_counter() - t0) * 1e3
        tiles = held
        n_held = sum(t is not None for t in tiles)
        executor = sharding.ShardedPairExecutor(rank, world, owners, device=local_rank)
    sims = build_sims(tiles, origins, local_rank, tile_shape=tile)
    torch.cuda.synchronize()

    do_register = not args.no_register
    registration = None
    if do_register:
        from multiview_stitcher_amd import registration  # noqa: F811

    out_holder = {}
    kernel_ms, reg_ms, fuse_ms, pair_ms, plan_ms = [], [], [], [], []
    if do_register:
        inner = registration.compute_pairwise_registrations

        depth = [0]

        def timed_pairs(*a, **k):      # (the sharded executor calls the function from inside the outer call: count once)
            t0 = time.perf_counter()
            depth[0] += 1
            try:
                return inner(*a, **k)
            finally:
                depth[0] -= 1
                if depth[0] == 0:
                    pair_ms[-1] += (time.perf_counter() - t0) * 1e3
        registration.compute_pairwise_registrations = timed_pairs

    def step():
        # the previous step's mosaic is released before the next one is fused (as a consumer would), so the 10 GB
        # output buffer is recycled by the library's pool instead of being hipMalloc'ed anew every other step
        out_holder.clear()
        t_reg0 = time.perf_counter()
        key = key_in
        pair_ms.append(0.0)
        if do_register:
            registration.register(sims, transform_key=key_in, new_transform_key=key_out, device=local_rank,
                                  pre_registration_pruning_method=args.pruning, pairwise_executor=executor,
                                  n_parallel_pairwise_regs=args.reg_threads)
            key = key_out
        t_reg1 = time.perf_counter()
        if shard:
            fused, box = sharding.fuse_shard(sims, rank, world, key, output_chunksize={d: 1 << 30 for d in "zyx"},
                                             output_on_backend=True, device=local_rank)
            out_holder["box"] = box
        else:
            fused = fusion.fuse(sims, transform_key=key, output_chunksize={d: 1 << 30 for d in "zyx"},
                                output_on_backend=True, device=local_rank)
        out_holder["fused"] = fused
        kernel_ms.append(_lib.last_kernel_ms(local_rank))   # blocks until the fuse kernels are done
        plan_ms.append(_lib.get_counter("fuse_plan_ms", local_rank))
        reg_ms.append((t_reg1 - t_reg0) * 1e3)
        fuse_ms.append((time.perf_counter() - t_reg1) * 1e3)
        return fused

    # Everything alive now (torch, numpy, scipy, the tiles' metadata) is long-lived: move it to the permanent generation so
    # that the cyclic collector's occasional full passes do not walk the import graph of torch inside a step (60-70 ms each,
    # twice in 20 steps without this).  Collection itself stays enabled.
    import gc
    gc.collect()
    gc.freeze()
    if os.environ.get("MVS_BENCH_TRACE"):
        gc_t = {}

        def gc_cb(phase, info):
            if phase == "start":
                gc_t["t"] = time.perf_counter()
            else:
                print(f"gc gen{info['generation']} {1e3 * (time.perf_counter() - gc_t['t']):.1f} ms collected {info['collected']}", file=sys.stderr)
        gc.callbacks.append(gc_cb)
    # the first step of the process is the COLD one: no plan / replay memo, no pooled device blocks, the mosaic-sized result
    # hipMalloc'ed, and -- the largest part, 300-430 ms for 144 pairs -- the crop-length question asked once per pair geometry
    # (the reference's linprog + Qhull sequence, ~2-3 ms of scipy per pair; registration._reference_crop_differs, memoised;
    # MVS_KNIFE_CHECK=0 or overlap_bbox="closed_form" skips it): what a caller with ONE mosaic sees (config.step_cold_ms)
    cold = {}
    for w in range(max(args.warmup, 0)):
        t_w = time.perf_counter()
        step()
        torch.cuda.synchronize()
        if w == 0:
            cold = {"step_cold_ms": (time.perf_counter() - t_w) * 1e3, "register_first_call_ms": reg_ms[0] if do_register else None,
                    "fuse_first_call_ms": fuse_ms[0], "pairwise_first_call_ms": pair_ms[0] if do_register else None}
    cold_plan_ms = plan_ms[0] if plan_ms else None      # the first call of a geometry builds (and caches) the decomposition
    for lst in (kernel_ms, reg_ms, fuse_ms, pair_ms, plan_ms):
        lst.clear()
    for lane in range(16):
        for key in ("reg_alg_bytes", "reg_alg_bytes_full", "reg_pairs", "reg_candidates", "reg_pruned", "reg_cand_volumes"):
            _lib.get_counter(key, local_rank | (lane << 8), reset=True)
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
    if os.environ.get("MVS_BENCH_TRACE"):
        print("per-step ms: register", [round(v, 1) for v in reg_ms], "fuse", [round(v, 1) for v in fuse_ms], file=sys.stderr)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    reg_bytes = sum(_lib.get_counter("reg_alg_bytes", local_rank | (lane << 8)) for lane in range(16))
    reg_bytes_full = sum(_lib.get_counter("reg_alg_bytes_full", local_rank | (lane << 8)) for lane in range(16))
    reg_pairs = sum(_lib.get_counter("reg_pairs", local_rank | (lane << 8)) for lane in range(16))
    reg_cands = sum(_lib.get_counter("reg_candidates", local_rank | (lane << 8)) for lane in range(16))
    reg_pruned = sum(_lib.get_counter("reg_pruned", local_rank | (lane << 8)) for lane in range(16))
    reg_cand_vols = sum(_lib.get_counter("reg_cand_volumes", local_rank | (lane << 8)) for lane in range(16))

    fused = out_holder["fused"]
    out_shape = fused.shape
    dump_dir = os.environ.get("MVS_BENCH_DUMP")
    if dump_dir:      # (tests) every rank leaves its fused sub-box and where it sits in the mosaic
        box = out_holder.get("box")
        off = [int(box["index_offset"][d]) for d in "zyx"] if box is not None else [0, 0, 0]
        np.save(os.path.join(dump_dir, f"fused_rank{rank}of{world}.npy"), np.asarray(fused.data))
        with open(os.path.join(dump_dir, f"fused_rank{rank}of{world}.json"), "w") as f:
            json.dump({"index_offset": off, "shape": [int(v) for v in out_shape], "pairs_registered": int(reg_pairs),
                       "tiles_held": int(n_held), "halo_exchange_ms": halo_ms, "mode": "shard" if shard else "replica"}, f)
    out_vox_local = float(np.prod(out_shape))
    es = 2
    # algorithmic bytes of THIS rank's fuse launch: every input voxel that reaches into its output box once + every
    # output voxel once (N = 1: all tiles + the whole mosaic)
    fo_ = si.get_origin_from_sim(fused, asarray=True)
    in_vox_local = 0.0
    from multiview_stitcher_amd import param_utils
    for s_, t_ in zip(sims, tiles):
        if t_ is None:
            continue
        p = param_utils.select_time(si.get_affine_from_sim(s_, key_out if do_register else key_in), 0)
        lo = si.get_origin_from_sim(s_, asarray=True) + p[:3, 3]
        hi = lo + (tile - 1)
        ilo, ihi = np.maximum(lo, fo_), np.minimum(hi, fo_ + np.asarray(out_shape) - 1)
        if np.all(ihi >= ilo):
            in_vox_local += float(np.prod(np.floor(ihi - ilo) + 1))
    alg_bytes = in_vox_local * es + out_vox_local * es
    k_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    if world > 1:
        tv = torch.tensor([out_vox_local], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tv, op=dist.ReduceOp.SUM)
        out_vox_total = float(tv.item())
    else:
        out_vox_total = out_vox_local
    value = out_vox_total / (elapsed / args.steps) / 1e6

    reg_err = None
    if do_register:
        rec = np.array([param_utils.select_time(si.get_affine_from_sim(s, key_out), 0)[:3, 3] for s in sims])
        reg_err = float(np.max(np.abs((rec - rec[0]) - (jitters - jitters[0]))))   # relative to tile 0: the resolver fixes its own reference view
    pair_wall_ms = float(np.mean(pair_ms)) if do_register else None      # (before the PCIe leg, which registers again)
    # fuse() as a user of the reference calls it: the default 256^3 output_chunksize (merged into launch blocks by fuse())
    # (two calls: the first may have to hipMalloc a second mosaic-sized result -- ~240 ms for 10.7 GB -- while the timed loop's result
    # is still alive; the pool hands the block back to the second call)
    default_chunks_ms = default_chunks_first_ms = None
    if world == 1:
        for rep in range(2):
            _lib.synchronize(local_rank)
            t_d = time.perf_counter()
            f_d = fusion.fuse(sims, transform_key=key_out if do_register else key_in, output_on_backend=True, device=local_rank)
            _lib.synchronize(local_rank)
            ms_d = (time.perf_counter() - t_d) * 1e3
            del f_d
            if rep == 0:
                default_chunks_first_ms = ms_d
            default_chunks_ms = ms_d
    # a mosaic whose GEOMETRY has not been seen before (all stage positions moved by 3 px: same tiles, same relative layout, other
    # absolute coordinates), process warm: nothing is replayed -- fuse() derives its plan, and register() asks the crop-length question
    # of every pair (~2 ms of scipy each, on a thread of its own next to the GPU's pair loop).  What a series of DIFFERENT mosaics pays
    # per mosaic, between the cold first call and the steady state of one repeated geometry that `value` is.
    new_geo = {}
    if world == 1 and do_register:
        try:
            sims_b = build_sims(tiles, origins + 3.0, local_rank, tile_shape=tile)
            out_holder.clear()
            _lib.synchronize(local_rank)
            t_a = time.perf_counter()
            registration.register(sims_b, transform_key=key_in, new_transform_key=key_out, device=local_rank,
                                  pre_registration_pruning_method=args.pruning, n_parallel_pairwise_regs=args.reg_threads)
            t_b = time.perf_counter()
            f_b = fusion.fuse(sims_b, transform_key=key_out, output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=local_rank)
            _lib.synchronize(local_rank)
            new_geo = {"step_new_geometry_ms": (time.perf_counter() - t_a) * 1e3, "register_new_geometry_ms": (t_b - t_a) * 1e3}
            del f_b, sims_b
        except Exception as e:   # noqa: BLE001 - an optional figure must not take the bench line down
            new_geo = {"step_new_geometry_error": repr(e)[:200]}
    by_class = None
    if world == 1:
        try:
            by_class = fuse_by_class(_lib, fusion, sims, key_out if do_register else key_in, local_rank)
        except Exception as e:   # noqa: BLE001 - an optional leg must not take the bench line down
            by_class = {"error": repr(e)[:300]}
    pcie = None
    if (world == 1 or shard) and do_register and not args.no_pcie:
        out_holder.clear()
        shard_ctx = {"rank": rank, "world": world, "executor": executor, "dist": dist, "backend": backend} if shard else None
        if shard:
            # (collective: every rank enters it; a rank that fails says so to the others before anybody reads the result, so that the
            # line of the main loop -- what a scaling run is after -- is printed either way.  MVS_BENCH_PCIE_SHARDED=0 skips the leg.)
            err = None
            if os.environ.get("MVS_BENCH_PCIE_SHARDED", "1") != "0":
                try:
                    pcie = pcie_pipeline(torch, dev, local_rank, sims, tiles, args, key_in, key_out, shard=shard_ctx)
                except Exception as e:   # noqa: BLE001
                    err = repr(e)[:300]
                flag = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag.item()) < 1.0:
                    pcie = {"error": err or "another rank failed in the PCIe-inclusive leg"}
        else:
            try:
                pcie = pcie_pipeline(torch, dev, local_rank, sims, tiles, args, key_in, key_out)
            except Exception as e:   # noqa: BLE001 - an optional leg must not take the bench line down
                pcie = {"error": repr(e)[:300]}
    c3 = None
    if world == 1 and not args.no_c3:
        try:
            c3 = c3_content_based_leg(torch, dev, local_rank, args)
        except Exception as e:   # noqa: BLE001 - an optional leg must not take the bench line down
            c3 = {"error": repr(e)[:300]}
    host_fuse = None
    if world == 1 and not args.no_pcie:
        out_holder.clear()
        try:
            with _original_affinity():
                host_fuse = host_arrays_leg(sims, key_out if do_register else key_in, local_rank, key_reg_in=key_in)
        except Exception as e:   # noqa: BLE001 - an optional leg must not take the bench line down
            host_fuse = {"error": repr(e)[:300]}
    c5 = None
    if world == 1 and not args.no_c5:
        out_holder.clear()
        try:
            with _original_affinity():
                c5 = c5_stream_leg(torch, dev, local_rank, args)
        except Exception as e:   # noqa: BLE001 - an optional leg must not take the bench line down
            c5 = {"error": repr(e)[:300]}
    traffic, traffic_src = fuse_traffic_bytes(grid, tile) if world == 1 else (None, "N > 1")
    # per-rank phase figures (own step time: register + fuse of this rank, without the other ranks' tail)
    own_ms = float(np.mean(reg_ms)) + float(np.mean(fuse_ms))
    own_serial = own_ms - (pair_wall_ms or 0.0) - k_ms
    own_pairs = reg_pairs / max(args.steps, 1) if do_register else 0.0
    if world > 1:
        tg = torch.tensor([own_serial, own_pairs], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(tg) for _ in range(world)]
        dist.all_gather(gathered, tg)
        serial_by_rank = [float(g[0].item()) for g in gathered]
        pairs_by_rank = [float(g[1].item()) for g in gathered]
    else:
        serial_by_rank, pairs_by_rank = [own_serial], [own_pairs]
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
            # what `--gpus N` does with N > 1: 'shard' (the default) cuts ONE mosaic over the ranks = strong scaling; the N = 1 line
            # carries the label of the mode its N > 1 siblings will be compared under
            "scaling": "weak" if mode == "replica" else "strong",
            "vs_baseline": None,
            "dtype": "u16 in/out, f32 accumulate, f64 coordinates",
            "data": "synthetic (seeded smoothed noise mosaic generated in HBM, integer jitter unknown to metadata)",
            "host": "python gc.freeze() after setup (garbage collection stays enabled); CPU affinity of the process: "
                    + _cpu_ranges(os.sched_getaffinity(0)) + (" (executors.pin_process_to_compact_cpus)" if pinned_cpus else "")
                    + ("; pair workers unpinned" if os.environ.get("MVS_PIN_CPUS", "") == "0" else "; pair workers on one block of CPUs"),
            "value_incl_pcie": pcie,
            "config": {
                "workload": f"{'x'.join(map(str, grid))} grid (z,y,x) of {'x'.join(map(str, tile))} uint16 tiles, "
                            f"{int(args.overlap_frac * 100)}% overlap, "
                            + (f"register (overlap graph, pre_registration_pruning_method={args.pruning!r}, phase-correlation "
                               f"registration of the kept pairs, global_optimization resolution) + " if do_register else "")
                            + "cosine-blend weighted-average fuse; "
                            + ("ONE mosaic sharded over the ranks (tile bricks + halo, pairs balanced over the owners of their two views, output sub-boxes)"
                               if shard else "one mosaic per GPU"),
                "mode": "shard" if shard else ("replica" if world > 1 else "single"),
                "output_shape_rank0": [int(s) for s in out_shape],
                "tiles_held_rank0": int(n_held),
                "halo_exchange_ms": halo_ms,
                "register_ms_per_step": float(np.mean(reg_ms)) if do_register else None,
                "pairwise_ms_per_step": pair_wall_ms,
                "pairs_per_step_rank0": reg_pairs / max(args.steps, 1) if do_register else None,
                "scored_candidates_per_pair": (reg_cands / reg_pairs) if reg_pairs else None,
                "ssim_prune": os.environ.get("MVS_SSIM_PRUNE", "1") != "0",
                "candidates_left_unfinished_per_pair": (reg_pruned / reg_pairs) if reg_pairs else None,
                "candidate_volumes_walked_per_pair": (reg_cand_vols / reg_pairs) if reg_pairs else None,
                "fuse_ms_per_step": float(np.mean(fuse_ms)),
                "fuse_kernel_ms": k_ms,
                # what the host does serially around the device work of a step: overlap graph + pruning, queueing the binning,
                # assembling the pair jobs, groupwise resolution, writing the transforms back, fuse()'s host path
                "serial_host_ms": (ms_per_step - pair_wall_ms - k_ms) if do_register else (ms_per_step - k_ms),
                "serial_host_ms_by_rank": serial_by_rank,
                "pairs_per_step_by_rank": pairs_by_rank,
                "fuse_host_replay": bool(fusion._REPLAY[0]),
                "fuse_plan_cold_ms": cold_plan_ms,
                # first register() + fuse() of this process (library loaded, nothing cached, result buffer not yet allocated):
                # the figures of a one-mosaic caller; `value` / ms_per_step are the steady state of a series of mosaics
                "step_cold_ms": cold.get("step_cold_ms"),
                "register_first_call_ms": cold.get("register_first_call_ms"),
                "pairwise_first_call_ms": cold.get("pairwise_first_call_ms"),
                "fuse_first_call_ms": cold.get("fuse_first_call_ms"),
                # warm process, a geometry not seen before (nothing replayed, the crop-length question asked for every pair)
                "step_new_geometry_ms": new_geo.get("step_new_geometry_ms"),
                "register_new_geometry_ms": new_geo.get("register_new_geometry_ms"),
                "fuse_default_chunksize_ms": default_chunks_ms,
                "fuse_default_chunksize_first_call_ms": default_chunks_first_ms,
                "registration_max_abs_error_px": reg_err,
                "c3_fuse_mvoxels_s": c3.get("mvoxels_s") if c3 else None,
            },
            "c3_content_based": c3,
            "c5_stream": c5,
            "fuse_host_arrays": host_fuse,
            "roofline": {
                "bound": "hbm",
                "kernel": "fuse launch of rank 0 = copy_region_kernel + fuse_region_kernel<1|2|4|8> (u16) side by side on forked streams, timed as one unit (first start to last end)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "copy_ceiling": HBM_COPY_CEILING_GBS,
                "frac_of_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes,
                "by_class": by_class,
            },
            "roofline_register": None if not (do_register and reg_pairs) else {
                "bound": "hbm",
                "kernel": "pairwise registrations of rank 0 (binning, crops, FFTs, cross power, argmax, upsampled DFT, shifts, SSIM, ranks)",
                "algorithmic_bytes_per_step": reg_bytes / args.steps,
                "model": "SURVEY 8d: per pair of n binned overlap voxels 2 x 28 n (phase correlation, two normalisations) + 20 n per candidate VOLUME the SSIM walk went through (a candidate the pruned arg-max search stopped counts the fraction it was scored on: config.candidate_volumes_walked_per_pair) + 64 n (rank correlation)",
                "duration_ms": pair_wall_ms,
                "achieved": reg_bytes / args.steps / (pair_wall_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": reg_bytes / args.steps / (pair_wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "reference_formulation": {"algorithmic_bytes_per_step": reg_bytes_full / args.steps,
                                          "frac": reg_bytes_full / args.steps / (pair_wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "note": "the same model with every scored candidate counted whole (20 n each), i.e. the bytes the reference's "
                                                  "formulation of the same result moves, over the same duration: the rate at which this path does the "
                                                  "reference's job, not a measure of the kernels' own traffic"},
                "note": "duration = wall time of compute_pairwise_registrations (kernels of the context lanes (8 by default) overlap; includes host round trips)",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, grid, tile, overlap)
        else:
            result["cpu_baseline"] = None
        if json_fd is not None:
            os.write(json_fd, (json.dumps(result) + "\n").encode())
        else:
            print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def _union_stack(sps):
    """Output stack of axis-aligned unit-spacing views: the union box (fusion.calc_fusion_stack_properties on metadata)."""
    lo = np.min([[sp["origin"][d] for d in "zyx"] for sp in sps], axis=0)
    hi = np.max([[sp["origin"][d] + (sp["shape"][d] - 1) * sp["spacing"][d] for d in "zyx"] for sp in sps], axis=0)
    shape = np.floor((hi - lo) + 1e-9).astype(int) + 1
    return {"origin": dict(zip("zyx", lo.tolist())), "spacing": dict(zip("zyx", [1.0] * 3)), "shape": dict(zip("zyx", shape.tolist()))}


if __name__ == "__main__":
    main()
