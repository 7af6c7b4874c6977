"""hipSetDeviceFlags(schedule mode) before anything else touches the device, then register() on the north-star grid:
python tools/sched_probe.py [auto|spin|yield|block] [lanes] [reps]     (MVS_NO_BATCH=1: the per-pair interpreter threads)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_stitcher_amd.executors import pin_process_to_compact_cpus
pin_process_to_compact_cpus()
mode = sys.argv[1] if len(sys.argv) > 1 else "auto"
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
flag = {"auto": 0, "spin": 1, "yield": 2, "block": 4, "blocking": 4}[mode]
hip = ctypes.CDLL("libamdhip64.so")
rc = hip.hipSetDeviceFlags(ctypes.c_uint(flag))
import numpy as np, torch
import bench
from multiview_stitcher_amd import _lib, registration
from multiview_stitcher_amd import spatial_image_utils as si
dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
if os.environ.get("MVS_RAW_CROPS") == "0":
    registration._raw_crops_enabled[0] = False
if os.environ.get("MVS_NO_BATCH"):
    registration._batch_enabled[0] = False
import gc
walls, cpus = [], []
for rep in range(reps + 3):
    c0, t0 = time.process_time(), time.perf_counter()
    registration.register(sims, transform_key=key, new_transform_key="reg", device=0, pre_registration_pruning_method="alternating_pattern", n_parallel_pairwise_regs=lanes)
    w, cpu = time.perf_counter() - t0, time.process_time() - c0
    if rep == 2:
        gc.collect(); gc.freeze()
    if rep >= 3:
        walls.append(w * 1e3); cpus.append(cpu * 1e3)
print(f"lanes {lanes} batch {registration._batch_enabled[0]} raw_crops {registration._raw_crops_enabled[0]} schedule {mode} (hipSetDeviceFlags rc {rc}): register ms min {min(walls):.1f} median {np.median(walls):.1f} max {max(walls):.1f}; {np.median(cpus) / np.median(walls):.1f} cores busy")
