"""Wall time and host CPU time of register() on the north-star grid (how busy are the lane threads while they wait?):
python tools/host_cpu_probe.py [lanes] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_stitcher_amd.executors import pin_process_to_compact_cpus
pin_process_to_compact_cpus()          # (before torch / HIP start their threads; MVS_PIN_CPUS=0: off)
import numpy as np, torch
import bench
from multiview_stitcher_amd import _lib, registration
from multiview_stitcher_amd import spatial_image_utils as si

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
walls, cpus = [], []
for rep in range(reps + 2):
    c0, t0 = time.process_time(), time.perf_counter()
    registration.register(sims, transform_key=key, new_transform_key="reg", device=0, pre_registration_pruning_method="alternating_pattern",
                          n_parallel_pairwise_regs=lanes)
    w, cpu = time.perf_counter() - t0, time.process_time() - c0
    if rep >= 2:
        walls.append(w * 1e3); cpus.append(cpu * 1e3)
print(f"lanes {lanes}: register wall ms min {min(walls):.1f} median {np.median(walls):.1f} max {max(walls):.1f}; process CPU ms median {np.median(cpus):.1f} "
      f"= {np.median(cpus) / np.median(walls):.1f} cores busy; env " + " ".join(f"{k}={os.environ[k]}" for k in ("ROC_ACTIVE_WAIT_TIMEOUT", "HSA_ENABLE_INTERRUPT", "GPU_MAX_HW_QUEUES", "MVS_SSIM_PRUNE") if k in os.environ))
