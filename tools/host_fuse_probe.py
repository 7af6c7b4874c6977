"""fuse() of plain HOST numpy tiles into a host numpy result (what a user of the reference calls): the serial path of the library
(mvs_fuse_chunk uploads from pageable memory, fuses, downloads -- per launch block) against the block pipeline (pinned staging,
asynchronous transfers around the launch blocks).  python tools/host_fuse_probe.py [grid z,y,x] [tile z,y,x]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from multiview_stitcher_amd import _lib, fusion
from multiview_stitcher_amd import spatial_image_utils as si

grid = np.array([int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4,4,4").split(",")])
tile = np.array([int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "512,512,512").split(",")])
dev = torch.device("cuda", 0); _lib.init(0)
tiles, _, org = bench.make_mosaic_on_device(torch, dev, grid, tile, np.round(tile * 0.2).astype(int), seed=3, max_jitter=0)
sims = []
for t, o in zip(tiles, org):
    h = t.view(torch.int16).cpu().numpy().view(np.uint16)
    s = si.to_spatial_image(h, dims=["z", "y", "x"], scale={d: 1.0 for d in "zyx"}, translation=dict(zip("zyx", o)))
    si.set_sim_affine(s, np.eye(4), si.DEFAULT_TRANSFORM_KEY)
    sims.append(s)
if os.environ.get("DEVICE_TILES"):       # tiles resident on the device, result still a host array
    sims = bench.build_sims(tiles, org, 0)
else:
    del tiles
    torch.cuda.empty_cache()
in_gb = sum(int(np.prod(s.data.shape)) * 2 for s in sims) / 1e9
for mode in (os.environ.get("MODES", "serial,pipeline,serial,pipeline").split(",")):
    fusion._HOST_STREAM[0] = mode == "pipeline"
    t0 = time.perf_counter()
    out = fusion.fuse(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, device=0)
    dt = time.perf_counter() - t0
    a = np.asarray(out.data)
    print("%-8s %.3f s  %.0f Mvoxels/s  in %.1f GB out %.1f GB -> %.1f GB/s  checksum %d" % (mode, dt, a.size / dt / 1e6, in_gb, a.nbytes / 1e9, (in_gb + a.nbytes / 1e9) / dt, int(a[::7, ::11, ::13].sum())), flush=True)
    del out, a
if os.environ.get("REGISTER"):
    from multiview_stitcher_amd import registration
    for rep in range(3):
        t0 = time.perf_counter()
        registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="r", device=0)
        print("register(host numpy tiles) %.3f s" % (time.perf_counter() - t0), flush=True)
