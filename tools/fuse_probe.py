"""Minimal driver for profiling the fuse kernel alone (few other kernels, so PMC passes stay short).
usage: python tools/fuse_probe.py [reps] [frac]   (frac=1: fractional offsets, 2: integer jitter of +-3 px)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multiview_stitcher_amd import _lib, fusion, spatial_image_utils as si
from multiview_stitcher_amd.device import DeviceArray

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frac = int(sys.argv[2]) if len(sys.argv) > 2 else 0
grid = np.array([int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "4,4,4").split(",")])
tile = np.array([int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "512,512,512").split(",")])
dev = torch.device("cuda", 0)
_lib.init(0)
ov = int(sys.argv[5]) if len(sys.argv) > 5 else None      # overlap in voxels (default 20 %); 104 makes every cell boundary of 512-wide tiles a multiple of 8
step = tile - (np.round(tile * 0.2).astype(int) if ov is None else ov)
rng = np.random.default_rng(0)
sims, keep = [], []
for idx in np.ndindex(*grid):
    t = torch.randint(0, 4096, tuple(int(s) for s in tile), dtype=torch.int16, device=dev).view(torch.uint16)
    keep.append(t)
    da = DeviceArray.from_pointer(t.data_ptr(), tuple(t.shape), np.uint16, 0, owner=t)
    o = np.asarray(idx) * step
    sim = si.to_spatial_image(da, dims=["z", "y", "x"], scale=dict(zip("zyx", [1.0] * 3)), translation=dict(zip("zyx", o.astype(float))))
    p = np.eye(4)
    if frac == 1:
        p[:3, 3] = np.round(rng.uniform(-2, 2, 3), 3)
    elif frac == 2:                                     # integer jitter, like a registered mosaic
        p[:3, 3] = rng.integers(-3, 4, 3).astype(float)
    si.set_sim_affine(sim, p, "k")
    sims.append(sim)
torch.cuda.synchronize()
_lib.set_option("ablate", int(os.environ.get("MVS_ABLATE", "0")))
_lib.set_option("rows_v1", int(os.environ.get("MVS_ROWS_V1", "0")))               # 1: direct-load row kernels before the region kernels
_lib.set_option("serial_classes", int(os.environ.get("MVS_SERIAL", "0")))   # 1: class kernels one after the other (A/B of the side streams)
osp = None
if os.environ.get("MVS_PAD_X") or os.environ.get("MVS_SHIFT_X"):      # alignment experiments: output rows padded to a multiple of
    osp = fusion.process_output_stack_properties(sims, transform_key="k")      # MVS_PAD_X voxels, output origin moved MVS_SHIFT_X voxels to the left
    osp = {k: dict(v) for k, v in osp.items()}
    sh = int(os.environ.get("MVS_SHIFT_X", "0"))
    osp["origin"]["x"] -= sh * osp["spacing"]["x"]
    pad = int(os.environ.get("MVS_PAD_X", "1"))
    osp["shape"]["x"] = -(-(int(osp["shape"]["x"]) + sh) // pad) * pad
ms = []
for _ in range(reps):
    out = fusion.fuse(sims, transform_key="k", output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=0, output_stack_properties=osp)
    ms.append(_lib.last_kernel_ms(0))
vox = float(np.prod(out.shape)) if osp is None else float(np.prod(out.shape[:2])) * float(out.shape[2] - (int(os.environ.get("MVS_PAD_X", "1")) > 1) * 0)
byt = len(sims) * float(np.prod(tile)) * 2 + vox * 2
print("shape", out.shape, "kernel ms", ms, "GB/s", byt / (min(ms) * 1e-3) / 1e9)
if os.environ.get("MVS_SERIAL", "0") == "1":      # the class kernels ran one after the other: their own durations and rates
    for k, name in ((4, "copy"), (0, "NV1"), (1, "NV2"), (2, "NV4"), (3, "NV8")):
        iv, ov_, t = (_lib.get_counter(f"fuse_class_{w}_{k}") for w in ("in_vox", "out_vox", "ms"))
        if ov_ > 0:
            print("   class %-4s out %7.1f Mvox  alg %6.2f GB  alone %6.3f ms  %6.0f GB/s" % (name, ov_ / 1e6, (iv + ov_) * 2 / 1e9, t, (iv + ov_) * 2 / (t * 1e-3) / 1e9))
if os.environ.get("MVS_COMPARE"):        # the same mosaic through the generic kernel: must agree up to the knife-edge voxels
    a = out.data.get().astype(np.int32)
    _lib.set_option("rows_v1", 0); _lib.set_option("force_generic", 1)
    ref = fusion.fuse(sims, transform_key="k", output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=0)
    print("generic kernel ms", _lib.last_kernel_ms(0))
    d = a - ref.data.get().astype(np.int32)
    nz = np.argwhere(d != 0)
    print("compare: differing voxels", int((d != 0).sum()), "max |d|", int(np.abs(d).max()), "of", d.size)
    for p in nz[:12]:
        print("   ", tuple(int(v) for v in p), int(a[tuple(p)]), int(a[tuple(p)] - d[tuple(p)]))
