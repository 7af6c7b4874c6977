"""BASELINE config C4: two 512^3 uint16 views, view 2 under a full affine (rotation + tilt + anisotropic scale + shift),
weighted-average fuse through the generic affine kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multiview_stitcher_amd import _lib, fusion, param_utils
from multiview_stitcher_amd import spatial_image_utils as si
from multiview_stitcher_amd.device import DeviceArray
dev = torch.device("cuda", 0); _lib.init(0)
n = 512
g = torch.Generator(device=dev); g.manual_seed(0)
sims = []
for v in range(2):
    t = (torch.rand((n, n, n), generator=g, device=dev) * 4095).to(torch.int32).to(torch.uint16)
    da = DeviceArray.from_pointer(t.data_ptr(), (n, n, n), np.uint16, 0, owner=t)
    spacing = {"z": 2.0 if v else 1.0, "y": 1.0, "x": 1.0}
    sim = si.to_spatial_image(da, dims=["z", "y", "x"], scale=spacing, translation={"z": 0.0, "y": 0.0, "x": 0.0})
    A = np.eye(4)
    if v:
        c, s = np.cos(np.pi / 2), np.sin(np.pi / 2)          # 90 degrees about x, 2 degree tilt about z, +-1 % scale, shift
        Rx = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        a = np.deg2rad(2.0)
        Rz = np.array([[1.0, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        A[:3, :3] = Rx @ Rz @ np.diag([1.01, 0.99, 1.0])
        ctr = np.array([255.5 * 2, 255.5, 255.5])
        A[:3, 3] = np.array([255.5, 255.5, 255.5]) - A[:3, :3] @ ctr + np.array([3.3, -2.1, 4.7])
    si.set_sim_affine(sim, A, "k")
    sims.append(sim)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    out = fusion.fuse(sims, transform_key="k", output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=0)
    _lib.synchronize(0)
    dt = time.perf_counter() - t0
    print("C4 %s: %.1f ms wall, kernel %.2f ms, %.0f Mvoxels/s" % (out.shape, dt * 1e3, _lib.last_kernel_ms(0), np.prod(out.shape) / dt / 1e6), flush=True)
