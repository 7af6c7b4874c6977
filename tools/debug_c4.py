"""C4 at 512^3: per-box comparison with the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multiview_stitcher_amd import _lib, fusion
from multiview_stitcher_amd import spatial_image_utils as si
from multiview_stitcher_amd.device import DeviceArray
from tests import at_size
dev = torch.device("cuda", 0); _lib.init(0)
n = 512
g = torch.Generator(device=dev); g.manual_seed(5)
sims, keep = [], []
for v in range(2):
    noise = torch.rand((1, 1, n, n, n), generator=g, device=dev)
    noise = torch.nn.functional.avg_pool3d(noise, 3, stride=1, padding=1, count_include_pad=False)[0, 0]
    t = (noise * 4095 + 600 * v).to(torch.int32).to(torch.uint16).contiguous()
    keep.append(t)
    da = DeviceArray.from_pointer(t.data_ptr(), (n, n, n), np.uint16, 0, owner=t)
    sim = si.to_spatial_image(da, dims=["z", "y", "x"], scale={"z": 2.0 if v else 1.0, "y": 1.0, "x": 1.0}, translation={"z": 0.0, "y": 0.0, "x": 0.0})
    A = np.eye(4)
    if v:
        c, s = np.cos(np.pi / 2), np.sin(np.pi / 2)
        Rx = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        a = np.deg2rad(2.0)
        Rz = np.array([[1.0, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        A[:3, :3] = Rx @ Rz @ np.diag([1.01, 0.99, 1.0])
        ctr = np.array([255.5 * 2, 255.5, 255.5])
        A[:3, 3] = np.array([255.5, 255.5, 255.5]) - A[:3, :3] @ ctr + np.array([3.3, -2.1, 4.7])
    si.set_sim_affine(sim, A, "k")
    sims.append(sim)
torch.cuda.synchronize()
fused = fusion.fuse(sims, transform_key="k", output_on_backend=True, device=0)
_lib.synchronize(0)
fo_, fs_ = si.get_origin_from_sim(fused, asarray=True), si.get_spacing_from_sim(fused, asarray=True)
shape = np.array(fused.shape)
print("fused", shape, fo_, fs_)
base = np.round(-fo_ / fs_).astype(int)
los = [base + 224, base + np.array([0, 200, 200]), base + np.array([448, 100, 300]), base + np.array([200, 448, 0]),
       np.zeros(3, int), shape - 64, base + np.array([-32, 224, 224]), base + np.array([224, 480, 224])]
los = [np.minimum(np.maximum(lo, 0), shape - 64) for lo in los]
for lo in los:
    task = at_size.fuse_box_task(sims, "k", fo_, fs_, lo, (64, 64, 64))
    res = at_size.run_fuse_task(task)
    got = at_size.fetch(fused.data, lo, lo + 64)
    if res is None:
        print(lo, "no views; got any", got.any()); continue
    want, want_f, floor = res
    d = got.astype(np.int64) - want.astype(np.int64)
    bad = np.argwhere(np.abs(d) > 1)
    print(lo.tolist(), "views", len(task["views"]), [v["data"].shape for v in task["views"]], "max|d|", np.abs(d).max(), "nbad", len(bad),
          "first", bad[:3].tolist(), [(int(got[tuple(b)]), int(want[tuple(b)])) for b in bad[:3]])
    if len(bad):
        # redo with a generous margin to tell slab clipping from a kernel difference
        import tests.at_size as A2
        t2 = at_size.fuse_box_task(sims, "k", fo_, fs_, lo, (64, 64, 64))
        for k, sim in enumerate(sims):
            pass
