"""North-star size (BASELINE.json: 4x4x4 grid of 512^3 uint16 tiles) on the GPU: the oracle cannot run this in
seconds, so the checks are size-independent properties of register + fuse on a mosaic with a known answer."""
import numpy as np
import pytest

try:   # torch brings its own HIP runtime: it has to be loaded before libmvs_hip.so pulls in the system one (as in bench.py)
    import torch
except ImportError:   # pragma: no cover
    torch = None

pytestmark = pytest.mark.gpu


class _SignedView:
    """torch's __cuda_array_interface__ import has no uint16: hand the bytes over as int16 and view them back."""

    def __init__(self, arr):
        self.__cuda_array_interface__ = dict(arr.__cuda_array_interface__, typestr="<i2")
        self.owner = arr


def _as_torch(torch, arr):
    assert arr.dtype == np.uint16
    return torch.as_tensor(_SignedView(arr), device="cuda")      # int16 view: the mosaic's values stay below 4096


def test_north_star_register_and_fuse_properties(hip_device):
    if torch is None or not torch.cuda.is_available():
        pytest.skip("needs torch on the GPU for the on-device mosaic")
    import bench
    from multiview_stitcher_amd import _lib, fusion, param_utils, registration
    from multiview_stitcher_amd import spatial_image_utils as si

    dev = torch.device("cuda", 0)
    grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
    overlap = np.round(tile * 0.2).astype(int)
    tiles, jitters, origins, gt, pad = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=4321, return_ground_truth=True)
    sims = bench.build_sims(tiles, origins, 0)
    torch.cuda.synchronize()

    # (1) every hidden integer jitter is recovered exactly, relative to tile 0
    registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="reg", device=0,
                          pre_registration_pruning_method="keep_axis_aligned")
    rec = np.array([param_utils.select_time(si.get_affine_from_sim(s, "reg"), 0)[:3, 3] for s in sims])
    # (the global optimisation keeps its best-connected view fixed, not tile 0, and its 500 bead sweeps -- the reference's
    # max_iter -- leave ~3e-7 px on this consistent graph; fuse snaps offsets within 1e-6 of the grid, transformation.py:72-83)
    np.testing.assert_allclose(rec - rec[0], jitters - jitters[0], atol=1e-6)
    ref_shift = np.round(rec[0]).astype(int)      # = -jitter of the fixed view: world x shows ground truth x + pad - ref_shift

    # (2) idempotence: registering the registered mosaic finds no further shift
    registration.register(sims, transform_key="reg", new_transform_key="reg2", device=0,
                          pre_registration_pruning_method="keep_axis_aligned")
    rec2 = np.array([param_utils.select_time(si.get_affine_from_sim(s, "reg2"), 0)[:3, 3] for s in sims])
    np.testing.assert_allclose(rec2 - rec2[0], rec - rec[0], atol=2e-6)

    # (3) the fused mosaic reproduces the ground truth: all tiles were cut from one volume, so every weighted mean is a
    # mean of identical values -- exact where one view or unit weights contribute, and at most one count low where the
    # float32 rounding of sum(w v) / sum(w) lands just under the integer (the reference truncates the same way)
    fuse_kw = dict(transform_key="reg", output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=0)
    fused = fusion.fuse(sims, **fuse_kw)
    _lib.synchronize(0)
    f = _as_torch(torch, fused.data)
    fo = np.round(si.get_origin_from_sim(fused, asarray=True)).astype(int)
    assert np.allclose(si.get_origin_from_sim(fused, asarray=True), fo, atol=1e-5)
    sl = tuple(slice(int(o + pad - r), int(o + pad - r + n)) for o, r, n in zip(fo, ref_shift, f.shape))
    want = gt.view(torch.int16)[sl]
    assert want.shape == f.shape
    # (the outermost voxels of the mosaic are excluded: next to the edges of a tile its blend weight rounds to 0 and the
    # reference -- and this library -- write 0 there, weights.py:502-507; the rim is ragged by the +-3 px jitter)
    m = 16
    inner = tuple(slice(m, n - m) for n in f.shape)
    diff = f[inner].to(torch.int32) - want[inner].to(torch.int32)
    assert int(diff.max()) <= 0 and int(diff.min()) >= -1
    frac_low = float((diff != 0).float().mean())
    assert frac_low < 0.05, frac_low
    # single-cover interior of tile (1, 1, 1): exact copy
    c = np.round(origins[21] + rec[21] - fo).astype(int) + 110 - m
    box = tuple(slice(int(a), int(a + 290)) for a in c)
    assert bool((diff[box] == 0).all())
    del diff, want

    # (4) the region fast path and the generic affine kernel (validated against the oracle at small sizes) agree at full size
    _lib.set_option("force_generic", 1)
    try:
        fused_g = fusion.fuse(sims, **fuse_kw)
        _lib.synchronize(0)
    finally:
        _lib.set_option("force_generic", 0)
    d2 = f.to(torch.int32) - _as_torch(torch, fused_g.data).to(torch.int32)
    assert int(d2.abs().max()) <= 1
    assert float((d2 != 0).float().mean()) < 0.05
