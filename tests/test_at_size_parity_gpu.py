"""Oracle parity AT BASELINE.json SIZES (north star, C2, C3, C4, C5): the runs are full size on the GPU, the oracle checks
SAMPLES of them -- output boxes of 64^3 (interior, seams, mosaic corner, edges and corners of the tile grid) recomputed by
``oracle.fuse_oracle.fuse_np`` from exactly the slabs the run used, and one pair per orientation recomputed by
``oracle.reg_oracle.phase_correlation_registration`` from exactly the (binned, resampled) crops ``register()`` makes.
Bars: selected shift bit-exact, quality 1e-5, float32 fused voxels 1e-4 relative, integer voxels +-1 LSB only at truncation
boundaries -- and the number of voxels that need the reference's own rounding-noise floor (tests/helpers.py) is asserted.

The tiles of a mosaic are cut from one ground truth, so for the FUSE checks every tile gets its own intensity offset first:
otherwise every weighted mean is a mean of identical values and the blend weights would not be exercised."""
import numpy as np
import pytest

try:   # torch brings its own HIP runtime: load it before libmvs_hip.so pulls in the system one (as in bench.py)
    import torch
except ImportError:   # pragma: no cover
    torch = None

from tests import at_size
from tests.helpers import SignedView

pytestmark = pytest.mark.gpu


def _need_torch():
    if torch is None or not torch.cuda.is_available():
        pytest.skip("needs torch on the GPU for the on-device mosaic")


def _offset_tiles(tiles, step=500, period=8):
    """tile i += step * (i % period) (uint16 tensors viewed as int16; values stay below 2^15)."""
    for i, t in enumerate(tiles):
        t.view(torch.int16).add_(int(step * (i % period)))
    torch.cuda.synchronize()


def _grid_boxes(grid, tile, overlap, out_shape, n=64):
    grid, tile, overlap, out_shape = (np.asarray(v) for v in (grid, tile, overlap, out_shape))
    step = tile - overlap
    seam = step + overlap // 2 - n // 2            # centred on the first overlap zone of an axis
    mid = tile // 2 - n // 2                       # interior of tile 0 along an axis
    boxes = [
        mid,                                       # one view
        np.array([mid[0], mid[1], seam[2]]),       # x seam (2 views)
        np.array([mid[0], seam[1], mid[2]]),       # y seam
        np.array([seam[0], mid[1], mid[2]]),       # z seam
        np.array([mid[0], seam[1], seam[2]]),      # edge of the tile grid (4 views)
        seam.copy(),                               # corner of the tile grid (8 views)
        np.zeros(3, int),                          # low corner of the mosaic (rim)
        out_shape - n,                             # high corner of the mosaic
        np.array([256 - n // 2, 256 - n // 2, step[2] - n // 2]),   # across chunk borders and the start of an x ramp
        np.array([step[0] - 8, mid[1], step[2] + overlap[2] - n + 8]),   # both ends of ramps
    ]
    ok = [np.minimum(np.maximum(b, 0), out_shape - n) for b in boxes if np.all(out_shape >= n)]
    return [b.astype(int) for b in ok]


# ---------------------------------------------------------------------------------------------------------------------
def test_north_star_sampled_oracle_parity(hip_device):
    """4 x 4 x 4 grid of 512^3 uint16: register() with the reference's default pruning, three pairs re-registered through
    the generic pairwise path with their crops captured for the oracle, then the whole mosaic fused and 10 boxes checked."""
    _need_torch()
    import bench
    from multiview_stitcher_amd import _lib, fusion, param_utils, registration
    from multiview_stitcher_amd import spatial_image_utils as si

    dev = torch.device("cuda", 0)
    grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
    overlap = np.round(tile * 0.2).astype(int)
    tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=2024)
    sims = bench.build_sims(tiles, origins, 0)
    torch.cuda.synchronize()
    key = si.DEFAULT_TRANSFORM_KEY

    registration.register(sims, transform_key=key, new_transform_key="reg", device=0)      # default: alternating_pattern
    rec = np.array([param_utils.select_time(si.get_affine_from_sim(s, "reg"), 0)[:3, 3] for s in sims])
    np.testing.assert_allclose(rec - rec[0], jitters - jitters[0], atol=1e-6)

    # one pair per orientation: lean default path == generic path (crops captured) == oracle on those crops
    cap = at_size.CapturePairs(keep=3)
    for i, j in [(0, 1), (0, 4), (0, 16), (21, 22), (21, 25), (21, 37)]:
        lean = registration.register_pair_of_msims(sims[i], sims[j], key, device=0)
        gen = registration.register_pair_of_msims(sims[i], sims[j], key, device=0, pairwise_reg_func=cap)
        assert np.array_equal(np.asarray(lean["transform"]), np.asarray(gen["transform"]))
        assert abs(lean["quality"] - gen["quality"]) <= 1e-9
    assert len(cap.records) == 3 and {r["orient"] for r in cap.records} == {0, 1, 2}
    assert min(r["fixed"].size for r in cap.records) > 3_000_000       # binned 51 x 256 x 256 crops
    assert cap.check() == 3

    # fuse: per-tile intensity offsets make the weights matter
    _offset_tiles(tiles)
    fused = fusion.fuse(sims, transform_key="reg", output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=0)
    _lib.synchronize(0)
    fo_, fs_ = si.get_origin_from_sim(fused, asarray=True), si.get_spacing_from_sim(fused, asarray=True)
    los = _grid_boxes(grid, tile, overlap, fused.shape)
    tasks = [at_size.fuse_box_task(sims, "reg", fo_, fs_, lo, (64, 64, 64)) for lo in los]
    assert max(len(t["views"]) for t in tasks) == 8
    st = at_size.check_boxes(fused.data, tasks, los, [(64,) * 3] * len(los))
    assert st["boxes"] == len(los) >= 8
    # how often the reference's own rounding noise had to be invoked: never on the registered (integer-offset) mosaic
    assert st["beyond_plain_bar"] <= 1e-3 * st["voxels"], st

    # the API's default chunking (256^3 chunks merged into launch blocks) gives the same mosaic
    fused_d = fusion.fuse(sims, transform_key="reg", output_on_backend=True, device=0)
    _lib.synchronize(0)
    for lo in los[:4]:
        np.testing.assert_array_equal(at_size.fetch(fused_d.data, lo, lo + 64), at_size.fetch(fused.data, lo, lo + 64))


def test_north_star_fractional_offsets_sampled_oracle_parity(hip_device):
    """The same grid with sub-pixel stage offsets (a sub-pixel-registered mosaic): 2 x 2 x 2 taps per view, float64
    coordinates like scipy at 1748 voxels per axis.  2 x 4 x 4 tiles keep the run short; the x / y extent is full size."""
    _need_torch()
    import bench
    from multiview_stitcher_amd import _lib, fusion
    from multiview_stitcher_amd import spatial_image_utils as si

    dev = torch.device("cuda", 0)
    grid, tile = np.array([2, 4, 4]), np.array([512, 512, 512])
    overlap = np.round(tile * 0.2).astype(int)
    tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=7, max_jitter=0)
    rng = np.random.default_rng(3)
    origins = origins + rng.uniform(-1.5, 1.5, origins.shape)
    sims = bench.build_sims(tiles, origins, 0)
    _offset_tiles(tiles)
    key = si.DEFAULT_TRANSFORM_KEY
    fused = fusion.fuse(sims, transform_key=key, output_on_backend=True, device=0)       # default chunking, merged blocks
    _lib.synchronize(0)
    fo_, fs_ = si.get_origin_from_sim(fused, asarray=True), si.get_spacing_from_sim(fused, asarray=True)
    los = _grid_boxes(grid, tile, overlap, fused.shape)
    tasks = [at_size.fuse_box_task(sims, key, fo_, fs_, lo, (64, 64, 64)) for lo in los]
    st = at_size.check_boxes(fused.data, tasks, los, [(64,) * 3] * len(los))
    assert st["boxes"] >= 8
    assert st["beyond_plain_bar"] <= 1e-3 * st["voxels"], st
    # merged launch blocks == chunk by chunk, voxel for voxel.  fuse() derives every view's parameters once, in the index
    # frame of the output stack, and chunks / slabs only shift integer indices (include/mvs_hip.h: index_origin); the
    # reference derives them per chunk and rounds to 10 decimals (transformation.py:72-83), so two of ITS chunkings differ
    # by ~1e-9 px in the blend weights -- measured here before the frame existed: one count on 1.4e-5 of the voxels.  (What
    # was left after that, 8e-7, came from the kernels: (base - fraction) + j rounded the support distance twice, so a
    # weight's last bit depended on where a lane's 8-voxel group started; it is (base + j) - fraction now, as in fold_u.)
    fused_c = fusion.fuse(sims, transform_key=key, output_on_backend=True, device=0, merge_chunks=False)
    _lib.synchronize(0)
    a = torch.as_tensor(SignedView(fused.data), device="cuda")
    b = torch.as_tensor(SignedView(fused_c.data), device="cuda")
    assert bool((a == b).all())
    d = None
    del a, b, d, fused_c
    _lib.set_option("force_generic", 1)
    try:
        sub = dict(transform_key=key, output_on_backend=True, device=0,
                   output_stack_properties={"origin": dict(zip("zyx", fo_ + np.array([100.0, 500.0, 700.0]))),
                                            "spacing": dict(zip("zyx", fs_)), "shape": {"z": 200, "y": 300, "x": 400}},
                   frame_origin=dict(zip("zyx", fo_)))
        g_m = fusion.fuse(sims, **sub)
        g_c = fusion.fuse(sims, merge_chunks=False, output_chunksize={"z": 64, "y": 128, "x": 96}, **sub)
        _lib.synchronize(0)
        np.testing.assert_array_equal(g_m.data.get(), g_c.data.get())
        # ... and the sub-stack equals the corresponding window of the whole mosaic's generic result
        whole = fusion.fuse(sims, transform_key=key, output_on_backend=True, device=0)
        _lib.synchronize(0)
        np.testing.assert_array_equal(g_m.data.get(), at_size.fetch(whole.data, [100, 500, 700], [300, 800, 1100]))
    finally:
        _lib.set_option("force_generic", 0)


def test_c2_sampled_oracle_parity(hip_device):
    """C2: 3 x 3 grid of 2D 2048^2 float32 tiles: register (12 pairs; two captured for the oracle), cosine-blend fuse in
    2048^2 chunks, 512^2 boxes against the oracle at 1e-4 relative."""
    _need_torch()
    from multiview_stitcher_amd import _lib, fusion, param_utils, registration
    from multiview_stitcher_amd import spatial_image_utils as si
    from multiview_stitcher_amd.device import DeviceArray
    from tests.test_full_size_gpu import _mosaic_2d_f32

    dev = torch.device("cuda", 0)
    tiles, jitters, origins, gt, pad = _mosaic_2d_f32(torch, dev, (3, 3), (2048, 2048), (410, 410), seed=78)
    torch.cuda.synchronize()
    sims = []
    for t, o in zip(tiles, origins):
        da = DeviceArray.from_pointer(t.data_ptr(), tuple(t.shape), np.float32, 0, owner=t)
        s = si.to_spatial_image(da, dims=["y", "x"], scale={"y": 1.0, "x": 1.0}, translation=dict(zip("yx", o)))
        si.set_sim_affine(s, np.eye(3), si.DEFAULT_TRANSFORM_KEY)
        sims.append(s)
    key = si.DEFAULT_TRANSFORM_KEY
    cap = at_size.CapturePairs(keep=6)
    registration.register(sims, transform_key=key, new_transform_key="reg", device=0, pairwise_reg_func=cap)
    rec = np.array([param_utils.select_time(si.get_affine_from_sim(s, "reg"), 0)[:2, 2] for s in sims])
    np.testing.assert_allclose(rec - rec[0], jitters - jitters[0], atol=1e-6)
    assert {r["orient"] for r in cap.records} == {0, 1} and cap.check() >= 2
    for i, t in enumerate(tiles):
        t.mul_(1.0 + 0.25 * (i % 4)).add_(0.1 * (i % 3))
    torch.cuda.synchronize()
    fused = fusion.fuse(sims, transform_key="reg", output_chunksize={"y": 2048, "x": 2048}, output_on_backend=True, device=0)
    _lib.synchronize(0)
    fo_, fs_ = si.get_origin_from_sim(fused, asarray=True), si.get_spacing_from_sim(fused, asarray=True)
    n, step = 512, 2048 - 410
    los = [np.array(v) for v in [(700, 700), (700, step - 50), (step - 50, 700), (step - 100, step - 100), (0, 0),
                                 (fused.shape[0] - n, fused.shape[1] - n), (2048 - 256, 2048 - 256), (2 * step - 30, step + 300)]]
    tasks = [at_size.fuse_box_task(sims, "reg", fo_, fs_, lo, (n, n)) for lo in los]
    st = at_size.check_boxes(fused.data, tasks, los, [(n, n)] * len(los))
    assert st["boxes"] == 8 and st["beyond_plain_bar"] <= 1e-3 * st["voxels"], st


def test_c3_register_and_content_based_sampled_oracle_parity(hip_device):
    """C3: 4 x 4 x 2 (x, y, z) grid of 256 x 512 x 512 uint16: register() at size (auto-binning {z: 2}; three pairs against
    the oracle), then content-based weights (sigma 5 / 11, halo 22) in 256^3 chunks.  The reference filters every halo
    chunk with ``mode="reflect"``, so a sampled box is recomputed by the oracle on a crop of ITS halo chunk: crop borders
    that coincide with the halo chunk's borders reproduce the reflection, artificial ones lie >= 64 px (the support of
    the two chained Gaussians, 20 + 44) away from the compared box."""
    _need_torch()
    import bench
    from multiview_stitcher_amd import _lib, fusion, param_utils, registration
    from multiview_stitcher_amd import spatial_image_utils as si

    dev = torch.device("cuda", 0)
    grid, tile = np.array([2, 4, 4]), np.array([256, 512, 512])
    overlap = np.round(tile * 0.2).astype(int)
    tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=31)
    sims = bench.build_sims(tiles, origins, 0)
    torch.cuda.synchronize()
    key = si.DEFAULT_TRANSFORM_KEY
    registration.register(sims, transform_key=key, new_transform_key="reg", device=0)
    rec = np.array([param_utils.select_time(si.get_affine_from_sim(s, "reg"), 0)[:3, 3] for s in sims])
    np.testing.assert_allclose(rec - rec[0], jitters - jitters[0], atol=1e-6)
    cap = at_size.CapturePairs(keep=3)
    for i, j in [(0, 1), (0, 4), (0, 16)]:
        lean = registration.register_pair_of_msims(sims[i], sims[j], key, device=0)
        gen = registration.register_pair_of_msims(sims[i], sims[j], key, device=0, pairwise_reg_func=cap)
        assert np.array_equal(np.asarray(lean["transform"]), np.asarray(gen["transform"]))
    assert len(cap.records) == 3 and cap.check() == 3

    _offset_tiles(tiles, step=300)
    halo, cs = 22, 256
    fused = fusion.fuse(sims, transform_key="reg", weights_func=fusion.content_based, output_chunksize={d: cs for d in "zyx"},
                        output_on_backend=True, device=0)
    _lib.synchronize(0)
    fo_, fs_ = si.get_origin_from_sim(fused, asarray=True), si.get_spacing_from_sim(fused, asarray=True)
    shape = np.array(fused.shape)
    n, reach = 64, 64
    # (chunk index, box offset inside the chunk): chunk corners (reflection at the halo border) and interiors, on seams
    samples = [((0, 1, 1), (0, 0, 0)), ((0, 1, 1), (96, 150, 150)), ((1, 1, 1), (0, 192, 0)), ((0, 0, 0), (30, 30, 30)),
               ((1, 3, 3), (96, 96, 96)), ((0, 2, 1), (192, 96, 140)), ((0, 1, 5), (100, 150, 10)), ((1, 4, 1), (140, 10, 150))]
    tasks, los, cmps = [], [], []
    for cidx, off in samples:
        c0 = np.array(cidx) * cs
        cn = np.minimum(c0 + cs, shape) - c0                      # chunk extent (the last chunk of an axis is partial)
        off = np.minimum(np.array(off), np.maximum(cn - n, 0))
        bn = np.minimum(n, cn)
        # halo chunk H = [c0 - halo, c0 + cn + halo); crop = box +- reach, snapped to H's borders when closer than reach
        h0, h1 = c0 - halo, c0 + cn + halo
        b0, b1 = c0 + off, c0 + off + bn
        k0 = np.where(b0 - reach <= h0 + 8, h0, b0 - reach)
        k1 = np.where(b1 + reach >= h1 - 8, h1, b1 + reach)
        task = at_size.fuse_box_task(sims, "reg", fo_, fs_, k0, k1 - k0, halo=0, weights="content_based",
                                     weights_kwargs={"sigma_1": 5, "sigma_2": 11})
        tasks.append(task)
        los.append(b0)
        cmps.append((b0 - k0, bn))
    # through check_boxes like the other configs (statistics -> profiles/): +-1 count only within 1e-4 of an integer boundary
    st = at_size.check_boxes(fused.data, tasks, los, [tuple(int(v) for v in bn) for _, bn in cmps], windows=[rel for rel, _ in cmps])
    assert st["boxes"] == len(samples) and st["voxels"] >= 8 * 64 ** 3 // 2
    assert st["lsb_flips"] <= 0.002 * st["voxels"], st         # (2 % allowed in round 3; measured 0-0.06 % elsewhere)
    assert st["beyond_plain_bar"] == 0, st


def test_c4_sampled_oracle_parity(hip_device):
    """C4: two 512^3 uint16 views, the second under a full affine (90 degrees about x, 2 degree tilt, +-1 % scale,
    sub-pixel shift, z spacing 2): weighted-average fuse of the union stack through the generic kernel, 64^3 boxes."""
    _need_torch()
    from multiview_stitcher_amd import _lib, fusion
    from multiview_stitcher_amd import spatial_image_utils as si
    from multiview_stitcher_amd.device import DeviceArray

    dev = torch.device("cuda", 0)
    n = 512
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    sims, keep = [], []
    for v in range(2):
        noise = torch.rand((1, 1, n, n, n), generator=g, device=dev)
        noise = torch.nn.functional.avg_pool3d(noise, 3, stride=1, padding=1, count_include_pad=False)[0, 0]
        t = (noise * 4095 + 600 * v).to(torch.int32).to(torch.uint16).contiguous()
        keep.append(t)
        da = DeviceArray.from_pointer(t.data_ptr(), (n, n, n), np.uint16, 0, owner=t)
        sim = si.to_spatial_image(da, dims=["z", "y", "x"], scale={"z": 2.0 if v else 1.0, "y": 1.0, "x": 1.0},
                                  translation={"z": 0.0, "y": 0.0, "x": 0.0})
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
    rng = np.random.default_rng(1)
    # world (0..511)^3 is view 0; boxes inside it, on its borders (where view 1 sticks out) and at the stack's corners
    base = np.round(-fo_ / fs_).astype(int)            # mosaic index of world 0
    los = [base + 224, base + np.array([0, 200, 200]), base + np.array([448, 100, 300]), base + np.array([200, 448, 0]),
           np.zeros(3, int), shape - 64, base + np.array([-32, 224, 224]), base + np.array([224, 480, 224])]
    los += [rng.integers(0, shape - 64) for _ in range(2)]
    los = [np.minimum(np.maximum(lo, 0), shape - 64) for lo in los]
    tasks = [at_size.fuse_box_task(sims, "k", fo_, fs_, lo, (64, 64, 64)) for lo in los]
    assert max(len(t["views"]) for t in tasks) == 2
    st = at_size.check_boxes(fused.data, tasks, los, [(64,) * 3] * len(los))
    assert st["boxes"] >= 8 and st["beyond_plain_bar"] <= 1e-2 * st["voxels"], st
    assert st["marginal_voxels"] <= 8, st


def test_c5_full_grid_sparse_store_sampled_oracle_parity(hip_device, tmp_path):
    """C5 at its FULL geometry: 8 x 8 x 4 (x, y, z) grid of 512 x 1024 x 1024 uint16 tiles in Zarr stores (128^3 chunks),
    output 1742 x 6757 x 6757 = 79.5 G voxels in 256^3 chunks (5103 of them) streamed into a Zarr array.  The planner, the
    slab windows and the chunk farm run on the whole grid; the voxel DATA is materialised only where the sampled output
    chunks look (a chunk file that was never written reads as the fill value), which keeps the store at a few GB
    instead of 275 GB.  Every sampled 256^3 chunk is recomputed by the oracle from the same lazily read slabs.  The
    registration leg is checked on three real pairs of full-size tiles (auto-binning {z: 3, y: 2, x: 2})."""
    _need_torch()
    import bench
    from multiview_stitcher_amd import fusion, registration, zarr_io
    from multiview_stitcher_amd import spatial_image_utils as si

    grid, tile = np.array([4, 8, 8]), np.array([512, 1024, 1024])
    overlap = np.round(tile * 0.2).astype(int)
    step = tile - overlap
    key = si.DEFAULT_TRANSFORM_KEY
    rng = np.random.default_rng(11)
    sims, arrays, origins = [], [], []
    for k, idx in enumerate(np.ndindex(*grid)):
        o = (np.array(idx) * step + rng.integers(-3, 4, 3) * (k > 0)).astype(float)
        za = zarr_io.ZarrArray.create(str(tmp_path / f"tile{k}.zarr"), tuple(tile), (128, 128, 128), np.uint16)
        s = si.to_spatial_image(za[...], dims=["z", "y", "x"], scale={d: 1.0 for d in "zyx"}, translation=dict(zip("zyx", o)))
        si.set_sim_affine(s, np.eye(4), key)
        sims.append(s)
        arrays.append(za)
        origins.append(o)
    origins = np.array(origins)
    lo_w = origins.min(0)
    out_shape = (origins.max(0) + tile - lo_w).astype(int)
    assert tuple(out_shape) == tuple(int(v) for v in (origins.max(0) - origins.min(0) + tile))
    cs = 256
    nblocks = -(-out_shape // cs)
    assert int(np.prod(nblocks)) >= 5000
    # sampled chunks: interior, seams, a corner of the tile grid, the far corner of the mosaic (partial chunk)
    sx = int(step[2]) // cs
    sampled = [(1, 1, 1), (0, 1, sx), (0, sx, sx), (int(step[0]) // cs, sx, sx), (3, 13, 13),
               tuple(int(v) for v in nblocks - 1), (2, 20, 7), (5, 9, 22)]
    sampled = [tuple(int(min(b, nb - 1)) for b, nb in zip(s_, nblocks)) for s_ in sampled]
    # materialise the tile data the sampled chunks read: value = hash of (tile, z, y, x) below 4096 + a per-tile offset
    written = 0
    for bi in sampled:
        c0 = lo_w + np.array(bi) * cs
        c1 = np.minimum(c0 + cs, lo_w + out_shape)
        for k, (za, o) in enumerate(zip(arrays, origins)):
            a = np.maximum(np.floor(c0 - o).astype(int) - 2, 0) // 128 * 128
            b = np.minimum(-(-(np.ceil(c1 - o).astype(int) + 3) // 128) * 128, tile)
            if np.any(b <= a):
                continue
            zz, yy, xx = np.meshgrid(*[np.arange(p, q, dtype=np.int64) for p, q in zip(a, b)], indexing="ij", sparse=True)
            vals = ((zz * 7919 + yy * 104729 + xx * 1299709 + k * 15485863) % 3001 + 150 * (k % 7)).astype(np.uint16)
            za.write(list(a), vals)
            written += vals.nbytes
    assert written < 12 << 30
    out_url = str(tmp_path / "fused.zarr")
    wanted = set(sampled)
    fused = fusion.fuse(sims, transform_key=key, output_chunksize={d: cs for d in "zyx"}, output_zarr_url=out_url,
                        chunk_filter=lambda bi: tuple(int(v) for v in bi) in wanted, device=0)
    assert zarr_io.is_zarr_backed(fused.data) and tuple(fused.data.shape[-3:]) == tuple(out_shape)
    fo_, fs_ = si.get_origin_from_sim(fused, asarray=True), si.get_spacing_from_sim(fused, asarray=True)
    np.testing.assert_allclose(fo_, lo_w)
    los = [np.array(bi) * cs for bi in sampled]
    shapes = [np.minimum(lo + cs, out_shape) - lo for lo in los]
    tasks = [at_size.fuse_box_task(sims, key, fo_, fs_, lo, shp) for lo, shp in zip(los, shapes)]
    assert max(len(t["views"]) for t in tasks) == 8
    st = at_size.check_boxes(fused.data, tasks, los, shapes)
    assert st["boxes"] == len(sampled) and st["beyond_plain_bar"] <= 1e-3 * st["voxels"], st

    # registration leg: one real pair of full-size tiles per orientation
    dev = torch.device("cuda", 0)
    cap = at_size.CapturePairs(keep=3)
    for axis in range(3):
        g = np.ones(3, int)
        g[axis] = 2
        tl, jit, org = bench.make_mosaic_on_device(torch, dev, g, tile, overlap, seed=50 + axis)
        ps = bench.build_sims(tl, org, 0)
        torch.cuda.synchronize()
        assert registration.get_optimal_registration_binning(ps[0], ps[1]) == {"z": 3, "y": 2, "x": 2}
        lean = registration.register_pair_of_msims(ps[0], ps[1], key, device=0)
        gen = registration.register_pair_of_msims(ps[0], ps[1], key, device=0, pairwise_reg_func=cap)
        assert np.array_equal(np.asarray(lean["transform"]), np.asarray(gen["transform"]))
        # the pairwise translation is the hidden jitter difference of the two tiles (up to the direction convention), as
        # far as the binned grid resolves it: multiples of bin / 2 (upsample factor 2 in 3D, registration.py:410-411)
        got_t, true_t = np.abs(np.asarray(lean["transform"])[:3, 3]), np.abs(jit[1] - jit[0])
        assert np.all(np.abs(got_t - true_t) <= np.array([3, 2, 2]) / 2 + 1e-6), (got_t, true_t)
        del tl, ps
        torch.cuda.empty_cache()
    assert cap.check() == 3


# ---------------------------------------------------------------------------------------------------------------------
def test_c1_two_tiles_2d_whole_mosaic_oracle_parity(hip_device):
    """C1 (BASELINE.json configs[0], the reference's own CPU-runnable case) at its size: a 2 x 1 grid of 2D 512 x 512 uint16
    tiles with 20 % overlap -- register() (translation-only phase correlation) + fuse() with the default cosine blending.
    Small enough for the oracle to do ALL of it: the pair's crops are rebuilt by the oracle from the raw tiles
    (registration.py:194-350) and equal the device's; the selected shift is bit-exact, the quality within 1e-5; the WHOLE
    fused mosaic is compared with oracle.fuse_oracle.fuse_np."""
    from multiview_stitcher_amd import fusion, param_utils, registration, sample_data
    from multiview_stitcher_amd import spatial_image_utils as si
    from oracle import fuse_oracle as fo
    from tests.helpers import assert_fused_close, sim_to_view, squeeze_field

    sims, jit, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(512, 512), tiles=(1, 2), overlap=(0, 102),
                                                      dtype=np.uint16, max_jitter=3, seed=11)
    key = sample_data.METADATA_TRANSFORM_KEY
    assert np.abs(jit[1]).max() > 0
    cap = at_size.CapturePairs(keep=3)
    res = registration.register(sims, transform_key=key, new_transform_key="reg", reg_channel_index=0, pairwise_reg_func=cap,
                                groupwise_resolution_kwargs={"reference_view": 0}, return_dict=True)
    assert res["pairwise_registration"]["edges"] == [(0, 1)] and len(cap.records) == 1
    assert cap.check() == 1                                         # oracle on the captured crops: shift bit-exact, quality 1e-5
    # the default (fused, one-call) pair path gives the same numbers as the generic path the crops were captured on
    lean = registration.register_pair_of_msims(squeeze_field(sims[0]), squeeze_field(sims[1]), key, device=0)
    np.testing.assert_array_equal(np.asarray(lean["transform"]), np.asarray(res["pairwise_registration"]["results"][0][0]["transform"]))
    # the crops themselves, recomputed by the oracle from the raw tiles (no binning at this size)
    flat = [squeeze_field(s) for s in sims]
    fixed, moving, binning = at_size.oracle_registration_crops(
        np.asarray(flat[0].data), np.asarray(flat[1].data), si.get_origin_from_sim(flat[0], asarray=True),
        si.get_origin_from_sim(flat[1], asarray=True), si.get_spacing_from_sim(flat[0], asarray=True))
    assert binning == {"y": 1, "x": 1}
    at_size.assert_crops_equal(cap.records[0]["fixed"], fixed)
    at_size.assert_crops_equal(cap.records[0]["moving"], moving)
    # hidden jitter recovered (upsample factor 10 in 2D: 0.1 px grid)
    got = np.array([param_utils.select_time(p, 0)[:-1, -1] for p in res["params"]])
    np.testing.assert_allclose(got[1] - got[0], jit[1] - jit[0], atol=0.11)

    fused = fusion.fuse(sims, transform_key="reg")
    out = squeeze_field(fused)
    params = [np.asarray(param_utils.select_time(si.get_affine_from_sim(s, "reg"), 0), dtype=np.float64) for s in flat]
    views, bbs = zip(*[sim_to_view(s) for s in flat])
    out_bb = fo.bb(si.get_origin_from_sim(out, asarray=True), si.get_spacing_from_sim(out, asarray=True), list(out.shape))
    want, want_f, dbg = fo.fuse_np(list(views), params, out_bb, full_view_bbs=list(bbs), return_debug=True)
    assert out.shape[0] >= 512 and out.shape[1] >= 2 * 512 - 102 - 6
    from tests.helpers import reference_noise_floor
    st = assert_fused_close(np.asarray(out.data), want, want_f, noise_floor=reference_noise_floor(dbg, want_f))
    at_size._record(dict(st, boxes=1, marginal_voxels=0))


def _knife_edge_pairs(n_knife=4, n_plain=2):
    """Small pairs of tiles cut from one ground truth at fractional stage positions: ``n_knife`` whose reference crop (Qhull
    vertices) is one sample shorter than the closed form's along some axis, ``n_plain`` where both agree."""
    from scipy import ndimage

    from multiview_stitcher_amd import device, registration
    from multiview_stitcher_amd import spatial_image_utils as si

    rng = np.random.default_rng(21)
    knife, plain = [], []
    for trial in range(400):
        nd = 2 if trial % 2 else 3
        shape = (96, 120) if nd == 2 else (40, 64, 72)
        sp = rng.choice([1.0, 0.7, 2.0], nd)
        ax = int(rng.integers(nd))
        off_px = np.zeros(nd, int)
        off_px[ax] = int(0.7 * shape[ax])
        jit = rng.integers(-2, 3, nd)
        base = rng.normal(0, 20, nd)
        sd = "zyx"[-nd:]
        sims = [si.to_spatial_image(np.zeros(shape, np.uint16), dims=list(sd), scale=dict(zip(sd, sp)), translation=dict(zip(sd, base + k * off_px * sp)))
                for k in range(2)]
        for s_ in sims:
            si.set_sim_affine(s_, np.eye(nd + 1), "k")
        a = registration._get_overlap_bboxes(sims[0], sims[1], "k", None, None, closed_form=False)
        b = registration._get_overlap_bboxes(sims[0], sims[1], "k", None, None, closed_form=True)
        spv = si.get_spacing_from_sim(sims[0], asarray=True)
        n_ref = np.floor((a["uppers"][0] - a["lowers"][0]) / spv + 1).astype(int)
        n_cf = np.floor((b["uppers"][0] - b["lowers"][0]) / spv + 1).astype(int)
        is_knife = bool(np.any(n_ref != n_cf))
        if (is_knife and len(knife) >= n_knife) or (not is_knife and len(plain) >= n_plain):
            continue
        pad = 4
        gt_shape = tuple(int(n + o + 2 * pad) for n, o in zip(shape, off_px))
        gt = (ndimage.gaussian_filter(rng.random(gt_shape), 1.5) * 60000 - 28000).clip(0, 4095).astype(np.uint16)
        tiles = [np.ascontiguousarray(gt[tuple(slice(pad, pad + n) for n in shape)]),
                 np.ascontiguousarray(gt[tuple(slice(pad + o + j, pad + o + j + n) for o, j, n in zip(off_px, jit, shape))])]
        for s_, t in zip(sims, tiles):
            s_.data = device.to_device(si.SpatialImage(t, list(sd)), 0).data
        (knife if is_knife else plain).append({"sims": sims, "tiles": tiles, "n_ref": n_ref.tolist(), "n_cf": n_cf.tolist()})
        if len(knife) >= n_knife and len(plain) >= n_plain:
            break
    assert len(knife) == n_knife and len(plain) == n_plain
    return knife + plain


def _routed_to_reference(st1, st2, closed_form_shape):
    """registration._reference_crop_differs on two stack-property records (zyx arrays) under identity transforms."""
    from multiview_stitcher_amd import registration
    from multiview_stitcher_amd import spatial_image_utils as si
    from multiview_stitcher_amd.sharding import RemoteArray

    geoms = []
    for st in (st1, st2):
        nd = len(st["origin"])
        sd = "zyx"[-nd:]
        s_ = si.to_spatial_image(RemoteArray(tuple(int(v) for v in st["shape"]), np.uint16), dims=list(sd), scale=dict(zip(sd, st["spacing"])),
                                 translation=dict(zip(sd, st["origin"])))
        si.set_sim_affine(s_, np.eye(nd + 1), "k")
        geoms.append(registration._TileGeom(s_, "k"))
    return registration._reference_crop_differs(geoms[0], geoms[1], [0.0] * len(st1["origin"]), closed_form_shape)


def test_crop_length_knife_edge_reference_mode_end_to_end(hip_device):
    """VERDICT round 4 item 2.  (a) How many pairs of every BASELINE.json geometry have a reference crop (Qhull vertices,
    registration.py:229-239, 314-316) that differs from the product's closed-form crop: recorded in the at-size statistics.
    (b) For every pair that does -- C1's own pair at size and synthetic pairs at fractional stage positions -- the oracle runs end to
    end from the RAW tiles on ITS crop, and the device in ``overlap_bbox="reference"`` mode registers crops of exactly that shape
    and content and returns the oracle's pixel translation bit for bit, the quality to 1e-5.  Round 6: so does the DEFAULT mode --
    it asks the reference's sequence once per pair geometry whether it lands on N - 1 and sends such pairs through it
    (registration._reference_crop_differs); with that question switched off (the plain closed form, rounds 4-5) the same pairs give
    the same integer shift, the sub-pixel refinement may move by one upsampling step because an N-long and an (N - 1)-long crop are
    different circular correlations -- recorded next to it."""
    from multiview_stitcher_amd import registration, sample_data
    from multiview_stitcher_amd import spatial_image_utils as si
    from oracle import reg_oracle as ro
    from tests import crop_length
    from tests.helpers import squeeze_field

    counts = {name: crop_length.count_differing_pairs(*geo)[:2] for name, geo in crop_length.CONFIG_GEOMETRIES.items()}
    assert counts["north_star"][0] == 144 and counts["C3"][0] == 64 and counts["C2"][0] == 12 and counts["C1"][0] == 1

    cases = []
    sims, jit, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(512, 512), tiles=(1, 2), overlap=(0, 102), dtype=np.uint16, max_jitter=3, seed=11)
    flat = [squeeze_field(s_) for s_ in sims]
    cases.append({"name": "C1", "sims": flat, "tiles": [np.asarray(f.data) for f in flat], "key": sample_data.METADATA_TRANSFORM_KEY})
    for k, c in enumerate(_knife_edge_pairs()):
        cases.append(dict(c, name=f"synthetic{k}", key="k"))
    n_knife = n_default_equal = n_default_checked = 0
    max_default_shift_diff = 0.0
    for c in cases:
        s0, s1 = c["sims"]
        fixed, moving, binning = at_size.oracle_registration_crops(
            c["tiles"][0], c["tiles"][1], si.get_origin_from_sim(s0, asarray=True), si.get_origin_from_sim(s1, asarray=True),
            si.get_spacing_from_sim(s0, asarray=True))
        assert max(binning.values()) == 1
        want = ro.phase_correlation_registration(fixed, moving)
        cap_ref, cap_def = at_size.CapturePairs(keep=300), at_size.CapturePairs(keep=300)
        r_ref = registration.register_pair_of_msims(s0, s1, c["key"], device=0, pairwise_reg_func=cap_ref, overlap_bbox="reference")
        # the default mode: through the generic path (captured crops) and through the lean path (built-in phase correlation)
        cap_auto = at_size.CapturePairs(keep=300)
        r_auto = registration.register_pair_of_msims(s0, s1, c["key"], device=0, pairwise_reg_func=cap_auto)
        r_lean = registration.register_pair_of_msims(s0, s1, c["key"], device=0)
        got_auto = cap_auto.records[0]
        assert got_auto["fixed"].shape == fixed.shape, (c["name"], got_auto["fixed"].shape, fixed.shape)
        np.testing.assert_array_equal(got_auto["got"]["affine_matrix"], want["affine_matrix"])
        np.testing.assert_array_equal(np.asarray(r_auto["transform"]), np.asarray(r_ref["transform"]))
        np.testing.assert_array_equal(np.asarray(r_lean["transform"]), np.asarray(r_ref["transform"]))
        n_default_checked += 1
        # ... and the plain closed form (the question switched off)
        registration._KNIFE_CHECK[0] = False
        try:
            r_def = registration.register_pair_of_msims(s0, s1, c["key"], device=0, pairwise_reg_func=cap_def)
        finally:
            registration._KNIFE_CHECK[0] = True
        got_ref, got_def = cap_ref.records[0], cap_def.records[0]
        # reference mode: the oracle's crop -- shape and samples -- and its result
        assert got_ref["fixed"].shape == fixed.shape and got_ref["moving"].shape == moving.shape, (c["name"], got_ref["fixed"].shape, fixed.shape)
        assert at_size.assert_crops_equal(got_ref["fixed"], fixed) and at_size.assert_crops_equal(got_ref["moving"], moving)
        np.testing.assert_array_equal(got_ref["got"]["affine_matrix"], want["affine_matrix"])
        assert abs(got_ref["got"]["quality"] - want["quality"]) <= 1e-5
        knife = got_def["fixed"].shape != fixed.shape
        n_knife += int(knife)
        if c["name"] == "C1":
            assert knife and counts["C1"][1] == 1                           # 511 rows in the reference, 512 in closed form
        if knife:
            assert all(abs(a - b) <= 1 for a, b in zip(got_def["fixed"].shape, fixed.shape))     # (at fractional positions either can be the longer)
            same = np.array_equal(got_def["got"]["affine_matrix"], want["affine_matrix"])
            n_default_equal += int(same)
            d = np.abs(np.asarray(got_def["got"]["affine_matrix"]) - np.asarray(want["affine_matrix"])).max()
            max_default_shift_diff = max(max_default_shift_diff, float(d))
            # the two crop lengths never disagree about the integer part of the shift
            assert d <= 0.5 + 1e-9, (c["name"], got_def["got"], want)
        else:        # no knife edge: both modes are the same registration
            np.testing.assert_array_equal(got_def["got"]["affine_matrix"], want["affine_matrix"])
            np.testing.assert_array_equal(np.asarray(r_ref["transform"]), np.asarray(r_def["transform"]))
    assert n_knife >= 5
    # per config, through the product's own rule: pairs whose DEFAULT-mode crop differs from the reference's (the N - 1 pairs are
    # routed through the reference's sequence, so none is left)
    left = {}
    for name, geo in crop_length.CONFIG_GEOMETRIES.items():
        _, _, differing = crop_length.count_differing_pairs(*geo)
        stacks = crop_length.grid_stacks(*geo)
        left[name] = sum(0 if _routed_to_reference(stacks[e[0]], stacks[e[1]], cf) else 1 for e, _, cf in differing)
    assert all(v == 0 for v in left.values()), left
    at_size._record({"knife_edge_pairs_by_config": {k: {"pairs": v[0], "closed_form_crop_differs": v[1], "reference_crop_differs": left[k]}
                                                    for k, v in counts.items()},
                     "end_to_end_cases": len(cases), "knife_edge_cases": n_knife, "default_mode_equal_to_reference": n_default_checked,
                     "plain_closed_form_equal_to_reference": n_default_equal,
                     "plain_closed_form_max_abs_shift_difference_px": max_default_shift_diff,
                     "voxels": 0, "beyond_plain_bar": 0, "lsb_flips": 0, "max_floor_used": 0.0, "boxes": 0})


def test_north_star_device_results_through_the_resolution_and_pruning_oracles(hip_device):
    """f3 / f4 on the driver's box: the pairs register() selects on the north-star mosaic equal the
    oracle's pruning of the same overlap graph (mv_graph.py:664-881 on networkx), and the DEVICE run's pairwise results
    (transform, quality, overlap box of all 144 pairs), fed to oracle/resolve_oracle.py's global optimisation
    (global_optimization.py:16-511), give the parameters register() returned.  Also a2 / a3 at size: the crops of one pair
    per orientation are rebuilt by the oracle from the RAW 512^3 tiles -- {2, 2, 2} mean binning with the truncating cast,
    overlap boxes, resample onto the fixed view's grid -- and equal the crops the device path registers."""
    _need_torch()
    nx = pytest.importorskip("networkx")
    import bench
    from multiview_stitcher_amd import mv_graph, param_utils, registration
    from multiview_stitcher_amd import spatial_image_utils as si
    from oracle import resolve_oracle

    dev = torch.device("cuda", 0)
    grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
    overlap = np.round(tile * 0.2).astype(int)
    tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=77)
    sims = bench.build_sims(tiles, origins, 0)
    torch.cuda.synchronize()
    key = si.DEFAULT_TRANSFORM_KEY
    res = registration.register(sims, transform_key=key, new_transform_key="reg", device=0, return_dict=True)
    edges = res["pairwise_registration"]["edges"]
    results = res["pairwise_registration"]["results"][0]
    assert len(edges) == 144

    # f4: overlap graph + default pruning against the oracle (same edges, same order = the registration work list)
    sps = [si.get_stack_properties_from_sim(s) for s in sims]
    g_views = mv_graph.build_view_adjacency_graph([dict(sp, transform=np.eye(4)) for sp in sps])
    h = nx.Graph()
    for n in g_views.nodes:
        h.add_node(n, **g_views.node_attrs[n])
    seen = set()
    for n in g_views.nodes:
        for m in g_views.adj[n]:
            if (m, n) not in seen:
                seen.add((n, m))
                h.add_edge(n, m, **dict(g_views.adj[n][m]))
    want_edges = [tuple(sorted(e)) for e in resolve_oracle.prune_view_adjacency_graph(h, "alternating_pattern").edges()]
    assert edges == want_edges

    # f3: the device's pairwise results through the oracle's groupwise resolution
    hr = nx.Graph()
    for v in range(len(sims)):
        hr.add_node(v, stack_props={"spacing": sps[v]["spacing"]})
    for (a, b), r in zip(edges, results):
        hr.add_edge(a, b, transform=np.asarray(r["transform"]), quality=float(r["quality"]), overlap=1.0, bbox=np.asarray(r["bbox"]))
    want_p, want_info = resolve_oracle.groupwise_resolution(hr)
    got_info = res["groupwise_resolution"]["info"][0]
    for v in range(len(sims)):
        np.testing.assert_allclose(param_utils.select_time(res["params"][v], 0), want_p[v], rtol=0, atol=1e-9)
    assert sorted(got_info["used_edges"][0]) == sorted(want_info["used_edges"])
    rec = np.array([param_utils.select_time(p, 0)[:3, 3] for p in res["params"]])
    np.testing.assert_allclose(rec - rec[0], jitters - jitters[0], atol=1e-6)

    # a2 / a3: crops of one pair per orientation from the raw tiles, by the oracle
    cap = at_size.CapturePairs(keep=3)
    for i, j in [(0, 1), (0, 4), (0, 16)]:
        cap.tag = (i, j)
        registration.register_pair_of_msims(sims[i], sims[j], key, device=0, pairwise_reg_func=cap)
    assert len(cap.records) == 3
    host = {v: tiles[v].cpu().numpy() for v in (0, 1, 4, 16)}
    n_equal = 0
    for r in cap.records:
        i, j = r["tag"]
        fixed, moving, binning = at_size.oracle_registration_crops(host[i], host[j], origins[i], origins[j], [1.0, 1.0, 1.0])
        assert binning == {"z": 2, "y": 2, "x": 2}
        assert min(fixed.shape) >= 50
        for got, want in ((r["fixed"], fixed), (r["moving"], moving)):
            n_equal += int(at_size.assert_crops_equal(got, want, exact=False))
    assert n_equal == 6          # half-pixel taps of integer-valued voxels are exact in float32: the crops are the same bits


def test_north_star_pruned_search_equals_full_scoring_on_all_pairs(hip_device):
    """The pruned arg-max search of mvs_register_crops (option "ssim_prune") on the north-star mosaic, every one of
    the 144 pairs, on all 16 context lanes: the pairwise translations, qualities and the resolved parameters are those of the run
    that scores every candidate in full -- bit for bit -- on the bench's clean tiles and on tiles with independent noise on top
    (the overlaps are then no longer copies of each other: the winner's SSIM drops and candidates leave later or not at all)."""
    _need_torch()
    import bench
    from multiview_stitcher_amd import _lib, param_utils, registration
    from multiview_stitcher_amd import spatial_image_utils as si

    dev = torch.device("cuda", 0)
    grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
    overlap = np.round(tile * 0.2).astype(int)
    tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=99)
    key = si.DEFAULT_TRANSFORM_KEY
    lanes = [lane << 8 for lane in range(16)]

    def run(flag):
        for d in lanes:
            _lib.init(d)
            _lib.set_option("ssim_prune", flag, d)
            for k in ("reg_pruned", "reg_cand_volumes", "reg_candidates"):
                _lib.get_counter(k, d, reset=True)
        res = registration.register(sims, transform_key=key, new_transform_key="reg", device=0, return_dict=True)
        stats = {k: sum(_lib.get_counter(k, d, reset=True) for d in lanes) for k in ("reg_pruned", "reg_cand_volumes", "reg_candidates")}
        return res, stats

    try:
        for noise in (0, 40):
            if noise:      # independent "camera noise" per tile (the mosaic's values span 0..4095)
                g = torch.Generator(device=dev)
                g.manual_seed(5)
                for t in tiles:
                    ti = t.view(torch.int16)          # (values stay far below 2^15: the signed view holds them)
                    n = torch.randn(t.shape, generator=g, device=dev, dtype=torch.float16) * noise
                    ti.copy_((ti.to(torch.float32) + n.to(torch.float32)).clamp_(0, 30000).to(torch.int16))
                    del n
            sims = bench.build_sims(tiles, origins, 0)
            torch.cuda.synchronize()
            got, st1 = run(1)
            want, st0 = run(0)
            assert got["pairwise_registration"]["edges"] == want["pairwise_registration"]["edges"]
            assert len(got["pairwise_registration"]["edges"]) == 144
            for a, b in zip(got["pairwise_registration"]["results"][0], want["pairwise_registration"]["results"][0]):
                np.testing.assert_array_equal(np.asarray(a["transform"]), np.asarray(b["transform"]))
                assert a["quality"] == b["quality"]
            for p, q in zip(got["params"], want["params"]):
                np.testing.assert_array_equal(param_utils.select_time(p, 0), param_utils.select_time(q, 0))
            assert st0["reg_pruned"] == 0 and st0["reg_cand_volumes"] == st0["reg_candidates"]
            assert st1["reg_cand_volumes"] < st1["reg_candidates"]
            at_size._record({"config": "north star, pruned arg-max search vs full scoring, noise sigma %d" % noise, "pairs": 144,
                             "candidates": st1["reg_candidates"], "left_unfinished": st1["reg_pruned"],
                             "candidate_volumes_walked": round(st1["reg_cand_volumes"], 2), "identical": True, "voxels": 0,
                             "beyond_plain_bar": 0})
    finally:
        for d in lanes:
            _lib.set_option("ssim_prune", 1, d)
