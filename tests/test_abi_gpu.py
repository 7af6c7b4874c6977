"""GPU: error behaviour and memory helpers of the C ABI (include/mvs_hip.h): 0 = ok, negative = error with a message in
mvs_last_error(device); the Python shim turns that into RuntimeError."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_invalid_arguments_return_codes_and_messages(hip_device):
    from multiview_stitcher_amd import _lib

    lib = _lib.init(0)
    a = np.zeros((4, 4), np.float32)
    shape = _lib.i64x3((1, 4, 4))
    shift, peak, pabs = (C.c_double * 3)(), (C.c_int64 * 3)(), C.c_float()
    rc = lib.mvs_phasecorr(0, a.ctypes.data, a.ctypes.data, _lib.MVS_MEM_HOST, 4, shape, 1, 1, shift, peak, C.byref(pabs))   # ndim 4
    assert rc < 0 and b"ndim" in lib.mvs_last_error(0)
    rc = lib.mvs_phasecorr(0, None, a.ctypes.data, _lib.MVS_MEM_HOST, 2, shape, 1, 1, shift, peak, C.byref(pabs))
    assert rc < 0 and lib.mvs_last_error(0)
    with pytest.raises(RuntimeError, match="mvs_phasecorr"):
        _lib.check(rc, 0, "mvs_phasecorr")
    # transform lengths beyond the four-step path (2^22) are refused, not silently mishandled
    big = np.zeros((1, 1, (1 << 22) + 2), np.complex64)
    rc = lib.mvs_fft_c2c(0, big.ctypes.data, _lib.MVS_MEM_HOST, 2, _lib.i64x3((1, 1, (1 << 22) + 2)), 0)
    assert rc < 0 and b"4194304" in lib.mvs_last_error(0)


def test_pool_recycles_blocks_and_copy_into_checks_bounds(hip_device):
    from multiview_stitcher_amd import _lib
    from multiview_stitcher_amd.device import DeviceArray

    lib = _lib.init(0)
    p1, p2 = C.c_void_p(), C.c_void_p()
    assert lib.mvs_malloc(0, 3 << 20, C.byref(p1)) == 0
    assert lib.mvs_free(0, p1) == 0
    # same size class: the cached block comes back -- possibly after other cached blocks of that class (left by earlier tests)
    got = []
    for _ in range(64):
        assert lib.mvs_malloc(0, 3 << 20, C.byref(p2)) == 0
        got.append(p2.value)
        if p2.value == p1.value:
            break
    assert p1.value in got
    for q in got:
        assert lib.mvs_free(0, C.c_void_p(q)) == 0
    small = DeviceArray.from_host(np.ones((4, 4), np.uint16), 0)
    dst = DeviceArray.empty((6, 6), np.uint16, 0)
    with pytest.raises(RuntimeError, match="does not fit"):
        small.copy_into(dst, (3, 0))
    dst.fill_zero()
    small.copy_into(dst, (2, 1))
    want = np.zeros((6, 6), np.uint16)
    want[2:6, 1:5] = 1
    np.testing.assert_array_equal(dst.get(), want)


def test_out_of_memory_has_its_own_code_and_exception(hip_device):
    """A HIP allocation that fails with hipErrorOutOfMemory comes back as MVS_ERR_OUT_OF_MEMORY (-5), which the Python shim raises
    as DeviceMemoryError -- the class fuse()'s fallback from merged launch blocks to the requested chunk grid keys on (ADVICE
    round 3: no matching of message text); other failures stay MvsError with their own code; unknown options are refused."""
    from multiview_stitcher_amd import _lib

    lib = _lib.init(0)
    free_b, total_b = _lib.mem_info(0)
    p = C.c_void_p()
    rc = lib.mvs_malloc(0, C.c_size_t(total_b * 4), C.byref(p))                  # four times the device: cannot succeed
    assert rc == _lib.ERR_OUT_OF_MEMORY, (rc, lib.mvs_last_error(0))
    with pytest.raises(_lib.DeviceMemoryError) as ei:
        _lib.DeviceBuffer(0, total_b * 4)
    assert ei.value.code == -5 and isinstance(ei.value, RuntimeError)
    # the context is still usable afterwards
    buf = _lib.DeviceBuffer(0, 1 << 20)
    buf.free()
    # ... and the failed allocation does not linger in the runtime's per-thread "last error": the next call that checks
    # hipGetLastError() after its launches (mvs_resample) must come back clean
    from multiview_stitcher_amd import transformation
    src = np.arange(64, dtype=np.float32).reshape(4, 4, 4)
    got = transformation.resample_array(src, np.eye(3), np.zeros(3), (4, 4, 4), 1, 0.0, 0)
    np.testing.assert_array_equal(got, src)
    with pytest.raises(_lib.MvsError) as ei:
        _lib.set_option("rowlds", 1)                                             # retired in round 3
    assert ei.value.code == -1 and not isinstance(ei.value, _lib.DeviceMemoryError)


def test_a_failing_composite_call_leaves_no_state_on_its_lane(hip_device):
    """mvs_register_views / mvs_register_crops hand their per-call knowledge (statistics parked by the crop kernels, "only the
    arg max is needed", "both crops are finite", deferred waits) to the inner steps as ARGUMENTS (MvsScoreOpts / MvsCropStats /
    MvsResampleOpts in csrc/mvs_internal.h), not as context fields: a composite call that fails half way -- here the second crop
    is refused after the first one has been queued with its statistics -- leaves nothing behind.  On the same lane afterwards the
    public mvs_score_candidates scores every candidate in full (nothing pruned), mvs_resample to host memory has waited for its
    result, and a registration equals the one of an untouched lane bit for bit."""
    from scipy import ndimage

    from multiview_stitcher_amd import _lib, _reg_ops
    from multiview_stitcher_amd.device import DeviceArray
    from oracle import reg_oracle as ro

    lane_a, lane_b = 0 | (3 << 8), 0 | (4 << 8)
    lib = _lib.init(lane_a)
    _lib.init(lane_b)
    rng = np.random.default_rng(4)
    big = (ndimage.gaussian_filter(rng.random((44, 88, 140)), 1.2) * 4000).astype(np.uint16)
    t0, t1 = np.ascontiguousarray(big[2:42, 4:84, 6:126]), np.ascontiguousarray(big[3:43, 2:82, 10:130])
    d0, d1 = DeviceArray.from_host(t0, lane_a), DeviceArray.from_host(t1, lane_a)
    shape = (40, 80, 120)

    def view(d, bad=False):
        v = _lib.mvs_view_t()
        v.data, v.dtype, v.mem = d.ptr, _lib.MVS_U16, _lib.MVS_MEM_DEVICE
        v.shape[:] = list(shape)
        v.stride[:] = [80 * 120, 120, 2 if bad else 1]            # an x stride of 2 is refused by the resampler
        v.matrix[:] = [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0]
        v.offset[:] = [0.0, 0.0, 0.0]
        return v

    t, q, status, ncand = (C.c_double * 3)(), C.c_double(), C.c_int32(), C.c_int32()

    def register(lane, v0, v1):
        return lib.mvs_register_views(lane, C.byref(v0), C.byref(v1), 3, _lib.i64x3(shape), 2, -1, 1, t, C.byref(q), C.byref(status), C.byref(ncand))

    assert register(lane_b, view(d0), view(d1)) == 0                 # reference result from an untouched lane
    want = (list(t), q.value, status.value, ncand.value)
    for key in ("reg_pruned", "reg_cand_volumes", "reg_candidates"):
        _lib.get_counter(key, lane_a, reset=True)
    assert register(lane_a, view(d0), view(d1, bad=True)) < 0       # fails after the fixed crop (and its statistics) were queued
    assert lib.mvs_last_error(lane_a)
    # (1) the public scoring call on that lane: every candidate in full
    a = ro.rescale_intensity_01(t0.astype(np.float32))
    b = ro.rescale_intensity_01(t1.astype(np.float32))
    cands = np.array([[-1.0, 2.0, -4.0], [1.0, -2.0, 4.0], [0.0, 0.0, 0.0], [-1.0, 2.0, 4.0]])
    ssim, spear, codes = _reg_ops.score_candidates(a, b, cands, "union", 1.0, 0.0, device=lane_a)
    assert _lib.get_counter("reg_pruned", lane_a) == 0
    ssim_b, spear_b, codes_b = _reg_ops.score_candidates(a, b, cands, "union", 1.0, 0.0, device=lane_b)
    np.testing.assert_array_equal(ssim, ssim_b)
    np.testing.assert_array_equal(spear, spear_b)
    # (2) a plain resample to HOST memory returns a finished result
    out = np.full(shape, -1.0, np.float32)
    v0 = view(d0)
    assert lib.mvs_resample(lane_a, C.byref(v0), _lib.i64x3(shape), 1, np.nan, out.ctypes.data, _lib.MVS_MEM_HOST) == 0
    np.testing.assert_array_equal(out, t0.astype(np.float32))
    # (3) the same registration on the lane that failed == the untouched lane's
    assert register(lane_a, view(d0), view(d1)) == 0
    assert (list(t), q.value, status.value, ncand.value) == want
    assert want[2] == 0 and [abs(v) for v in want[0]] == [1.0, 2.0, 4.0]


def test_measurement_counters_of_the_fuse_classes_and_the_pool(hip_device):
    """mvs_get_counter: "fuse_class_{in_vox,out_vox,ms}_<k>" of the last region-kernel launch -- the classes' boxes tile the chunk (their
    voxels add up to it; views x voxels is what the launch cannot avoid reading), a class kernel's own time exists only for a launch
    made with option serial_classes -- and "pool_misses" / "pool_miss_bytes" / "pool_releases" of a lane's allocation cache."""
    from multiview_stitcher_amd import _lib, fusion, sample_data
    from multiview_stitcher_amd import spatial_image_utils as si
    from multiview_stitcher_amd.device import to_device
    from tests.helpers import squeeze_field

    lib = _lib.init(0)
    sims, _, _ = sample_data.generate_tiled_dataset(ndim=3, tile_shape=(40, 48, 56), tiles=(2, 2, 2), overlap=(8, 10, 12), dtype=np.uint16, seed=2)
    sims = [to_device(squeeze_field(s), 0) for s in sims]
    key = si.DEFAULT_TRANSFORM_KEY
    for serial in (1, 0):
        _lib.set_option("serial_classes", serial, 0)
        try:
            out = fusion.fuse(sims, transform_key=key, output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=0)
            _lib.synchronize(0)
        finally:
            _lib.set_option("serial_classes", 0, 0)
        ov = [_lib.get_counter(f"fuse_class_out_vox_{k}", 0) for k in range(5)]
        iv = [_lib.get_counter(f"fuse_class_in_vox_{k}", 0) for k in range(5)]
        ms = [_lib.get_counter(f"fuse_class_ms_{k}", 0) for k in range(5)]
        assert sum(ov) == float(np.prod(out.shape))
        assert ov[4] > 0 and iv[4] == ov[4]                       # copy class: one view per voxel
        assert iv[1] == 2 * ov[1] and ov[1] > 0                   # faces: exactly two views
        assert ov[2] > 0 and 2 * ov[2] < iv[2] <= 4 * ov[2]      # edges: three or four views
        assert ov[3] == 0 or 4 * ov[3] < iv[3] <= 8 * ov[3]     # corners: five to eight
        for k in range(5):
            assert (ms[k] >= 0.0) if (serial and ov[k] > 0) else (ms[k] == -1.0)
    v = C.c_double()
    assert lib.mvs_get_counter(0, b"fuse_class_ms_5", 0, C.byref(v)) < 0 and b"unknown key" in lib.mvs_last_error(0)
    _lib.get_counter("pool_misses", 0, reset=True)
    _lib.get_counter("pool_miss_bytes", 0, reset=True)
    p = C.c_void_p()
    odd = (7 << 20) + 12345 * 4096                                 # a size class nothing else in the suite uses
    assert lib.mvs_malloc(0, odd, C.byref(p)) == 0
    assert _lib.get_counter("pool_misses", 0) == 1 and _lib.get_counter("pool_miss_bytes", 0) >= odd
    assert lib.mvs_free(0, p) == 0
    assert lib.mvs_malloc(0, odd, C.byref(p)) == 0 and _lib.get_counter("pool_misses", 0) == 1      # served from the cache
    assert lib.mvs_free(0, p) == 0
    assert _lib.get_counter("pool_releases", 0) >= 0
