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
