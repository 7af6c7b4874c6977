import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from multiview_stitcher_amd import _lib, _reg_ops
from multiview_stitcher_amd.device import DeviceArray
_lib.init(0)
for shape in [(51, 256, 256), (256, 256, 51), (256, 51, 256)]:
    a = (np.random.default_rng(0).random(shape) + 1j * np.random.default_rng(1).random(shape)).astype(np.complex64)
    lib = _lib.load()
    import ctypes as C
    d = DeviceArray.from_host(a.view(np.float32).reshape(shape + (2,)), 0) if hasattr(DeviceArray, "from_host") else None
    s3 = (C.c_int64 * 3)(*shape)
    for r in range(3):
        rc = lib.mvs_fft_c2c(0, C.c_void_p(d.ptr), 1, 3, s3, 0)
    _lib.synchronize(0)
    t = time.perf_counter()
    for r in range(20):
        rc = lib.mvs_fft_c2c(0, C.c_void_p(d.ptr), 1, 3, s3, 0)
    _lib.synchronize(0)
    print(shape, "fft3 %.1f us" % ((time.perf_counter() - t) / 20 * 1e6), rc)
