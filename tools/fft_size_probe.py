import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, ctypes as C
from multiview_stitcher_amd import _lib
from multiview_stitcher_amd.device import DeviceArray
_lib.init(0)
lib = _lib.load()
for shape in [(16, 256, 256), (64, 256, 256), (256, 256, 256), (1024, 256, 256)]:
    a = np.zeros(shape + (2,), np.float32)
    d = DeviceArray.from_host(a, 0)
    s3 = (C.c_int64 * 3)(*shape)
    for r in range(3): lib.mvs_fft_c2c(0, C.c_void_p(d.ptr), 1, 3, s3, 0)
    _lib.synchronize(0)
    t = time.perf_counter()
    for r in range(20): lib.mvs_fft_c2c(0, C.c_void_p(d.ptr), 1, 3, s3, 0)
    _lib.synchronize(0)
    dt = (time.perf_counter() - t) / 20
    n = np.prod(shape)
    print(shape, "fft3 %.1f us, %.2f ns per kilo-element, %.2f TB/s over 3 passes" % (dt * 1e6, dt * 1e9 / (n / 1e3), 3 * 2 * n * 8 / dt / 1e12))
