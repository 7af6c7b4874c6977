import sys, time
sys.path.insert(0, ".")
import numpy as np
from multiview_stitcher_amd import _lib
from multiview_stitcher_amd.device import DeviceArray
lane = 15 << 8
_lib.init(0); _lib.init(lane)
for rep in range(5):
    t0 = time.perf_counter(); a = DeviceArray.empty((64, 256, 256, 256), np.uint16, lane); t1 = time.perf_counter()
    views = [a[i] for i in range(64)]; t2 = time.perf_counter()
    del a; del views; t3 = time.perf_counter()
    print("alloc %.3f ms  views %.3f ms  free %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
