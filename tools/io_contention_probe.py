"""Host-side I/O stages of the streamed fuse alone and together (no GPU work): chunk-file reads into a buffer, chunk-file writes out
of one, and both at once -- with the cgroup's CPU throttling counters around each leg.  python tools/io_contention_probe.py [dir]"""
import os, shutil, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multiview_stitcher_amd import streaming, zarr_io

base = sys.argv[1] if len(sys.argv) > 1 else tempfile.gettempdir()
tmp = tempfile.mkdtemp(prefix="mvs_io_", dir=base)
shape = (512, 1024, 1024)
rng = np.random.default_rng(0)
data = rng.integers(0, 4096, shape, dtype=np.uint16)
tiles = []
for i in range(4):
    a = zarr_io.ZarrArray.create(os.path.join(tmp, f"t{i}.zarr"), shape, [128] * 3, np.uint16)
    streaming.write_region(a, [0, 0, 0], data)
    tiles.append(a)
buf = np.empty(shape, np.uint16)


def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)) / 1e3
    except OSError:
        return 0, 0.0


def reads():
    t0 = time.perf_counter()
    for a in tiles:
        streaming.read_window(a, buf)
    return time.perf_counter() - t0


def writes(tag):
    t0 = time.perf_counter()
    for i in range(4):
        p = os.path.join(tmp, f"o{tag}{i}.zarr")
        o = zarr_io.ZarrArray.create(p, shape, [256] * 3, np.uint16)
        streaming.write_region(o, [0, 0, 0], data)
    return time.perf_counter() - t0


gb = 4 * data.nbytes / 1e9
print("dir", base, "io threads", os.environ.get("MVS_IO_THREADS", "8"), "affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
for rep in range(2):
    n0 = throttled(); tr = reads(); n1 = throttled()
    print("reads alone  %.3f s  %.1f GB/s  throttled +%d periods %.0f ms" % (tr, gb / tr, n1[0] - n0[0], n1[1] - n0[1]))
    n0 = throttled(); tw = writes("a%d" % rep); n1 = throttled()
    print("writes alone %.3f s  %.1f GB/s  throttled +%d periods %.0f ms" % (tw, gb / tw, n1[0] - n0[0], n1[1] - n0[1]))
    res = {}
    th = threading.Thread(target=lambda: res.__setitem__("r", reads()))
    n0 = throttled(); t0 = time.perf_counter(); th.start(); res["w"] = writes("b%d" % rep); th.join(); both = time.perf_counter() - t0; n1 = throttled()
    print("both at once %.3f s (reads %.3f, writes %.3f)  throttled +%d periods %.0f ms" % (both, res["r"], res["w"], n1[0] - n0[0], n1[1] - n0[1]))
shutil.rmtree(tmp, ignore_errors=True)
