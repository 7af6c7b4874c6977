"""crop_bin_kernel / crop_int_kernel durations by pair orientation from a 1-lane kernel trace of register() (pairs run in edge order,
two crops each): python tools/crop_by_orientation.py kernel_trace.csv"""
import csv, sys, collections
import numpy as np
sys.path.insert(0, ".")
from multiview_stitcher_amd import mv_graph
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
step = tile - np.round(tile * 0.2).astype(int)
sps = [{"origin": dict(zip("zyx", (np.array(i) * step).astype(float))), "spacing": dict(zip("zyx", [1.0] * 3)), "shape": dict(zip("zyx", [512] * 3)),
        "transform": np.eye(4)} for i in np.ndindex(*grid)]
edges = mv_graph.registration_edges_native(sps, None, None, "alternating_pattern")
orient = [{1: "x", 4: "y", 16: "z"}[b - a] for a, b in edges]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "crop_bin_kernel" in r["Kernel_Name"] or "crop_int_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
for k, r in enumerate(rows):
    pair = (k // 2) % len(edges)
    acc[(orient[pair], "fixed" if k % 2 == 0 else "moving")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key in sorted(acc):
    v = acc[key]
    print(key, "n", len(v), "median us %.1f" % np.median(v), "mean %.1f" % np.mean(v))
