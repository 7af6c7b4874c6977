"""Per-pair kernel time by kernel and pair orientation from a 1-lane kernel trace of register() on the north-star mosaic (pairs run in
edge order; a pair starts with its two crop kernels): python tools/kernels_by_orientation.py kernel_trace.csv"""
import csv, re, sys, collections
import numpy as np
sys.path.insert(0, ".")
from multiview_stitcher_amd import mv_graph
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
step = tile - np.round(tile * 0.2).astype(int)
sps = [{"origin": dict(zip("zyx", (np.array(i) * step).astype(float))), "spacing": dict(zip("zyx", [1.0] * 3)), "shape": dict(zip("zyx", [512] * 3)),
        "transform": np.eye(4)} for i in np.ndindex(*grid)]
edges = mv_graph.registration_edges_native(sps, None, None, "alternating_pattern")
orient = [{1: "x", 4: "y", 16: "z"}[b - a] for a, b in edges]
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
ncrop, pair = 0, -1
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]
    if "crop_bin_kernel" in name or "crop_int_kernel" in name:
        if ncrop % 2 == 0:
            pair += 1
            cnt[orient[pair % len(edges)]] += 1
        ncrop += 1
    if pair < 0:
        continue
    acc[name][orient[pair % len(edges)]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("pairs per orientation:", dict(cnt), "(crop shapes: x = 256 x 256 x 51, y = 256 x 51 x 256, z = 51 x 256 x 256)")
print(f"{'kernel':44s} {'x us/pair':>10s} {'y us/pair':>10s} {'z us/pair':>10s}")
tot = collections.defaultdict(float)
for name, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
    if not any(k in name for k in ("ssim", "fft", "dft", "slab", "long_xp", "hist", "rank", "updft", "crop", "shift", "rescale", "finish", "peek", "small_copy", "fold")):
        continue
    vals = [d.get(o, 0.0) / max(cnt[o], 1) for o in "xyz"]
    for o, v in zip("xyz", vals):
        tot[o] += v
    print(f"{name[:44]:44s} {vals[0]:10.1f} {vals[1]:10.1f} {vals[2]:10.1f}")
print(f"{'sum':44s} {tot['x']:10.1f} {tot['y']:10.1f} {tot['z']:10.1f}")
