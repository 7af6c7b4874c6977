"""Where a context lane's time goes between its kernels: from a rocprofv3 kernel_trace.csv, per stream (or queue) the gaps between
consecutive registration kernels, grouped by the kernel that PRECEDES the gap (a host round trip after it shows as a long gap)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
print("columns:", list(rows[0].keys()))
reg = re.compile(r"ssim|fft_|dft_|hist_|shift|updft|rank|rescale|crop_int|finish_region|peek|small_copy|fold")
by = collections.defaultdict(list)
for r in rows:
    if reg.search(r["Kernel_Name"]):
        by[(r.get("Queue_Id"), r.get(key))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:28]))
print("streams with registration kernels:", len(by), "grouped by", key, "; queues:", len({k[0] for k in by}))
gaps = collections.defaultdict(list)
busy = idle = 0
for k, ks in by.items():
    ks.sort()
    for (a0, b0, n0), (a1, b1, n1) in zip(ks, ks[1:]):
        g = a1 - b0
        if g > 3_000_000:      # between steps
            continue
        gaps[n0 + " -> " + n1].append(g)
        idle += max(g, 0)
    busy += sum(b - a for a, b, _ in ks)
print(f"per-stream busy {busy / 1e6:.1f} ms, idle between kernels {idle / 1e6:.1f} ms")
tot = sorted(((sum(v), len(v), k) for k, v in gaps.items()), reverse=True)
print("    total ms   count   median us   transition")
for s, n, k in tot[:40]:
    v = sorted(gaps[k])
    print(f"  {s / 1e6:9.2f}  {n:6d}  {v[len(v) // 2] / 1e3:9.1f}   {k}")
