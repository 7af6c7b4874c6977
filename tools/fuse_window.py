"""Wall window of every fuse launch in a rocprofv3 kernel_trace.csv: the five class kernels (copy_region_kernel, fuse_region_kernel<1|2|4|8>)
run side by side on forked streams, so the launch's duration is first start -> last end, not the sum of the kernels' durations."""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(r"copy_region_kernel|fuse_region_kernel", r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups, cur = [], []
for r in rows:
    if cur and int(r["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in cur) > 1_000_000:   # > 1 ms apart: next launch
        groups.append(cur); cur = []
    cur.append(r)
if cur: groups.append(cur)
print("launch,n_kernels,window_ms,sum_of_kernel_durations_ms")
for k, g in enumerate(groups):
    t0 = min(int(x["Start_Timestamp"]) for x in g); t1 = max(int(x["End_Timestamp"]) for x in g)
    tot = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in g)
    print(f"{k},{len(g)},{(t1 - t0) / 1e6:.3f},{tot / 1e6:.3f}")
