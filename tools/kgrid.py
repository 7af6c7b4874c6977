"""Durations of one kernel grouped by launch geometry (grid size, LDS) from a rocprofv3 kernel_trace.csv: which of its shapes cost what."""
import collections, csv, re, sys
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if pat and not pat.search(n):
        continue
    key = (re.sub(r"\(anonymous namespace\)::|^void ", "", n)[:36], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in acc.values())
print(f"total {tot / 1e3:.2f} ms over {sum(len(v) for v in acc.values())} launches")
for k, v in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{k[0]:36s} grid {k[1]:>9s} wg {k[2]:>4s} lds {k[3]:>6s}  n={len(v):4d}  mean {sum(v) / len(v):8.1f} us  sum {sum(v) / 1e3:7.2f} ms")
