"""Compact sequence view of a rocprofv3 kernel_trace.csv: consecutive identical kernels are merged."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|ROCPRIM_\d+_NS::|detail::|rocprim::|at::native::|^void ", "", n)
    return n[:48]
t_first = int(rows[0]["Start_Timestamp"])
prev, cnt, dur, start = None, 0, 0, 0
out = []
for r in rows:
    n = short(r["Kernel_Name"]); d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if n == prev:
        cnt += 1; dur += d
    else:
        if prev: out.append((start, prev, cnt, dur))
        prev, cnt, dur, start = n, 1, d, int(r["Start_Timestamp"]) - t_first
if prev: out.append((start, prev, cnt, dur))
for s, n, c, d in out[skip:]:
    print(f"{s/1e3:12.1f} us  {n:48s} x{c:<3d} {d/1e3:9.1f} us")
