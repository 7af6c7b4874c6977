"""GPU occupancy in time from a rocprofv3 kernel_trace.csv: per step (gaps > 5 ms separate steps) the union of the kernel
intervals (busy), the sum of durations per kernel name, mean concurrency.  python tools/kbusy2.py trace.csv [name regex to keep]"""
import csv, re, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
skip = re.compile(r"elementwise|avg_pool|distribution|fuse_region|copy_region|bin_mean")
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if not skip.search(r["Kernel_Name"]))
steps, cur, end = [], [], None
for a, b, n in iv:
    if end is not None and a - end > 5_000_000:
        steps.append(cur); cur = []
    cur.append((a, b, n)); end = b if end is None else max(end, b)
    if not cur[:-1]: end = b
steps.append(cur)
steps = [s for s in steps if len(s) > 500]
for k, st in enumerate(steps[-3:]):
    t0, t1 = st[0][0], max(b for _, b, _ in st)
    busy, end = 0, t0
    for a, b, _ in st:
        if b > end:
            busy += b - max(a, end); end = b
    tot = sum(b - a for a, b, _ in st)
    print(f"step {k}: {len(st)} kernels, window {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({100 * busy / (t1 - t0):.0f} %), sum of durations {tot / 1e6:.1f} ms "
          f"(mean concurrency {tot / max(busy, 1):.2f})")
st = steps[-1]
per = collections.defaultdict(lambda: [0, 0])
for a, b, n in st:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)[:60]
    per[n][0] += 1; per[n][1] += b - a
for n, (cnt, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {n:60s} {cnt:5d} {t / 1e6:8.2f} ms {t / cnt / 1e3:8.1f} us")
