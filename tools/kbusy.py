"""GPU occupancy in time from a rocprofv3 kernel_trace.csv: union of kernel intervals (busy), mean concurrency, and the
idle share inside the window of the registration kernels (between the first and last ssim / fft / sort kernel of a step)."""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = []
reg = re.compile(r"ssim|fft|dft_line|slab_kernel|long_xp|hist_|rank|updft|crop_|shift_|rescale|finish_region|peek|small_copy|fold|radix|argmax|xpower|compact|image_stats|bin_mean|pack_pair")
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if reg.search(r["Kernel_Name"]))
if not iv:
    sys.exit("no registration kernels")
# split into steps: gaps > 8 ms separate them
steps, cur = [], [iv[0]]
for a, b in iv[1:]:
    if a - max(x[1] for x in cur[-50:]) > 8_000_000:
        steps.append(cur); cur = []
    cur.append((a, b))
steps.append(cur)
for k, st in enumerate(steps):
    t0, t1 = st[0][0], max(b for _, b in st)
    busy, end = 0, t0
    for a, b in st:
        if b > end:
            busy += b - max(a, end); end = b
    tot = sum(b - a for a, b in st)
    gaps = []
    end = t0
    for a, b in st:
        if a > end: gaps.append(a - end)
        end = max(end, b)
    print(f"step {k}: {len(st)} kernels, window {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({100 * busy / (t1 - t0):.0f} %), "
          f"sum of durations {tot / 1e6:.1f} ms (mean concurrency {tot / max(busy, 1):.2f}), {len(gaps)} idle gaps, median gap {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us")
