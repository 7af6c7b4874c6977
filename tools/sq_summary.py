"""Per-kernel SQ picture from a rocprofv3 --pmc counter_collection.csv (one lane, kernels alone): share of the wave cycles spent issuing
(ACTIVE_INST_ANY, of which VALU / LDS), parked at s_waitcnt / barriers (WAIT_ANY) and stalled at issue (WAIT_INST_ANY); VALU instructions
per wave.  WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:44]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (name, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[name] += 1
print(f"{'kernel':44s} {'launches':>8s} {'issuing':>8s} {'VALU':>6s} {'LDS':>6s} {'parked':>7s} {'stalled':>8s} {'VALU inst/wave':>15s} {'waves/launch':>13s}")
rows = []
for name, c in acc.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0: continue
    rows.append((wc, name, c))
for wc, name, c in sorted(rows, reverse=True)[:24]:
    w = max(c.get("SQ_WAVES", 0.0), 1.0)
    print(f"{name:44s} {cnt[name]:8d} {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:8.2f} {c.get('SQ_ACTIVE_INST_VALU', 0) / wc:6.2f} {c.get('SQ_ACTIVE_INST_LDS', 0) / wc:6.2f} "
          f"{c.get('SQ_WAIT_ANY', 0) / wc:7.2f} {c.get('SQ_WAIT_INST_ANY', 0) / wc:8.2f} {c.get('SQ_INSTS_VALU', 0) / w:15.0f} {w / max(cnt[name], 1):13.0f}")
