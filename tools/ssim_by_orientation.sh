#!/bin/bash
# durations of the per-pair kernels grouped by launch geometry (the three pair orientations of the north-star mosaic), one lane
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=/tmp/ori; rm -rf $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --reg-threads 1 > /tmp/ori.log 2>&1
python - <<PY
import csv, glob, collections, re
f = glob.glob("/tmp/ori/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(ssim_fused_batch_kernel|bluestein_reg_kernel|fft_reg2_kernel|updft_yx2_kernel|hist_rank_kernel<false>|hist_rank_kernel<true>|shift_batch_kernel|ssim_yx_fused_kernel|ssim_first_pass_kernel|crop_int_kernel|rescale_pair_kernel)", n)
    if not m: continue
    key = (m.group(1), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y", ""))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    v = acc[k]
    print("%-28s grid %8s x %-4s n %5d  avg %7.1f us  min %7.1f  max %7.1f" % (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v), max(v)))
PY
