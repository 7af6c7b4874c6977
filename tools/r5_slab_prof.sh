#!/bin/bash
# solo (one lane) kernel times per pair orientation with the three-pass phase correlation, and 8-lane stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5slab; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python $R/tools/sched_probe.py auto 1 3 > $O/log.txt 2>&1
cd $R; python tools/kernels_by_orientation.py $(find $O/t -name "*kernel_trace.csv") | tee $O/kernels_by_orientation_slab.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t8 -- python $R/tools/sched_probe.py auto 8 5 > $O/log8.txt 2>&1
cd $R; tail -1 $O/log8.txt
python - $(find $O/t8 -name "*kernel_stats.csv") <<'PY' | tee $O/kstats8.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.1f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
