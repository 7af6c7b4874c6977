#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests/test_reg_gpu.py -x -q -m gpu 2>&1 | tail -3
for t in 16 8 16 8 12; do
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pcie --reg-threads $t > $O/bench_$t.json 2> $O/bench.err; python - $t <<'PY'
import json,sys
d=json.load(open("gpurun_out/r2d/bench_%s.json"%sys.argv[1]))
print(sys.argv[1], d["value"], d["ms_per_step"], {k:d["config"][k] for k in ("register_ms_per_step","pairwise_ms_per_step","fuse_ms_per_step","fuse_kernel_ms","registration_max_abs_error_px")})
PY
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --reg-threads 1 > $R/$O/prof.log 2>&1
find $R/$O/prof -name "*kernel_trace.csv" -delete; find $R/$O/prof -name "*.db" -delete
python $R/tools/kstats.py $(find $R/$O/prof -name "*kernel_stats.csv") 30 | grep -v "elementwise\|avg_pool"
