#!/bin/bash
# single-lane kernel statistics of the bench (isolated kernel durations) + a default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/ks; mkdir -p $O; rm -rf $O/prof
[ -n "$TESTS" ] && timeout 900 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --reg-threads 1 > $R/$O/prof.log 2>&1
find $R/$O/prof -name "*kernel_trace.csv" -delete; find $R/$O/prof -name "*.db" -delete
python $R/tools/kstats.py $(find $R/$O/prof -name "*kernel_stats.csv") ${NK:-24} | grep -v "elementwise\|avg_pool"
cd $R; timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pcie > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/ks/bench.json"))
print(d["value"], d["ms_per_step"], {k:d["config"][k] for k in ("register_ms_per_step","pairwise_ms_per_step","fuse_ms_per_step","fuse_kernel_ms","registration_max_abs_error_px")})
PY
