#!/bin/bash
# Whole -m gpu suite with the at-size parity records, then the bench line with default arguments.  O=gpurun_out/<dir>
cd $GRAFT_REPO_ROOT
O=${O:-gpurun_out/full}; mkdir -p $O; rm -f $O/at_size_parity.jsonl
MVS_AT_SIZE_STATS=$GRAFT_REPO_ROOT/$O/at_size_parity.jsonl timeout ${PYTEST_TIMEOUT:-2400} python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.txt
timeout 900 python bench.py ${BENCH_ARGS} > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"; tail -3 $O/bench_line.err; head -c 1200 $O/bench_line.json; echo
python - <<PY
import json
d = json.load(open("$O/bench_line.json"))
c = d["config"]
print({k: c.get(k) for k in ("pairwise_ms_per_step", "fuse_kernel_ms", "step_cold_ms", "register_first_call_ms", "fuse_first_call_ms", "c3_fuse_mvoxels_s")})
print("by_class", json.dumps(d["roofline"].get("by_class"))[:1500])
print("c5", json.dumps(d.get("c5_stream"))[:1800])
PY
