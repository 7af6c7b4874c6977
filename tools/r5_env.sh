#!/bin/bash
# runtime environment knobs against the pair loop (one box, alternating): signal waits by polling, kernel arguments in device memory, queues
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5_env.txt
: > $O
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
  python - "$label" <<'PY' | tee -a $O
import json, sys
try:
    r = json.load(open("gpurun_out/b.json"))
    c = r["config"]
    print("%-40s ms_per_step %.2f register %.2f pairwise %.2f fuse %.2f serial_host %.2f" % (sys.argv[1], r["ms_per_step"], c.get("register_ms_per_step", float("nan")), c.get("pairwise_ms_per_step"), c.get("fuse_ms_per_step"), c.get("serial_host_ms", float("nan"))))
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/b.err").read()[-300:])
PY
}
for rep in 1 2; do
  run "default" A=1
  run "HSA_ENABLE_INTERRUPT=0" HSA_ENABLE_INTERRUPT=0
  run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
  run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
  run "GPU_MAX_HW_QUEUES=6" GPU_MAX_HW_QUEUES=6
  HSA_ENABLE_INTERRUPT=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --reg-threads 12 > gpurun_out/b.json 2> gpurun_out/b.err; run_parse() { :; }; python - "INTERRUPT=0 + 12 lanes" <<'PY' | tee -a $O
import json, sys
r = json.load(open("gpurun_out/b.json")); c = r["config"]
print("%-40s ms_per_step %.2f register %.2f pairwise %.2f" % (sys.argv[1], r["ms_per_step"], c.get("register_ms_per_step"), c.get("pairwise_ms_per_step")))
PY
done
