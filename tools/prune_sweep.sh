#!/bin/bash
# sweep of the pruned arg-max search: hardware queues, work items per candidate, context lanes (bench line fields)
mkdir -p gpurun_out/prune
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline ${EXTRA} > gpurun_out/prune/$tag.json 2> gpurun_out/prune/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/prune/$tag.json")); c=d["config"]
    print("$tag", "step %.1f reg %.1f pair %.1f fuse %.1f vols %.2f" % (d["ms_per_step"], c["register_ms_per_step"], c["pairwise_ms_per_step"], c["fuse_ms_per_step"], c["candidate_volumes_walked_per_pair"]))
except Exception as e: print("$tag", "ERR", e)
PY
}
run base A=1
run q8 GPU_MAX_HW_QUEUES=8
run q16 GPU_MAX_HW_QUEUES=16
run items160 MVS_SSIM_PRUNE_ITEMS=160
run items640 MVS_SSIM_PRUNE_ITEMS=640
run q8_items640 GPU_MAX_HW_QUEUES=8 MVS_SSIM_PRUNE_ITEMS=640
EXTRA="--reg-threads 8" run lanes8 A=1
EXTRA="--reg-threads 12" run lanes12 A=1
