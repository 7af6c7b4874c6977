#!/bin/bash
mkdir -p gpurun_out/prune
for rep in 1 2; do
for L in 4 6 8 10 16; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --reg-threads $L > gpurun_out/prune/l$L.json 2> gpurun_out/prune/l$L.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/prune/l$L.json")); c=d["config"]
    print("lanes $L", "step %.1f reg %.1f pair %.1f fuse %.1f" % (d["ms_per_step"], c["register_ms_per_step"], c["pairwise_ms_per_step"], c["fuse_ms_per_step"]))
except Exception as e: print("$L", "ERR", e)
PY
done; done
