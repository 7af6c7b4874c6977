#!/bin/bash
# the overlap test in a loop (flakiness), the streaming / fuse tests, the transfer legs of the bench
cd $GRAFT_REPO_ROOT
for i in $(seq ${LOOPS:-8}); do timeout 300 python -m pytest tests/test_stream_gpu.py -x -q -m gpu -k "overlap" 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_zarr_gpu.py tests/test_fuse_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-c3 2>/dev/null > gpurun_out/sc.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/sc.json")); p=d["value_incl_pcie"]; t=p["timeline"]; c5=d["c5_stream"]
print("pcie", round(p["ms"],1), t["slab_fused_ms"], "c5", round(c5["wall_s"],3), round(c5["gb_per_s"],1), "step", round(d["ms_per_step"],2), "fuse kernel", round(d["config"]["fuse_kernel_ms"],2))
PY
done
