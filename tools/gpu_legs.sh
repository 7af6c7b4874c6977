#!/bin/bash
# the side legs of the bench line on a short main loop (PCIe-inclusive, C3, C5) + the launch tests.  usage: [LEGS="..."] bash tools/gpu_legs.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/legs; mkdir -p $O
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${LEGS} > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
p = d.get("value_incl_pcie") or {}
print("pcie", {k: p.get(k) for k in ("value", "ms", "register_phase_ms", "h2d_gb_per_s", "d2h_gb_per_s", "error")})
c5 = d.get("c5_stream") or {}
print("c5", {k: c5.get(k) for k in ("wall_s", "gb_per_s", "stages_alone", "slowest_stage", "frac_of_slowest_stage", "launch_blocks", "error", "skipped")})
for b in c5.get("block_timeline_s") or []: print("   ", b)
c3 = d.get("c3_content_based") or {}
print("c3", {k: c3.get(k) for k in ("ms", "mvoxels_s", "line_launches_per_fuse", "error")})
PY
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -5; fi
