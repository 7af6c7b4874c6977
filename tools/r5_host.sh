#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 300 python tools/host_profile.py > $O/host_profile.txt 2>&1; echo "host_profile rc $?"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5c/bench.json"))
c=d["config"]
print("ms/step", d["ms_per_step"], "register", c["register_ms_per_step"], "pairwise", c["pairwise_ms_per_step"], "fuse", c["fuse_ms_per_step"], "kernel", c["fuse_kernel_ms"], "err", c["registration_max_abs_error_px"])
PY
timeout 600 python -m pytest tests/test_register_fuse_gpu.py tests/test_sharding_gpu.py tests/test_bench_launch_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
