#!/bin/bash
# round 5, first GPU session: the changed paths (pruned search fix, native graph / resolution, pre-binning tickets) + bench + host profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_reg_gpu.py tests/test_register_fuse_gpu.py tests/test_abi_gpu.py -x -q -m gpu > $O/pytest_a.log 2>&1; echo "pytest_a rc $?"; tail -3 $O/pytest_a.log
timeout 900 python -m pytest tests/test_at_size_parity_gpu.py -x -q -m gpu -k "north_star" > $O/pytest_b.log 2>&1; echo "pytest_b rc $?"; tail -3 $O/pytest_b.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5a/bench.json"))
c=d["config"]
print("ms/step", d["ms_per_step"], "register", c["register_ms_per_step"], "pairwise", c["pairwise_ms_per_step"], "fuse", c["fuse_ms_per_step"], "kernel", c["fuse_kernel_ms"], "err", c["registration_max_abs_error_px"])
PY
timeout 300 python tools/host_profile.py > $O/host_profile.txt 2>&1; echo "host_profile rc $?"; grep -E "^(register|fuse) ms" $O/host_profile.txt
timeout 300 python tools/register_phases.py > $O/register_phases.txt 2>&1; tail -3 $O/register_phases.txt
