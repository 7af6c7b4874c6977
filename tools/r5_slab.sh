#!/bin/bash
# three-pass phase correlation (mvs_fft_slab.hip): parity tests, then the bench A/B on one box (alternating runs)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5_slab.txt
: > $O
timeout 1500 python -m pytest tests/test_reg_gpu.py tests/test_register_fuse_gpu.py tests/test_pair_batch.py tests/test_abi_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee -a $O
for rep in 1 2; do
  for flag in 1 0; do
    MVS_FFT_NO_SLAB=$flag timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
    python - $flag <<'PY' | tee -a $O
import json, sys
r = json.load(open("gpurun_out/b.json"))
c = r["config"]
print("MVS_FFT_NO_SLAB=%s ms_per_step %.2f register %.2f pairwise %s fuse %s serial_host %.2f" % (sys.argv[1], r["ms_per_step"], c.get("register_ms_per_step", float("nan")), c.get("pairwise_ms_per_step"), c.get("fuse_ms_per_step"), c.get("serial_host_ms", float("nan"))))
PY
  done
done
bash tools/r5_slab_prof.sh 2>&1 | tail -45 | tee -a $O
