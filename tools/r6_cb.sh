#!/bin/bash
# round 6: content-based fast path -- parity tests, then the C3-like probe on both paths
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r6
python -m pytest tests/test_fuse_gpu.py -x -q -k "content" 2>&1 | tail -15 > gpurun_out/r6/cb_tests.txt
cat gpurun_out/r6/cb_tests.txt
rm -f gpurun_out/r6/cb_probe.txt
for mode in 0 1; do
  echo "== MVS_CB_EXACT=$mode" >> gpurun_out/r6/cb_probe.txt
  MVS_CB_EXACT=$mode python tools/cb_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6/cb_probe.txt
done
echo "== MVS_CB_TAPS_F64=1" >> gpurun_out/r6/cb_probe.txt
MVS_CB_TAPS_F64=1 python tools/cb_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6/cb_probe.txt
cat gpurun_out/r6/cb_probe.txt
