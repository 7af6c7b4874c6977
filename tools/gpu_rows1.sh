#!/bin/bash
# parity of the row-owning fuse kernels + A/B timing against the region kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/rows1; mkdir -p $O
timeout 900 python -m pytest tests/test_fuse_gpu.py tests/test_register_fuse_gpu.py -x -q -k "rowlds or kat" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for mode in ${MODES:-0 2}; do
  for ab in ${ABL:-0}; do echo "== rowlds mode $mode ablate $ab"; MVS_ROWLDS=1 MVS_ABLATE=$ab MVS_PLAN_STATS=1 timeout 300 python tools/fuse_probe.py 4 $mode 2>&1 | tail -2; done
  echo "== regions mode $mode"; timeout 300 python tools/fuse_probe.py 4 $mode 2>&1 | tail -1
done
