#!/bin/bash
# SQ counters of the registration kernels, alone (one lane, one register() after warm-up)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5sq; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-include-regex "ssim|fft|dft_line|slab|long_xp|hist|rank|updft|crop|shift|rescale" --output-format csv -d $O/t -- python $R/tools/sched_probe.py auto 1 1 > $O/log.txt 2>&1
tail -2 $O/log.txt
cd $R; python tools/sq_summary.py $(find $O/t -name "*counter_collection.csv") | tee $O/sq_summary.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
