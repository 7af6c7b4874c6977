#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun):
#   1. kernel-trace --stats of the default bench command        -> gpurun_out/prof_bench/
#   2. PMC passes (kernel-filtered, separate runs) of the fuse kernel: FETCH_SIZE, WRITE_SIZE
#      on the north-star mosaic and on a single-tile calibration case (known byte count)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_bench $O/pmc_fetch $O/pmc_write $O/pmc_cal_fetch $O/pmc_cal_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
timeout 600 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/tools/fuse_probe.py 2 1 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/tools/fuse_probe.py 2 1 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc FETCH_SIZE --output-format csv -d $O/pmc_cal_fetch -- python $R/tools/fuse_probe.py 2 0 1,1,1 512,512,512 > $O/pmc_cal_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc WRITE_SIZE --output-format csv -d $O/pmc_cal_write -- python $R/tools/fuse_probe.py 2 0 1,1,1 512,512,512 > $O/pmc_cal_write.log 2>&1
grep -h shape $O/pmc_fetch.log $O/pmc_cal_fetch.log
tail -2 $O/prof_bench.log
