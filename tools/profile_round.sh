#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun):
#   1. kernel-trace --stats of the default bench command        -> gpurun_out/prof_bench/
#   2. PMC passes (kernel-filtered, ONE counter per run) of the fuse kernels: FETCH_SIZE, WRITE_SIZE on the
#      north-star mosaic (integer offsets = what the registered bench mosaic has, and fractional offsets)
#      and on a single-tile calibration case (known byte count)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_bench $O/pmc_*
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
python $R/tools/fuse_window.py $(find $O/prof_bench -name "*kernel_trace.csv") > $O/prof_bench/fuse_launch_windows.csv
find $O/prof_bench -name "*kernel_trace.csv" -delete; find $O/prof_bench -name "*.db" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_int_$c -- python $R/tools/fuse_probe.py 2 0 > $O/pmc_int_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_frac_$c -- python $R/tools/fuse_probe.py 2 1 > $O/pmc_frac_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_cal_$c -- python $R/tools/fuse_probe.py 2 0 1,1,1 512,512,512 > $O/pmc_cal_$c.log 2>&1
done
for d in int frac cal; do for c in FETCH_SIZE WRITE_SIZE; do echo "== $d $c"; grep -h "kernel ms" $O/pmc_${d}_$c.log; python $R/tools/pmc_summary.py $(find $O/pmc_${d}_$c -name "*counter_collection.csv"); done; done
tail -1 $O/prof_bench.log
python $R/tools/kstats.py $(find $O/prof_bench -name "*kernel_stats.csv") 30
cat $O/prof_bench/fuse_launch_windows.csv
