#!/bin/bash
# Round-4 evidence, collected on the GPU box through gpurun (everything lands in gpurun_out/r4prof; the summaries are copied
# to profiles/round4_* by tools/collect_round4.py):
#   1. rocprofv3 --kernel-trace --stats of the bench command (16 pair lanes = the default; and with 1 lane = isolated kernel durations)
#   2. PMC passes (ONE counter per run, kernel-filtered) of the fuse launch: FETCH_SIZE / WRITE_SIZE for the region kernels
#      (integer and fractional offsets, single-tile calibration) and for the content-based chunk pipeline
#   3. HIP-event timings of the fuse launch, the content-based probe, host overheads, the bench line itself
#   4. the at-size parity tests with their statistics recorded
#   5. per-class alone-times and SQ instruction counts of the fuse launch (serial classes), the host-link ceilings (h2d_probe)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4prof
rm -rf $O; mkdir -p $O
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- $BENCH > $O/bench.log 2>&1
python $R/tools/fuse_window.py $(find $O/bench -name "*kernel_trace.csv") > $O/fuse_launch_windows.csv
python $R/tools/kstats.py $(find $O/bench -name "*kernel_stats.csv") 60 > $O/bench_kstats.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench1 -- $BENCH --reg-threads 1 > $O/bench1.log 2>&1
python $R/tools/kstats.py $(find $O/bench1 -name "*kernel_stats.csv") 60 > $O/bench1_kstats.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cbstats -- python $R/tools/cb_probe.py > $O/cbstats.log 2>&1
python $R/tools/kstats.py $(find $O/cbstats -name "*kernel_stats.csv") 30 > $O/cb_kstats.txt
python $R/tools/kgrid.py $(find $O/cbstats -name "*kernel_trace.csv") "gauss" 24 >> $O/cb_kstats.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_int_$c -- python $R/tools/fuse_probe.py 2 2 > $O/pmc_int_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_frac_$c -- python $R/tools/fuse_probe.py 2 1 > $O/pmc_frac_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_cal_$c -- python $R/tools/fuse_probe.py 2 0 1,1,1 512,512,512 > $O/pmc_cal_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-include-regex "gauss|cb_fuse|prep_kernel|ng_finish|mask_normalize|resample_kernel|blend_kernel|crop_int" --pmc $c --output-format csv -d $O/pmc_cb_$c -- python $R/tools/cb_probe.py > $O/pmc_cb_$c.log 2>&1
done
for d in int frac cal cb; do for c in FETCH_SIZE WRITE_SIZE; do echo "== $d $c"; grep -h "kernel ms\|Mvoxels" $O/pmc_${d}_$c.log | tail -2; python $R/tools/pmc_summary.py $(find $O/pmc_${d}_$c -name "*counter_collection.csv"); done; done > $O/pmc_summary.txt 2>&1
cd $R
{
  for f in 0 2 1; do echo "== region kernels frac=$f (0: exact grid, 2: +-3 px integer jitter = the bench geometry, 1: fractional offsets)"; python tools/fuse_probe.py 5 $f 2>&1 | grep "kernel ms" | tail -1; done
  echo "== rows_v1 exact grid"; MVS_ROWS_V1=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms" | tail -1
  echo "== serial classes"; MVS_SERIAL=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms" | tail -1
} > $O/fuse_variants.txt 2>&1
python tools/cb_probe.py > $O/cb_probe.txt 2>&1
python tools/h2d_probe.py > $O/h2d_probe.txt 2>&1
{ echo "== jittered geometry (the bench's)"; bash tools/gpu_fuse_classes.sh 2; echo "== exact grid"; bash tools/gpu_fuse_classes.sh 0; } > $O/fuse_classes.txt 2>&1
cd $R
python tools/pair_overhead.py > $O/pair_overhead.txt 2>&1
python tools/host_profile.py 2>&1 | cut -c1-170 | grep -v "^$" | head -70 > $O/host_profile.txt
timeout 1500 python bench.py --steps 10 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err
# 4. parity at BASELINE sizes: the statistics of every sampled-oracle check (voxels compared, how many needed the noise floor)
rm -f $O/at_size_parity.jsonl
MVS_AT_SIZE_STATS=$O/at_size_parity.jsonl timeout 900 python -m pytest tests/test_at_size_parity_gpu.py -q -m gpu > $O/at_size_parity.log 2>&1
tail -2 $O/at_size_parity.log
tail -3 $O/bench.log; cat $O/fuse_variants.txt; cat $O/cb_probe.txt | tail -3; head -c 2500 $O/bench_line.json
