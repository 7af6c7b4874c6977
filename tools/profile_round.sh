#!/bin/bash
# Evidence of a round, collected on the GPU box through gpurun: everything lands in gpurun_out/prof_$ROUND, the summaries are copied to
# profiles/${ROUND}_* by tools/collect_round.py $ROUND.   usage (gpurun command): ROUND=round6 bash tools/profile_round.sh
#   1. rocprofv3 --kernel-trace --stats of the bench command (8 lanes = the default; 1 lane = isolated kernel durations), GPU busy
#      share inside the registration window, per-lane gaps
#   2. PMC passes (ONE counter per run, kernel-filtered) of the fuse launch: FETCH_SIZE / WRITE_SIZE (integer offsets = the bench's
#      geometry, fractional offsets, single-tile calibration, the 128-byte-aligned geometry) and of the content-based chunk pipeline
#   3. HIP-event timings: fuse launch variants + the alignment bound (overlap 102 / 104 / 128 px, forked and class by class),
#      content-based probe, host phases
#   4. the bench line itself (cpu_baseline, PCIe leg, C3 and C5 legs, by_class)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-round6}
O=$R/gpurun_out/prof_$ROUND
rm -rf $O; mkdir -p $O
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pcie --no-c3 --no-c5"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- $BENCH > $O/bench.log 2>&1
python $R/tools/fuse_window.py $(find $O/bench -name "*kernel_trace.csv") > $O/fuse_launch_windows.csv
python $R/tools/kstats.py $(find $O/bench -name "*kernel_stats.csv") 60 > $O/bench_kstats.txt
python $R/tools/kbusy.py $(find $O/bench -name "*kernel_trace.csv") 2>&1 | grep -v " 1 kernels" > $O/bench_busy.txt
python $R/tools/lane_gaps.py $(find $O/bench -name "*kernel_trace.csv") > $O/lane_gaps.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench1 -- $BENCH --reg-threads 1 > $O/bench1.log 2>&1
python $R/tools/kstats.py $(find $O/bench1 -name "*kernel_stats.csv") 60 > $O/bench1_kstats.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cbstats -- python $R/tools/cb_probe.py > $O/cbstats.log 2>&1
python $R/tools/kstats.py $(find $O/cbstats -name "*kernel_stats.csv") 30 > $O/cb_kstats.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_int_$c -- python $R/tools/fuse_probe.py 2 2 > $O/pmc_int_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_frac_$c -- python $R/tools/fuse_probe.py 2 1 > $O/pmc_frac_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_cal_$c -- python $R/tools/fuse_probe.py 2 0 1,1,1 512,512,512 > $O/pmc_cal_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_al128_$c -- python $R/tools/fuse_probe.py 2 0 4,4,4 512,512,512 128 > $O/pmc_al128_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-include-regex "gauss|cb_|prep_kernel|ng_finish|mask_normalize|resample_kernel|blend_kernel|crop_int" --pmc $c --output-format csv -d $O/pmc_cb_$c -- python $R/tools/cb_probe.py > $O/pmc_cb_$c.log 2>&1
done
for d in int frac cal al128 cb; do for c in FETCH_SIZE WRITE_SIZE; do echo "== $d $c"; grep -h "kernel ms\|Mvoxels" $O/pmc_${d}_$c.log | tail -2; python $R/tools/pmc_summary.py $(find $O/pmc_${d}_$c -name "*counter_collection.csv"); done; done > $O/pmc_summary.txt 2>&1
cd $R
{
  for f in 0 2 1; do echo "== region kernels frac=$f (0: exact grid, 2: +-3 px integer jitter = the bench geometry, 1: fractional offsets)"; python tools/fuse_probe.py 5 $f 2>&1 | grep "kernel ms" | tail -1; done
  echo "== serial classes"; MVS_SERIAL=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms\|class" | tail -6
} > $O/fuse_variants.txt 2>&1
{
  echo "Alignment bound of the fuse launch (VERDICT round 5 item 4): the SAME kernels on geometries whose rows are better and better aligned."
  echo "overlap 102 px (20 %): cell boundaries at multiples of 2 voxels, output pitch 3484 B; 104 px: every cell boundary a multiple of 8 voxels"
  echo "(16 B: every lane's load and store is 16-byte aligned), pitch 3472 B; 128 px: every cell boundary and the pitch (3328 B) a multiple of 64"
  echo "voxels = 128 B -- what sector-aligned brick cuts + a padded pitch + in-register realignment could reach at best."
  for ov in 102 104 128; do
    echo "== overlap $ov px, exact grid, forked classes"; python tools/fuse_probe.py 6 0 4,4,4 512,512,512 $ov 2>&1 | grep "kernel ms" | tail -1
    echo "== overlap $ov px, exact grid, class by class"; MVS_SERIAL=1 python tools/fuse_probe.py 6 0 4,4,4 512,512,512 $ov 2>&1 | grep "kernel ms\|class" | tail -6
  done
  echo "== overlap 102 px, +-3 px jitter (bench geometry), class by class"; MVS_SERIAL=1 python tools/fuse_probe.py 6 2 2>&1 | grep "kernel ms\|class" | tail -6
} > $O/fuse_alignment.txt 2>&1
python tools/cb_probe.py > $O/cb_probe.txt 2>&1
python tools/host_profile.py 2>&1 | cut -c1-170 | grep -v "^$" | head -110 > $O/host_profile.txt
python tools/register_phases.py > $O/register_phases.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_line.err
tail -3 $O/bench.log | head -c 400; echo; cat $O/fuse_variants.txt; cat $O/fuse_alignment.txt; tail -2 $O/cb_probe.txt; head -c 1500 $O/bench_line.json
