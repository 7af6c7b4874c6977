#!/bin/bash
# Round-2 evidence, collected on the GPU box through gpurun (everything lands in gpurun_out/r2prof; the summaries are copied
# to profiles/round2_* by tools/collect_round2.py):
#   1. rocprofv3 --kernel-trace --stats of the bench command (16 pair lanes = the default; and with 1 lane = isolated kernel durations)
#   2. PMC passes (ONE counter per run, kernel-filtered) of the fuse launch: FETCH_SIZE / WRITE_SIZE for the region kernels
#      (integer and fractional offsets, single-tile calibration), for the opt-in row kernel (rowlds) and for the content-based chunk
#   3. HIP-event timings of the fuse variants, the LDS-DMA / store microbenchmark, the content-based probe, host overheads
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2prof
rm -rf $O; mkdir -p $O
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- $BENCH > $O/bench.log 2>&1
python $R/tools/fuse_window.py $(find $O/bench -name "*kernel_trace.csv") > $O/fuse_launch_windows.csv
python $R/tools/kstats.py $(find $O/bench -name "*kernel_stats.csv") 60 > $O/bench_kstats.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench1 -- $BENCH --reg-threads 1 > $O/bench1.log 2>&1
python $R/tools/kstats.py $(find $O/bench1 -name "*kernel_stats.csv") 60 > $O/bench1_kstats.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_int_$c -- python $R/tools/fuse_probe.py 2 0 > $O/pmc_int_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_frac_$c -- python $R/tools/fuse_probe.py 2 1 > $O/pmc_frac_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_cal_$c -- python $R/tools/fuse_probe.py 2 0 1,1,1 512,512,512 > $O/pmc_cal_$c.log 2>&1
  MVS_ROWLDS=1 timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_rowlds_$c -- python $R/tools/fuse_probe.py 2 0 > $O/pmc_rowlds_$c.log 2>&1
  MVS_STREAM=1 timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_stream_$c -- python $R/tools/fuse_probe.py 2 0 > $O/pmc_stream_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-include-regex "gauss|cb_fuse|prep_kernel|ng_finish|mask_normalize|resample_kernel|blend_kernel" --pmc $c --output-format csv -d $O/pmc_cb_$c -- python $R/tools/cb_probe.py > $O/pmc_cb_$c.log 2>&1
done
for d in int frac cal rowlds stream cb; do for c in FETCH_SIZE WRITE_SIZE; do echo "== $d $c"; grep -h "kernel ms\|Mvoxels" $O/pmc_${d}_$c.log | tail -2; python $R/tools/pmc_summary.py $(find $O/pmc_${d}_$c -name "*counter_collection.csv"); done; done > $O/pmc_summary.txt 2>&1
cd $R
{
  for f in 0 2 1; do echo "== regions frac=$f"; python tools/fuse_probe.py 5 $f 2>&1 | grep "kernel ms" | tail -1; done
  echo "== rowlds exact grid"; MVS_ROWLDS=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms" | tail -1
  echo "== rowlds jittered (falls back)"; MVS_ROWLDS=1 python tools/fuse_probe.py 5 2 2>&1 | grep "kernel ms" | tail -1
  echo "== rows_v1 exact grid"; MVS_ROWS_V1=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms" | tail -1
  echo "== serial classes"; MVS_SERIAL=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms" | tail -1
  echo "== stream rows exact grid"; MVS_STREAM=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms" | tail -1
  echo "== stream rows jittered"; MVS_STREAM=1 python tools/fuse_probe.py 5 2 2>&1 | grep "kernel ms" | tail -1
  echo "== stream rows exact grid, ablation 1 (weights = clamp(W), no ramp polynomial)"; MVS_ABLATE=1 MVS_STREAM=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms" | tail -1
  echo "== stream rows exact grid, ablation 2 (no tile loads)"; MVS_ABLATE=2 MVS_STREAM=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms" | tail -1
  echo "== stream rows exact grid, ablation 4 (all weights 1: loads, sums and stores only)"; MVS_ABLATE=4 MVS_STREAM=1 python tools/fuse_probe.py 5 0 2>&1 | grep "kernel ms" | tail -1
} > $O/fuse_variants.txt 2>&1
[ -x tools/dma_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/dma_probe tools/dma_probe.hip
./tools/dma_probe > $O/dma_probe.txt 2>&1
[ -x tools/ubench/valu_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip
./tools/ubench/valu_rate > $O/valu_rate.txt 2>&1
python tools/cb_probe.py > $O/cb_probe.txt 2>&1
python tools/pair_overhead.py > $O/pair_overhead.txt 2>&1
python tools/host_profile.py 2>&1 | cut -c1-170 | grep -v "^$" | head -70 > $O/host_profile.txt
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err
tail -3 $O/bench.log; cat $O/fuse_variants.txt; head -c 1500 $O/bench_line.json
