#!/bin/bash
# Ablations and instruction counters of the opt-in row kernel (fuse_rowlds_kernel) on the exact north-star grid.
# MVS_ABLATE bits: 1 no LDS-DMA loads, 2 no arithmetic (stores of zeros), 4 no stores, 8 every cell treated as a one-view copy.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2prof; mkdir -p $O
cd $R
{
for a in 0 1 2 4 8 9 10 12; do echo "== rowlds ablate=$a"; MVS_ROWLDS=1 MVS_ABLATE=$a python tools/fuse_probe.py 4 0 2>&1 | grep "kernel ms" | tail -1; done
echo "== rowlds, overlap 104 px (every cell boundary a multiple of 8 px), ablate 0 and 8"
MVS_ROWLDS=1 python tools/fuse_probe.py 4 0 4,4,4 512,512,512 104 2>&1 | grep "kernel ms" | tail -1
MVS_ROWLDS=1 MVS_ABLATE=8 python tools/fuse_probe.py 4 0 4,4,4 512,512,512 104 2>&1 | grep "kernel ms" | tail -1
} > $O/rowlds_ablation.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  for rl in 1 0; do
    rm -rf $O/sq; MVS_ROWLDS=$rl timeout 300 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $set --output-format csv -d $O/sq -- python $R/tools/fuse_probe.py 2 0 > $O/sq.log 2>&1
    echo "== rowlds=$rl counters: $set"; python $R/tools/pmc_summary.py $(find $O/sq -name "*counter_collection.csv")
  done
done > $O/rowlds_sq_counters.txt 2>&1
rm -rf $O/sq
cat $O/rowlds_ablation.txt; cat $O/rowlds_sq_counters.txt | head -80
