#!/bin/bash
# per-class alone-times of the fuse launch as it is and with every view counted as a unit view (timing floor, wrong results)
# the "unit" leg needs a profiling build: make -C multiview-stitcher_amd/csrc EXTRA_CXXFLAGS=-DMVS_PROFILING_ABLATIONS (touch mvs_fuse_region.hip first)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/floor; rm -rf $O; mkdir -p $O
for tag in asis unit; do
  [ $tag = unit ] && export MVS_FUSE_ALL_UNIT=1
  MVS_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --kernel-include-regex "fuse|copy_region" --output-format csv -d $O/$tag -- python $R/tools/fuse_probe.py 4 2 > $O/$tag.log 2>&1
  echo "== $tag (serial classes)"; grep -h "kernel ms" $O/$tag.log | tail -1
  python $R/tools/kstats.py $(find $O/$tag -name "*kernel_stats.csv") 8 | grep -v distribution
  echo "== $tag (side by side)"; python $R/tools/fuse_probe.py 5 2 2>/dev/null | grep "kernel ms"
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
