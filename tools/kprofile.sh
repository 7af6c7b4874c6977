#!/bin/bash
# rocprofv3 of one command: kernel statistics (default) or counters.   usage (gpurun command):
#   bash tools/kprofile.sh [-o NAME] [-k KERNEL_REGEX] [-p "COUNTER ..."] [-e "ENV=V ..."] [-n ROWS] -- <command ...>
#   without -p: --kernel-trace --stats; prints the top ROWS kernels (tools/kstats.py), the GPU-busy share and concurrency
#               (tools/kbusy.py) and, for the fuse launch, its windows (tools/fuse_window.py)
#   with    -p: ONE --pmc pass per counter group given (never combined with a trace, see the gpurun rule); per-kernel means
#               (tools/pmc_summary.py; SQ_* groups: tools/sq_summary.py).  FETCH_SIZE counts half the bytes on gfx950.
# commands used for the records under profiles/:
#   bench (8 lanes / 1 lane)   -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pcie --no-c3 --no-c5 [--reg-threads 1]
#   fuse launch                -k "fuse|copy_region" -- python tools/fuse_probe.py 4 2      (MVS_SERIAL=1: classes one after the other)
#   content-based chain        -k "gauss|cb_" -- python tools/cb_probe.py                   (MVS_CB_EXACT=1: the bit-faithful passes)
#   registration kernels alone -k "ssim|fft|dft_line|slab|long_xp|hist|rank|updft|crop|shift|rescale" -p "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU" -- python tools/sched_probe.py auto 1 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; NAME=kprofile; REGEX=""; PMC=""; ENVS=""; ROWS=40
while getopts "o:k:p:e:n:" o; do case $o in o) NAME=$OPTARG;; k) REGEX=$OPTARG;; p) PMC=$OPTARG;; e) ENVS=$OPTARG;; n) ROWS=$OPTARG;; esac; done; shift $((OPTIND - 1))
[ "$1" = "--" ] && shift
O=$R/gpurun_out/$NAME; rm -rf $O; mkdir -p $O
FILTER=(); [ -n "$REGEX" ] && FILTER=(--kernel-include-regex "$REGEX")
cd $R
if [ -z "$PMC" ]; then
  env $ENVS timeout 600 rocprofv3 --kernel-trace --stats "${FILTER[@]}" --output-format csv -d $O/t -- "$@" > $O/log.txt 2>&1
  tail -3 $O/log.txt | cut -c1-400
  python tools/kstats.py $(find $O/t -name "*kernel_stats.csv") $ROWS | grep -v "elementwise\|avg_pool\|distribution" | tee $O/kstats.txt
  python tools/kbusy.py $(find $O/t -name "*kernel_trace.csv") 2>&1 | grep -v " 1 kernels" | tee $O/busy.txt
  python tools/fuse_window.py $(find $O/t -name "*kernel_trace.csv") > $O/fuse_launch_windows.csv 2>/dev/null
  cp $(find $O/t -name "*kernel_stats.csv") $O/kernel_stats.csv
else
  env $ENVS timeout 600 rocprofv3 --pmc $PMC "${FILTER[@]}" --output-format csv -d $O/t -- "$@" > $O/log.txt 2>&1
  tail -3 $O/log.txt | cut -c1-400
  f=$(find $O/t -name "*counter_collection.csv")
  case "$PMC" in SQ_*) python tools/sq_summary.py $f | tee $O/summary.txt;; *) python tools/pmc_summary.py $f | tee $O/summary.txt;; esac
  cp $f $O/counter_collection.csv
fi
find $O/t -name "*kernel_trace.csv" -delete; find $O/t -name "*.db" -delete
