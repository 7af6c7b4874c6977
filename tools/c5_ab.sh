#!/bin/bash
# C5 stream leg under launch-block sizes / I/O thread counts (same box).  usage: bash tools/c5_ab.sh "1024 8" "512 8" "256 8" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "${@:-1024 8}"; do set -- $v
echo "== launch block $1 MiB, $2 I/O threads per direction"
MVS_MAX_STREAM_BYTES=$(($1 << 20)) MVS_IO_THREADS=$2 LEGS="--no-c3 --no-pcie" bash tools/gpu_legs.sh 2>&1 | grep "^c5" | cut -c1-330
done
