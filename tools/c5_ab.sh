#!/bin/bash
# C5 stream leg under pipeline knobs (same box).  usage: bash tools/c5_ab.sh "READERS WRITERS DEPTH BLOCK_MiB" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "${@:-1 1 2 1024}"; do set -- $v
echo "== readers $1 writers $2 depth $3 block $4 MiB"
MVS_STREAM_READERS=$1 MVS_STREAM_WRITERS=$2 MVS_STREAM_DEPTH=$3 MVS_MAX_STREAM_BYTES=$(($4 << 20)) LEGS="--no-c3 --no-pcie" bash tools/gpu_legs.sh 2>&1 | grep "^c5" | cut -c1-300
done
