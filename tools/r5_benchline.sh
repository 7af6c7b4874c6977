#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5line; mkdir -p $O
timeout 1500 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_line.err; echo rc $?
head -c 700 $O/bench_line.json
