#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/bench; mkdir -p $O
timeout 900 python bench.py --steps ${STEPS:-5} --warmup 1 > $O/bench1.json 2> $O/bench1.err; echo "rc $?"; tail -3 $O/bench1.err; cat $O/bench1.json
if [ -n "$TWO" ]; then
MVS_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench2.json 2> $O/bench2.err; echo "rc $?"; tail -5 $O/bench2.err; cat $O/bench2.json
fi
