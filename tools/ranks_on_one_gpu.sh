#!/bin/bash
# N ranks of the north-star bench SHARING the one GPU (gloo control plane): NOT a scaling measurement -- it exercises the sharded
# launch path (tile bricks + halo, balanced pair ownership, sub-box fuse) and yields the per-rank host figures of the line.
# usage (gpurun command): bash tools/ranks_on_one_gpu.sh 2 4
cd $GRAFT_REPO_ROOT
O=gpurun_out/ranks; mkdir -p $O
for n in "${@:-2}"; do
MVS_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 5 --warmup 2 > $O/bench$n.json 2> $O/bench$n.err; echo "rc $?"
python - <<PY
import json
d=json.load(open("$O/bench$n.json")); c=d["config"]
print($n, "ranks: ms/step", round(d["ms_per_step"],1), "register", round(c["register_ms_per_step"],1), "pairwise", round(c["pairwise_ms_per_step"],1), "fuse", round(c["fuse_ms_per_step"],1), "kernel", round(c["fuse_kernel_ms"],1), "serial_host_ms_by_rank", [round(v,2) for v in c["serial_host_ms_by_rank"]], "pairs", c["pairs_per_step_by_rank"], "err", c["registration_max_abs_error_px"], "scaling", d["scaling"])
PY
done
