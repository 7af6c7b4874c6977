#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MVS_PRUNE_DEBUG=1 timeout 200 python tools/sched_probe.py auto 1 1 2> gpurun_out/prune_dbg.txt | tail -1
grep "^prune" gpurun_out/prune_dbg.txt | tail -144 > gpurun_out/prune_dbg_last.txt
rm gpurun_out/prune_dbg.txt
python - <<'PY'
import re
tot = 0; n = 0; comp = 0
for l in open("gpurun_out/prune_dbg_last.txt"):
    fr = [int(a) / int(b) for a, b in re.findall(r"(\d+)/(\d+):", l)]
    tot += sum(fr); n += 1; comp += sum(1 for f in fr if f == 1.0)
print("pairs", n, "candidate volumes per pair %.3f" % (tot / n), "complete candidates per pair %.3f" % (comp / n))
PY
timeout 300 python -m pytest tests/test_reg_gpu.py tests/test_register_fuse_gpu.py -x -q -m gpu -k "prun or north_star or batched" 2>&1 | tail -3
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); c=r['config']; print('ms_per_step %.2f pairwise %.2f volumes/pair %.3f' % (r['ms_per_step'], c['pairwise_ms_per_step'], c['candidate_volumes_walked_per_pair']))"
