#!/bin/bash
run() { env "$@" python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('   step %.1f register %.1f pairwise %.1f fuse %.1f | %s' % (d['ms_per_step'], c['register_ms_per_step'], c['pairwise_ms_per_step'], c['fuse_ms_per_step'], d['host'][67:]))"; }
for rep in 1 2 3 4; do
echo "default (process + workers)"; run A=1
echo "MVS_PIN_PROCESS=0 (workers only)"; run MVS_PIN_PROCESS=0
echo "MVS_PIN_CPUS=0 (nothing)"; run MVS_PIN_CPUS=0
done
