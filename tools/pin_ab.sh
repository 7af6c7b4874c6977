#!/bin/bash
# A/B of the CPU block bench.py pins itself to (alternating runs on one box)
run() { env "$@" python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('   step %.1f register %.1f pairwise %.1f fuse %.1f | %s' % (d['ms_per_step'], c['register_ms_per_step'], c['pairwise_ms_per_step'], c['fuse_ms_per_step'], d['host'][-70:]))"; }
for rep in 1 2 3; do
echo "default (self-pinned)"; run A=1
echo "MVS_PIN_CPUS=0"; run MVS_PIN_CPUS=0
echo "MVS_PIN_CPUS=0-7"; run MVS_PIN_CPUS=0-7
echo "MVS_PIN_CPUS=0-31"; run MVS_PIN_CPUS=0-31
echo "MVS_PIN_CPUS=64-79 (GPU's NUMA node)"; run MVS_PIN_CPUS=64-79
echo "MVS_PIN_CPUS=64-71,192-199 (8 cores + their SMT siblings)"; run MVS_PIN_CPUS=64-71,192-199
done
