#!/bin/bash
lscpu | grep -E "NUMA|Socket|Thread|Core|Model name" 
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3
rocm-smi --showtoponuma 2>/dev/null | grep -i numa | head -4
run() { python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('   step %.1f register %.1f pairwise %.1f fuse %.1f' % (d['ms_per_step'], c['register_ms_per_step'], c['pairwise_ms_per_step'], c['fuse_ms_per_step']))"; }
for rep in 1 2; do
echo "no affinity"; run
for cpus in 0-15 16-31 32-47 64-79 96-111 128-143; do echo "taskset $cpus"; taskset -c $cpus bash -c "$(declare -f run); run"; done
done
