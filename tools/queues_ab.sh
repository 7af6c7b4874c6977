#!/bin/bash
# A/B of the number of hardware queues (and lanes) for register() with the pruned search: alternating runs, medians of 12
for rep in 1 2 3; do
python tools/host_cpu_probe.py 16 12 2>/dev/null
GPU_MAX_HW_QUEUES=8 python tools/host_cpu_probe.py 16 12 2>/dev/null
GPU_MAX_HW_QUEUES=12 python tools/host_cpu_probe.py 16 12 2>/dev/null
done
python tools/host_cpu_probe.py 12 12 2>/dev/null
python tools/host_cpu_probe.py 10 12 2>/dev/null
python tools/host_cpu_probe.py 8 12 2>/dev/null
