#!/bin/bash
python tools/host_cpu_probe.py 16 10 2>/dev/null
python tools/host_cpu_probe.py 16 10 2>/dev/null
ROC_ACTIVE_WAIT_TIMEOUT=100 python tools/host_cpu_probe.py 16 10 2>/dev/null
ROC_ACTIVE_WAIT_TIMEOUT=1000 python tools/host_cpu_probe.py 16 10 2>/dev/null
HSA_ENABLE_INTERRUPT=0 python tools/host_cpu_probe.py 16 10 2>/dev/null
HSA_ENABLE_INTERRUPT=0 python tools/host_cpu_probe.py 8 10 2>/dev/null
python tools/host_cpu_probe.py 8 10 2>/dev/null
MVS_SSIM_PRUNE=0 python tools/host_cpu_probe.py 16 10 2>/dev/null
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
