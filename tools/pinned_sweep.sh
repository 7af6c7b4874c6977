#!/bin/bash
# with the process pinned (run-to-run spread ~1 ms) the small knobs can be read: lanes, work items per candidate, hardware queues
p() { env "$@" python tools/host_cpu_probe.py ${LANES:-16} 10 2>/dev/null | sed 's/process CPU.*busy;//'; }

p A=1; p A=1
LANES=8 p A=1; LANES=12 p A=1
p MVS_SSIM_PRUNE_ITEMS=160; p MVS_SSIM_PRUNE_ITEMS=640
p GPU_MAX_HW_QUEUES=8; p GPU_MAX_HW_QUEUES=2
p MVS_SSIM_PRUNE=0
