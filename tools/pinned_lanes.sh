#!/bin/bash
for rep in 1 2; do for L in 16 14 12 10 8; do python tools/host_cpu_probe.py $L 12 2>/dev/null | sed 's/process CPU.*//'; done; done
