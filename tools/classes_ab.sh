#!/bin/bash
# A/B of the residue classes of the pruned search (16 / 32), alternating runs on one box
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('   step %.1f register %.1f pairwise %.1f fuse %.1f volumes %.2f' % (d['ms_per_step'], c['register_ms_per_step'], c['pairwise_ms_per_step'], c['fuse_ms_per_step'], c['candidate_volumes_walked_per_pair']))"; }
for rep in 1 2 3 4; do echo "32 classes"; run A=1; echo "16 classes"; run MVS_SSIM_PRUNE_CLASSES=16; done
