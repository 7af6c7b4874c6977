#!/bin/bash
# A/B of the pruned arg-max search on ONE box: alternating bench runs with MVS_SSIM_PRUNE=1 / 0, then the search's own report for a few pairs
mkdir -p gpurun_out/prune_ab
for rep in 1 2 3; do for p in 1 0; do
  MVS_SSIM_PRUNE=$p python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie > gpurun_out/prune_ab/p${p}_$rep.json 2> gpurun_out/prune_ab/p${p}_$rep.err
  python - <<PY
import json
d=json.load(open("gpurun_out/prune_ab/p${p}_$rep.json")); c=d["config"]
print("MVS_SSIM_PRUNE=$p run $rep: %.0f Mvoxels/s, step %.1f ms, register %.1f, pairwise %.1f, fuse %.1f; candidates scored / left unfinished / volumes walked per pair: %.2f / %.2f / %.2f; roofline_register.frac %.3f; max |error| %.1e px" % (d["value"], d["ms_per_step"], c["register_ms_per_step"], c["pairwise_ms_per_step"], c["fuse_ms_per_step"], c["scored_candidates_per_pair"], c["candidates_left_unfinished_per_pair"], c["candidate_volumes_walked_per_pair"], d["roofline_register"]["frac"], c["registration_max_abs_error_px"]))
PY
done; done
echo "== MVS_PRUNE_DEBUG=1: per pair the best mean SSIM | per candidate: sixteenths of the volume walked : mean SSIM over them (x = dropped)"
MVS_PRUNE_DEBUG=1 python tools/host_cpu_probe.py 16 1 2>&1 | grep "prune:" | tail -10
