#!/bin/bash
# A/B of the bench step under variants, alternating on ONE box (box-to-box spread is 3-5 %, so only same-box runs compare).
# usage (gpurun command):  bash tools/ab_bench.sh [-r REPS] [-a "bench args"] "label|ENV=V ENV2=V|extra bench args|lib.so" ...
#   label   printed in front of the figures
#   ENV=V   environment of that variant (empty: the defaults)
#   extra   bench.py arguments of that variant (e.g. --reg-threads 12)
#   lib.so  another build of the library swapped in for that variant (e.g. tools/variants/libmvs_hip_prof.so built with
#           make CXXFLAGS+=-DMVS_PROFILING_ABLATIONS; MVS_DUP_KERNELS=<tag> then measures a kernel's marginal cost in the pair loop)
# examples of the experiments recorded under profiles/ that this script reproduces:
#   runtime knobs   "default||" "irq0|HSA_ENABLE_INTERRUPT=0|" "kernarg|HIP_FORCE_DEV_KERNARG=1|" "queues6|GPU_MAX_HW_QUEUES=6|"
#   lanes           "8||" "12|| --reg-threads 12" "16|| --reg-threads 16"
#   pruned search   "on||" "off|MVS_SSIM_PRUNE=0|" "f64 walk|MVS_SSIM_F32=0|" "16 classes|MVS_SSIM_PRUNE_CLASSES=16|"
#   three-pass FFT  "x only||" "all axes|MVS_FFT_SLAB_AXES=7|" "off|MVS_FFT_NO_SLAB=1|"
#   fuse list       "classes||" "mixed|MVS_FUSE_MIXED=1|"
#   CPU pinning     "pinned||" "unpinned|MVS_PIN_PROCESS=0 MVS_PIN_CPUS=0|"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
REPS=3; ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-c3 --no-c5"
while getopts "r:a:" o; do case $o in r) REPS=$OPTARG;; a) ARGS=$OPTARG;; esac; done; shift $((OPTIND - 1))
O=gpurun_out/ab_bench.txt; : > $O
cp multiview-stitcher_amd/libmvs_hip.so /tmp/mvs_tree.so
for rep in $(seq $REPS); do
  for v in "$@"; do
    IFS='|' read -r label envs extra lib <<< "$v"
    cp "${lib:-/tmp/mvs_tree.so}" multiview-stitcher_amd/libmvs_hip.so
    env $envs timeout 600 python bench.py $ARGS $extra > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - "$label" <<'PY' | tee -a $O
import json, sys
try:
    r = json.load(open("gpurun_out/ab.json")); c = r["config"]
    f = lambda k: "%.2f" % c[k] if c.get(k) is not None else "-"
    print("%-28s ms_per_step %.2f register %s pairwise %s fuse %s kernel %s serial_host %s volumes/pair %s" % (
        sys.argv[1], r["ms_per_step"], f("register_ms_per_step"), f("pairwise_ms_per_step"), f("fuse_ms_per_step"), f("fuse_kernel_ms"),
        f("serial_host_ms"), f("candidate_volumes_walked_per_pair")))
except Exception as e:
    print(sys.argv[1], "failed:", e, open("gpurun_out/ab.err").read()[-300:])
PY
  done
done
cp /tmp/mvs_tree.so multiview-stitcher_amd/libmvs_hip.so
