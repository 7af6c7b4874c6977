#!/bin/bash
# SQ counters of one kernel (default: the fused SSIM walk) on ONE north-star sized pair: bash tools/ssim_pmc.sh [kernel regex]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ssim_pmc; rm -rf $O; mkdir -p $O
K=${1:-ssim_fused_batch}
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-include-regex "$K" --pmc $set --output-format csv -d $O/p$i -- python $R/tools/reg_probe.py 2 > $O/probe$i.log 2>&1
  tail -2 $O/probe$i.log | cut -c1-300
done
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'][:40], r.get('Grid_Size', r.get('Grid_Size_X')))
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        n[(k,r['Counter_Name'])]+=1
    for k,d in acc.items():
        print(k, {c: '%.4g'%(v/n[(k,c)]) for c,v in d.items()})
PY
