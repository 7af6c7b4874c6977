#!/bin/bash
# SQ instruction-mix counters of the fuse kernels (one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; rm -rf $O; mkdir -p $O
MVS_ABLATE=${ABL:-0} timeout 300 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc ${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES} --output-format csv -d $O/p -- python $R/tools/fuse_probe.py 2 ${1:-0} > $O/probe.log 2>&1
grep -h "kernel ms" $O/probe.log | tail -1
python - <<PY
import csv,glob,collections
f=glob.glob("$O/p/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=(r['Kernel_Name'][:34], r.get('Grid_Size', r.get('Grid_Size_X')))
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    n[(k,r['Counter_Name'])]+=1
for k,d in acc.items():
    print(k, {c: '%.3g'%(v/n[(k,c)]) for c,v in d.items()})
PY
