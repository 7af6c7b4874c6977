#!/bin/bash
# per-class alone-times (serial class kernels) and SQ instruction counts of the fuse launch; arg 1: fuse_probe geometry (0 exact, 2 jitter, 1 frac)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fcls; rm -rf $O; mkdir -p $O
G=${1:-2}
MVS_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --kernel-include-regex "fuse|copy_region" --output-format csv -d $O/t -- python $R/tools/fuse_probe.py 4 $G > $O/trace.log 2>&1
grep -h "kernel ms" $O/trace.log | tail -1
python $R/tools/kstats.py $(find $O/t -name "*kernel_stats.csv") 8
find $O/t -name "*kernel_trace.csv" -delete; find $O/t -name "*.db" -delete
MVS_SERIAL=1 timeout 300 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/p -- python $R/tools/fuse_probe.py 2 $G > $O/pmc.log 2>&1
python - <<PY
import csv,glob,collections,re
f=glob.glob("$O/p/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=re.sub(r'\(anonymous namespace\)::|^void |unsigned short, unsigned short','',r['Kernel_Name'])[:30]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k,d in acc.items():
    print(k, {c: '%.4g'%(v/n[(k,c)]) for c,v in d.items()})
PY
