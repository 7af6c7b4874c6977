#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rows2; rm -rf $O; mkdir -p $O
MVS_ROWLDS=${ROWLDS:-1} MVS_ABLATE=${ABL:-0} MVS_PLAN_STATS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/fuse_probe.py 3 ${1:-0} > $O/probe.log 2>&1
grep -h "plan\|kernel ms" $O/probe.log | tail -2
python - <<PY
import csv,glob
f=glob.glob("$O/kt/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'fuse' in r['Kernel_Name'] or 'copy_region' in r['Kernel_Name']]
for r in rows[-8:]: print(r['Kernel_Name'][:40], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,'ms grid',r.get('Grid_Size_X', r.get('Grid_Size')), 'wg', r.get('Workgroup_Size_X', r.get('Workgroup_Size')), 'lds', r.get('LDS_Block_Size'), 'vgpr', r.get('VGPR_Count'), 'accum', r.get('Accum_VGPR_Count'), 'sgpr', r.get('SGPR_Count'))
PY
find $O/kt -name "*kernel_trace.csv" -delete; find $O/kt -name "*.db" -delete
