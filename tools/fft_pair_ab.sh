#!/bin/bash
# partner-line pairs in the inverse x pass: HBM fetch of the FFT passes of a z-neighbour pair (51 x 256 x 256 crops) and the bench, with and without
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fftpair; rm -rf $O; mkdir -p $O
for flag in 0 1; do
MVS_FFT_NO_PAIR=$flag timeout 300 rocprofv3 --kernel-include-regex "fft_reg2|dft_line" --pmc FETCH_SIZE --output-format csv -d $O/f$flag -- python $R/tools/reg_probe.py 2 2,1,1 > $O/f$flag.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$O/f$flag/**/*counter_collection.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
print("fft_no_pair=$flag: FETCH_SIZE KB of the last 6 passes:", ["%.0f" % float(r["Counter_Value"]) for r in rows[-6:]])
PY
done
cd $R
for rep in 1 2 3; do for flag in 0 1; do
MVS_FFT_NO_PAIR=$flag python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('fft_no_pair=$flag   step %.1f register %.1f pairwise %.1f' % (d['ms_per_step'], c['register_ms_per_step'], c['pairwise_ms_per_step']))"
done; done
