#!/bin/bash
# HBM fetch per FFT pass of one north-star pair (FETCH_SIZE, KB; x 2 on gfx950): does the inverse x pass read its array twice?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fftfetch; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-include-regex "fft_reg2|dft_line" --pmc $c --output-format csv -d $O/$c -- python $R/tools/reg_probe.py 2 > $O/$c.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$O/$c/**/*counter_collection.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
print("$c", len(rows), "dispatches; last 12 (one pair = 6 passes: fwd axis0, axis1, axis2, inv axis2, axis1, axis0):")
for r in rows[-12:]:
    print("   %-28s grid %8s  %10.0f KB" % (r["Kernel_Name"][:28], r.get("Grid_Size",""), float(r["Counter_Value"])))
PY
done
