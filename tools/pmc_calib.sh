#!/bin/bash
# GPU box: calibrate FETCH_SIZE / WRITE_SIZE (KiB) against kernels that move exactly 1 GiB each
# (tools/ubench/hbm_calib.hip).  Writes gpurun_out/hbm_calib.txt.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/calib_$c -o c --output-format csv -- $R/tools/ubench/hbm_calib > $R/gpurun_out/calib_$c.log 2>&1
done
cd $R
python3 - <<PY > gpurun_out/hbm_calib.txt
import csv, glob
print('# tools/pmc_calib.sh: rocprofv3 counter (KiB) / bytes actually moved (1 GiB per launch), mean of 3 launches')
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc = {}
    for f in glob.glob('gpurun_out/calib_%s/*counter_collection.csv' % c):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == c:
                acc.setdefault(row['Kernel_Name'].split('(')[0], []).append(float(row['Counter_Value']))
    for k in sorted(acc):
        v = sum(acc[k]) / len(acc[k])
        print('%-10s %-12s %14.0f KiB per launch = %.3f x 1 GiB' % (c, k, v, v * 1024 / 2 ** 30))
PY
cat gpurun_out/hbm_calib.txt
