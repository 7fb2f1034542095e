#!/bin/bash
# GPU box: HBM traffic of the persistent rollout kernel.  FETCH_SIZE and WRITE_SIZE need separate
# passes (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"); units are KiB.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/hbm_$c -o h --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/hbm_$c.log 2>&1
done
python3 - <<PY
import csv, glob
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = []
    for f in glob.glob('$R/gpurun_out/hbm_%s/*counter_collection.csv' % c):
        for row in csv.DictReader(open(f)):
            if 'rollout_persistent_kernel' in row['Kernel_Name'] and row['Counter_Name'] == c:
                vals.append(float(row['Counter_Value']))
    print(c, 'launches', len(vals), 'mean per launch [KiB]', sum(vals) / max(len(vals), 1))
PY
