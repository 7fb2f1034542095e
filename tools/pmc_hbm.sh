#!/bin/bash
# GPU box: HBM traffic of the persistent rollout kernel.  FETCH_SIZE and WRITE_SIZE need separate
# passes (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"); units are KiB.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/hbm_$c -o h --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $R/gpurun_out/hbm_$c.log 2>&1
done
python3 - <<PY
import csv, glob, json
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = []
    for f in glob.glob('$R/gpurun_out/hbm_%s/*counter_collection.csv' % c):
        for row in csv.DictReader(open(f)):
            if 'rollout_persistent_kernel' in row['Kernel_Name'] and row['Counter_Name'] == c:
                vals.append(float(row['Counter_Value']))
    print(c, 'launches', len(vals), 'mean per launch [KiB]', sum(vals) / max(len(vals), 1))
    res[c] = (len(vals), sum(vals) / max(len(vals), 1))
json.dump({'launches_averaged': res['FETCH_SIZE'][0], 'FETCH_SIZE_KiB_per_launch': res['FETCH_SIZE'][1],
           'WRITE_SIZE_KiB_per_launch': res['WRITE_SIZE'][1],
           'hbm_bytes_per_launch': 1024.0 * (2.0 * res['FETCH_SIZE'][1] + res['WRITE_SIZE'][1])},
          open('$R/gpurun_out/hbm_traffic.json', 'w'), indent=1)
PY
