#!/bin/bash
# GPU box: HBM-side traffic of the persistent rollout kernel for the default bench workload.  FETCH_SIZE and
# WRITE_SIZE need separate passes (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"); units are KiB.
# Writes gpurun_out/hbm_traffic.json, stamped with the hash of the kernel sources so bench.py only
# quotes it for the library it was measured on (copy it to profiles/ to have it reported).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/hbm_$c -o h --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $R/gpurun_out/hbm_$c.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, json, sys
sys.path.insert(0, '$R')
import bench
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = []
    for f in glob.glob('$R/gpurun_out/hbm_%s/*counter_collection.csv' % c):
        for row in csv.DictReader(open(f)):
            if 'rollout_persistent_kernel' in row['Kernel_Name'] and row['Counter_Name'] == c:
                vals.append(float(row['Counter_Value']))
    print(c, 'launches', len(vals), 'mean per launch [KiB]', sum(vals) / max(len(vals), 1))
    res[c] = (len(vals), sum(vals) / max(len(vals), 1))
total = 1024.0 * (2.0 * res['FETCH_SIZE'][1] + res['WRITE_SIZE'][1])
# SURVEY 8(d) algorithmic bytes of one C2 rollout: 200 x 14 sample-steps x 2.49 MB of LSTM state round trip
# + 200 x 13 predicted frames and distributions written + the 33 MB weight set once per step
algorithmic = 200 * 14 * 2 * 311296 * 4 + 200 * 13 * 64 * 64 * 4 * 4 + 14 * 33.0e6
json.dump({'kernel': 'rollout_persistent_kernel<1>', 'workload': 'c2', 'precision': 'fp32',
           'lib_sources_sha16': bench.library_hash(),
           'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py '
                      '--steps 2 --warmup 1 --no-cpu-baseline --no-alt   (tools/pmc_hbm.sh)',
           'launches_averaged': res['FETCH_SIZE'][0], 'FETCH_SIZE_KiB_per_launch': res['FETCH_SIZE'][1],
           'WRITE_SIZE_KiB_per_launch': res['WRITE_SIZE'][1], 'hbm_bytes_per_launch': total,
           'algorithmic_bytes_per_launch': algorithmic, 'ratio_to_algorithmic': total / algorithmic,
           'correction': 'MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports half the bytes of a wide '
                         'coalesced read, so the read side is doubled; both counters calibrated with '
                         'tools/ubench/hbm_calib.hip on the rollout\'s access patterns (FETCH 0.500x, WRITE 1.000x, '
                         'profiles/r03_hbm_calib.txt); Infinity-Cache hits are included in both'},
          open('$R/gpurun_out/hbm_traffic.json', 'w'), indent=1)
PY
