#!/bin/bash
# GPU box: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the persistent rollout kernel for one
# variant of the default bench workload.  usage: pmc_variant.sh <tag> [ENV=VAL ...] [-- bench args]
# The variant is selected through environment variables (VF_XCD_QUEUES=0, VF_LIBRARY=<other build>, ...), which
# are exported HERE, before rocprofv3 starts: the program after `--` is python3 itself.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
extra=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; extra=("$@"); break; fi
  export "$1"; shift
done
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmcv_${tag}_$c -o h --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt "${extra[@]}" > $R/gpurun_out/pmcv_${tag}_$c.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob
tot = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = []
    for f in glob.glob('$R/gpurun_out/pmcv_${tag}_%s/*counter_collection.csv' % c):
        for row in csv.DictReader(open(f)):
            if 'rollout_persistent_kernel' in row['Kernel_Name'] and row['Counter_Name'] == c:
                vals.append(float(row['Counter_Value']))
    tot[c] = sum(vals) / max(len(vals), 1) / 1048576.0
    print('$tag', c, 'launches', len(vals), 'GiB per launch %.2f' % tot[c])
print('$tag', 'HBM-side GiB per launch (2 x FETCH + WRITE) %.2f' % (2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']))
PY
