#!/bin/bash
# GPU box: LDS / wait counters per kernel with per-layer launches (VF_PERSISTENT=0), two passes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
export VF_PERSISTENT=${VF_PERSISTENT:-0}
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/lds_pmc$i -o m --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $R/gpurun_out/lds_pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob('$R/gpurun_out/lds_pmc*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][-60:]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, m in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CU_CYCLES', 0))[:14]:
    print(k)
    for c, v in sorted(m.items()):
        print('   %-28s %16.0f' % (c, v))
    if m.get('SQ_LDS_IDX_ACTIVE'):
        print('   bank conflict cycles / LDS active cycles = %.3f' % (m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']))
    if m.get('SQ_BUSY_CU_CYCLES'):
        print('   LDS active / CU busy = %.3f   MFMA busy / (4 x CU busy) = %.3f' % (
            m['SQ_LDS_IDX_ACTIVE'] / m['SQ_BUSY_CU_CYCLES'], m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES'])))
PY
