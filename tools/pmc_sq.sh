#!/bin/bash
# GPU box: instruction mix and per-pipe activity of the persistent rollout kernel (SQ counters, three passes of the
# default bench workload); extra arguments go to bench.py.  Writes gpurun_out/sq_mix.txt.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/sq_mix$i -o m --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt "$@" > $R/gpurun_out/sq_mix$i.log 2>&1
done
python3 - <<PY > $R/gpurun_out/sq_mix.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$R/gpurun_out/sq_mix*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        if 'rollout_persistent_kernel' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print('# tools/pmc_sq.sh: SQ counters of rollout_persistent_kernel, mean per launch (default bench workload)')
for k, v in sorted(m.items()):
    print('%-28s %18.0f  (%d launches)' % (k, v, len(agg[k])))
if 'GRBM_GUI_ACTIVE' in m:
    cyc = m['GRBM_GUI_ACTIVE'] / 8.0
    print('kernel cycles %.0f; per SIMD (1024): non-MFMA VALU instructions %.0f, MFMA %.0f, LDS %.0f, VMEM %.0f, SALU %.0f' % (
        cyc, (m.get('SQ_INSTS_VALU', 0) - m.get('SQ_INSTS_MFMA', 0)) / 1024, m.get('SQ_INSTS_MFMA', 0) / 1024,
        m.get('SQ_INSTS_LDS', 0) / 1024, (m.get('SQ_INSTS_VMEM_RD', 0) + m.get('SQ_INSTS_VMEM_WR', 0)) / 1024,
        m.get('SQ_INSTS_SALU', 0) / 1024))
PY
cat $R/gpurun_out/sq_mix.txt
