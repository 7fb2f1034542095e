#!/bin/bash
# GPU box: MFMA utilisation of the persistent rollout kernel (SQ counters, one pass); extra arguments go to bench.py
# (e.g. --precision bf16x6)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
   -d $R/gpurun_out/mfma_pmc -o m --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt "$@" > $R/gpurun_out/mfma_pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$R/gpurun_out/mfma_pmc/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        if 'rollout_persistent_kernel' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in agg.items()}
for k, v in sorted(m.items()):
    print('%-32s %18.0f per launch (%d launches)' % (k, v, len(agg[k])))
gui = m['GRBM_GUI_ACTIVE'] / 8.0          # summed over the 8 XCDs
print('kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs)      %.0f' % gui)
print('MFMA pipe utilisation = MFMA_BUSY / (1024 SIMDs x cycles) = %.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * gui)))
print('fp32 MFMA FLOPs (MOPS x 512)                  %.3e' % (m['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512))
print('bf16 MFMA FLOPs (MOPS x 512)                  %.3e' % (m.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) * 512))
PY
