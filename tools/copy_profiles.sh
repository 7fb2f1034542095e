#!/bin/bash
# copy the summaries of a tools/run_profiles.sh pass (gpurun_out/prof_<tag>/) into profiles/${2:-r06}_*
O=gpurun_out/prof_$1
cp $O/bench_default.json profiles/${2:-r06}_bench_default.json
cp $O/kernel_stats_fp32.txt profiles/${2:-r06}_kernel_stats_fp32.txt
cp $O/mfma_pmc.txt profiles/${2:-r06}_mfma_pmc.txt
cp $O/kernel_stats_bf16x6.txt profiles/${2:-r06}_kernel_stats_bf16x6.txt
cp $O/mfma_pmc_bf16x6.txt profiles/${2:-r06}_mfma_pmc_bf16x6.txt
cp $O/hbm_traffic.json profiles/${2:-r06}_hbm_traffic.json
for n in 200 25 125 c5_shard c5_shard_savp2 c5_shard_savp3 c5_shard_savp3_spec64 200_savp3; do
  [ -f $O/phase_stats_$n.txt ] && grep -v amdgpu.ids $O/phase_stats_$n.txt > profiles/${2:-r06}_phase_stats_$n.txt
done
for f in c1 c3 c4_shard125 c4_1gpu c2_shard25 c2_shard50 c2_nd4 c5_shard125 2ranks_gloo_dryrun c5_shard125_savp2 c5_shard125_savp3 \
         c5_shard125_savp3_spec64 c2_savp3; do
  [ -f $O/bench_$f.json ] && cp $O/bench_$f.json profiles/${2:-r06}_bench_$f.json
done
[ -f $O/kernel_stats_c5_savp3.txt ] && cp $O/kernel_stats_c5_savp3.txt profiles/${2:-r06}_kernel_stats_c5_savp3.txt
[ -f $O/gputests.log ] && cp $O/gputests.log profiles/${2:-r06}_gpu_tests.log
[ -f $O/fingerprint.txt ] && grep -v amdgpu.ids $O/fingerprint.txt > profiles/${2:-r06}_fingerprints.txt
for f in cu_trace_200 cu_trace_200_timeline cu_trace_25 chain_25 chain_200 mfma_shadow_ubench hbm_calib sq_mix precision_check stress_repeat soak; do
  [ -f $O/$f.txt ] && cp $O/$f.txt profiles/${2:-r06}_$f.txt
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']; s=r.get('survey_8d') or {}
    print('%-34s ms/step %.1f frames/s %.0f it/s %.2f launch %.2f ms frac %.3f s8d %.3f traffic %s' % (f.split('/')[-1], d['ms_per_step'], d['value'], d['cem_iters_per_sec'], r['avg_launch_us']/1e3, r['frac'], s.get('frac_of_fp32_mfma_peak',0), r.get('traffic')))
    if 'alt_precision' in d:
        a=d['alt_precision']; print('    alt %.0f frames/s %.1f ms/step launch %.2f ms bf16 frac %s' % (a['value'], a['ms_per_step'], a['roofline']['avg_launch_us']/1e3, a['roofline'].get('bf16_mfma_frac_of_peak')))
    if d.get('cpu_baseline'): print('    cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('at_all_physical_cores',{}).get('value'))
PY
