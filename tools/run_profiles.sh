#!/bin/bash
# GPU box: every measurement profiles/ quotes for the current tree, written to gpurun_out/prof_<tag>/ (copy the
# summaries into profiles/ afterwards).  usage: tools/run_profiles.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r05}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd $R
# 1. the default bench line (fp32 primary, split-bf16 alt, CPU legs)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/bench_line.py $O/bench_default.json default
# 2. kernel trace of the same workload
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ktrace -o k -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/ktrace_bench.json 2> $O/ktrace.err
cd $R
db=$(ls $O/ktrace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/rocprof_summary.py $db > $O/kernel_stats_fp32.txt && head -6 $O/kernel_stats_fp32.txt
# 3. counters (separate runs, --kernel-trace only)
bash tools/pmc_mfma.sh > $O/mfma_pmc.txt 2>&1; tail -4 $O/mfma_pmc.txt
bash tools/pmc_hbm.sh > $O/pmc_hbm.log 2>&1; cp gpurun_out/hbm_traffic.json $O/hbm_traffic.json; tail -3 $O/pmc_hbm.log
# 3b. the split-bf16 mode: its own kernel trace and MFMA counters
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ktrace16 -o k -- python3 $R/bench.py --precision bf16x6 --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/ktrace_bench_bf16x6.json 2> $O/ktrace16.err
cd $R
db=$(ls $O/ktrace16/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/rocprof_summary.py $db > $O/kernel_stats_bf16x6.txt && head -4 $O/kernel_stats_bf16x6.txt
bash tools/pmc_mfma.sh --precision bf16x6 > $O/mfma_pmc_bf16x6.txt 2>&1; tail -5 $O/mfma_pmc_bf16x6.txt
# 4. the other BASELINE shapes
python bench.py --workload c1 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c1.json; python tools/bench_line.py $O/bench_c1.json c1
python bench.py --workload c3 --no-alt --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3.json; python tools/bench_line.py $O/bench_c3.json c3
python bench.py --workload c4 --samples 125 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_c4_shard125.json; python tools/bench_line.py $O/bench_c4_shard125.json c4-shard
python bench.py --workload c4 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c4_1gpu.json; python tools/bench_line.py $O/bench_c4_1gpu.json c4-1gpu
python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_c2_shard25.json; python tools/bench_line.py $O/bench_c2_shard25.json c2-shard25
python bench.py --workload c5 --samples 125 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c5_shard125.json; python tools/bench_line.py $O/bench_c5_shard125.json c5-shard
python bench.py --ndesig 4 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_c2_nd4.json; python tools/bench_line.py $O/bench_c2_nd4.json c2-nd4
python bench.py --samples 50 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_c2_shard50.json; python tools/bench_line.py $O/bench_c2_shard50.json c2-shard50
# 5. phase statistics, two-rank dry run
python tools/persist_stats.py 200 > $O/phase_stats_200.txt 2>&1
python tools/persist_stats.py 25 > $O/phase_stats_25.txt 2>&1
python tools/persist_stats.py 125 64 cdna 15 > $O/phase_stats_125.txt 2>&1
python tools/persist_stats.py 625 128 savp 15 > $O/phase_stats_c5_shard.txt 2>&1
python bench.py --gpus 2 --no-alt --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 > $O/bench_2ranks_gloo_dryrun.json; python tools/bench_line.py $O/bench_2ranks_gloo_dryrun.json 2ranks-gloo
# 6. what the two workgroups of a CU are doing (event-log build, tools/build_variants.sh), micro-benchmarks, counter
#    calibration, instruction mix
if [ -f build/ab/trace.so ]; then
  VF_LIBRARY=build/ab/trace.so timeout 300 python tools/trace_cu.py 200 > $O/cu_trace_200.txt 2>&1
  VF_TRACE_TIMELINE=1 VF_LIBRARY=build/ab/trace.so timeout 300 python tools/trace_cu.py 200 2>&1 | awk '/^timeline/{f=1} f' > $O/cu_trace_200_timeline.txt
  VF_LIBRARY=build/ab/trace.so timeout 300 python tools/trace_cu.py 25 > $O/cu_trace_25.txt 2>&1
  VF_LIBRARY=build/ab/trace.so timeout 300 python tools/trace_chain.py 25 6 1 > $O/chain_25.txt 2>&1
  VF_LIBRARY=build/ab/trace.so timeout 300 python tools/trace_chain.py 200 6 1 > $O/chain_200.txt 2>&1
fi
[ -x tools/ubench/mfma_shadow ] && timeout 120 tools/ubench/mfma_shadow > $O/mfma_shadow_ubench.txt 2>&1
[ -x tools/ubench/hbm_calib ] && timeout 300 bash tools/pmc_calib.sh > /dev/null 2>&1 && cp gpurun_out/hbm_calib.txt $O/hbm_calib.txt
timeout 600 bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/sq_mix.txt $O/sq_mix.txt
python tools/precision_check.py > $O/precision_check.txt 2>&1
rm -rf $O/ktrace $O/ktrace16 gpurun_out/hbm_FETCH_SIZE gpurun_out/hbm_WRITE_SIZE gpurun_out/mfma_pmc gpurun_out/sq_mix1 gpurun_out/sq_mix2 gpurun_out/sq_mix3 gpurun_out/calib_FETCH_SIZE gpurun_out/calib_WRITE_SIZE
