# GPU box, round 6: the whole GPU suite, the round's profile pass (tools/run_profiles.sh) and the arch-3 (published SAVP
# generator) measurements.  usage: bash tools/r6_full.sh [tag]
cd $GRAFT_REPO_ROOT
tag=${1:-r06}
O=gpurun_out/prof_$tag; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; echo "tests rc $?"; tail -3 $O/gputests.log
timeout 300 python tools/fingerprint.py $tag > $O/fingerprint.txt 2>&1
bash tools/run_profiles.sh $tag
# arch 3: one rank's share of BASELINE configs[4] on the table the public code selects at 128 pixels, on the paper's
# five-cell table, and the paper's network at its own size (64 x 64, the C2 shape)
python bench.py --workload c5 --samples 125 --network savp2 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c5_shard125_savp2.json; python tools/bench_line.py $O/bench_c5_shard125_savp2.json c5-savp2
python bench.py --workload c5 --samples 125 --network savp3 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c5_shard125_savp3.json; python tools/bench_line.py $O/bench_c5_shard125_savp3.json c5-savp3
python bench.py --workload c5 --samples 125 --network savp3 --layer-spec 64 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c5_shard125_savp3_spec64.json; python tools/bench_line.py $O/bench_c5_shard125_savp3_spec64.json c5-savp3-spec64
python bench.py --workload c2 --network savp3 --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_c2_savp3.json; python tools/bench_line.py $O/bench_c2_savp3.json c2-savp3
python tools/persist_stats.py 625 128 savp2 15 > $O/phase_stats_c5_shard_savp2.txt 2>&1
python tools/persist_stats.py 625 128 savp3 15 > $O/phase_stats_c5_shard_savp3.txt 2>&1
python tools/persist_stats.py 625 128 savp3 15 64 > $O/phase_stats_c5_shard_savp3_spec64.txt 2>&1
python tools/persist_stats.py 200 64 savp3 13 > $O/phase_stats_200_savp3.txt 2>&1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/ktrace_s3 -o k -- python3 $R/bench.py --workload c5 --samples 125 --network savp3 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/ktrace_bench_c5_savp3.json 2> $R/$O/ktrace_s3.err
cd $R
db=$(ls $O/ktrace_s3/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/rocprof_summary.py $db > $O/kernel_stats_c5_savp3.txt && head -6 $O/kernel_stats_c5_savp3.txt
rm -rf $O/ktrace_s3
# the same planning calls many times (must reproduce themselves bit for bit) and long bench loops, arch 3 included
timeout 1500 python tools/stress_repeat.py 60 2>&1 | grep -v amdgpu.ids > $O/stress_repeat.txt; tail -1 $O/stress_repeat.txt
timeout 1400 bash tools/soak.sh > $O/soak.txt 2>&1
