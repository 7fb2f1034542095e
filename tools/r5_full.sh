cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_full; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; echo "tests rc $?"; tail -3 $O/gputests.log
timeout 300 python tools/fingerprint.py r5 > $O/fingerprint.txt 2>&1
VF_LIBRARY=build/ab/gatediv.so timeout 300 python tools/fingerprint.py gatediv > $O/fingerprint_gatediv.txt 2>&1
bash tools/run_profiles.sh r05
python bench.py --workload c5 --samples 125 --network savp2 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/prof_r05/bench_c5_shard125_savp2.json; python tools/bench_line.py gpurun_out/prof_r05/bench_c5_shard125_savp2.json c5-savp2
