cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_final2; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; echo "tests rc $?"; tail -2 $O/gputests.log
timeout 400 python tools/fingerprint.py r5-final 2>&1 | grep -v amdgpu > $O/fingerprint.txt
bash tools/pmc_hbm.sh > $O/pmc_hbm.log 2>&1; cp gpurun_out/hbm_traffic.json $O/hbm_traffic.json; cp $O/hbm_traffic.json profiles/r05_hbm_traffic.json
python bench.py > $O/bench_default.json 2>$O/bench_default.err; python tools/bench_line.py $O/bench_default.json default
python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_c2_shard25.json; python tools/bench_line.py $O/bench_c2_shard25.json c2-shard25
python bench.py --workload c1 --no-alt --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_c1.json; python tools/bench_line.py $O/bench_c1.json c1
