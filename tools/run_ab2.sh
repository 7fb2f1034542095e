cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2_convt_tests.log 2>&1; echo "tests rc $?"
tail -2 gpurun_out/r2_convt_tests.log
for rep in 1 2; do
 out=$(python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1)
 echo "convt-skip $(echo "$out" | python tools/bench_line.py -)"
done
python tools/persist_stats.py 200 > gpurun_out/r2_stats200d.log 2>&1
tail -14 gpurun_out/r2_stats200d.log
