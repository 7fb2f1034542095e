cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_k; mkdir -p $O
VF_LIBRARY=build/ab/wt7.so timeout 300 python tools/fingerprint.py wt7 > $O/fingerprint_wt7.txt 2>&1; tail -10 $O/fingerprint_wt7.txt
VF_LIBRARY=build/ab/wt7.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_savp.py tests/test_gpu_repeat.py -x -q -m gpu > $O/tests_wt7.log 2>&1; tail -3 $O/tests_wt7.log
for n in wt3 wt7 wt3 wt7; do
  export VF_LIBRARY=build/ab/$n.so
  python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_25.json; python tools/bench_line.py $O/bench_${n}_25.json $n-25
  python bench.py --samples 50 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_50.json; python tools/bench_line.py $O/bench_${n}_50.json $n-50
  python bench.py --workload c4 --samples 125 --no-alt --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_125.json; python tools/bench_line.py $O/bench_${n}_125.json $n-125
  python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_200.json; python tools/bench_line.py $O/bench_${n}_200.json $n-200
done
unset VF_LIBRARY
VF_LIBRARY=build/ab/wt7.so timeout 900 python tools/stress_repeat.py > $O/stress_wt7.txt 2>&1; tail -3 $O/stress_wt7.txt
VF_LIBRARY=build/ab/trace.so timeout 300 python tools/trace_chain.py 25 > $O/chain_25.txt 2>&1; tail -20 $O/chain_25.txt
