cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_j; mkdir -p $O
VF_LIBRARY=build/ab/wt3.so timeout 300 python tools/fingerprint.py wt3 > $O/fingerprint_wt3.txt 2>&1; tail -10 $O/fingerprint_wt3.txt
VF_LIBRARY=build/ab/wt3.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_savp.py -x -q -m gpu > $O/tests_wt3.log 2>&1; tail -3 $O/tests_wt3.log
for n in wt1 wtv1 wt3 wt1 wtv1 wt3; do
  export VF_LIBRARY=build/ab/$n.so
  python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_25.json; python tools/bench_line.py $O/bench_${n}_25.json $n-25
  python bench.py --workload c4 --samples 125 --no-alt --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_125.json; python tools/bench_line.py $O/bench_${n}_125.json $n-125
  python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_200.json; python tools/bench_line.py $O/bench_${n}_200.json $n-200
done
unset VF_LIBRARY
VF_LIBRARY=build/ab/wt3.so timeout 600 python tools/stress_repeat.py > $O/stress_wt3.txt 2>&1; tail -3 $O/stress_wt3.txt
