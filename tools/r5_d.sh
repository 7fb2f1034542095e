cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_d; mkdir -p $O
VF_LIBRARY=build/ab/fcw.so timeout 300 python tools/fingerprint.py fcw > $O/fingerprint_fcw.txt 2>&1; tail -10 $O/fingerprint_fcw.txt
for n in yrow fcw yrow fcw; do
  export VF_LIBRARY=build/ab/$n.so
  python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_25.json; python tools/bench_line.py $O/bench_${n}_25.json $n-25
  python bench.py --samples 50 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_50.json; python tools/bench_line.py $O/bench_${n}_50.json $n-50
  python bench.py --workload c4 --samples 125 --no-alt --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_125.json; python tools/bench_line.py $O/bench_${n}_125.json $n-125
  python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_200.json; python tools/bench_line.py $O/bench_${n}_200.json $n-200
  python bench.py --workload c1 --no-alt --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_${n}_c1.json; python tools/bench_line.py $O/bench_${n}_c1.json $n-c1
done
unset VF_LIBRARY
VF_LIBRARY=build/ab/trace_fcw.so timeout 300 python tools/trace_chain.py 25 > $O/chain_25_fcw.txt 2>&1; tail -20 $O/chain_25_fcw.txt
VF_LIBRARY=build/ab/trace_fcw.so timeout 300 python tools/trace_cu.py 25 > $O/cu_trace_25_fcw.txt 2>&1; head -14 $O/cu_trace_25_fcw.txt
VF_LIBRARY=build/ab/trace_fcw.so timeout 300 python tools/trace_cu.py 200 > $O/cu_trace_200_fcw.txt 2>&1; head -14 $O/cu_trace_200_fcw.txt
