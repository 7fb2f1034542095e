cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_c; mkdir -p $O
for n in y0 yrow ytap y0 yrow ytap; do
  export VF_LIBRARY=build/ab/$n.so
  python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_25.json; python tools/bench_line.py $O/bench_${n}_25.json $n-25
  python bench.py --samples 50 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_50.json; python tools/bench_line.py $O/bench_${n}_50.json $n-50
  python bench.py --workload c4 --samples 125 --no-alt --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_${n}_125.json; python tools/bench_line.py $O/bench_${n}_125.json $n-125
done
unset VF_LIBRARY
VF_LIBRARY=build/ab/ytap.so timeout 300 python tools/fingerprint.py ytap > $O/fingerprint_ytap.txt 2>&1; tail -10 $O/fingerprint_ytap.txt
VF_LIBRARY=build/ab/trace_ytap.so timeout 300 python tools/trace_chain.py 25 > $O/chain_25_ytap.txt 2>&1; tail -24 $O/chain_25_ytap.txt
VF_LIBRARY=build/ab/trace_ytap.so timeout 300 python tools/trace_cu.py 25 > $O/cu_trace_25_ytap.txt 2>&1; head -12 $O/cu_trace_25_ytap.txt
