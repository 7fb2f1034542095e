# same-box A/B of the cooperative-yield budgets (production builds with -DVF_YIELD_DEFAULT=n)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_b; mkdir -p $O
VF_LIBRARY=build/ab/y120.so timeout 300 python tools/fingerprint.py y120 > $O/fingerprint_y120.txt 2>&1; tail -10 $O/fingerprint_y120.txt
for n in 0 40 120 400 0; do
  L=build/ab/y$n.so
  export VF_LIBRARY=$L
  python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_y${n}_25.json; python tools/bench_line.py $O/bench_y${n}_25.json y$n-25
  python bench.py --workload c1 --no-alt --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_y${n}_c1.json; python tools/bench_line.py $O/bench_y${n}_c1.json y$n-c1
  python bench.py --samples 50 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_y${n}_50.json; python tools/bench_line.py $O/bench_y${n}_50.json y$n-50
  python bench.py --workload c4 --samples 125 --no-alt --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_y${n}_125.json; python tools/bench_line.py $O/bench_y${n}_125.json y$n-125
  python bench.py --no-alt --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_y${n}_200.json; python tools/bench_line.py $O/bench_y${n}_200.json y$n-200
done
unset VF_LIBRARY
VF_YIELD=120 VF_LIBRARY=build/ab/trace_yield.so timeout 300 python tools/trace_cu.py 25 > $O/cu_trace_25_y120.txt 2>&1
VF_YIELD=0 VF_LIBRARY=build/ab/trace_yield.so timeout 300 python tools/trace_cu.py 25 > $O/cu_trace_25_y0.txt 2>&1
head -40 $O/cu_trace_25_y120.txt
