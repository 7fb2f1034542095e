# GPU box: short same-box A/B (previous build in build/libvf_prev.so); extra environment for both sides in $AB_ENV
cd $GRAFT_REPO_ROOT
export $AB_ENV
for rep in 1 2; do
for v in prev new; do
 if [ $v == prev ]; then export VF_LIBRARY=$PWD/build/libvf_prev.so; else unset VF_LIBRARY; fi
 python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-$v
done; done
for v in prev new; do
 if [ $v == prev ]; then export VF_LIBRARY=$PWD/build/libvf_prev.so; else unset VF_LIBRARY; fi
 python bench.py --precision bf16x6 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - bf16-$v
 python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-25-$v
 python bench.py --workload c5 --samples 125 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - c5s-$v
done
