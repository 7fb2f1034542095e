# GPU box: same-box A/B of two library builds (build/libvf_prev.so = the previous build, copied there before rebuilding)
# over the C2 bench in both precision modes, the 25-sample share and the C5 shard.  Boxes of the pool differ by 1-3 %,
# so only pairs measured by one call of this script are compared (profiles/r02_ab_experiments.log).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in prev new; do
 if [ $v == prev ]; then export VF_LIBRARY=$PWD/build/libvf_prev.so; else unset VF_LIBRARY; fi
 python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-$v
 python bench.py --precision bf16x6 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - bf16-$v
done; done
for v in prev new; do
 if [ $v == prev ]; then export VF_LIBRARY=$PWD/build/libvf_prev.so; else unset VF_LIBRARY; fi
 python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-25-$v
 python bench.py --workload c5 --samples 125 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - c5s-$v
done
