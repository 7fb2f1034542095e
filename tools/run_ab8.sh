cd $GRAFT_REPO_ROOT
for n in 100 50 25; do
for v in prev new; do
 if [ $v == prev ]; then export VF_LIBRARY=$PWD/build/libvf_prev.so; else unset VF_LIBRARY; fi
 python bench.py --samples $n --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-$n-$v
done; done
for v in prev new; do
 if [ $v == prev ]; then export VF_LIBRARY=$PWD/build/libvf_prev.so; else unset VF_LIBRARY; fi
 python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-$v
done
