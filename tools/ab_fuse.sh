cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in 0 1; do
 VF_FUSE_TOP=$f python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-fuse$f
done; done
for f in 0 1; do
 VF_FUSE_TOP=$f python bench.py --precision bf16x6 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - bf16-fuse$f
 VF_FUSE_TOP=$f python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-25-fuse$f
 VF_FUSE_TOP=$f python bench.py --workload c5 --samples 125 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - c5s-fuse$f
 VF_FUSE_TOP=$f python bench.py --workload c3 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - c3-fuse$f
done
VF_FUSE_TOP=1 python tools/persist_stats.py 200 2>&1 | tail -12
