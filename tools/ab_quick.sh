# GPU box: short same-box A/B (previous build in build/libvf_prev.so): C2 twice, C5 shard, composite item times
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in prev new; do
 if [ $v == prev ]; then export VF_LIBRARY=$PWD/build/libvf_prev.so; else unset VF_LIBRARY; fi
 python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-$v
done; done
for v in prev new; do
 if [ $v == prev ]; then export VF_LIBRARY=$PWD/build/libvf_prev.so; else unset VF_LIBRARY; fi
 python bench.py --workload c5 --samples 125 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - c5s-$v
 python tools/persist_stats.py 200 2>&1 | grep -E "COMPOSITE" | tail -2
done
unset VF_LIBRARY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_savp.py -x -q -m gpu 2>&1 | tail -2
