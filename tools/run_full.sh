# GPU box: the whole -m gpu suite, then bench lines for the default workload and for the shards of configs 4 / 5
cd $GRAFT_REPO_ROOT
tag=${1:-r2}
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/${tag}_gputests.log 2>&1; echo "tests rc $?"
tail -3 gpurun_out/${tag}_gputests.log
python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_c2_quick.json
python tools/bench_line.py gpurun_out/${tag}_bench_c2_quick.json c2
python bench.py --workload c5 --samples 125 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>gpurun_out/${tag}_c5.err | tail -1 > gpurun_out/${tag}_bench_c5_shard.json
python tools/bench_line.py gpurun_out/${tag}_bench_c5_shard.json c5-shard-125x5
python bench.py --workload c5 --samples 1000 --no-alt --no-cpu-baseline --steps 2 --warmup 1 2>>gpurun_out/${tag}_c5.err | tail -1 > gpurun_out/${tag}_bench_c5_1gpu.json
python tools/bench_line.py gpurun_out/${tag}_bench_c5_1gpu.json c5-1000x5-on-1gpu
