cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_savp.py -x -q -m gpu > gpurun_out/r2_ab3_tests.log 2>&1; echo "tests rc $?"
tail -2 gpurun_out/r2_ab3_tests.log
for rep in 1 2; do
 python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2
done
python bench.py --precision bf16x6 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-bf16x6
python bench.py --workload c5 --samples 125 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - c5s
python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c2-25
tools/prof_layers.sh c5s_b --workload c5 --samples 125 > /dev/null 2>&1; grep -E "composite|conv_mfma_kernel<(1, 2|4, 4|4, 3|1, 1)" gpurun_out/layers_c5s_b.txt | tail -16
