cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k config5 > $O/tests_c5.log 2>&1; tail -12 $O/tests_c5.log
python bench.py --workload c5 --samples 125 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>$O/c5.err | tail -1 > $O/bench_c5_savp.json; python tools/bench_line.py $O/bench_c5_savp.json c5-savp
python bench.py --workload c5 --samples 125 --network savp2 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>$O/c5b.err | tail -1 > $O/bench_c5_savp2.json; python tools/bench_line.py $O/bench_c5_savp2.json c5-savp2
tail -3 $O/c5b.err
