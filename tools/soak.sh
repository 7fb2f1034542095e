cd $GRAFT_REPO_ROOT
python bench.py --no-alt --no-cpu-baseline --steps 150 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - soak-c2
python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 300 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - soak-25
python bench.py --workload c5 --samples 125 --no-alt --no-cpu-baseline --steps 15 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - soak-c5
python bench.py --workload c5 --samples 125 --network savp2 --no-alt --no-cpu-baseline --steps 10 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - soak-c5-savp2
python bench.py --samples 50 --no-alt --no-cpu-baseline --steps 150 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - soak-50
python bench.py --workload c1 --no-alt --no-cpu-baseline --steps 600 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - soak-c1
python bench.py --workload c3 --no-alt --no-cpu-baseline --steps 12 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - soak-c3
python bench.py --precision bf16x6 --no-alt --no-cpu-baseline --steps 100 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - soak-bf16
python bench.py --gpus 2 --no-alt --no-cpu-baseline --steps 30 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - soak-2ranks
python bench.py --workload c5 --samples 125 --network savp3 --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - soak-c5-savp3
python bench.py --workload c2 --network savp3 --no-cpu-baseline --steps 60 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - soak-c2-savp3
