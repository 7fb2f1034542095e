cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py::test_role_mode_is_invisible_in_the_results tests/test_gpu_savp.py::test_savp_launch_strategies_and_chunking_are_bit_identical -x -q -m gpu > gpurun_out/r2_ab4_tests.log 2>&1; echo "tests rc $?"
tail -15 gpurun_out/r2_ab4_tests.log
for rep in 1 2; do
for r in 0 1; do
 VF_ROLE_MODE=$r timeout 300 python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - role$r
done; done
VF_ROLE_MODE=1 timeout 300 python bench.py --workload c4 --samples 125 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - c4s-role1
VF_ROLE_MODE=1 timeout 300 python bench.py --workload c3 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - c3-role1
VF_ROLE_MODE=1 timeout 300 python bench.py --workload c5 --samples 125 --no-alt --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python tools/bench_line.py - c5s-role1
