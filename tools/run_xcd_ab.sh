cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py -x -q -m gpu > gpurun_out/r2_xcd_tests.log 2>&1; echo "tests rc $?" 
tail -3 gpurun_out/r2_xcd_tests.log
for rep in 1 2; do
for q in 0 1; do
 out=$(VF_XCD_QUEUES=$q python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1)
 echo "xcd_queues=$q $(echo "$out" | python tools/bench_line.py -)"
done; done
tools/pmc_variant.sh q0 VF_XCD_QUEUES=0
tools/pmc_variant.sh q1 VF_XCD_QUEUES=1
