cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_a; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu -x > $O/gputests.log 2>&1; echo "tests rc $?"; tail -4 $O/gputests.log
timeout 300 python tools/fingerprint.py r5base > $O/fingerprint.txt 2>&1; cat $O/fingerprint.txt | tail -11
bash tools/ab.sh $O visual_foresight_amd/libvf_hip.so base
timeout 300 python bench.py --ndesig 4 --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/bench_c2_nd4.json; python tools/bench_line.py $O/bench_c2_nd4.json c2-nd4
timeout 300 python bench.py --workload c1 --no-alt --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_c1.json; python tools/bench_line.py $O/bench_c1.json c1
