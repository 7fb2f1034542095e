cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_p; mkdir -p $O
python bench.py > $O/bench_default.json 2>$O/bench_default.err; python tools/bench_line.py $O/bench_default.json default
timeout 1500 python tools/stress_repeat.py 60 > $O/stress_repeat.txt 2>&1; tail -16 $O/stress_repeat.txt
timeout 1500 bash tools/soak.sh > $O/soak.txt 2>&1; cat $O/soak.txt
