cd $GRAFT_REPO_ROOT
L=$PWD/build/libvf_knobs.so
for plan in auto 2211h11 2211h12 2211h21 2211hh1 221hh11 22h1h11 2212h11 2211211 2211h1h 22hhh11 2211hhh; do
  if [ $plan == auto ]; then unset VF_LSTM_MREP; else export VF_LSTM_MREP=$plan; fi
  out=$(VF_LIBRARY=$L python bench.py --no-alt --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1)
  echo "$plan $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f us' % d['roofline']['avg_launch_us'])")"
done
