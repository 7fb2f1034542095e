#!/bin/bash
# GPU box: conv-LSTM tile-plan sweep with the -DVF_DEBUG_KNOBS build (build/ab/knobs.so).  VF_LSTM_MREP picks the plan
# per layer lstm1..7: 2 = 256 rows, 1 = 128 (gate-split), h = 64, q = 32; "auto" = the cost model of vf_engine.hip.
L=build/ab/knobs.so
run() {  # run <samples> <extra bench args> -- plans...
  local M=$1; shift; local extra=$1; shift
  echo "# samples $M $extra"
  local v=("auto=$L")
  for pl in "$@"; do v+=("p$pl=$L:VF_LSTM_MREP=$pl"); done
  bash tools/ab_bench.sh "${v[@]}" -- --samples $M $extra
}
run 25 "" hhhhqhh 11hhhh1 1111h11 hhhhhhh 11hhqh1
run 50 "" 11hhhh1 1111h11 hhhhqhh 1111q11
run 100 "" 1111h11 1111111 11hhhh1 2211h11
run 125 "--workload c4" 1111h11 2211h12 2211h11 1111111
run 200 "" 1111h11 2211h11 1111111
run 1000 "--workload c4 --steps 3 --warmup 1" 2222222 2211112 1111111 2211h11 1111h11
