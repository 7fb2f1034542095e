#!/bin/bash
# GPU box, -DVF_DEBUG_KNOBS build (build/ab/knobs.so): conv-LSTM tile plans forced per layer (VF_LSTM_MREP: one
# character per conv-LSTM, 2 / 1 / h / q = 256 / 128 / 64 / 32 rows, anything else = automatic).
# usage: tools/sweep_plans.sh <samples> <workload> <plan> [<plan> ...]
n=$1; wl=$2; shift 2
export VF_LIBRARY=build/ab/knobs.so
for plan in "$@"; do
  VF_LSTM_MREP=$plan python bench.py --workload $wl --samples $n --no-alt --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > /tmp/sweep.json
  python tools/bench_line.py /tmp/sweep.json "$wl-$n-$plan"
done
