#!/bin/bash
# GPU box: persistent-schedule knobs on the headline workload
for cfg in "1 9" "2 9" "2 5" "2 13" "3 6" "4 4"; do
  set -- $cfg
  VF_PERSISTENT=1 VF_GROUPS=$1 VF_GROUP_OFFSET=$2 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "groups=$1 offset=$2"
done
