#!/bin/bash
# GPU box: persistent-schedule knobs on the headline workload: "groups offset wgs_per_cu"
for cfg in "1 9 2" "1 9 3" "2 9 3" "2 5 3" "3 6 3"; do
  set -- $cfg
  VF_PERSISTENT=1 VF_GROUPS=$1 VF_GROUP_OFFSET=$2 VF_PERSIST_WGS_PER_CU=$3 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py - "groups=$1 offset=$2 wgs/cu=$3"
done
