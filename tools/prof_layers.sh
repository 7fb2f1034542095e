#!/bin/bash
# GPU box: per-layer kernel times (VF_PERSISTENT=0: one launch per layer per step) of a bench workload.
# usage: prof_layers.sh <tag> [bench args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
export VF_PERSISTENT=0
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/layers_$tag -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt "$@" > $R/gpurun_out/layers_$tag.log 2>&1
cd $R
db=$(ls gpurun_out/layers_$tag/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then python3 tools/rocprof_summary.py $db > gpurun_out/layers_$tag.txt; else ls -R gpurun_out/layers_$tag | head; fi
head -40 gpurun_out/layers_$tag.txt
