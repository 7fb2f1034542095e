#!/bin/bash
# Build experiment variants of the library next to this script: VF_EXP_* give WRONG results (timing only);
# VARIANTS=TILE_STATS builds the instrumented library tools/tile_stats.py reads (correct results, slower).
cd "$(dirname "$0")"
SRC=../../visual_foresight_amd/csrc/vf_engine.hip
for v in ${VARIANTS:-NO_A NO_B NO_STAGE}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -DVF_EXP_$v -DVF_$v -o libvf_exp_$v.so $SRC &
done
wait
ls -la libvf_exp_*.so
