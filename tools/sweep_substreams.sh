#!/bin/bash
# GPU box: bench the headline workload for several sub-batch stream counts
for n in 1 2 3 4 6 8; do
  VF_SUBSTREAMS=$n python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
rf = r['roofline']
print('substreams %d: %.0f frames/s  %.2f iters/s  %.1f ms/step  sum-dur TF %.1f  busy TF %.1f  busy share %.3f' % (
    rf['substreams'], r['value'], r['cem_iters_per_sec'], r['ms_per_step'], rf['achieved'], rf['achieved_while_busy'], rf['kernel_time_share']))"
done
