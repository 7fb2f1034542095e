#!/usr/bin/env python
"""Condense a bench.py JSON line to one readable line:  bench_line.py FILE [label]   (FILE '-' = stdin)."""
import json
import sys
if len(sys.argv) < 2:
    sys.exit('usage: bench_line.py FILE|- [label]')     # never block on a terminal-less stdin by accident
src = sys.stdin if sys.argv[1] == '-' else open(sys.argv[1])
r = json.loads(src.read().strip().splitlines()[-1])
sys.argv = sys.argv[:1] + sys.argv[2:]
rf = r['roofline']
print('%s M=%d: %.0f frames/s  %.2f iters/s  %.1f ms/step | %s: %.1f TF/s frac %.3f  launches %d avg %.1f us  share %.3f host %.1f ms' % (
    sys.argv[1] if len(sys.argv) > 1 else '', r['config']['num_samples'], r['value'], r['cem_iters_per_sec'],
    r['ms_per_step'], rf['kernel'].split(' ')[0][:26], rf['achieved'], rf['frac'], rf['launches'], rf['avg_launch_us'],
    rf['kernel_time_share'], r.get('host_ms_per_step_outside_predictor', -1)))
if 'alt_precision' in r:
    a = r['alt_precision']; ar = a['roofline']
    print('   alt %s: %.0f frames/s  %.2f iters/s  %.1f ms/step | %.1f TF/s fp32-equiv (frac %.3f) avg %.1f us  same elites: %s' % (
        a['precision'], a['value'], a['cem_iters_per_sec'], a['ms_per_step'], ar['achieved'], ar['frac'], ar['avg_launch_us'],
        a['elites_identical_to_primary']))
