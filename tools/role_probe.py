#!/usr/bin/env python
"""GPU box diagnostic: run one rollout in role mode and report how the workgroups landed on the CUs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visual_foresight_amd import _lib
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
from oracle import pixel_cost
M, T = int(sys.argv[1]) if len(sys.argv) > 1 else 96, 4
pred = HipVPredEvaluation('', dict(designated_pixel_count=1, run_batch_size=M, sequence_length=T + 2, role_mode=1)).restore()
if os.environ.get('VF_PROBE_STATS'):
    _lib.check(_lib.load_library().vf_set_phase_stats(pred._handle, 1))
rs = np.random.RandomState(0)
ctx = {'context_frames': rs.randint(0, 256, (2, 1, 64, 64, 3)).astype(np.uint8), 'context_actions': np.zeros((1, 4)),
       'context_states': np.zeros((2, 5)), 'context_pixel_distributions': pixel_cost.one_hot_distrib([[[32, 32]]], 2, 1, 64, 64, 1)}
acts = rs.normal(0, 0.05, (M, T, 4))
t0 = time.time()
try:
    s, _ = pred.score(ctx, {'actions': acts}, [[[16, 48]]])
    print('ok', s[:3], '%.1f s' % (time.time() - t0))
except Exception as e:
    print('FAILED after %.1f s: %s' % (time.time() - t0, str(e)[:80]))
print('census (active, CUs with k arrivals):', pred.role_census())
if os.environ.get('VF_PROBE_STATS'):
    import ctypes
    lib = _lib.load_library()
    N = 120
    types, items, wr = (ctypes.c_int32 * N)(), (ctypes.c_int32 * N)(), (ctypes.c_uint64 * (2 * N))()
    n = lib.vf_debug_phase_stats(pred._handle, N, types, items, wr)
    names = ['LSTM', 'CONV_RELU', 'CONV_RAW', 'CONVT_RELU', 'CONVT_RAW', 'FC', 'SA', 'FIN', 'COMPOSITE']
    for i in range(min(n, 40)):
        print('%3d %-10s items %5d wait %10.3f ms run %10.3f ms' % (i, names[types[i]], items[i], wr[2 * i] * 1e-5, wr[2 * i + 1] * 1e-5))
