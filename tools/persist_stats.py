#!/usr/bin/env python
"""GPU box diagnostic: per-phase wait/run time of one persistent rollout (VF_PERSIST_STATS=1)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['VF_PERSISTENT'] = '1'
import numpy as np, torch
from visual_foresight_amd import _lib
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
from oracle import pixel_cost

# usage: persist_stats.py [M [H [arch [T [layer_spec]]]]]   (arch: cdna | savp | savp2 | savp3; the SAVP-class networks run with 12
# action channels, 8 of them latent for savp3)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
arch = sys.argv[3] if len(sys.argv) > 3 else 'cdna'
T = int(sys.argv[4]) if len(sys.argv) > 4 else 13
adim = 12 if arch in ('savp', 'savp2', 'savp3') else 4
extra = dict(zdim=8, layer_spec=int(sys.argv[5]) if len(sys.argv) > 5 else 0) if arch == 'savp3' else {}
pred = HipVPredEvaluation('', dict(designated_pixel_count=1, run_batch_size=M, sequence_length=T + 2, image_height=H,
                                   image_width=H, arch=arch, adim=adim, **extra)).restore()
lib = _lib.load_library()
_lib.check(lib.vf_set_phase_stats(pred._handle, 1))
rs = np.random.RandomState(0)
ctx = {'context_frames': rs.randint(0, 256, (2, 1, H, H, 3)).astype(np.uint8), 'context_actions': np.zeros((1, adim)),
       'context_states': np.zeros((2, 5)), 'context_pixel_distributions': pixel_cost.one_hot_distrib([[[32, 32]]], 2, 1, H, H, 1)}
acts = rs.normal(0, 0.05, (M, T, adim))
for _ in range(2):
    pred.score(ctx, {'actions': acts}, [[[16, 48]]])
N = 2000
types, items, wr = (ctypes.c_int32 * N)(), (ctypes.c_int32 * N)(), (ctypes.c_uint64 * (2 * N))()
n = lib.vf_debug_phase_stats(pred._handle, N, types, items, wr)
names = {i: n for i, n in enumerate(['LSTM', 'CONV_RELU', 'CONV_RAW', 'CONVT_RELU', 'CONVT_RAW', 'FC', 'SA', 'FIN', 'COMPOSITE', 'TOP_FUSED',
                                     'CONV_PAIR', 'COND', 'CONV_RAW3', 'GATES_RAW', 'EW', 'CONV_RAW3G2', 'CONV_RAW3G4'])}
names.update({100 + i: n for i, n in enumerate(['EW_SA3', 'EW_COND3', 'EW_INORM', 'EW_INCELL', 'EW_UPSAMPLE', 'EW_TRANSFORM', 'EW_COMPOSE', 'EW_TOP3'])})
tick = 1e-8     # wall_clock64: 100 MHz
agg = {}
print('phase type items  wait_ms(sum over items)  run_ms(sum)  run_us/item')
for i in range(n):
    w, r = wr[2 * i] * tick * 1e3, wr[2 * i + 1] * tick * 1e3
    if i < 16 or n - 44 <= i < n - 22 or (arch == 'savp3' and n - 120 <= i < n - 60):
        print('%3d %-10s %5d %10.2f %10.2f %10.1f' % (i, names[types[i]], items[i], w, r, 1e3 * r / max(items[i], 1)))
    a = agg.setdefault(names[types[i]], [0, 0., 0.])
    a[0] += items[i]; a[1] += w; a[2] += r
print('--- totals per type (ms summed over items; 512 workgroup slots)')
tw = tr = 0
for k, (it, w, r) in agg.items():
    print('%-10s items %7d wait %10.1f run %10.1f' % (k, it, w, r)); tw += w; tr += r
print('total wait %.1f ms run %.1f ms -> per slot: wait %.2f run %.2f ms' % (tw, tr, tw / 512, tr / 512))
