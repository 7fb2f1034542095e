#!/usr/bin/env python
"""GPU box diagnostic: time split of conv-LSTM items inside the persistent rollout.

Needs the instrumented library (hipcc ... -DVF_TILE_STATS, tools/ubench/build_variants.sh with
VARIANTS=TILE_STATS); run as   VF_LIBRARY=build/ab/tilestats.so python tools/tile_stats.py [M]
(K loop column = first chunk staged -> epilogue, INCLUDING the staging of later chunks shown in the last column)
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_foresight_amd import _lib
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation

# usage: tile_stats.py [M [precision [H [arch [T]]]]]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
H = int(sys.argv[3]) if len(sys.argv) > 3 else 64
arch = sys.argv[4] if len(sys.argv) > 4 else 'cdna'
T = int(sys.argv[5]) if len(sys.argv) > 5 else 13
adim = 12 if arch in ('savp', 'savp2') else 4
pred = HipVPredEvaluation('', dict(designated_pixel_count=1, run_batch_size=M, sequence_length=T + 2, precision=prec,
                                   image_height=H, image_width=H, arch=arch, adim=adim)).restore()
rs = np.random.RandomState(0)
d = np.zeros((2, 1, H, H, 1), np.float32); d[:, 0, H // 2, H // 2, 0] = 1
ctx = {'context_frames': rs.randint(0, 256, (2, 1, H, H, 3)).astype(np.uint8), 'context_actions': np.zeros((1, adim)),
       'context_states': np.zeros((2, 5)), 'context_pixel_distributions': d}
acts = rs.normal(0, 0.05, (M, T, adim))
lib = _lib.load_library()
lib.vf_debug_tile_clocks.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int32]
buf = (ctypes.c_uint64 * 256)()
pred.score(ctx, {'actions': acts}, [[[16, 48]]])
lib.vf_debug_tile_clocks(buf, 1)
pred.set_profiling(True)
pred.score(ctx, {'actions': acts}, [[[16, 48]]])
k_ms = pred.get_profile()[0]
lib.vf_debug_tile_clocks(buf, 0)
a = np.array(buf[:], dtype=np.float64).reshape(32, 8)
tick_us = 0.01
names = {2: 'lstm1/2 (Ctot 64 @32)', 3: 'lstm7 (96 @32)', 11: 'lstm3 (96 @16)', 12: 'lstm4 (128 @16)',
         14: 'lstm6 (192 @16)', 8: 'lstm5 (256 @8)', 16: 'enc0', 17: 'enc3', 18: 'enc1', 19: 'enc2', 20: 'convt1',
         21: 'convt2', 22: 'FC', 23: 'fused top', 24: 'unfused top',
         25: 'fused epilogue: stats | halo | wait+LN | (items) | compose'}
print('kernel %.2f ms' % k_ms)
print('%-24s %7s %9s %9s %9s %9s | per-slot ms: pro kloop epi' % ('layer', 'items', 'pro us', 'kloop us', 'epi us', 'stage us'))
tot = np.zeros(3)
for key in list(range(15)) + list(range(16, 32)):
    n = a[key, 3]
    if n == 0:
        continue
    pro, kl, ep, st = (a[key, i] * tick_us / n for i in (0, 1, 2, 4))
    if key < 15:
        tot += np.array([a[key, 0], a[key, 1], a[key, 2]]) * tick_us / 512 / 1e3
    print('%-24s %7d %9.1f %9.1f %9.1f %9.1f | %6.2f %6.2f %6.2f' % (
        names.get(key, 'key %d' % key), n, pro, kl, ep, st, a[key, 0] * tick_us / 512e3, a[key, 1] * tick_us / 512e3,
        a[key, 2] * tick_us / 512e3))
print('LSTM totals per slot [ms]: prologue %.2f  K loop %.2f  epilogue %.2f' % tuple(tot))
n = a[15, 7]
print('scheduler per item: ticket+lookup %.2f us, publish %.2f us  (%d items) -> per slot %.2f + %.2f ms' % (
    a[15, 5] * tick_us / n, a[15, 6] * tick_us / n, n, a[15, 5] * tick_us / 512e3, a[15, 6] * tick_us / 512e3))
