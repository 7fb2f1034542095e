#!/usr/bin/env python
"""Staged probe of large-batch / 128x128 rollouts: prints a timestamped line per stage (unbuffered).

    python tools/c5_probe.py SIZE BATCH T [precision]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
T0 = time.time()


def log(*a):
    print('[%7.2fs]' % (time.time() - T0), *a, flush=True)


def main():
    size, B, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    prec = sys.argv[4] if len(sys.argv) > 4 else 'fp32'
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    import torch
    log('imports done')
    hp = dict(designated_pixel_count=1, run_batch_size=B, adim=12, sdim=5, image_height=size, image_width=size,
              sequence_length=T + 2, precision=prec)
    pred = HipVPredEvaluation('', hp)
    log('engine created; device memory in use %.1f GB' % ((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9))
    pred.restore()
    log('weights loaded')
    rs = np.random.RandomState(0)
    ctx = {'context_frames': rs.randint(0, 256, (2, 1, size, size, 3)).astype(np.uint8),
           'context_actions': np.zeros((1, 12)), 'context_states': rs.normal(0, .1, (2, 5)),
           'context_pixel_distributions': None}
    d = np.zeros((2, 1, size, size, 1), np.float32)
    d[:, 0, size // 2, size // 2, 0] = 1
    ctx['context_pixel_distributions'] = d
    acts = rs.normal(0, .1, (B, T, 12))
    goal = np.array([[[size // 4, size // 4]]])
    pred.set_profiling(True)
    for i in range(3):
        t = time.time()
        s, _ = pred.score(ctx, {'actions': acts}, goal)
        log('score call %d: %.1f ms, status %s, score[0] %.5f' % (i, 1e3 * (time.time() - t), pred.device_status(), s[0]))
    k_ms, n, fl, busy = pred.get_profile()
    log('kernel %.1f ms/launch, %.1f TF/s' % (k_ms / max(n, 1), fl / (k_ms * 1e-3) / 1e12 if k_ms else 0))


if __name__ == '__main__':
    main()
