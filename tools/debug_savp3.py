"""Development aid: arch 'savp3' on the GPU against OracleSavp3 (prints the errors; tests/test_gpu_savp3.py asserts them)."""
import sys
import time

import numpy as np
import torch

from oracle import pixel_cost
from oracle.savp3_predictor import OracleSavp3
from visual_foresight_amd.video_prediction.savp3_arch import Savp3Config, CdnaWeights
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation


def run(H, W, T, M, nd, persistent, layer_spec=0, adim=12, zdim=8, seed=3):
    hp = dict(designated_pixel_count=nd, run_batch_size=M, adim=adim, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, arch='savp3', zdim=zdim, layer_spec=layer_spec, persistent=persistent)
    pred = HipVPredEvaluation('', hp)
    cfg = Savp3Config(height=H, width=W, adim=adim, ndesig=nd, sequence_length=T + 2, zdim=zdim, layer_spec=layer_spec)
    w = CdnaWeights.random(cfg, seed=seed, bias_scale=0.05, ln_jitter=0.1)
    pred.restore(w)
    rs = np.random.RandomState(H + W + T + M)
    desig = rs.randint(0, min(H, W), (1, nd, 2))
    d = pixel_cost.one_hot_distrib(desig, 2, 1, H, W, nd)
    d[1] = 0.5 * d[1] + 0.5 / (H * W)
    a_env = adim - zdim
    ctx = {'context_frames': rs.randint(0, 256, (3, 1, H, W, 3)).astype(np.uint8),
           'context_actions': np.concatenate([rs.normal(0, 0.05, (2, a_env)), np.zeros((2, zdim))], axis=1),
           'context_states': rs.normal(0, 0.1, (3, 5)), 'context_pixel_distributions': d}
    actions = np.concatenate([rs.normal(0, 0.1, (M, T, a_env)), rs.normal(0, 1.0, (M, T, zdim))], axis=2)
    goal = rs.randint(-2, max(H, W) + 2, (1, nd, 2))
    t0 = time.time()
    scores, per_task = pred.score(ctx, {'actions': actions}, goal, finalweight=10.)
    got = pred(ctx, {'actions': actions})
    t1 = time.time()
    f, dd, s = OracleSavp3(w).rollout(ctx['context_frames'], ctx['context_actions'], ctx['context_pixel_distributions'],
                                      ctx['context_states'], actions)
    want, want_pt = pixel_cost.eval_pixel_cost(dd, goal, 10.)
    ef = np.abs(got['predicted_frames'] - f).max(axis=(0, 2, 3, 4, 5))
    dmax = dd.max(axis=(3, 4), keepdims=True)
    ed = (np.abs(got['predicted_pixel_distributions'] - dd) / dmax).max(axis=(0, 2, 3, 4, 5))
    es = np.abs(got['predicted_states'] - s).max()
    esc = np.abs(scores - want).max() / np.abs(want).max()
    print('%dx%d T%d M%d nd%d spec%d persistent=%d: frame err per step %s, distrib %s, state %.2g, score %.2g, status %d, gpu %.2fs'
          % (H, W, T, M, nd, layer_spec, persistent, np.array2string(ef, precision=2), np.array2string(ed, precision=2), es, esc,
             pred.device_status(), t1 - t0), flush=True)
    return scores, got


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'small'
    if which == 'small':
        a, ga = run(32, 32, 3, 5, 1, 0)
        b, gb = run(32, 32, 3, 5, 1, 1)
        print('persistent == per-layer:', np.array_equal(a, b), np.array_equal(ga['predicted_frames'], gb['predicted_frames']))
        a, ga = run(64, 64, 3, 4, 2, 0)
        b, gb = run(64, 64, 3, 4, 2, 1)
        print('persistent == per-layer:', np.array_equal(a, b), np.array_equal(ga['predicted_frames'], gb['predicted_frames']))
    elif which == 'more':
        run(48, 64, 2, 7, 4, 1)
        run(64, 80, 2, 3, 1, 1)
        run(128, 128, 2, 2, 1, 1)
        run(128, 128, 2, 2, 1, 1, layer_spec=64)
        run(64, 64, 3, 33, 1, 1)
