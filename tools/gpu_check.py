#!/usr/bin/env python
"""Diagnostic (GPU box): per-step error of the HIP predictor vs the fp32 and fp64 oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
from oracle.cdna_predictor import OracleCdna
from oracle import pixel_cost


def main(H=64, W=64, T=4, M=6, nd=1):
    hp = dict(designated_pixel_count=nd, run_batch_size=M, adim=4, sdim=5, image_height=H,
              image_width=W, sequence_length=T + 2)
    pred = HipVPredEvaluation('', hp)
    cfg = CdnaConfig(height=H, width=W, ndesig=nd, sequence_length=T + 2)
    weights = CdnaWeights.random(cfg, seed=3, bias_scale=0.05, ln_jitter=0.1)
    pred.restore(weights)
    rs = np.random.RandomState(0)
    desig = rs.randint(0, min(H, W), (1, nd, 2))
    ctx = {'context_frames': rs.randint(0, 256, (3, 1, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (2, 4)),
           'context_states': rs.normal(0, 0.1, (3, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib(desig, 2, 1, H, W, nd)}
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = rs.randint(0, min(H, W), (1, nd, 2))
    t0 = time.time()
    scores, per_task = pred.score(ctx, {'actions': actions}, goal, finalweight=10.)
    torch.cuda.synchronize()
    print('score call %.3fs' % (time.time() - t0))
    got = pred(ctx, {'actions': actions})
    for dt in (torch.float32, torch.float64):
        ora = OracleCdna(weights, dt)
        f, d, s = ora.rollout(ctx['context_frames'], ctx['context_actions'],
                              ctx['context_pixel_distributions'], ctx['context_states'], actions)
        ws, wpt = pixel_cost.eval_pixel_cost(d.astype(np.float32), goal, 10.)
        print('--- oracle', dt, 'H%d W%d T%d M%d nd%d' % (H, W, T, M, nd))
        for t in range(T):
            print(' t=%d frame abs err %.3g  distrib rel err %.3g  state abs err %.3g' % (
                t, np.abs(got['predicted_frames'][:, t] - f[:, t]).max(),
                np.abs(got['predicted_pixel_distributions'][:, t] - d[:, t]).max() / d[:, t].max(),
                np.abs(got['predicted_states'][:, t] - s[:, t]).max()))
        print(' scores', scores, '\n want  ', ws)
        print(' score rel err %.3g ; argsort equal: %s' % (np.abs(scores - ws).max() / np.abs(ws).max(),
                                                          np.array_equal(scores.argsort(), ws.argsort())))
    print(' distrib sums', got['predicted_pixel_distributions'].sum((3, 4))[0, :, 0])


if __name__ == '__main__':
    main()
    main(H=48, W=64, T=2, M=5, nd=2)
    main(H=32, W=32, T=2, M=9, nd=1)
