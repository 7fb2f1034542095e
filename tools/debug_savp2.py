import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pixel_cost
from oracle.savp_predictor import OracleSavp2
from visual_foresight_amd.video_prediction.savp_arch import Savp2Config, CdnaWeights
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation

H = W = 64; T, M, nd, adim = 3, 5, 1, 6
cfg = Savp2Config(height=H, width=W, adim=adim, ndesig=nd, sequence_length=T + 2)
base = CdnaWeights.random(cfg, seed=3, bias_scale=0.05, ln_jitter=0.1)
rs = np.random.RandomState(1)
desig = rs.randint(0, H, (1, nd, 2))
d = pixel_cost.one_hot_distrib(desig, 2, 1, H, W, nd); d[1] = 0.5 * d[1] + 0.5 / (H * W)
ctx = {'context_frames': rs.randint(0, 256, (3, 1, H, W, 3)).astype(np.uint8), 'context_actions': rs.normal(0, 0.05, (2, adim)),
       'context_states': rs.normal(0, 0.1, (3, 5)), 'context_pixel_distributions': d}
actions = rs.normal(0, 0.1, (M, T, adim))

def run(weights, tag, **kw):
    hp = dict(designated_pixel_count=nd, run_batch_size=M, adim=adim, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, arch='savp2', **kw)
    pred = HipVPredEvaluation('', hp); pred.restore(weights)
    got = pred(ctx, {'actions': actions})
    f, dd, s = OracleSavp2(weights).rollout(ctx['context_frames'], ctx['context_actions'], ctx['context_pixel_distributions'], ctx['context_states'], actions)
    ef = np.abs(got['predicted_frames'] - f).max(axis=(0, 2, 3, 4, 5))
    ed = (np.abs(got['predicted_pixel_distributions'] - dd) / dd.max(axis=(3, 4), keepdims=True)).max(axis=(0, 2, 3, 4, 5))
    es = np.abs(got['predicted_states'] - s).max(axis=(0, 2))
    print('%-40s frames per t %s  distrib %s  states %s' % (tag, ef, ed, es), flush=True)

run(base, 'full')
run(base, 'full per-layer launches', persistent=0)
t = dict(base.tensors)
for k, cx in enumerate((32, 32, 32, 64, 64, 128, 64)):
    w = t['lstm%d/w' % (k + 1)].copy(); w[:, :, cx:cx + adim + 5] = 0; t['lstm%d/w' % (k + 1)] = w
run(CdnaWeights(cfg, t), 'conditioning rows zeroed')
for hot in range(7):
    t2 = dict(t); t2['masks/w'] = np.zeros_like(t2['masks/w']); b = np.full(7, -80., np.float32); b[hot] = 80.; t2['masks/b'] = b
    run(CdnaWeights(cfg, t2), 'no cond, one-hot mask %d' % hot)
for k in range(7):
    t3 = dict(t); t3['lstm%d/w' % (k + 1)] = base.tensors['lstm%d/w' % (k + 1)]
    run(CdnaWeights(cfg, t3), 'cond only in lstm%d' % (k + 1))
