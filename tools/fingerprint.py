#!/usr/bin/env python
"""GPU box: sha256 fingerprints of scores / predicted frames / distributions for a fixed set of seeded workloads.

Used to check that a kernel change which is meant to be bit-neutral (index arithmetic, instruction scheduling, where a
LayerNorm gain is fetched from) really leaves every output bit where it was: run before and after, diff the lines.
Covers both architectures, both precision modes, several tile plans (batch sizes) and two camera views.
    python tools/fingerprint.py [tag] > gpurun_out/fingerprint_<tag>.txt
"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import pixel_cost  # noqa: E402  (one_hot_distrib only: input construction, nothing is checked against it)
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights  # noqa: E402
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation  # noqa: E402
from visual_foresight_amd.video_prediction.savp_arch import SavpConfig  # noqa: E402
from visual_foresight_amd.video_prediction.savp_arch import CdnaWeights as SavpWeights  # noqa: E402
from visual_foresight_amd.video_prediction.savp_arch import Savp2Config  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def run(name, arch, H, W, T, M, nd, prec, export, seed, ncam=1):
    adim = 4 if arch == 'cdna' else 6
    hp = dict(designated_pixel_count=nd, run_batch_size=M, adim=adim, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, precision=prec, arch=arch, ncam=ncam)
    if arch == 'cdna':
        cfg = CdnaConfig(height=H, width=W, ndesig=nd, sequence_length=T + 2)
        weights = [CdnaWeights.random(cfg, seed=seed + v, bias_scale=0.05, ln_jitter=0.1) for v in range(ncam)]
    else:
        cfg = (Savp2Config if arch == 'savp2' else SavpConfig)(height=H, width=W, adim=adim, ndesig=nd, sequence_length=T + 2)
        weights = [SavpWeights.random(cfg, seed=seed + v, bias_scale=0.05, ln_jitter=0.1) for v in range(ncam)]
    pred = HipVPredEvaluation('', hp)
    pred.restore(weights if ncam > 1 else weights[0])
    rs = np.random.RandomState(seed)
    desig = rs.randint(0, min(H, W), (ncam, nd, 2))
    ctx = {'context_frames': rs.randint(0, 256, (2, ncam, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (1, adim)), 'context_states': rs.normal(0, 0.1, (2, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib(desig, 2, ncam, H, W, nd)}
    actions = rs.normal(0, 0.1, (M, T, adim))
    goal = rs.randint(0, min(H, W), (ncam, nd, 2))
    sc, pt = pred.score(ctx, {'actions': actions}, goal)
    line = '%-28s scores %s' % (name, digest(sc, pt))
    if export:
        out = pred(ctx, {'actions': actions[:4]})
        line += ' frames %s distrib %s' % (digest(out['predicted_frames']), digest(out['predicted_pixel_distributions']))
    print(line, flush=True)


if __name__ == '__main__':
    print('# tools/fingerprint.py %s' % (sys.argv[1] if len(sys.argv) > 1 else ''))
    run('cdna 64x64 M200 T13', 'cdna', 64, 64, 13, 200, 1, 'fp32', False, 1)
    run('cdna 64x64 M120 T4 nd2', 'cdna', 64, 64, 4, 120, 2, 'fp32', True, 2)
    run('cdna 64x64 M25 T5', 'cdna', 64, 64, 5, 25, 1, 'fp32', True, 3)
    run('cdna 64x64 M7 T3', 'cdna', 64, 64, 3, 7, 1, 'fp32', True, 4)
    run('cdna 48x80 M30 T3 nd2', 'cdna', 48, 80, 3, 30, 2, 'fp32', True, 5)
    run('cdna 64x64 M60 T4 2 views', 'cdna', 64, 64, 4, 60, 1, 'fp32', True, 6, ncam=2)
    run('cdna 64x64 M100 T5 bf16x6', 'cdna', 64, 64, 5, 100, 1, 'bf16x6', True, 7)
    run('savp 128x128 M40 T3', 'savp', 128, 128, 3, 40, 1, 'fp32', True, 8)
    run('savp 64x64 M150 T3 nd2', 'savp', 64, 64, 3, 150, 2, 'fp32', True, 9)
    run('savp 128x128 M20 T2 bf16x6', 'savp', 128, 128, 2, 20, 1, 'bf16x6', False, 10)
    run('savp2 128x128 M30 T3', 'savp2', 128, 128, 3, 30, 1, 'fp32', True, 11)
    run('savp2 64x64 M101 T3 nd2', 'savp2', 64, 64, 3, 101, 2, 'fp32', True, 12)
    run('savp2 64x64 M7 T4', 'savp2', 64, 64, 4, 7, 1, 'fp32', True, 13)
