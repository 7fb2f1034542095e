#!/usr/bin/env python
"""GPU box: the same planning call many times - every repetition must reproduce the first one bit for bit.

A race between workgroups of the persistent rollout shows up here as a handful of samples whose cost sums change from
call to call (that is how two experimental variants of the scheduler loop were caught, profiles/r03_tile_plan_sweep.txt).
    python tools/stress_repeat.py [repetitions]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import pixel_cost  # noqa: E402  (one_hot_distrib only: input construction)
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights  # noqa: E402
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation  # noqa: E402
from visual_foresight_amd.video_prediction.savp_arch import Savp2Config, SavpConfig  # noqa: E402
from visual_foresight_amd.video_prediction.savp_arch import CdnaWeights as SavpWeights  # noqa: E402
from visual_foresight_amd.video_prediction.savp3_arch import Savp3Config  # noqa: E402


def run(arch, H, W, T, M, nd, prec, seed, reps, ncam=1):
    adim = 4 if arch == 'cdna' else (12 if arch == 'savp3' else 6)
    hp = dict(designated_pixel_count=nd, run_batch_size=M, adim=adim, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, precision=prec, arch=arch, ncam=ncam)
    if arch == 'savp3':     # (8 of the 12 action channels are the latent)
        hp['zdim'] = 8
        cfg = Savp3Config(height=H, width=W, adim=adim, ndesig=nd, sequence_length=T + 2, zdim=8)
        weights = [SavpWeights.random(cfg, seed=seed + v, bias_scale=0.05, ln_jitter=0.1) for v in range(ncam)]
    elif arch == 'cdna':
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
    ref, bad = None, 0
    for it in range(reps):
        try:
            sc, pt = pred.score(ctx, {'actions': actions}, goal)
        except Exception as e:      # NaN scores raise (hip_predictor._check_scores)
            if ref is None:         # never accept a failure as the reference value
                raise
            print('  repetition %d raised: %s' % (it, str(e)[:100]), flush=True)
            bad += 1
            continue
        sc, pt = np.array(sc), np.array(pt)
        if not (np.isfinite(sc).all() and np.isfinite(pt).all()):
            if ref is None:
                raise RuntimeError('the first repetition produced non-finite scores')
            bad += 1
            continue
        cur = (sc.tobytes(), pt.tobytes())
        if ref is None:
            ref = cur
        elif cur != ref:
            bad += 1
    print('%-5s %3dx%-3d M%-4d T%-2d nd%d views %d %-6s: %d of %d repetitions differ from the first' % (
        arch, H, W, M, T, nd, ncam, prec, bad, reps - 1), flush=True)
    return bad


if __name__ == '__main__':
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    total = 0
    total += run('cdna', 64, 64, 4, 120, 2, 'fp32', 2, reps)
    total += run('cdna', 64, 64, 4, 200, 4, 'fp32', 3, reps)
    total += run('cdna', 64, 64, 6, 150, 3, 'fp32', 4, reps)
    total += run('cdna', 64, 64, 13, 200, 1, 'fp32', 1, max(reps // 3, 3))
    total += run('cdna', 64, 64, 4, 100, 2, 'fp32', 5, reps, ncam=2)
    total += run('cdna', 64, 64, 4, 120, 2, 'bf16x6', 6, reps)
    total += run('savp', 64, 64, 3, 150, 2, 'fp32', 9, reps)
    total += run('savp', 128, 128, 3, 40, 2, 'fp32', 8, max(reps // 3, 3))
    # small shards: the recurrent halves yield to their CU partner there (fewer than 96 samples per view), and every
    # item publishes write-through
    total += run('cdna', 64, 64, 13, 25, 1, 'fp32', 11, reps)
    total += run('cdna', 64, 64, 6, 50, 2, 'fp32', 12, reps)
    total += run('cdna', 64, 64, 5, 7, 4, 'fp32', 13, reps)
    total += run('cdna', 64, 64, 4, 40, 2, 'fp32', 14, reps, ncam=2)
    total += run('savp2', 64, 64, 3, 60, 2, 'fp32', 15, reps)
    total += run('savp2', 128, 128, 3, 20, 1, 'fp32', 16, max(reps // 3, 3))
    # the published generator: element-wise items between the GEMMs, every one of them publishing write-through
    total += run('savp3', 64, 64, 3, 40, 2, 'fp32', 17, reps)
    total += run('savp3', 32, 32, 4, 100, 1, 'fp32', 18, reps)
    total += run('savp3', 128, 128, 2, 12, 1, 'fp32', 19, max(reps // 3, 3))
    print('TOTAL differing repetitions:', total)
    sys.exit(1 if total else 0)
