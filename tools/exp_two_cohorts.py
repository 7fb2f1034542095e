#!/usr/bin/env python
"""GPU box experiment: does running a small shard as TWO independent persistent launches (one workgroup per CU each,
co-resident, the second one offset in time) beat one launch with two workgroups per CU?  -DVF_DEBUG_KNOBS build,
VF_PERSIST_WGS_PER_CU=1 for the cohort engines.
    VF_LIBRARY=build/ab/knobs.so python tools/exp_two_cohorts.py [M=25]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights  # noqa: E402
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 25
T = 13
cfg = CdnaConfig(sequence_length=T + 2)
w = CdnaWeights.random(cfg, seed=0)
rs = np.random.RandomState(0)
d = np.zeros((2, 1, 64, 64, 1), np.float32)
d[:, 0, 32, 32, 0] = 1
ctx = {'context_frames': rs.randint(0, 256, (2, 1, 64, 64, 3)).astype(np.uint8), 'context_actions': np.zeros((1, 4)),
       'context_states': np.zeros((2, 5)), 'context_pixel_distributions': d}
acts = rs.normal(0, 0.05, (M, T, 4))
goal = [[[16, 48]]]


def make(m, wgs):
    os.environ['VF_PERSIST_WGS_PER_CU'] = str(wgs)
    p = HipVPredEvaluation('', dict(designated_pixel_count=1, run_batch_size=m, sequence_length=T + 2))
    p.restore(w)
    return p


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


one = make(M, 2)
ref = one.score(ctx, {'actions': acts}, goal)[0]


def run_one():
    s, _ = one._score_prepared(ctx, acts, M, goal, 10.)
    return s


print('one launch, 2 workgroups per CU, %d samples: %.3f ms per rollout' % (M, timed(run_one)))
for wgs in (1, 2):
    mA = (M + 1) // 2
    A, B = make(mA, wgs), make(M - mA, wgs)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    for delay in (0, 100000, 400000, 800000):
        out = {}

        def run_two():
            with torch.cuda.stream(sA):
                out['a'] = A._score_prepared(ctx, acts[:mA], mA, goal, 10.)[0]
            with torch.cuda.stream(sB):
                if delay:
                    torch.cuda._sleep(delay)
                out['b'] = B._score_prepared(ctx, acts[mA:], M - mA, goal, 10.)[0]
        ms = timed(run_two)
        got = np.concatenate([out['a'].cpu().numpy(), out['b'].cpu().numpy()])
        print('two launches (%d + %d samples), %d workgroup(s) per CU each, second delayed %7d cycles: %.3f ms, scores %s'
              % (mA, M - mA, wgs, delay, ms, 'identical' if np.array_equal(got, ref) else 'DIFFER'))
