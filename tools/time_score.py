#!/usr/bin/env python
"""GPU box diagnostic: where does the wall time of one score() call go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
from oracle import pixel_cost
M, T = 200, 13
pred = HipVPredEvaluation('', dict(designated_pixel_count=1, run_batch_size=M, sequence_length=T + 2)).restore()
rs = np.random.RandomState(0)
ctx = {'context_frames': rs.randint(0, 256, (2, 1, 64, 64, 3)).astype(np.uint8), 'context_actions': np.zeros((1, 4)),
       'context_states': np.zeros((2, 5)), 'context_pixel_distributions': pixel_cost.one_hot_distrib([[[32, 32]]], 2, 1, 64, 64, 1)}
acts = rs.normal(0, 0.05, (M, T, 4))
goal = np.array([[16, 48]])
for mode in ('cpu()', 'event-poll', 'synchronize'):
    for rep in range(4):
        t0 = time.perf_counter()
        pred._set_context(ctx)
        local = torch.from_numpy(np.ascontiguousarray(acts, dtype=np.float32)).to(pred.device)
        scores = torch.empty(M, dtype=torch.float32, device=pred.device)
        per = torch.empty((M, 1), dtype=torch.float32, device=pred.device)
        t1 = time.perf_counter()
        pred._rollout_chunk(local, goal, 10., scores, per)
        t2 = time.perf_counter()
        if mode == 'cpu()':
            s = scores.cpu()
        elif mode == 'event-poll':
            ev = torch.cuda.Event(); ev.record()
            while not ev.query():
                pass
            s = scores.cpu()
        else:
            torch.cuda.synchronize(); s = scores.cpu()
        t3 = time.perf_counter()
        print('%-12s rep %d: prep %.2f ms  enqueue %.2f ms  wait %.2f ms  total %.2f ms' % (mode, rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0)))
