"""Fake predictor exposing the fused ``score`` entry point with the product's sample sharding.

Mirrors what ``HipVPredEvaluation.score`` does around the device call - shard by rank, score the
local samples, all-gather the score rows - with the fake NumPy predictor standing in for the
GPU rollout, so the multi-rank control flow can be exercised on CPU with gloo.
"""
import numpy as np
import torch

from oracle import pixel_cost
from tests.helpers.fake_predictor import make_fake_predictor_class
from visual_foresight_amd.video_prediction.sharding import dist_info, shard_bounds, all_gather_rows


def make_sharded_fake_class(T, height, width):
    Base = make_fake_predictor_class(T, height, width)

    class ShardedFake(Base):
        evaluated = []

        def score(self, context, inputs, goal_pix, finalweight=10., only_take_first_view=False):
            actions = np.asarray(inputs['actions'])
            M = actions.shape[0]
            rank, world = dist_info()
            lo, hi = shard_bounds(M, rank, world)
            type(self).evaluated.append((lo, hi))
            out = Base.__call__(self, context, {'actions': actions[lo:hi]})
            self._local = (lo, out['predicted_pixel_distributions'])
            scores, per_task = pixel_cost.eval_pixel_cost(out['predicted_pixel_distributions'],
                                                          np.asarray(goal_pix), finalweight)
            packed = torch.from_numpy(np.concatenate([scores[:, None], per_task], axis=1))
            full = all_gather_rows(packed, M).numpy()
            return full[:, 0].copy(), full[:, 1:].copy()

        def fetch_pixel_distributions(self, sample_index):
            import torch.distributed as dist
            lo, d = self._local
            rank, world = dist_info()
            local = sample_index - lo
            out = torch.zeros(d.shape[1:], dtype=torch.float32)
            if 0 <= local < d.shape[0]:
                out = torch.from_numpy(d[local].copy())
            if world > 1:
                dist.all_reduce(out)
            return out.numpy()

    return ShardedFake
