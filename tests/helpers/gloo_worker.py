"""One rank of the 2-rank gloo test: run a few planning calls and dump what the rank saw."""
import contextlib
import io
import os
import pickle
import sys

import numpy as np


def run(rank, world, port, out_dir, num_samples, propagation):
    import torch.distributed as dist
    from tests.helpers.sharded_fake import make_sharded_fake_class
    from visual_foresight_amd.policy.cem_controllers import PixelCostController
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    T, H, W = 5, 16, 16
    fake = make_sharded_fake_class(T, H, W)
    pol = {'predictor_class': fake, 'verbose': False, 'rejection_sampling': False, 'repeat': 1,
           'num_samples': num_samples, 'replan_interval': 2}
    if propagation:
        pol['predictor_propagation'] = True
    ag = {'adim': 4, 'sdim': 5, 'image_height': H, 'image_width': W}
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl = PixelCostController(ag, pol, 0, 1)
        ctrl.reset()
    np.random.seed(123)         # every rank draws the identical candidate set
    rs = np.random.RandomState(5)
    images = rs.randint(0, 256, (5, 1, H, W, 3)).astype(np.uint8)
    states = rs.normal(0, 0.1, (5, 5))
    log = []
    for t in range(4):
        with contextlib.redirect_stdout(io.StringIO()):
            out = ctrl.act(t=t, i_tr=0, desig_pix=[[8, 8]], goal_pix=[[3, 12]], images=images[:t + 1],
                           state=states[:t + 1])
        log.append({'action': np.array(out['actions']),
                    'plan_stat': {k: np.array(v) for k, v in out['plan_stat'].items()},
                    'best': None if ctrl._best_indices is None else np.array(ctrl._best_indices)})
    # what bench.py prints at N > 1: the per-call digests compared across the ranks of THIS job
    from visual_foresight_amd.video_prediction.sharding import plan_digest, gather_plan_digests
    same, shas = gather_plan_digests([plan_digest(e['plan_stat'], e['best'], e['action']) for e in log])
    with open(os.path.join(out_dir, 'rank%d_of%d.pkl' % (rank, world)), 'wb') as f:
        pickle.dump({'log': log, 'evaluated': list(fake.evaluated), 'identical_across_ranks': same,
                     'scores_sha_per_rank': shas}, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    rank, world, port, out_dir, num_samples, propagation = sys.argv[1:7]
    run(int(rank), int(world), int(port), out_dir, int(num_samples), propagation == '1')
