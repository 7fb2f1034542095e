"""One rank of the multi-rank GPU test: the REAL HipVPredEvaluation / StochasticHipPredictor behind the
controller, ranks sharing one GPU over gloo (RCCL needs one GPU per rank; the sharding, the all-gather of
score rows and the propagation fetch are backend-independent)."""
import contextlib
import io
import os
import pickle
import sys

import numpy as np


def run(rank, world, port, out_dir, num_samples, stochastic, vpred_batch_size=0):
    import torch.distributed as dist
    from visual_foresight_amd.policy.cem_controllers import PixelCostController
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    from visual_foresight_amd.video_prediction.stochastic_predictor import StochasticHipPredictor
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    H = W = 32
    cls = StochasticHipPredictor.with_options(n_latent=3, zdim=4, latent_seed=9) if stochastic else HipVPredEvaluation
    pol = {'predictor_class': cls, 'verbose': False, 'rejection_sampling': False, 'repeat': 1, 'nactions': 3,
           'num_samples': num_samples, 'predictor_propagation': True, 'iterations': 2}
    if vpred_batch_size:        # shards larger than the engine's batch: only a rank's LAST chunk stays resident
        pol['vpred_batch_size'] = vpred_batch_size
    ag = {'adim': 4, 'sdim': 5, 'image_height': H, 'image_width': W}
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl = PixelCostController(ag, pol, 0, 1)
        ctrl.reset()
    np.random.seed(123)         # every rank draws the identical candidate set
    rs = np.random.RandomState(5)
    images = rs.randint(0, 256, (4, 1, H, W, 3)).astype(np.uint8)
    states = rs.normal(0, 0.1, (4, 5))
    log = []
    for t in range(4):
        with contextlib.redirect_stdout(io.StringIO()):
            out = ctrl.act(t=t, i_tr=0, desig_pix=[[16, 16]], goal_pix=[[5, 25]], images=images[:t + 1],
                           state=states[:t + 1])
        log.append({'action': np.array(out['actions']),
                    'plan_stat': {k: np.array(v) for k, v in out['plan_stat'].items()},
                    'best': None if ctrl._best_indices is None else np.array(ctrl._best_indices),
                    'chosen': None if ctrl._chosen_distrib is None else np.array(ctrl._chosen_distrib)})
    with open(os.path.join(out_dir, 'gpu_rank%d_of%d.pkl' % (rank, world)), 'wb') as f:
        pickle.dump({'log': log, 'rolled': int(ctrl.predictor._last_M)}, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    rank, world, port, out_dir, num_samples, stochastic = sys.argv[1:7]
    run(int(rank), int(world), int(port), out_dir, int(num_samples), stochastic == '1',
        int(sys.argv[7]) if len(sys.argv) > 7 else 0)
