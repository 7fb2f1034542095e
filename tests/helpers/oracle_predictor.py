"""VPredEvaluation duck-type around the CPU oracle (test infrastructure only)."""
import numpy as np
import torch

from oracle.cdna_predictor import OracleCdna
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig


def make_oracle_predictor_class(weights_factory, dtype=torch.float32):
    class OracleVPredEvaluation(object):
        wants_agent_params = True
        n_context_default = 2

        def __init__(self, model_path, hparams, n_gpus=1, first_gpu=0):
            hp = dict(hparams)
            self.n_context = 2
            self.sequence_length = hp['sequence_length']
            self.cfg = CdnaConfig(height=hp['image_height'], width=hp['image_width'], adim=hp['adim'],
                                  sdim=hp['sdim'], ndesig=hp['designated_pixel_count'],
                                  sequence_length=hp['sequence_length'])
            self.weights = None

        def restore(self):
            self.weights = weights_factory(self.cfg)
            self.oracle = OracleCdna(self.weights, dtype)

        def __call__(self, context, inputs):
            f, d, s = self.oracle.rollout(context['context_frames'], context['context_actions'],
                                          context['context_pixel_distributions'],
                                          context['context_states'], np.asarray(inputs['actions']))
            return {'predicted_frames': f.astype(np.float32),
                    'predicted_pixel_distributions': d.astype(np.float32)}

    return OracleVPredEvaluation
