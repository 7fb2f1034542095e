"""VPredEvaluation duck-type around the CPU oracle (test infrastructure only)."""
import numpy as np
import torch

from oracle.cdna_predictor import OracleCdna
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig


def make_oracle_predictor_class(weights_factory, dtype=torch.float32):
    """``weights_factory(cfg)`` -> one weight set, or (``weights_factory(cfg, view)`` when the controller passes
    ``ncam > 1``) one per view: independent per-view networks stacked on the camera axis, as the reference's
    multi-view models are (``vpred_model_interface.py:60-88``)."""
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
            self.n_cam = int(hp.get('ncam', 1))
            self.weights = None
            self.calls = 0

        def restore(self):
            if self.n_cam == 1:
                self.weights = weights_factory(self.cfg)
                self.oracles = [OracleCdna(self.weights, dtype)]
            else:
                self.weights = [weights_factory(self.cfg, v) for v in range(self.n_cam)]
                self.oracles = [OracleCdna(w, dtype) for w in self.weights]

        def __call__(self, context, inputs):
            self.calls += 1
            frames, distribs = [], []
            for v, oracle in enumerate(self.oracles):
                f, d, s = oracle.rollout(np.asarray(context['context_frames'])[:, v:v + 1],
                                         context['context_actions'],
                                         np.asarray(context['context_pixel_distributions'])[:, v:v + 1],
                                         context['context_states'], np.asarray(inputs['actions']))
                frames.append(f.astype(np.float32))
                distribs.append(d.astype(np.float32))
            return {'predicted_frames': np.concatenate(frames, axis=2),
                    'predicted_pixel_distributions': np.concatenate(distribs, axis=2)}

    return OracleVPredEvaluation
