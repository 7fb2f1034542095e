"""Several independent single-view HIP predictors behind one ``VPredEvaluation`` duck-type.

The reference's multi-view models are one network per view sharing actions and states
(``IndepMultiSAVPVideoPredictionModel``; outputs stacked on a camera axis,
``visual_mpc/video_prediction/vpred_model_interface.py:60-88``).  Here every view gets its own
``HipVPredEvaluation`` engine (own weights, own device buffers); the views are rolled one after
the other on the same GPU and their per-task scores are concatenated camera-major, which is the
order ``PixelCostController`` stacks them in (``pixel_cost_controller.py:138-149``).
"""
import os

import numpy as np

from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation


class MultiViewHipPredictor(object):
    wants_agent_params = True
    n_context_default = HipVPredEvaluation.n_context_default

    def __init__(self, model_path, hparams, n_gpus=1, first_gpu=0):
        hp = dict(hparams)
        self.n_cam = int(hp.pop('ncam', 1))
        seed = int(hp.get('seed', 0))
        self.views = []
        for c in range(self.n_cam):
            path = model_path
            if isinstance(model_path, (list, tuple)):
                path = model_path[c]
            elif model_path and os.path.isdir(os.path.join(os.path.expanduser(model_path), 'view%d' % c)):
                path = os.path.join(model_path, 'view%d' % c)
            self.views.append(HipVPredEvaluation(path, dict(hp, ncam=1, seed=seed + c), n_gpus=n_gpus,
                                                 first_gpu=first_gpu))
        self.n_context = self.views[0].n_context
        self.sequence_length = self.views[0].sequence_length

    def restore(self, weights=None):
        for c, v in enumerate(self.views):
            v.restore(None if weights is None else weights[c])
        return self

    @property
    def weights(self):
        return [v.weights for v in self.views]

    @staticmethod
    def _view_context(context, c):
        return {'context_frames': np.asarray(context['context_frames'])[:, c:c + 1],
                'context_pixel_distributions': np.asarray(context['context_pixel_distributions'])[:, c:c + 1],
                'context_actions': context['context_actions'], 'context_states': context['context_states']}

    def score(self, context, inputs, goal_pix, finalweight=10., only_take_first_view=False):
        goal = np.asarray(goal_pix).reshape(self.n_cam, -1, 2)
        per_task = []
        for c, v in enumerate(self.views):
            _, pt = v.score(self._view_context(context, c), inputs, goal[c:c + 1], finalweight=finalweight)
            per_task.append(pt)
            if only_take_first_view:
                break
        per_task = np.concatenate(per_task, axis=1)
        if only_take_first_view:
            per_task = per_task[:, :1]
        return np.mean(per_task, axis=1), per_task

    def fetch_pixel_distributions(self, sample_index):
        return np.concatenate([v.fetch_pixel_distributions(sample_index) for v in self.views], axis=1)

    def __call__(self, context, inputs):
        outs = [v(self._view_context(context, c), inputs) for c, v in enumerate(self.views)]
        return {'predicted_frames': np.concatenate([o['predicted_frames'] for o in outs], axis=2),
                'predicted_pixel_distributions': np.concatenate(
                    [o['predicted_pixel_distributions'] for o in outs], axis=2),
                'predicted_states': outs[0]['predicted_states']}
