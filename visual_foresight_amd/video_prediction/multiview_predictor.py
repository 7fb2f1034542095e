"""Multi-view predictor: ``ncam`` independent single-view networks behind one ``VPredEvaluation`` duck-type.

The reference's multi-view models are one network per view sharing actions and states
(``IndepMultiSAVPVideoPredictionModel``; outputs stacked on a camera axis,
``visual_mpc/video_prediction/vpred_model_interface.py:60-88``).  ``HipVPredEvaluation`` holds one
weight set per view and rolls every view of every sample in the SAME persistent launch (the views
are extra rows of the item space, ``csrc/vf_engine.hip`` ``build_schedule``); the per-task scores
come back camera-major, which is the order ``PixelCostController`` stacks them in
(``pixel_cost_controller.py:138-149``), in one ``[M, 1 + ncam*ndesig]`` all-gather.

This class only fixes the construction conventions of a multi-view model: ``ncam`` from the
hyper-parameters (default 2), ``model_path`` either a list of per-view directories or a directory
with ``view0/``, ``view1/`` ... inside, random-init seeds ``seed + view``.
"""
import numpy as np

from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation


class MultiViewHipPredictor(HipVPredEvaluation):
    def __init__(self, model_path, hparams, n_gpus=1, first_gpu=0):
        hp = dict(hparams)
        hp.setdefault('ncam', 2)
        super(MultiViewHipPredictor, self).__init__(model_path, hp, n_gpus=n_gpus, first_gpu=first_gpu)

    @staticmethod
    def view_context(context, c):
        """The single-view slice of a multi-view context (what one view's network sees)."""
        return {'context_frames': np.asarray(context['context_frames'])[:, c:c + 1],
                'context_pixel_distributions': np.asarray(context['context_pixel_distributions'])[:, c:c + 1],
                'context_actions': context['context_actions'], 'context_states': context['context_states']}
