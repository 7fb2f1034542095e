"""A deterministic stand-in video predictor with the ``VPredEvaluation`` duck-type.

Used on BOTH sides of the golden fixtures: ``tools/make_golden.py`` plugs it into the
stub-imported reference controller, the tests plug it into this repo's controller.  Its
outputs are a pure NumPy function of (context, actions), so the two runs see identical
distributions without storing them.
"""
import numpy as np


def make_fake_predictor_class(T, height, width, ncam=1, n_context=2):
    class FakeVPredEvaluation(object):
        wants_agent_params = False
        n_context_default = n_context
        calls = []

        def __init__(self, model_path, hparams, n_gpus=1, first_gpu=0):
            self.model_path = model_path
            self.hparams = dict(hparams)
            self.n_gpus, self.first_gpu = n_gpus, first_gpu
            self.n_context = n_context
            self.sequence_length = T + n_context
            self.restored = False

        def restore(self):
            self.restored = True

        def __call__(self, context, inputs):
            actions = np.asarray(inputs['actions'], dtype=np.float64)
            M, Tp = actions.shape[:2]
            assert Tp == T
            ctx = np.asarray(context['context_pixel_distributions'])
            ndesig = ctx.shape[-1]
            rr = np.arange(height, dtype=np.float64)[:, None]
            cc = np.arange(width, dtype=np.float64)[None, :]
            distrib = np.zeros((M, T, ncam, height, width, ndesig), dtype=np.float32)
            path = np.cumsum(actions[:, :, :2], axis=1) * 40.0
            for c in range(ncam):
                for p in range(ndesig):
                    start = np.unravel_index(np.argmax(ctx[-1, c, :, :, p]), (height, width))
                    pr = start[0] + path[:, :, 0] * (1 + c) + p
                    pc = start[1] + path[:, :, 1] - c
                    d2 = (rr[None, None] - pr[:, :, None, None]) ** 2 + (cc[None, None] - pc[:, :, None, None]) ** 2
                    distrib[:, :, c, :, :, p] = (np.exp(-d2 / 18.0) + 1e-3).astype(np.float32)
            frames = np.zeros((M, T, ncam, height, width, 3), dtype=np.float32)
            type(self).calls.append({'n_frames': len(context['context_frames']),
                                     'n_ctx_actions': len(context['context_actions'])})
            return {'predicted_frames': frames, 'predicted_pixel_distributions': distrib}

    return FakeVPredEvaluation
