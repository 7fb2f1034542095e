"""Latent-conditioned (stochastic) rollouts on top of the HIP predictor.

BASELINE config 5 asks for a stochastic predictor evaluated with several latent draws per action
sequence ("n_latent = 5 z-draws ~ N(0, I) per action ... latent draws fold into
the sample axis; per-action cost = mean over its draws").  The SAVP architecture is external to
the reference (``visual_mpc/video_prediction/vpred_model_interface.py:52-58`` only instantiates
it), so this module implements the conditioning scheme on the network this repo defines: the
latent ``z_t`` enters exactly where the action does (tiled and concatenated at the bottleneck,
``cdna_arch.py`` enc3), i.e. the engine is built for ``adim + zdim`` "action" channels.

Every action sequence is rolled ``n_latent`` times with common random numbers (draw ``d`` uses the
same ``z[d, t]`` for every action, redrawn per planning call from ``latent_seed``), which keeps the
comparison between candidates low-variance; the reference's vestigial hook repeats each action
``stochastic_planning[0]`` times (``samplers/gaussian_sampler.py:140-141``).  Draws of one action
stay on one rank, so the mean is local and the all-gather still moves one row per action.
"""
import numpy as np

from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
from visual_foresight_amd.video_prediction.sharding import dist_info, shard_bounds, all_gather_rows


class StochasticHipPredictor(object):
    wants_agent_params = True
    n_context_default = HipVPredEvaluation.n_context_default

    options = {}            # class-level defaults, see with_options()

    @classmethod
    def with_options(cls, **options):
        """A subclass with n_latent / zdim / latent_seed baked in, for use as ``predictor_class`` (the
        controller's hyper-parameter set stays exactly the reference's)."""
        return type(cls.__name__, (cls,), {'options': dict(cls.options, **options)})

    def __init__(self, model_path, hparams, n_gpus=1, first_gpu=0):
        hp = dict(self.options, **hparams)
        self.n_latent = int(hp.pop('n_latent', 5))
        self.zdim = int(hp.pop('zdim', 8))
        self.latent_seed = int(hp.pop('latent_seed', 0))
        self.adim = int(hp.get('adim', 4))
        batch = int(hp.get('run_batch_size', 200))
        inner = dict(hp, adim=self.adim + self.zdim, run_batch_size=batch * self.n_latent)
        self.engine = HipVPredEvaluation(model_path, inner, n_gpus=n_gpus, first_gpu=first_gpu)
        self.n_context, self.sequence_length = self.engine.n_context, self.engine.sequence_length
        self.n_cam = 1
        self._calls = 0

    def restore(self, weights=None):
        self.engine.restore(weights)
        return self

    @property
    def weights(self):
        return self.engine.weights

    def draw_latents(self, T):
        """z[n_latent, T, zdim] for this planning call (deterministic in latent_seed and call count)."""
        rs = np.random.RandomState(self.latent_seed + self._calls)
        return rs.normal(0.0, 1.0, (self.n_latent, T, self.zdim))

    def _augment(self, context, actions, z):
        M, T = actions.shape[:2]
        tiled = np.repeat(actions, self.n_latent, axis=0)                                   # [M*n, T, adim]
        zz = np.tile(z, (M, 1, 1))                                                          # draw-minor order
        ctx_actions = np.asarray(context['context_actions'], dtype=np.float64).reshape(-1, self.adim)
        ctx = dict(context, context_actions=np.concatenate(
            [ctx_actions, np.zeros((ctx_actions.shape[0], self.zdim))], axis=1))
        return ctx, np.concatenate([tiled, zz], axis=2)

    def score(self, context, inputs, goal_pix, finalweight=10., only_take_first_view=False):
        import torch
        actions = np.asarray(inputs['actions'], dtype=np.float64)
        M, T = actions.shape[:2]
        z = self.draw_latents(T)
        self._calls += 1
        rank, world = dist_info()
        lo, hi = shard_bounds(M, rank, world)            # shard ACTIONS; their draws stay together
        ctx, aug = self._augment(context, actions[lo:hi], z)
        # the engine must not shard again: temporarily score the local block as a whole
        scores, per_task = self.engine._score_local(ctx, aug, goal_pix, finalweight)
        scores = scores.reshape(hi - lo, self.n_latent).mean(axis=1)
        per_task = per_task.reshape(hi - lo, self.n_latent, -1).mean(axis=1)
        if world > 1:
            packed = torch.from_numpy(np.concatenate([scores[:, None], per_task], axis=1)).to(self.engine.device)
            full = all_gather_rows(packed, M).cpu().numpy()
            scores, per_task = full[:, 0].copy(), full[:, 1:].copy()
        return scores, per_task

    def fetch_pixel_distributions(self, sample_index):
        """Distributions of the first latent draw of the given action."""
        return self.engine.fetch_pixel_distributions_local(sample_index * self.n_latent)

    def __call__(self, context, inputs):
        actions = np.asarray(inputs['actions'], dtype=np.float64)
        z = self.draw_latents(actions.shape[1])
        ctx, aug = self._augment(context, actions, z)
        out = self.engine(ctx, {'actions': aug})
        n = self.n_latent       # report the first draw of every action, like a deterministic predictor would
        return {k: v[::n] for k, v in out.items()}
