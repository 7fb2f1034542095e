"""Latent-conditioned (stochastic) rollouts on top of the HIP predictor.

BASELINE config 5 asks for a stochastic predictor evaluated with several latent draws per action
sequence ("n_latent = 5 z-draws ~ N(0, I) per action ... latent draws fold into
the sample axis; per-action cost = mean over its draws").  The SAVP architecture is external to
the reference (``visual_mpc/video_prediction/vpred_model_interface.py:52-58`` only instantiates
it); the generator this repo implements for it is specified in ``savp_arch.py`` (``arch='savp'``, the
default here: four scales, first-frame compositing, per-step latent; parity unpinned) and
``arch='cdna'`` conditions the plain CDNA network of ``cdna_arch.py`` the same way.  In both, the latent
``z_t`` enters exactly where the action does (tiled and concatenated at the bottleneck, enc3), i.e. the
engine is built for ``adim + zdim`` "action" channels.

Every action sequence is rolled ``n_latent`` times with common random numbers (draw ``d`` uses the
same ``z[d, t]`` for every action, redrawn per planning call from ``latent_seed``), which keeps the
comparison between candidates low-variance; the reference's vestigial hook repeats each action
``stochastic_planning[0]`` times (``samplers/gaussian_sampler.py:140-141``).  The draws of one
action are consecutive rows of the rolled batch, the engine averages their costs on the device
(``vf_config.n_draws``), so one score row per ACTION leaves the GPU and the all-gather and the
propagation fetch work exactly as for the deterministic predictor (sharded by action, owner rank
exports, all-reduce).
"""
import numpy as np

from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation


class StochasticHipPredictor(HipVPredEvaluation):
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
        self._calls = 0
        self._z = None
        inner = dict(hp, adim=self.adim + self.zdim, n_draws=self.n_latent, arch=hp.get('arch', 'savp'), zdim=self.zdim)
        super(StochasticHipPredictor, self).__init__(model_path, inner, n_gpus=n_gpus, first_gpu=first_gpu)

    def draw_latents(self, T):
        """z[n_latent, T, zdim] for this planning call (deterministic in latent_seed and call count)."""
        rs = np.random.RandomState(self.latent_seed + self._calls)
        return rs.normal(0.0, 1.0, (self.n_latent, T, self.zdim))

    def _adim_in(self):
        return self.adim

    def _prepare(self, context, actions):
        """Fold the latent draws into the sample axis, draw-minor: row ``a * n_latent + d`` = action a, draw d."""
        actions = np.asarray(actions, dtype=np.float64)
        n, T = actions.shape[:2]
        z = self._z if self._z is not None else self.draw_latents(T)
        tiled = np.repeat(actions, self.n_latent, axis=0)                                   # [n*nl, T, adim]
        zz = np.tile(z, (n, 1, 1))
        ctx_actions = np.asarray(context['context_actions'], dtype=np.float64).reshape(-1, self.adim)
        ctx = dict(context, context_actions=np.concatenate(
            [ctx_actions, np.zeros((ctx_actions.shape[0], self.zdim))], axis=1))
        return ctx, np.concatenate([tiled, zz], axis=2)

    def score(self, context, inputs, goal_pix, finalweight=10., only_take_first_view=False, task_weights=None):
        T = np.asarray(inputs['actions']).shape[1]
        self._z = self.draw_latents(T)          # one set of draws per scoring call, shared by every rank
        self._calls += 1
        try:
            return super(StochasticHipPredictor, self).score(
                context, inputs, goal_pix, finalweight=finalweight, only_take_first_view=only_take_first_view,
                task_weights=task_weights)
        finally:
            self._z = None
