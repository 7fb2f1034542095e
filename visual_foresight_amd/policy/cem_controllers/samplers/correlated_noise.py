"""Time-correlated noise proposal (reference ``samplers/correlated_noise.py:10-80``).

Actions are an AR(1) filter over i.i.d. Gaussian noise,
``a_i = beta_0 * n_i + beta_1 * a_{i-1}``, and the refit is a soft (exponentially weighted)
mean of the elites instead of a Gaussian fit.  This is what the RoboNet-era experiment files
use (``experiments/robonet/franka/franka.py:54``).
"""
import numpy as np

from .cem_sampler import CEMSampler


class CorrelatedNoiseSampler(CEMSampler):
    def __init__(self, hp, adim, sdim, **kwargs):
        self._hp = hp
        # the action dimension is defined by the std list, not by the agent
        self._adim, self._sdim = len(self._hp.initial_std), sdim
        self._chosen_actions = []
        self._best_action_plans = []

    @staticmethod
    def get_default_hparams():
        return {
            'nactions': 15,
            'initial_std': [0.05, 0.05, 0.2, np.pi / 10],
            'mean_bias': None,
            'kappa': 1,
            'beta_0': 0.5,
            'beta_1': 0.5,
            'smooth_across_last_action': False,
            'refit_cov': False,
        }

    def _sample_noise(self, n_samples, cov=None):
        hp = self._hp
        noise = np.random.normal(size=(n_samples, hp.nactions, self._adim))
        if hp.mean_bias is not None:
            mean_bias = hp.mean_bias
            print('mean bias', mean_bias)
        else:
            mean_bias = np.zeros(self._adim)

        if cov is None:
            noise = noise * np.array(hp.initial_std).reshape((1, 1, -1)) + np.asarray(mean_bias)[None, None]
        else:
            noise = np.matmul(noise.reshape((n_samples, -1)), cov).reshape(
                (n_samples, hp.nactions, self._adim))

        final_actions = noise.copy()
        for i in range(hp.nactions):
            if hp.smooth_across_last_action and i == 0 and len(self._chosen_actions):
                # NOTE: the reference reads ``self._hp._chosen_actions`` here
                # (correlated_noise.py:33), an attribute that does not exist; the executed
                # action history lives on the sampler.
                prev = self._chosen_actions[-1][None]
            else:
                # i == 0 wraps around to the (still un-filtered) last step, as in the reference
                prev = final_actions[:, i - 1, :]
            final_actions[:, i, :] = hp.beta_0 * noise[:, i, :] + hp.beta_1 * prev
        return final_actions

    def sample_initial_actions(self, t, n_samples, current_state):
        return self._sample_noise(n_samples)

    def sample_next_actions(self, n_samples, best_actions, scores):
        rewards = -scores
        weights = np.exp(self._hp.kappa * (rewards - np.max(rewards)))
        mean_act = np.sum(best_actions * weights[:, None, None], 0) / (np.sum(weights) + 1e-4)

        cov = None
        if self._hp.refit_cov:
            cov = np.cov(np.transpose(best_actions.reshape(best_actions.shape[0], -1)))
        return self._sample_noise(n_samples, cov) + mean_act.reshape((1, best_actions.shape[1], self._adim))
