"""Time-correlated noise proposal (reference ``samplers/correlated_noise.py:10-80``).

Candidates are an AR(1) filter over i.i.d. Gaussian noise, ``a_i = beta_0 * n_i + beta_1 * a_{i-1}``, and the
refit is a soft (exponentially weighted) mean of the elites instead of a Gaussian fit.  This is what the
RoboNet-era experiment files use (``experiments/robonet/franka/franka.py:54``).  The draw order from the global
NumPy generator, the order of the floating-point operations and two quirks of the reference are part of the
contract (``tests/golden/sampler.*`` and ``act.*`` were minted from the reference and are reproduced bit for bit):

* the filter's first step has no predecessor and wraps around to the LAST, still unfiltered, noise step
  (``final_actions[:, -1]`` at ``i = 0``, reference ``:33-35``);
* with ``smooth_across_last_action`` the reference reads ``self._hp._chosen_actions`` (``:33``), an attribute that
  does not exist; the executed-action history lives on the sampler, which is what is used here.
"""
import numpy as np

from .cem_sampler import CEMSampler

DEFAULTS = (('nactions', 15), ('initial_std', [0.05, 0.05, 0.2, np.pi / 10]), ('mean_bias', None), ('kappa', 1),
            ('beta_0', 0.5), ('beta_1', 0.5), ('smooth_across_last_action', False), ('refit_cov', False))


class CorrelatedNoiseSampler(CEMSampler):
    def __init__(self, hp, adim, sdim, **kwargs):
        self._hp = hp
        self._sdim = sdim
        self._adim = len(hp.initial_std)        # the std list, not the agent, fixes the action dimension
        self._chosen_actions, self._best_action_plans = [], []

    @staticmethod
    def get_default_hparams():
        return dict(DEFAULTS)

    # ------------------------------------------------------------------ pieces of one proposal
    def _shaped_noise(self, n_samples, cov):
        """i.i.d. draws [n, nactions, adim], scaled per action dimension or coloured by a refitted covariance."""
        hp = self._hp
        shape = (n_samples, hp.nactions, self._adim)
        white = np.random.normal(size=shape)
        if hp.mean_bias is None:
            bias = np.zeros(self._adim)
        else:
            bias = hp.mean_bias
            print('mean bias', bias)
        if cov is not None:
            return np.matmul(white.reshape((n_samples, -1)), cov).reshape(shape)
        return white * np.array(hp.initial_std).reshape((1, 1, -1)) + np.asarray(bias)[None, None]

    def _first_predecessor(self, filtered):
        """What step 0 of the filter is smoothed against (see the module docstring)."""
        if self._hp.smooth_across_last_action and len(self._chosen_actions):
            return self._chosen_actions[-1][None]
        return filtered[:, -1, :]

    def _ar1(self, noise):
        b0, b1 = self._hp.beta_0, self._hp.beta_1
        out = noise.copy()
        prev = self._first_predecessor(out)
        for i in range(self._hp.nactions):
            out[:, i, :] = b0 * noise[:, i, :] + b1 * prev
            prev = out[:, i, :]
        return out

    def _sample_noise(self, n_samples, cov=None):
        return self._ar1(self._shaped_noise(n_samples, cov))

    # ------------------------------------------------------------------ CEMSampler interface
    def sample_initial_actions(self, t, n_samples, current_state):
        return self._sample_noise(n_samples)

    def sample_next_actions(self, n_samples, best_actions, scores):
        gain = -scores
        soft = np.exp(self._hp.kappa * (gain - np.max(gain)))
        centre = np.sum(best_actions * soft[:, None, None], 0) / (np.sum(soft) + 1e-4)
        cov = None
        if self._hp.refit_cov:
            cov = np.cov(np.transpose(best_actions.reshape(best_actions.shape[0], -1)))
        return self._sample_noise(n_samples, cov) + centre.reshape((1, best_actions.shape[1], self._adim))
