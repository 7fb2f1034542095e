"""Sampler interface of the CEM controller (reference ``samplers/cem_sampler.py:7-55``)."""
import numpy as np


class CEMSampler(object):
    """Proposal distribution over action sequences ``[M, T, adim]``.

    A sampler also keeps the history of executed actions (``chosen_actions``) and of the
    remaining best plans, because some proposals are conditioned on what was executed.
    """

    def __init__(self, hp, adim, sdim, **kwargs):
        self._hp = hp
        self._adim, self.b_sdim = adim, sdim
        self._chosen_actions = []
        self._best_action_plans = []

    def sample_initial_actions(self, t, nsamples, current_state):
        """First proposal of a planning call -> float64 ``[nsamples, T, adim]``."""
        raise NotImplementedError

    def sample_next_actions(self, n_samples, best_actions, scores):
        """Refit on the elites (ascending cost) and draw the next proposal."""
        raise NotImplementedError

    def log_best_action(self, action, best_action_plans):
        """Record the executed action and the tails of the elite plans (ascending cost)."""
        self._chosen_actions.append(action.copy())
        self._best_action_plans.append(best_action_plans)

    @property
    def chosen_actions(self):
        return np.array(self._chosen_actions)

    @staticmethod
    def get_default_hparams():
        return {}
