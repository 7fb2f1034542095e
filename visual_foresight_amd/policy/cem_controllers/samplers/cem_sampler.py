"""Interface between the CEM controller and its proposal distribution.

Counterpart of the reference's ``visual_mpc/policy/cem_controllers/samplers/cem_sampler.py:7-55``.
A sampler proposes float64 action sequences ``[M, T, adim]`` - once from scratch at the start of a
planning call and once per refit - and keeps two histories the proposals may condition on: the
actions that were actually executed and the tails of the elite plans at the time.
"""
import numpy as np


class CEMSampler(object):
    def __init__(self, hp, adim, sdim, **kwargs):
        self._hp = hp
        self._adim = adim
        self.b_sdim = sdim
        self._chosen_actions = []       # one [adim] array per control step
        self._best_action_plans = []    # per control step: [K, remaining T, adim] or None

    @staticmethod
    def get_default_hparams():
        """Hyper-parameters this sampler adds to the controller's (name -> default)."""
        return {}

    def sample_initial_actions(self, t, nsamples, current_state):
        """First proposal of a planning call at time ``t`` -> ``[nsamples, T, adim]``."""
        raise NotImplementedError

    def sample_next_actions(self, n_samples, best_actions, scores):
        """Refit on ``best_actions`` (ascending ``scores``) and draw ``[n_samples, T, adim]``."""
        raise NotImplementedError

    def log_best_action(self, action, best_action_plans):
        """Remember the executed action and what is left of the elite plans after it."""
        self._best_action_plans.append(best_action_plans)
        self._chosen_actions.append(action.copy())

    @property
    def chosen_actions(self):
        """Executed actions so far as one ``[t, adim]`` array."""
        return np.array(self._chosen_actions)
