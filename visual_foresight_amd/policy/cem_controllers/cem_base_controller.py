"""Cross-entropy-method planner loop.

API-compatible restatement of the reference's
``visual_mpc/policy/cem_controllers/cem_base_controller.py`` (``CEMBaseController`` :7,
defaults :42-64, sampler-default merge :66-76, ``perform_CEM`` :85-116, ``act`` :127-169).
Subclasses provide ``evaluate_rollouts(actions, cem_itr) -> scores[M]``; everything here is
small float64 host math.  Elite selection is ``scores.argsort()[:K]`` on the host - in the
multi-GPU build the score vector has already been all-gathered by the predictor, so every
rank selects the same elites.
"""
import numpy as np

from visual_foresight_amd.utils.logger import Logger
from visual_foresight_amd.policy.policy import Policy
from .samplers import GaussianCEMSampler


class CEMBaseController(Policy):
    """Cross Entropy Method stochastic optimizer over action sequences."""

    def __init__(self, ag_params, policyparams):
        self._hp = self._default_hparams()
        self._override_defaults(policyparams)
        self.agentparams = ag_params

        if self._hp.logging_dir:
            self._logger = Logger(self._hp.logging_dir,
                                  'cem{}log.txt'.format(self.agentparams['gpu_id']))
        else:
            self._logger = Logger(printout=True, mute=not self._hp.verbose)
        self._logger.log('init CEM controller')

        self._t_since_replan = None
        self._t = None
        self._n_iter = self._hp.iterations

        self._adim = self.agentparams['adim']
        self._sdim = self.agentparams['sdim']

        self._sampler = None
        self._best_indices, self._best_actions = None, None
        self._state = None
        assert self._hp.minimum_selection > 0, "must take at least 1 sample for refitting"

    def _default_hparams(self):
        defaults = [
            ('append_action', None),
            ('verbose', True),
            ('verbose_every_iter', False),
            ('logging_dir', ''),
            ('hard_coded_start_action', None),
            ('context_action_weight', [0.5, 0.5, 0.05, 1]),
            ('zeros_for_start_frames', True),
            ('replan_interval', 0),
            ('sampler', GaussianCEMSampler),
            ('T', 15),                      # planning horizon
            ('iterations', 3),
            ('num_samples', 200),
            ('selection_frac', 0.),         # fraction of samples refit on (0 -> minimum_selection)
            ('start_planning', 0),
            ('minimum_selection', 10),
        ]
        params = super(CEMBaseController, self)._default_hparams()
        for name, value in defaults:
            params.add_hparam(name, value)
        return params

    def _override_defaults(self, policyparams):
        # the sampler contributes its own hyper-parameters before user overrides are applied
        sampler_class = policyparams.get('sampler', GaussianCEMSampler)
        for name, value in sampler_class.get_default_hparams().items():
            if name in self._hp:
                print('Warning default value for {} already set!'.format(name))
                self._hp.set_hparam(name, value)
            else:
                self._hp.add_hparam(name, value)
        super(CEMBaseController, self)._override_defaults(policyparams)
        self._hp.sampler = sampler_class

    def reset(self):
        self._best_indices = None
        self._best_actions = None
        self._t_since_replan = None
        self._sampler = self._hp.sampler(self._hp, self._adim, self._sdim)
        self.plan_stat = {}     # planning statistics, returned from act()

    # ------------------------------------------------------------------ the CEM loop
    def _n_elites(self):
        K = self._hp.minimum_selection
        if self._hp.selection_frac:
            K = max(int(self._hp.selection_frac * self._hp.num_samples), K)
        return K

    def perform_CEM(self, state):
        hp = self._hp
        self._logger.log('starting cem at t{}...'.format(self._t))
        K = self._n_elites()
        actions = self._sampler.sample_initial_actions(self._t, hp.num_samples, state[-1])
        for itr in range(self._n_iter):
            if hp.append_action:
                tail = np.tile(np.array(hp.append_action)[None, None],
                               [hp.num_samples, actions.shape[1], 1])
                actions = np.concatenate((actions, tail), axis=-1)

            self._logger.log('iteration: ', itr)
            scores = self.evaluate_rollouts(actions, itr)
            assert scores.shape == (actions.shape[0],), "score shape should be (n_actions,)"

            self._best_indices = scores.argsort()[:K]
            self._best_actions = actions[self._best_indices]
            self.plan_stat['scores_itr{}'.format(itr)] = scores

            if itr < self._n_iter - 1:
                elites = self._best_actions.copy()
                if hp.append_action:
                    elites = elites[:, :, :-len(hp.append_action)]
                actions = self._sampler.sample_next_actions(
                    hp.num_samples, elites, scores[self._best_indices].copy())
        self._t_since_replan = 0

    def evaluate_rollouts(self, actions, cem_itr):
        raise NotImplementedError

    def _verbose_condition(self, cem_itr):
        return bool(self._hp.verbose and
                    (self._hp.verbose_every_iter or cem_itr == self._n_iter - 1))

    # ------------------------------------------------------------------ policy entry point
    def act(self, t=None, i_tr=None, state=None):
        hp = self._hp
        self._state = state
        self.i_tr = i_tr
        self._t = t

        if t < hp.start_planning:
            if hp.zeros_for_start_frames:
                assert hp.hard_coded_start_action is None
                action = np.zeros(self.agentparams['adim'])
            elif hp.hard_coded_start_action:
                action = np.array(hp.hard_coded_start_action)
            else:
                warmup_sampler = hp.sampler(hp, self._adim, self._sdim)
                action = warmup_sampler.sample_initial_actions(t, 1, state[-1])[0, 0] \
                    * hp.context_action_weight
                if hp.append_action:
                    action = np.concatenate((action, hp.append_action), axis=0)
        else:
            must_replan = (not hp.replan_interval or self._t_since_replan is None
                           or self._t_since_replan + 1 >= hp.replan_interval)
            if must_replan:
                self.perform_CEM(state)
            else:
                self._t_since_replan += 1
            action = self._best_actions[0, self._t_since_replan]

        assert action.shape == (self.agentparams['adim'],), "action shape does not match adim!"
        self._logger.log('time {}, action - {}'.format(t, action))

        if self._best_actions is not None:
            remaining = self._best_actions[:, min(self._t_since_replan + 1, hp.T - 1):]
            self._sampler.log_best_action(action, remaining)
        else:
            self._sampler.log_best_action(action, None)
        return {'actions': action, 'plan_stat': self.plan_stat}
