"""Cross-entropy-method planner: propose action sequences, score them, refit on the best.

API-compatible with the reference's ``visual_mpc/policy/cem_controllers/cem_base_controller.py``
(class ``CEMBaseController`` :7; hyper-parameter defaults :42-64; sampler-default merge :66-76;
``perform_CEM`` :85-116; ``act`` :127-169) - same constructor, hyper-parameter names and defaults,
``plan_stat`` keys and replan schedule, so reference experiment files drive it unchanged (pinned
by tests/test_host_golden.py against traces of the reference itself).

Subclasses implement ``evaluate_rollouts(actions, cem_itr) -> scores[M]`` (lower is better).  All
math here is small float64 host work.  The elites are ``argsort(scores)[:K]`` on the host; in the
multi-GPU build the predictor has already all-gathered the scores, so every rank picks the same.
"""
import numpy as np

from visual_foresight_amd.policy.policy import Policy
from visual_foresight_amd.utils.logger import Logger
from .samplers import GaussianCEMSampler

# (name, default) in registration order
CEM_HPARAMS = (
    ('append_action', None),                        # constant tail appended to every sampled action
    ('verbose', True),
    ('verbose_every_iter', False),
    ('logging_dir', ''),
    ('hard_coded_start_action', None),
    ('context_action_weight', [0.5, 0.5, 0.05, 1]),  # scale of the random actions before planning starts
    ('zeros_for_start_frames', True),
    ('replan_interval', 0),                          # 0: plan at every step
    ('sampler', GaussianCEMSampler),
    ('T', 15),                                       # planning horizon
    ('iterations', 3),
    ('num_samples', 200),
    ('selection_frac', 0.),                          # elite fraction; 0 -> minimum_selection
    ('start_planning', 0),
    ('minimum_selection', 10),
)


class CEMBaseController(Policy):
    def __init__(self, ag_params, policyparams):
        self._hp = self._default_hparams()
        self._override_defaults(policyparams)
        self.agentparams = ag_params
        self._adim, self._sdim = ag_params['adim'], ag_params['sdim']
        assert self._hp.minimum_selection > 0, "must take at least 1 sample for refitting"

        if self._hp.logging_dir:
            log_name = 'cem{}log.txt'.format(ag_params['gpu_id'])
            self._logger = Logger(self._hp.logging_dir, log_name)
        else:
            self._logger = Logger(printout=True, mute=not self._hp.verbose)
        self._logger.log('init CEM controller')

        self._n_iter = self._hp.iterations
        self._sampler = None
        self._state = None
        self._t = None
        self._t_since_replan = None
        self._best_indices = None
        self._best_actions = None

    # ------------------------------------------------------------------ configuration
    def _default_hparams(self):
        params = super(CEMBaseController, self)._default_hparams()
        for name, default in CEM_HPARAMS:
            params.add_hparam(name, default)
        return params

    def _override_defaults(self, policyparams):
        """Sampler defaults join the hyper-parameters first, then the user's overrides apply."""
        sampler_class = policyparams.get('sampler', GaussianCEMSampler)
        for name, default in sampler_class.get_default_hparams().items():
            if name in self._hp:
                print('Warning default value for {} already set!'.format(name))
                self._hp.set_hparam(name, default)
            else:
                self._hp.add_hparam(name, default)
        super(CEMBaseController, self)._override_defaults(policyparams)
        self._hp.sampler = sampler_class       # the class object itself, not a type-coerced copy

    def reset(self):
        """Fresh sampler and empty planning statistics; called before every rollout."""
        self._sampler = self._hp.sampler(self._hp, self._adim, self._sdim)
        self._t_since_replan = None
        self._best_indices, self._best_actions = None, None
        self.plan_stat = {}

    # ------------------------------------------------------------------ planning
    def _elite_count(self):
        hp = self._hp
        if not hp.selection_frac:
            return hp.minimum_selection
        return max(int(hp.selection_frac * hp.num_samples), hp.minimum_selection)

    def _append_constant(self, actions):
        tail = np.array(self._hp.append_action)[None, None]
        tail = np.tile(tail, [self._hp.num_samples, actions.shape[1], 1])
        return np.concatenate((actions, tail), axis=-1)

    def perform_CEM(self, state):
        hp = self._hp
        self._logger.log('starting cem at t{}...'.format(self._t))
        n_elite = self._elite_count()
        n_tail = len(hp.append_action) if hp.append_action else 0

        candidates = self._sampler.sample_initial_actions(self._t, hp.num_samples, state[-1])
        for itr in range(self._n_iter):
            if n_tail:
                candidates = self._append_constant(candidates)
            self._logger.log('iteration: ', itr)

            scores = self.evaluate_rollouts(candidates, itr)
            assert scores.shape == (candidates.shape[0],), "score shape should be (n_actions,)"
            self.plan_stat['scores_itr{}'.format(itr)] = scores

            self._best_indices = scores.argsort()[:n_elite]
            self._best_actions = candidates[self._best_indices]

            if itr + 1 < self._n_iter:      # refit on the elites (without the appended constants)
                elites = self._best_actions.copy()
                if n_tail:
                    elites = elites[:, :, :-n_tail]
                candidates = self._sampler.sample_next_actions(
                    hp.num_samples, elites, scores[self._best_indices].copy())
        self._t_since_replan = 0

    def evaluate_rollouts(self, actions, cem_itr):
        raise NotImplementedError

    def _verbose_condition(self, cem_itr):
        if not self._hp.verbose:
            return False
        return bool(self._hp.verbose_every_iter or cem_itr == self._n_iter - 1)

    # ------------------------------------------------------------------ acting
    def _action_before_planning(self, t, state):
        """What to execute while t < start_planning: zeros, a fixed action, or weighted noise."""
        hp = self._hp
        if hp.zeros_for_start_frames:
            assert hp.hard_coded_start_action is None
            return np.zeros(self.agentparams['adim'])
        if hp.hard_coded_start_action:
            return np.array(hp.hard_coded_start_action)
        throwaway = hp.sampler(hp, self._adim, self._sdim)
        action = throwaway.sample_initial_actions(t, 1, state[-1])[0, 0] * hp.context_action_weight
        if hp.append_action:
            action = np.concatenate((action, hp.append_action), axis=0)
        return action

    def _action_from_plan(self, state):
        """Replan when due, then read the next action off the best plan."""
        interval = self._hp.replan_interval
        due = (not interval) or self._t_since_replan is None or self._t_since_replan + 1 >= interval
        if due:
            self.perform_CEM(state)
        else:
            self._t_since_replan += 1
        return self._best_actions[0, self._t_since_replan]

    def act(self, t=None, i_tr=None, state=None):
        """One control step -> ``{'actions': [adim], 'plan_stat': {...}}``."""
        self._state, self.i_tr, self._t = state, i_tr, t
        if t < self._hp.start_planning:
            action = self._action_before_planning(t, state)
        else:
            action = self._action_from_plan(state)
        assert action.shape == (self.agentparams['adim'],), "action shape does not match adim!"
        self._logger.log('time {}, action - {}'.format(t, action))

        remaining = None
        if self._best_actions is not None:      # tails of the elite plans, for samplers that reuse them
            first_unused = min(self._t_since_replan + 1, self._hp.T - 1)
            remaining = self._best_actions[:, first_unused:]
        self._sampler.log_best_action(action, remaining)
        return {'actions': action, 'plan_stat': self.plan_stat}
