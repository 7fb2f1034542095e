"""Policy base class and the agent -> policy argument binding.

Same public surface as the reference's ``visual_mpc/policy/policy.py``:

* ``get_policy_args`` (reference :9-46) - the agent never passes positional arguments to
  ``policy.act``; it looks at the *names* of ``act``'s parameters and fills each one from the
  observation dict, the per-step agent data or the loop counters.  A parameter nobody can
  supply and that has no default is an error.
* ``Policy._override_defaults`` (reference :51-63) - lays a ``policyparams`` dict over the HParams
  defaults.  Two quirks are part of the contract because existing experiment files rely on
  them: an override *equal* to the default raises ``ValueError``, and a parameter whose default
  is ``None`` is assigned without a type check.
"""
import abc
import inspect

import numpy as np

from visual_foresight_amd.hparams import HParams

_MISSING = inspect.Parameter.empty


def _lookup(name, default, obs, t, i_tr, step_data):
    """Value for one ``act`` parameter, in the reference's precedence order."""
    if name in obs:
        return obs[name]
    if step_data is not None and name in step_data:
        return step_data[name]
    specials = {'t': t, 'i_tr': i_tr, 'obs': obs, 'step_data': step_data}
    if name in specials:
        return specials[name]
    if name == 'goal_pos':
        return step_data['goal_pos']
    return default


def get_policy_args(policy, obs, t, i_tr, step_data=None):
    """kwargs for ``policy.act`` bound by parameter name (reference ``policy.py:9-46``).

    :param obs: observation dict of the agent (``images``, ``state``, ...)
    :param step_data: per-step agent data (``desig_pix``, ``goal_pix``, ``verbose_worker``, ...)
    """
    kwargs = {}
    for name, param in inspect.signature(policy.act).parameters.items():
        value = _lookup(name, param.default, obs, t, i_tr, step_data)
        if value is _MISSING:
            raise ValueError("Required Policy Param {} not set in agent".format(name))
        kwargs[name] = value
    return kwargs


class Policy(abc.ABC):
    """``act`` returns a dict whose ``'actions'`` entry is the action for this time step."""

    def _default_hparams(self):
        return HParams()

    def _override_defaults(self, policyparams):
        hp = self._hp
        for name, value in policyparams.items():
            if name == 'type':                  # names the policy class, not a hyper-parameter
                continue
            print('overriding param {} to value {}'.format(name, value))
            current = getattr(hp, name)         # AttributeError for names that are not hyper-parameters
            if np.all(value == current):
                raise ValueError("attribute is {} is identical to default value!!".format(name))
            if name in hp and hp.get(name) is None:
                setattr(hp, name, value)        # a None default carries no type to check against
            else:
                hp.set_hparam(name, value)

    @abc.abstractmethod
    def act(self, *args):
        raise NotImplementedError("Must be implemented in subclass.")

    def reset(self):
        pass


class DummyPolicy(object):
    """Accepts the standard constructor arguments and does nothing."""

    def __init__(self, ag_params, policyparams, gpu_id, ngpu):
        pass

    def act(self, *args):
        pass

    def reset(self):
        pass


class NullPolicy(Policy):
    """Zero action at every step (reference ``policy.py:97-118``)."""

    def __init__(self, ag_params, policyparams, gpu_id, ngpu):
        self._adim = ag_params['adim']
        self._hp = self._default_hparams()
        self._override_defaults(policyparams)

    def _default_hparams(self):
        params = super(NullPolicy, self)._default_hparams()
        params.add_hparam('wait_for_user', False)
        return params

    def act(self):
        if self._hp.wait_for_user:
            input('NullPolicy: press enter to continue')
        return {'actions': np.zeros(self._adim)}
