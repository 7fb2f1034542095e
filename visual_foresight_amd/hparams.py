"""HParams work-alike for the policy / controller configuration surface.

The reference builds every controller's configuration from
``tensorflow.contrib.training.HParams`` objects (``visual_mpc/policy/policy.py:4,65-66``;
``visual_mpc/policy/cem_controllers/cem_base_controller.py:42-76``).  TensorFlow is not
part of this stack, so this module provides the small subset of that class the
controllers rely on:

* ``add_hparam(name, value)``  - register a new parameter, remembering its type
* ``set_hparam(name, value)``  - type-checked override (``KeyError`` for unknown names)
* ``name in hp`` / ``hp.get(name, default)`` / ``hp.values()`` / attribute access
* plain ``setattr`` bypasses the type check (the reference uses this for ``None`` defaults,
  ``policy.py:60-61``)

Type rules follow the TF implementation: the type of a parameter is the type of its
default (for list/tuple defaults: the type of element 0 and "is a list"); a list may not be
assigned to a scalar parameter and vice versa; bools only accept bools; ints accept only
integral values; floats accept any real number (and are stored as float); everything else
is passed through the parameter's type constructor when that is safe.
"""
import numbers

_RESERVED = ('_hparam_types',)


class HParams(object):
    def __init__(self, **kwargs):
        object.__setattr__(self, '_hparam_types', {})
        for name, value in kwargs.items():
            self.add_hparam(name, value)

    # ------------------------------------------------------------------ registration
    def add_hparam(self, name, value):
        if name in self._hparam_types:
            raise ValueError('Hyperparameter name already exists: %s' % name)
        if name in _RESERVED or hasattr(self, name):
            raise ValueError('Hyperparameter name is reserved: %s' % name)
        if isinstance(value, (list, tuple)):
            if not value:
                raise ValueError('Multi-valued hyperparameters cannot be empty: %s' % name)
            self._hparam_types[name] = (type(value[0]), True)
        else:
            self._hparam_types[name] = (type(value), False)
        object.__setattr__(self, name, value)

    def del_hparam(self, name):
        if name in self._hparam_types:
            delattr(self, name)
            del self._hparam_types[name]

    # ------------------------------------------------------------------ overrides
    @staticmethod
    def _coerce(name, param_type, value):
        """Cast ``value`` to ``param_type`` if the two are compatible, else ValueError: the four refusals of TF 1.6's
        ``_cast_to_type_if_compatible`` (``tensorflow/contrib/training/python/training/hparam.py``) in their order.
        One deliberate departure: TF ends with an unconditional ``param_type(value)``, which turns a class-valued
        parameter into its metaclass (the reference repairs ``sampler`` by hand for that reason,
        ``cem_base_controller.py:76``, and never overrides ``predictor_class``); here values of non-scalar parameter
        types (classes, callables, dicts, arrays) are stored as given - the predictor plug-in seam
        (``predictor_class=HipVPredEvaluation``) depends on it.  tests/test_hparams_tf_rules.py holds the table."""
        def fail():
            raise ValueError("Could not cast hparam '%s' of type '%s' from value %r"
                             % (name, param_type, value))

        if issubclass(param_type, type(None)):      # a None default carries no type
            return value
        if issubclass(param_type, (str, bytes)) and not isinstance(value, (str, bytes)):
            fail()
        # bools are never mixed with anything else
        if issubclass(param_type, bool) != isinstance(value, bool):
            fail()
        if issubclass(param_type, numbers.Integral) and not isinstance(value, numbers.Integral):
            fail()
        if issubclass(param_type, numbers.Number) and not isinstance(value, numbers.Number):
            fail()
        if issubclass(param_type, (bool, str, bytes, numbers.Number)):
            return param_type(value)
        # classes, callables, arrays ...: store as given
        return value

    def set_hparam(self, name, value):
        param_type, is_list = self._hparam_types[name]      # KeyError for unknown names
        if isinstance(value, list):
            if not is_list:
                raise ValueError('Must not pass a list for single-valued parameter: %s' % name)
            object.__setattr__(self, name, [self._coerce(name, param_type, v) for v in value])
        else:
            if is_list:
                raise ValueError('Must pass a list for multi-valued parameter: %s.' % name)
            object.__setattr__(self, name, self._coerce(name, param_type, value))

    def override_from_dict(self, values_dict):
        for name, value in values_dict.items():
            self.set_hparam(name, value)
        return self

    # ------------------------------------------------------------------ queries
    def __contains__(self, key):
        return key in self._hparam_types

    def get(self, key, default=None):
        if key in self._hparam_types:
            return getattr(self, key)
        return default

    def values(self):
        return {n: getattr(self, n) for n in self._hparam_types.keys()}

    def __repr__(self):
        return 'HParams(%s)' % ', '.join('%s=%r' % kv for kv in sorted(self.values().items(),
                                                                     key=lambda kv: kv[0]))

    __str__ = __repr__
