"""TEST INFRASTRUCTURE - restatement of ``tensorflow.contrib.training.HParams`` (TensorFlow 1.6).

The reference builds every controller configuration from this class (``visual_mpc/policy/policy.py:4,51-66``;
``visual_mpc/policy/cem_controllers/cem_base_controller.py:42-76``) and pins ``tensorflow-gpu==1.6.0``
(``requirements.txt:18``).  TensorFlow is not installable in this project, so the part of
``tensorflow/contrib/training/python/training/hparam.py`` the reference exercises is restated here from the published
source of that release: ``HParams.__init__ / add_hparam / set_hparam / __contains__ / get / values`` and the module-level
``_cast_to_type_if_compatible`` with its four refusals in their original order (non-string -> string, bool <-> non-bool,
non-integral -> integer, non-number -> number) followed by the UNCONDITIONAL ``param_type(value)``.

Used by ``tools/make_golden.py`` as the ``HParams`` the imported reference runs on (so the committed fixtures pin this
repository's ``visual_foresight_amd.hparams.HParams`` against an independent statement of TF's rules, not against
itself) and by ``tests/test_hparams_tf_rules.py``.  Never imported by the product.
"""
import numbers


def _cast_to_type_if_compatible(name, param_type, value):
    """hparam.py (r1.6) ``_cast_to_type_if_compatible``: cast ``value`` to ``param_type`` if compatible."""
    fail_msg = "Could not cast hparam '%s' of type '%s' from value %r" % (name, param_type, value)
    # "Some callers use None, for which we can't do any casting/checking."
    if issubclass(param_type, type(None)):
        return value
    # "Avoid converting a non-string type to a string."  (six.string_types + six.binary_type on Python 3)
    if issubclass(param_type, (str, bytes)) and not isinstance(value, (str, bytes)):
        raise ValueError(fail_msg)
    # "Avoid converting a number or string type to a boolean or vice versa."
    if issubclass(param_type, bool) != isinstance(value, bool):
        raise ValueError(fail_msg)
    # "Avoid converting float to an integer (the reverse is fine)."
    if issubclass(param_type, numbers.Integral) and not isinstance(value, numbers.Integral):
        raise ValueError(fail_msg)
    # "Avoid converting a non-numeric type to a numeric type."
    if issubclass(param_type, numbers.Number) and not isinstance(value, numbers.Number):
        raise ValueError(fail_msg)
    return param_type(value)


class HParams(object):
    def __init__(self, **kwargs):
        self._hparam_types = {}
        for name, value in kwargs.items():
            self.add_hparam(name, value)

    def add_hparam(self, name, value):
        # "'name' could be the name of a pre-existing attribute of this object.  In that case we refuse to use it"
        if getattr(self, name, None) is not None:
            raise ValueError('Hyperparameter name is reserved: %s' % name)
        if isinstance(value, (list, tuple)):
            if not value:
                raise ValueError('Multi-valued hyperparameters cannot be empty: %s' % name)
            self._hparam_types[name] = (type(value[0]), True)
        else:
            self._hparam_types[name] = (type(value), False)
        setattr(self, name, value)

    def set_hparam(self, name, value):
        param_type, is_list = self._hparam_types[name]
        if isinstance(value, list):
            if not is_list:
                raise ValueError('Must not pass a list for single-valued parameter: %s' % name)
            setattr(self, name, [_cast_to_type_if_compatible(name, param_type, v) for v in value])
        else:
            if is_list:
                raise ValueError('Must pass a list for multi-valued parameter: %s.' % name)
            setattr(self, name, _cast_to_type_if_compatible(name, param_type, value))

    def del_hparam(self, name):
        # (later TF releases; here only for the minting script's own set-up code - the reference never deletes)
        if hasattr(self, name):
            delattr(self, name)
            del self._hparam_types[name]

    def __contains__(self, key):
        return key in self._hparam_types

    def get(self, key, default=None):
        if key in self._hparam_types:
            return getattr(self, key)
        return default

    def values(self):
        return {n: getattr(self, n) for n in self._hparam_types.keys()}
