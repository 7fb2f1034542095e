"""Import network weights from a checkpoint given as named arrays (``.npz``) -> ``CdnaWeights``.

Replaces the restore path of the legacy boundary: ``variable_checkpoint_matcher``
(``visual_mpc/video_prediction/checkpoint_matcher.py:4-39``: every model variable is matched to
the checkpoint variable whose name ENDS with the model variable's name parts) and the
newest-iteration glob / ``saver.restore`` of ``setup_predictor.py:12-28,130-145``.

A TensorFlow-1 checkpoint cannot be read here (no TensorFlow); the supported input is the
``.npz`` a user dumps from it, one entry per variable under its checkpoint name, e.g.::

    reader = tf.train.NewCheckpointReader(path)
    np.savez('ckpt.npz', **{n: reader.get_tensor(n) for n in reader.get_variable_to_shape_map()})

``TF_NAMES`` maps this repo's tensor names to the variable names of the public CDNA
``prediction_model`` (slim scopes ``scale1_conv1``, ``layer_norm1..9``, ``state1..7``,
``conv2..4``, ``convt1..4``, ``convt7``, ``cdna_params``, ``state_pred``).  Matching is by name
suffix, exactly like the reference's matcher, so arbitrary outer scopes (``model/``,
``generator/``, optimiser slots stripped by the caller) do not matter.  Layout conversions:
conv kernels are ``[kh, kw, cin, cout]`` on both sides; ``conv2d_transpose`` kernels are
``[kh, kw, cout, cin]`` in TensorFlow and get their last two axes swapped; FC weights are
``[in, out]`` on both sides.  UNVALIDATED against a real checkpoint (none exists in this
project); what is tested is the matching rule, the layout conversions and the round trip
``CdnaWeights -> export_named_arrays -> import_named_arrays``.
"""
from collections import OrderedDict

import numpy as np

from visual_foresight_amd.video_prediction.cdna_arch import CdnaWeights

_CONV = {'enc0': 'scale1_conv1', 'enc1': 'conv2', 'enc2': 'conv3', 'enc3': 'conv4'}
_CONVT = {'convt1': 'convt1', 'convt2': 'convt2', 'convt3': 'convt3', 'rgb': 'convt4', 'masks': 'convt7'}
_FC = {'cdna': 'cdna_params', 'state': 'state_pred'}


def tf_name(name):
    """This repo's tensor name (``'lstm3/w'``) -> (TensorFlow variable name suffix, is_transposed_conv)."""
    layer, kind = name.split('/')
    if layer.startswith('lstm'):
        return 'state%s/Gates/%s' % (layer[4:], 'weights' if kind == 'w' else 'biases'), False
    if layer.startswith('ln'):
        return 'layer_norm%s/%s' % (layer[2:], 'gamma' if kind == 'g' else 'beta'), False
    for table, transposed in ((_CONV, False), (_CONVT, True), (_FC, False)):
        if layer in table:
            return '%s/%s' % (table[layer], 'weights' if kind == 'w' else 'biases'), transposed and kind == 'w'
    raise KeyError(name)




def match_suffix(wanted, available):
    """The reference's matching rule (``checkpoint_matcher.py:27-36``): the first checkpoint name whose
    trailing ``/``-separated parts equal the wanted name's parts."""
    parts = wanted.split('/')
    for ck in available:
        if ck.split('/')[-len(parts):] == parts:
            return ck
    raise ValueError('did not find variable %s' % wanted)


# convolutions that are followed by a LayerNorm: TF-slim builds those WITHOUT a bias variable
# (``use_bias = not normalizer_fn`` when ``normalizer_fn=layer_norm``), so a real checkpoint dump has no
# ``<scope>/biases`` for them; the LayerNorm's beta plays that role
_BIAS_OPTIONAL = ('enc0/b', 'convt3/b')


def import_named_arrays(arrays, cfg, log=None):
    """``arrays``: mapping checkpoint-variable-name -> ndarray (e.g. an open ``np.load(...npz)``).

    Only the CDNA architecture (``CdnaConfig``) has a TensorFlow name table; a ``SavpConfig`` is refused.
    A missing bias of a LayerNorm-followed convolution (``_BIAS_OPTIONAL``) is filled with zeros and reported
    through ``log`` (a callable taking one string; default: print)."""
    if getattr(cfg, 'arch_id', 0) != 0:
        raise ValueError("only arch 'cdna' checkpoints can be imported (no TensorFlow name table for %s)"
                         % type(cfg).__name__)
    log = log or print
    names = list(arrays.keys())
    tensors = OrderedDict()
    for name, shape in cfg.tensor_shapes().items():
        suffix, transposed = tf_name(name)
        try:
            found = match_suffix(suffix, names)
        except ValueError:
            if name not in _BIAS_OPTIONAL:
                raise
            log('checkpoint has no %s (convolution built with a normalizer): %s set to zeros' % (suffix, name))
            tensors[name] = np.zeros(shape, np.float32)
            continue
        arr = np.asarray(arrays[found], dtype=np.float32)
        if transposed:                                  # conv2d_transpose: [kh, kw, cout, cin] -> [kh, kw, cin, cout]
            arr = arr.transpose(0, 1, 3, 2)
        if tuple(arr.shape) != tuple(shape):
            raise ValueError('%s (from %s): shape %s, expected %s' % (name, suffix, arr.shape, shape))
        tensors[name] = np.ascontiguousarray(arr)
    return CdnaWeights(cfg, tensors)


def export_named_arrays(weights, scope='model'):
    """Inverse of ``import_named_arrays``: the arrays under TensorFlow-style names (for round-trip tests and
    for handing weights to a TF-side tool)."""
    out = OrderedDict()
    for name, arr in weights.tensors.items():
        suffix, transposed = tf_name(name)
        out['%s/%s' % (scope, suffix) if scope else suffix] = arr.transpose(0, 1, 3, 2) if transposed else arr
    return out


def convert_npz(npz_path, model_dir, cfg):
    """``ckpt.npz`` -> ``model_dir/manifest.json + weights.bin`` (what ``HipVPredEvaluation(model_path)`` loads)."""
    with np.load(npz_path) as data:
        weights = import_named_arrays(data, cfg)
    weights.save(model_dir)
    return weights


if __name__ == '__main__':
    import argparse
    from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0] + "  (arch 'cdna' only)")
    ap.add_argument('npz')
    ap.add_argument('model_dir')
    for k, d in (('height', 64), ('width', 64), ('adim', 4), ('sdim', 5)):
        ap.add_argument('--' + k, type=int, default=d)
    a = ap.parse_args()
    convert_npz(a.npz, a.model_dir, CdnaConfig(height=a.height, width=a.width, adim=a.adim, sdim=a.sdim))
    print('wrote', a.model_dir)
