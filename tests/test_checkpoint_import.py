"""Checkpoint import (reference ``checkpoint_matcher.py:4-39``, ``setup_predictor.py:130-145``): suffix
matching, layout conversion and round trip on named arrays."""
import os

import numpy as np
import pytest

from visual_foresight_amd.video_prediction import checkpoint_import as ci
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights, tensor_shapes


def test_every_tensor_has_a_distinct_tf_name():
    cfg = CdnaConfig(height=32, width=32)
    names = [ci.tf_name(n)[0] for n in tensor_shapes(cfg)]
    assert len(set(names)) == len(names)
    assert ci.tf_name('lstm3/w') == ('state3/Gates/weights', False)
    assert ci.tf_name('ln9/g') == ('layer_norm9/gamma', False)
    assert ci.tf_name('convt2/w') == ('convt2/weights', True) and ci.tf_name('convt2/b') == ('convt2/biases', False)
    assert ci.tf_name('rgb/w')[0] == 'convt4/weights' and ci.tf_name('masks/b')[0] == 'convt7/biases'


def test_suffix_matching_follows_the_reference_rule():
    avail = ['generator/model/state1/Gates/weights', 'generator/model/state11/Gates/weights', 'model/conv2/biases']
    assert ci.match_suffix('state1/Gates/weights', avail) == avail[0]          # whole parts, not substrings
    assert ci.match_suffix('conv2/biases', avail) == avail[2]
    with pytest.raises(ValueError, match='did not find variable'):
        ci.match_suffix('conv3/biases', avail)


def test_roundtrip_through_npz_and_model_dir(tmp_path):
    cfg = CdnaConfig(height=32, width=32, adim=3, sdim=3)
    w = CdnaWeights.random(cfg, seed=7, bias_scale=0.1, ln_jitter=0.1)
    named = ci.export_named_arrays(w, scope='generator/model')
    # transposed convs are stored [kh, kw, cout, cin] on the TensorFlow side
    assert named['generator/model/convt2/weights'].shape == (3, 3, 64, 96)
    assert named['generator/model/conv2/weights'].shape == (3, 3, 32, 32)
    # optimiser slots and unrelated variables in the checkpoint are ignored
    named['generator/model/state1/Gates/weights/Adam'] = np.zeros(1)
    named['global_step'] = np.zeros(())
    npz = os.path.join(str(tmp_path), 'ckpt.npz')
    np.savez(npz, **named)
    back = ci.convert_npz(npz, os.path.join(str(tmp_path), 'model'), cfg)
    loaded = CdnaWeights.load(os.path.join(str(tmp_path), 'model'), cfg)
    for k in w.tensors:
        np.testing.assert_array_equal(back.tensors[k], w.tensors[k])
        np.testing.assert_array_equal(loaded.tensors[k], w.tensors[k])
    # a wrong shape is refused with the offending names
    bad = dict(named)
    bad['generator/model/convt1/weights'] = bad['generator/model/convt1/weights'][:, :, :-1]
    with pytest.raises(ValueError, match='convt1/w'):
        ci.import_named_arrays(bad, cfg)


def test_missing_bias_of_a_normalised_conv_becomes_zeros_and_savp_is_refused():
    """TF-slim creates no bias for a conv built with normalizer_fn=layer_norm (scale1_conv1, convt3): a real
    checkpoint dump lacks those two arrays.  Any other missing variable is still an error."""
    import pytest
    from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights
    from visual_foresight_amd.video_prediction.checkpoint_import import export_named_arrays, import_named_arrays
    from visual_foresight_amd.video_prediction.savp_arch import SavpConfig
    cfg = CdnaConfig(height=32, width=32)
    w = CdnaWeights.random(cfg, seed=2, bias_scale=0.05)
    arrays = export_named_arrays(w)
    dump = {k: v for k, v in arrays.items() if k not in ('model/scale1_conv1/biases', 'model/convt3/biases')}
    assert len(dump) == len(arrays) - 2
    said = []
    back = import_named_arrays(dump, cfg, log=said.append)
    assert len(said) == 2
    for name, arr in w.tensors.items():
        if name in ('enc0/b', 'convt3/b'):
            assert not back.tensors[name].any()
        else:
            np.testing.assert_array_equal(back.tensors[name], arr)
    del dump['model/conv2/biases']
    with pytest.raises(ValueError, match='conv2/biases'):
        import_named_arrays(dump, cfg, log=said.append)
    with pytest.raises(ValueError, match='cdna'):
        import_named_arrays(arrays, SavpConfig(height=64, width=64))


def test_public_decoder_checkpoint_roundtrip(tmp_path):
    """The TensorFlow variable names are the same for both decoder tables; the shapes decide.  A dump of the public code's
    widths (convt2 96 -> 96, convt3 64 -> 64) imports into ``CdnaConfig(decoder='public')`` and is refused by the default."""
    cfg = CdnaConfig(height=32, width=32, decoder='public')
    w = CdnaWeights.random(cfg, seed=5, bias_scale=0.1, ln_jitter=0.1)
    named = ci.export_named_arrays(w, scope='model')
    assert named['model/convt2/weights'].shape == (3, 3, 96, 96) and named['model/convt3/weights'].shape == (3, 3, 64, 64)
    assert named['model/state7/Gates/weights'].shape == (5, 5, 128, 128) and named['model/convt4/weights'].shape == (1, 1, 3, 64)
    npz = os.path.join(str(tmp_path), 'ckpt.npz')
    np.savez(npz, **named)
    back = ci.convert_npz(npz, os.path.join(str(tmp_path), 'model'), cfg)
    loaded = CdnaWeights.load(os.path.join(str(tmp_path), 'model'), cfg)
    assert loaded.cfg.decoder == 'public'
    for k in w.tensors:
        np.testing.assert_array_equal(back.tensors[k], w.tensors[k])
        np.testing.assert_array_equal(loaded.tensors[k], w.tensors[k])
    with pytest.raises(ValueError, match='shape'):
        ci.import_named_arrays(named, CdnaConfig(height=32, width=32))
    with pytest.raises(ValueError, match='decoder'):
        CdnaWeights.load(os.path.join(str(tmp_path), 'model'), CdnaConfig(height=32, width=32))
