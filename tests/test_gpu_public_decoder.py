"""The CDNA predictor on the decoder widths of the PUBLIC ``prediction_model.py`` of arXiv:1605.07157 (``CdnaConfig(decoder=
'public')``, ``vf_config.arch = 0, layer_spec = 1``): ``convt2`` 96 -> 96, ``convt3`` 64 -> 64, so ``lstm7`` convolves 96 + 32
channels and the 1 x 1 heads read 64 feature channels (the compositing tile's two-round form).  A checkpoint has either these
widths or SURVEY a14's ('survey', the default); both load.  Parity unpinned (the network is not part of the reference,
``visual_mpc/video_prediction/checkpoint_matcher.py:4-39`` only matches variable names); tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import pixel_cost                                           # noqa: E402
from oracle.cdna_predictor import OracleCdna                            # noqa: E402
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights   # noqa: E402


def _predictor(H, W, T, nd, bs, seed=3, **extra):
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    hp = dict(designated_pixel_count=nd, run_batch_size=bs, adim=4, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, decoder='public', **extra)
    pred = HipVPredEvaluation('', hp)
    cfg = CdnaConfig(height=H, width=W, ndesig=nd, sequence_length=T + 2, decoder='public')
    weights = CdnaWeights.random(cfg, seed=seed, bias_scale=0.05, ln_jitter=0.1)
    pred.restore(weights)
    assert pred.cfg.decoder == 'public' and pred.cfg.tensor_shapes()['lstm7/w'] == (5, 5, 128, 128)
    return pred, weights


def _context(H, W, nd, rs, hist=3):
    desig = rs.randint(0, min(H, W), (1, nd, 2))
    return {'context_frames': rs.randint(0, 256, (hist, 1, H, W, 3)).astype(np.uint8),
            'context_actions': rs.normal(0, 0.05, (hist - 1, 4)), 'context_states': rs.normal(0, 0.1, (hist, 5)),
            'context_pixel_distributions': pixel_cost.one_hot_distrib(desig, 2, 1, H, W, nd)}


def _oracle(weights, ctx, actions):
    return OracleCdna(weights, torch.float32).rollout(ctx['context_frames'], ctx['context_actions'],
                                                      ctx['context_pixel_distributions'], ctx['context_states'], actions)


@pytest.mark.parametrize('H,W,T,M,nd', [(64, 64, 3, 5, 1), (48, 64, 2, 7, 2), (32, 32, 3, 9, 4), (64, 64, 2, 37, 1)])
def test_public_decoder_rollout_matches_oracle(H, W, T, M, nd):
    pred, weights = _predictor(H, W, T, nd, bs=M)
    rs = np.random.RandomState(H + W + T + M)
    ctx = _context(H, W, nd, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = rs.randint(-2, max(H, W) + 2, (1, nd, 2))
    scores, per_task = pred.score(ctx, {'actions': actions}, goal, finalweight=10.)
    got = pred(ctx, {'actions': actions})
    f, d, s = _oracle(weights, ctx, actions)
    assert np.abs(got['predicted_frames'] - f).max() <= 1e-5
    dmax = d.max(axis=(3, 4), keepdims=True)
    assert (np.abs(got['predicted_pixel_distributions'] - d) / dmax).max() <= 2e-5
    assert np.abs(got['predicted_states'] - s).max() <= 1e-6
    want, want_pt = pixel_cost.eval_pixel_cost(d, goal, 10.)
    np.testing.assert_allclose(scores, want, rtol=1e-5)
    np.testing.assert_allclose(per_task, want_pt, rtol=1e-5)
    assert pred.device_status() == 0


def test_public_decoder_launch_strategies_are_bit_identical_and_elites_match_oracle():
    """One CEM iteration of 48 candidates x T13 on the public table: persistent == per-layer == chunked == one queue bit for
    bit (with and without context de-duplication), and the K = 10 elite set equals the oracle's, with a margin assert."""
    H = W = 64
    T, M, K = 13, 48, 10
    rs = np.random.RandomState(17)
    ctx = _context(H, W, 1, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = np.array([[[16, 48]]])
    pred, weights = _predictor(H, W, T, 1, bs=M)
    base, _ = pred.score(ctx, {'actions': actions}, goal)
    for kw in (dict(persistent=0), dict(xcd_queues=0), dict(dedup=0), dict(run_batch_size=11)):
        hp = dict(kw)
        bs = hp.pop('run_batch_size', M)
        other, _ = _predictor(H, W, T, 1, bs=bs, **hp)
        np.testing.assert_array_equal(other.score(ctx, {'actions': actions}, goal)[0], base, err_msg=str(kw))
    _, d, _ = _oracle(weights, ctx, actions)
    want, _ = pixel_cost.eval_pixel_cost(d, goal, 10.)
    np.testing.assert_allclose(base, want, rtol=1e-5)
    gap = np.diff(np.sort(want))[K - 1]
    assert gap > 4 * np.abs(base - want).max(), 'fixture seeds give an ambiguous elite boundary'
    np.testing.assert_array_equal(np.sort(np.argsort(base)[:K]), np.sort(np.argsort(want)[:K]))


def test_a_checkpoint_of_the_other_table_is_refused(tmp_path):
    """'survey' weights do not load into a 'public' engine and vice versa (tensor shapes differ); the manifest carries the table."""
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    cfg_s = CdnaConfig(height=32, width=32, sequence_length=4)
    CdnaWeights.random(cfg_s, seed=0).save(str(tmp_path / 'survey'))
    cfg_p = CdnaConfig(height=32, width=32, sequence_length=4, decoder='public')
    CdnaWeights.random(cfg_p, seed=0).save(str(tmp_path / 'public'))
    hp = dict(designated_pixel_count=1, run_batch_size=2, image_height=32, image_width=32, sequence_length=4)
    with pytest.raises(ValueError, match='decoder'):
        HipVPredEvaluation(str(tmp_path / 'survey'), dict(hp, decoder='public')).restore()
    with pytest.raises(ValueError, match='decoder'):
        HipVPredEvaluation(str(tmp_path / 'public'), hp).restore()
    HipVPredEvaluation(str(tmp_path / 'public'), dict(hp, decoder='public')).restore()
