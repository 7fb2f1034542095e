"""GPU parity of the SAVP-class generators against their CPU oracles, both architectures: ``savp`` (vf_config.arch = 1,
SavpConfig / OracleSavp) and ``savp2`` (arch = 2: + the conditioning vector in every conv-LSTM and the published
seven-layer compositing with four CDNA kernels; Savp2Config / OracleSavp2, restated from arXiv:1804.01523 appendix A).

Parity unpinned (the SAVP source is not part of the reference, oracle/savp_predictor.py); tolerances as
in test_gpu_parity.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import pixel_cost                                           # noqa: E402
from oracle.savp_predictor import OracleSavp, OracleSavp2               # noqa: E402
from visual_foresight_amd.video_prediction.savp_arch import SavpConfig, Savp2Config, CdnaWeights   # noqa: E402

ARCHS = {'savp': (SavpConfig, OracleSavp), 'savp2': (Savp2Config, OracleSavp2)}
both_archs = pytest.mark.parametrize('arch', sorted(ARCHS))


def _predictor(H, W, T, nd, bs, adim=6, seed=3, arch='savp', **extra):
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    hp = dict(designated_pixel_count=nd, run_batch_size=bs, adim=adim, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, arch=arch, **extra)
    pred = HipVPredEvaluation('', hp)
    cfg = ARCHS[arch][0](height=H, width=W, adim=adim, ndesig=nd, sequence_length=T + 2)
    weights = CdnaWeights.random(cfg, seed=seed, bias_scale=0.05, ln_jitter=0.1)
    pred.restore(weights)
    return pred, weights


def _context(H, W, nd, adim, rs, hist=3):
    desig = rs.randint(0, min(H, W), (1, nd, 2))
    d = pixel_cost.one_hot_distrib(desig, 2, 1, H, W, nd)
    d[1] = 0.5 * d[1] + 0.5 / (H * W)        # the two context distributions differ: the first-frame term is visible
    return {'context_frames': rs.randint(0, 256, (hist, 1, H, W, 3)).astype(np.uint8),
            'context_actions': rs.normal(0, 0.05, (hist - 1, adim)),
            'context_states': rs.normal(0, 0.1, (hist, 5)),
            'context_pixel_distributions': d}


def _oracle(weights, ctx, actions, dtype=torch.float32):
    return ARCHS[weights.cfg.arch][1](weights, dtype).rollout(ctx['context_frames'], ctx['context_actions'],
                                              ctx['context_pixel_distributions'], ctx['context_states'], actions)


@pytest.mark.parametrize('arch,H,W,T,M,nd', [('savp', 64, 64, 3, 5, 1), ('savp', 48, 80, 2, 4, 2), ('savp', 128, 128, 2, 3, 1),
                                             ('savp', 32, 32, 3, 7, 4), ('savp2', 64, 64, 3, 5, 1), ('savp2', 64, 80, 2, 4, 2),
                                             ('savp2', 128, 128, 2, 3, 1), ('savp2', 64, 64, 2, 7, 4),
                                             ('savp2', 64, 64, 3, 33, 1)])
def test_savp_rollout_matches_oracle(arch, H, W, T, M, nd):
    adim = 6
    pred, weights = _predictor(H, W, T, nd, bs=M, adim=adim, arch=arch)
    rs = np.random.RandomState(H + W + T + M)
    ctx = _context(H, W, nd, adim, rs)
    actions = rs.normal(0, 0.1, (M, T, adim))
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
    np.testing.assert_allclose(got['predicted_pixel_distributions'].sum(axis=(3, 4)), 1.0, atol=5e-6)
    assert pred.device_status() == 0


@both_archs
def test_savp_launch_strategies_and_chunking_are_bit_identical(arch):
    """Persistent launch == per-layer launches == ragged chunks == one XCD queue, bit for bit; context
    de-duplication off gives the same bits too."""
    H = W = 64
    T, M = 3, 23
    rs = np.random.RandomState(5)
    ctx = _context(H, W, 1, 6, rs)
    actions = rs.normal(0, 0.1, (M, T, 6))
    goal = np.array([[[10, 50]]])
    pred, weights = _predictor(H, W, T, 1, bs=M, arch=arch)
    base, _ = pred.score(ctx, {'actions': actions}, goal)
    base_out = pred(ctx, {'actions': actions})
    for kw in (dict(persistent=0), dict(xcd_queues=0), dict(dedup=0), dict(run_batch_size=9), dict(fuse_top=0)):
        hp = dict(kw)
        bs = hp.pop('run_batch_size', M)
        other, _ = _predictor(H, W, T, 1, bs=bs, arch=arch, **hp)
        got, _ = other.score(ctx, {'actions': actions}, goal)
        np.testing.assert_array_equal(got, base, err_msg=str(kw))
        out = other(ctx, {'actions': actions})
        np.testing.assert_array_equal(out['predicted_frames'], base_out['predicted_frames'], err_msg=str(kw))
    # permutation invariance and duplicates
    perm = rs.permutation(M)
    got, _ = pred.score(ctx, {'actions': actions[perm]}, goal)
    np.testing.assert_array_equal(got, base[perm])


def test_savp_split_bf16_mode_matches_oracle():
    H = W = 64
    T, M = 3, 6
    rs = np.random.RandomState(9)
    ctx = _context(H, W, 1, 6, rs)
    actions = rs.normal(0, 0.1, (M, T, 6))
    goal = np.array([[[40, 12]]])
    pred, weights = _predictor(H, W, T, 1, bs=M, precision='bf16x6')
    scores, _ = pred.score(ctx, {'actions': actions}, goal)
    got = pred(ctx, {'actions': actions})
    f, d, _ = _oracle(weights, ctx, actions)
    assert np.abs(got['predicted_frames'] - f).max() <= 1e-5
    want, _ = pixel_cost.eval_pixel_cost(d, goal, 10.)
    np.testing.assert_allclose(scores, want, rtol=1e-5)


@both_archs
def test_savp_two_views_one_launch(arch):
    """Two views of the SAVP-class network in one engine (own weights per view, one launch), against the per-view
    oracle; persistent and per-layer launches agree bit for bit."""
    from visual_foresight_amd.video_prediction.hip_predictor import MultiViewHipPredictor
    H, W = (32, 48) if arch == 'savp' else (64, 80)
    T, M, nd, ncam, adim = 2, 5, 2, 2, 6
    hp = dict(designated_pixel_count=nd, run_batch_size=M, adim=adim, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, ncam=ncam, arch=arch)
    pred = MultiViewHipPredictor('', hp)
    cfg = ARCHS[arch][0](height=H, width=W, adim=adim, ndesig=nd, sequence_length=T + 2)
    weights = [CdnaWeights.random(cfg, seed=20 + c, bias_scale=0.05, ln_jitter=0.1) for c in range(ncam)]
    pred.restore(weights)
    rs = np.random.RandomState(31)
    desig = rs.randint(0, H, (ncam, nd, 2))
    ctx = {'context_frames': rs.randint(0, 256, (3, ncam, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (2, adim)), 'context_states': rs.normal(0, 0.1, (3, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib(desig, 2, ncam, H, W, nd)}
    actions = rs.normal(0, 0.1, (M, T, adim))
    goal = rs.randint(0, H, (ncam, nd, 2))
    want_d = np.concatenate([_oracle(weights[c], MultiViewHipPredictor.view_context(ctx, c), actions)[1]
                             for c in range(ncam)], axis=2)
    want, want_pt = pixel_cost.eval_pixel_cost(want_d, goal, 10.)
    first = None
    for persistent in (1, 0):
        pred.set_persistent(persistent)
        scores, per_task = pred.score(ctx, {'actions': actions}, goal)
        got = pred(ctx, {'actions': actions})
        dmax = want_d.max(axis=(3, 4), keepdims=True)
        assert (np.abs(got['predicted_pixel_distributions'] - want_d) / dmax).max() <= 2e-5
        np.testing.assert_allclose(per_task, want_pt, rtol=1e-5)
        np.testing.assert_allclose(scores, want, rtol=1e-5)
        if first is None:
            first = (scores, got['predicted_frames'])
        else:
            np.testing.assert_array_equal(scores, first[0])
            np.testing.assert_array_equal(got['predicted_frames'], first[1])
    assert pred.device_status() == 0


@both_archs
def test_savp_fused_top_at_128_matches_the_per_layer_launches(arch):
    """128x128: 32 transposed-conv tiles per sample wait for each other inside the fused decoder top; the
    persistent launch (fused) and the per-layer launches (never fused) agree bit for bit."""
    H = W = 128
    T, M = 2, 5
    rs = np.random.RandomState(17)
    ctx = _context(H, W, 2, 6, rs)
    actions = rs.normal(0, 0.1, (M, T, 6))
    goal = np.array([[[100, 20], [7, 77]]])
    pred, _ = _predictor(H, W, T, 2, bs=M, arch=arch)
    fused, fused_pt = pred.score(ctx, {'actions': actions}, goal)
    out_fused = pred(ctx, {'actions': actions})
    pred.set_persistent(0)
    plain, plain_pt = pred.score(ctx, {'actions': actions}, goal)
    out_plain = pred(ctx, {'actions': actions})
    np.testing.assert_array_equal(fused, plain)
    np.testing.assert_array_equal(fused_pt, plain_pt)
    np.testing.assert_array_equal(out_fused['predicted_frames'], out_plain['predicted_frames'])
    np.testing.assert_array_equal(out_fused['predicted_pixel_distributions'], out_plain['predicted_pixel_distributions'])
    assert pred.device_status() == 0


@both_archs
def test_savp_tile_plans_are_invisible_in_the_results(arch):
    """Arch 1 at 128x128 (64x64 conv-LSTM core): the same 160 sequences rolled as one batch (256-row tiles on the two
    widest conv-LSTMs), in chunks of 70 (128-row tiles) and in chunks of 30 (64- / 32-row tiles) give identical bits
    for scores and materialised predictions - the arch-1 counterpart of
    test_gpu_parity.py::test_tile_plans_are_invisible_in_the_results (the all-256-row plan is covered at 625
    sequences by test_gpu_configs.py::test_config5_full_rank_share_625_sequences_elites_match_oracle)."""
    H = W = 128
    T, M = 2, 160
    rs = np.random.RandomState(27)
    ctx = _context(H, W, 2, 6, rs)
    actions = rs.normal(0, 0.1, (M, T, 6))
    goal = np.array([[[13, 100], [90, 9]]])
    outs = []
    for bs in (M, 70, 30):
        pred, _ = _predictor(H, W, T, 2, bs=bs, arch=arch)
        s, pt = pred.score(ctx, {'actions': actions}, goal)
        got = pred(ctx, {'actions': actions[:20]})
        outs.append((s, pt, got['predicted_frames'], got['predicted_pixel_distributions']))
        assert pred.device_status() == 0
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)
