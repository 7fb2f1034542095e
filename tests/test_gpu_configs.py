"""BASELINE configs 3, 4 and 5 at full size on the GPU (configs 1 and 2 live in test_gpu_parity.py).

The CPU oracle is far too slow to roll every candidate of these shapes, so each test combines
  * oracle parity on the samples that matter (all of CEM iteration 0 where the elite set is
    asserted, a spot-check subset elsewhere), within the fp32 tolerances of test_gpu_parity.py;
  * size-independent properties over the FULL batch: bit-exact invariance to chunking and
    permutation, duplicate candidates scoring identically, planes summing to 1, run-to-run
    determinism.
Reference shapes: experiments/robonet/pixel_cost/hparams.py:31-42 (selection_frac .05),
experiments/sawyer/registration_experiments/conf.py:23-24 (ncam 2, ndesig 2),
cem_base_controller.py:53 (T = 15), register_gtruth_controller.py:88-94,175-195.
"""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import pixel_cost                                           # noqa: E402
from oracle.cdna_predictor import OracleCdna                            # noqa: E402
from tests.helpers.flow_warper import make_flow_warper                  # noqa: E402
from tests.helpers.oracle_predictor import make_oracle_predictor_class  # noqa: E402
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights   # noqa: E402


def _oracle_rollout(weights, ctx, actions, view=None):
    if view is not None:
        ctx = {'context_frames': np.asarray(ctx['context_frames'])[:, view:view + 1],
               'context_pixel_distributions': np.asarray(ctx['context_pixel_distributions'])[:, view:view + 1],
               'context_actions': ctx['context_actions'], 'context_states': ctx['context_states']}
    return OracleCdna(weights, torch.float32).rollout(ctx['context_frames'], ctx['context_actions'],
                                                      ctx['context_pixel_distributions'], ctx['context_states'],
                                                      actions)


_ORACLE_CACHE = {}


class _Recorder(object):
    """Wraps predictor.score and keeps what every CEM iteration asked for and got."""

    def __init__(self, predictor):
        self.calls = []
        self._inner = predictor.score
        predictor.score = self

    def __call__(self, context, inputs, **kw):
        scores, per_task = self._inner(context, inputs, **kw)
        self.calls.append({'context': {k: np.array(v) for k, v in context.items()},
                           'actions': np.array(inputs['actions']), 'kw': dict(kw),
                           'scores': scores.copy(), 'per_task': per_task.copy()})
        return scores, per_task


# ---------------------------------------------------------------------------------------- config 3
@pytest.mark.parametrize('trade_off', [False, True])
def test_config3_two_view_registration_planning_call(trade_off):
    """BASELINE configs[2]: 2 views x 64x64, 600 samples x horizon 13, flow-registration cost, 3 CEM iterations,
    selection_frac .05 (K = 30), RegisterGtruthController on the HIP predictor (one launch rolls both views)."""
    from visual_foresight_amd.policy.cem_controllers import RegisterGtruthController
    torch.set_num_threads(min(32, torch.get_num_threads()))     # the oracle is fastest at 32 threads (bench.py)
    H = W = 64
    ncam, M, T = 2, 600, 13
    ag = {'adim': 4, 'sdim': 5, 'image_height': H, 'image_width': W, 'ncam': ncam}
    pol = {'nactions': T, 'repeat': 1, 'rejection_sampling': False, 'verbose': False, 'num_samples': M,
           'vpred_batch_size': M, 'selection_frac': 0.05, 'designated_pixel_count': 2,
           'registration_warper': make_flow_warper(), 'register_region': True}
    if trade_off:
        pol['trade_off_reg'] = True
    rs = np.random.RandomState(11)
    frames = rs.randint(0, 256, (2, ncam, H, W, 3)).astype(np.uint8)
    states = rs.normal(0, .1, (2, 5))
    goal_image = rs.uniform(0, 1, (1, ncam, H, W, 3)).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl = RegisterGtruthController(dict(ag), pol, 0, 1)
        ctrl.reset()
        weights = ctrl.predictor.weights
        rec = _Recorder(ctrl.predictor)
        np.random.seed(3)       # a candidate set whose K / K+1 score gap is not razor thin (asserted below)
        kw = dict(goal_image=goal_image, i_tr=0, desig_pix=[[32, 32], [30, 36]], goal_pix=[[16, 48], [20, 44]])
        ctrl.act(t=0, images=frames[:1], state=states[:1], **kw)
        out = ctrl.act(t=1, images=frames, state=states, **kw)
    assert len(rec.calls) == 3 and ctrl.predictor.device_status() == 0
    tradeoff = out['plan_stat']['tradeoff']
    assert tradeoff.shape == (ncam, 2) and abs(tradeoff.sum() - 1.0) < 1e-6
    goal = ctrl._goal_pix                                              # [ncam, ndesig, 2], goal tiled per registration

    def oracle_scores(call, idx):
        # iteration 0 is the same candidate set with and without the trade-off (same seed, same registration):
        # the 2 x 600 oracle rollouts are shared by the two parametrisations of this test
        key = (call['actions'][idx].tobytes(), call['context']['context_pixel_distributions'].tobytes())
        if key not in _ORACLE_CACHE:
            per_view = [_oracle_rollout(weights[c], call['context'], call['actions'][idx], view=c)[1]
                        for c in range(ncam)]
            d = np.concatenate(per_view, axis=2)                       # [n, T, ncam, H, W, nd]
            _ORACLE_CACHE[key] = pixel_cost.eval_pixel_cost(d, goal, 10.)[1]
        pt = _ORACLE_CACHE[key]
        total = np.sum(pt * tradeoff.reshape(1, -1), axis=1) if trade_off else np.mean(pt, axis=1)
        return total, pt

    # iteration 0: every candidate through the oracle -> identical elite set (K = 30 of 600)
    c0 = rec.calls[0]
    want, want_pt = oracle_scores(c0, np.arange(M))
    np.testing.assert_allclose(c0['per_task'], want_pt, rtol=1e-5)
    np.testing.assert_allclose(c0['scores'], want, rtol=1e-5)
    K = 30
    order = np.sort(want)
    assert order[K] - order[K - 1] > 4 * np.abs(c0['scores'] - want).max(), 'ambiguous elite boundary for these seeds'
    np.testing.assert_array_equal(np.sort(c0['scores'].argsort()[:K]), np.sort(want.argsort()[:K]))
    # later iterations (sampled from the HIP elites): a spread of candidates incl. the best and the worst
    for call in rec.calls[1:]:
        rank = call['scores'].argsort()
        idx = np.unique(np.concatenate([rank[:2], rank[-2:], [7, 301]]))
        want, want_pt = oracle_scores(call, idx)
        np.testing.assert_allclose(call['per_task'][idx], want_pt, rtol=1e-5)
        np.testing.assert_allclose(call['scores'][idx], want, rtol=1e-5)
    # the chosen plan is the best candidate of the last iteration
    np.testing.assert_array_equal(out['actions'], rec.calls[2]['actions'][rec.calls[2]['scores'].argmin(), 0])


def test_config3_full_size_properties():
    """600 samples x 2 views x 2 pixels: invariances over the whole batch, both views in one launch."""
    from visual_foresight_amd.video_prediction.hip_predictor import MultiViewHipPredictor
    H = W = 64
    ncam, nd, M, T = 2, 2, 600, 13
    hp = dict(designated_pixel_count=nd, adim=4, sdim=5, image_height=H, image_width=W, sequence_length=T + 2)
    pred = MultiViewHipPredictor('', dict(hp, run_batch_size=M)).restore()
    chunked = MultiViewHipPredictor('', dict(hp, run_batch_size=256)).restore()        # 256 + 256 + 88
    rs = np.random.RandomState(5)
    ctx = {'context_frames': rs.randint(0, 256, (3, ncam, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (2, 4)), 'context_states': rs.normal(0, 0.1, (3, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib(rs.randint(0, H, (ncam, nd, 2)), 2, ncam, H, W, nd)}
    actions = rs.normal(0, 0.08, (M, T, 4))
    actions[500:] = actions[:100]
    goal = rs.randint(0, H, (ncam, nd, 2))
    w = rs.dirichlet(np.ones(ncam * nd)).reshape(ncam, nd)
    s, pt = pred.score(ctx, {'actions': actions}, goal)
    s2, pt2 = chunked.score(ctx, {'actions': actions}, goal)
    np.testing.assert_array_equal(pt, pt2)
    np.testing.assert_array_equal(s, s2)
    np.testing.assert_array_equal(s[500:], s[:100])
    perm = rs.permutation(M)
    sp, ptp = pred.score(ctx, {'actions': actions[perm]}, goal)
    np.testing.assert_array_equal(ptp, pt[perm])
    best = int(np.argmin(s))
    d = pred.fetch_pixel_distributions(best)
    assert d.shape == (T, ncam, H, W, nd) and (d >= 0).all()
    np.testing.assert_allclose(d.sum(axis=(2, 3)), 1.0, atol=5e-6)
    # trade-off weights on the device == the same weights applied to the per-task scores
    sw, ptw = pred.score(ctx, {'actions': actions}, goal, task_weights=w)
    np.testing.assert_array_equal(ptw, pt)
    np.testing.assert_allclose(sw, pt @ w.reshape(-1), rtol=2e-7)
    np.testing.assert_allclose(s, pt.mean(axis=1), rtol=2e-7)
    # views are independent networks: view 1 alone reproduces its columns bit for bit
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    single = HipVPredEvaluation('', dict(hp, run_batch_size=M)).restore(pred.weights[1])
    s1, pt1 = single.score(MultiViewHipPredictor.view_context(ctx, 1), {'actions': actions}, goal[1:2])
    np.testing.assert_array_equal(pt1, pt[:, nd:])
    assert pred.device_status() == 0


# ---------------------------------------------------------------------------------------- config 4
def test_config4_thousand_samples_horizon15_properties():
    """BASELINE configs[3] on one GPU: 1000 samples x horizon 15 x 64x64, run_batch_size 200 (five chunks)."""
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    H = W = 64
    M, T = 1000, 15
    hp = dict(designated_pixel_count=1, adim=4, sdim=5, image_height=H, image_width=W, sequence_length=T + 2)
    cfg = CdnaConfig(sequence_length=T + 2)
    weights = CdnaWeights.random(cfg, seed=4, bias_scale=0.05, ln_jitter=0.1)
    pred = HipVPredEvaluation('', dict(hp, run_batch_size=200)).restore(weights)
    shard = HipVPredEvaluation('', dict(hp, run_batch_size=125)).restore(weights)      # the 8-GPU shard size
    rs = np.random.RandomState(15)
    ctx = {'context_frames': rs.randint(0, 256, (2, 1, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (1, 4)), 'context_states': rs.normal(0, 0.1, (2, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib([[[32, 32]]], 2, 1, H, W, 1)}
    actions = rs.normal(0, 0.08, (M, T, 4))
    actions[900:] = actions[100:200]                       # duplicates that land in another chunk
    goal = np.array([[[16, 48]]])
    s, pt = pred.score(ctx, {'actions': actions}, goal)
    assert s.shape == (M,) and np.isfinite(s).all() and (s > 0).all() and (s < np.hypot(H, W)).all()
    np.testing.assert_array_equal(s[900:], s[100:200])
    np.testing.assert_array_equal(pt[:, 0], s)
    # the 8 contiguous 125-sample shards of the multi-GPU run, rolled one after the other
    parts = [shard.score(ctx, {'actions': actions[r * 125:(r + 1) * 125]}, goal)[0] for r in range(8)]
    np.testing.assert_array_equal(np.concatenate(parts), s)
    perm = rs.permutation(M)
    np.testing.assert_array_equal(pred.score(ctx, {'actions': actions[perm]}, goal)[0], s[perm])
    np.testing.assert_array_equal(pred.score(ctx, {'actions': actions}, goal)[0], s)       # run-to-run
    d = pred.fetch_pixel_distributions(999)                # the last chunk is resident
    np.testing.assert_allclose(d.sum(axis=(2, 3)), 1.0, atol=5e-6)
    idx = [0, 199, 200, 555, 999]                          # chunk borders included
    _, dd, _ = _oracle_rollout(weights, ctx, actions[idx])
    want, _ = pixel_cost.eval_pixel_cost(dd, goal, 10.)
    np.testing.assert_allclose(s[idx], want, rtol=1e-5)
    np.testing.assert_allclose(d[:, 0], dd[4, :, 0], atol=2e-5 * dd[4].max())
    assert pred.device_status() == 0


def test_config4_shard_elites_match_oracle():
    """One rank's share of configs[3]: 125 samples x horizon 15, 3 CEM iterations; elites identical to the
    oracle-driven controller in every iteration."""
    from visual_foresight_amd.policy.cem_controllers import PixelCostController
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    torch.set_num_threads(min(32, torch.get_num_threads()))     # the oracle is fastest at 32 threads (bench.py)
    ag = {'adim': 4, 'sdim': 5, 'image_height': 64, 'image_width': 64}
    base = {'nactions': 15, 'repeat': 1, 'rejection_sampling': False, 'verbose': False, 'num_samples': 125}
    factory = lambda cfg: CdnaWeights.random(cfg, seed=0)
    frames = np.random.RandomState(1).randint(0, 256, (2, 1, 64, 64, 3)).astype(np.uint8)
    states = np.random.RandomState(2).normal(0, .1, (2, 5))

    def run(predictor_class):
        with contextlib.redirect_stdout(io.StringIO()):
            ctrl = PixelCostController(dict(ag), dict(base, predictor_class=predictor_class), 0, 1)
            ctrl.reset()
            np.random.seed(0)
            ctrl.act(t=0, i_tr=0, desig_pix=[[32, 32]], goal_pix=[[16, 48]], images=frames[:1], state=states[:1])
            out = ctrl.act(t=1, i_tr=0, desig_pix=[[32, 32]], goal_pix=[[16, 48]], images=frames, state=states)
        return out, ctrl._best_indices.copy()

    ora, ora_idx = run(make_oracle_predictor_class(factory))
    hip, hip_idx = run(HipVPredEvaluation)
    for itr in range(3):
        key = 'scores_itr%d' % itr
        s_hip, s_ora = hip['plan_stat'][key], ora['plan_stat'][key]
        np.testing.assert_allclose(s_hip, s_ora, rtol=1e-5)
        gap = np.diff(np.sort(s_ora))[9]                    # margin at the K / K+1 boundary (K = 10)
        assert gap > 4 * np.abs(s_hip - s_ora).max(), 'fixture seeds give an ambiguous elite boundary'
    np.testing.assert_array_equal(hip_idx, ora_idx)
    np.testing.assert_array_equal(hip['actions'], ora['actions'])


# ---------------------------------------------------------------------------------------- config 5
def test_config5_fifth_of_a_shard_latent_draws_128():
    """A fifth of one rank's share of configs[4] (the full share - 125 actions x 5 draws - is the next test):
    25 actions x 5 latent draws x horizon 15 x 128x128 on the SAVP-class generator (savp_arch.py); the mean over
    draws is taken on the device."""
    from oracle.savp_predictor import OracleSavp
    from visual_foresight_amd.video_prediction.savp_arch import SavpConfig
    from visual_foresight_amd.video_prediction.stochastic_predictor import StochasticHipPredictor
    H = W = 128
    T, M, nl, zd = 15, 25, 5, 8
    hp = dict(designated_pixel_count=1, run_batch_size=M, adim=4, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, n_latent=nl, zdim=zd, latent_seed=5)
    pred = StochasticHipPredictor('', hp)
    assert pred.arch == 'savp'
    cfg = SavpConfig(height=H, width=W, adim=4 + zd, sdim=5, sequence_length=T + 2)
    weights = CdnaWeights.random(cfg, seed=2, bias_scale=0.05, ln_jitter=0.1)
    pred.restore(weights)
    rs = np.random.RandomState(3)
    ctx = {'context_frames': rs.randint(0, 256, (2, 1, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (1, 4)), 'context_states': rs.normal(0, 0.1, (2, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib([[[64, 64]]], 2, 1, H, W, 1)}
    actions = rs.normal(0, 0.1, (M, T, 4))
    actions[20:] = actions[:5]
    goal = np.array([[[32, 96]]])
    z = pred.draw_latents(T)
    scores, per_task = pred.score(ctx, {'actions': actions}, goal)
    assert scores.shape == (M,) and per_task.shape == (M, 1)
    np.testing.assert_array_equal(scores[20:], scores[:5])
    # oracle: the same network with z appended to every action, mean over the 5 draws, 3 actions
    idx = [0, 11, 24]
    ctx_o = dict(ctx, context_actions=np.concatenate([ctx['context_actions'], np.zeros((1, zd))], axis=1))
    aug = np.concatenate([np.repeat(actions[idx], nl, axis=0), np.tile(z, (len(idx), 1, 1))], axis=2)
    _, d, _ = OracleSavp(weights, torch.float32).rollout(
        ctx_o['context_frames'], ctx_o['context_actions'], ctx_o['context_pixel_distributions'],
        ctx_o['context_states'], aug)
    want, _ = pixel_cost.eval_pixel_cost(d, goal, 10.)
    np.testing.assert_allclose(scores[idx], want.reshape(len(idx), nl).mean(axis=1), rtol=1e-5)
    # the draws really differ (a mean over identical rollouts would hide a broken latent path)
    assert np.ptp(want.reshape(len(idx), nl), axis=1).min() > 1e-4 * want.mean()
    best = pred.fetch_pixel_distributions(int(np.argmin(scores)))
    assert best.shape == (T, 1, H, W, 1)
    np.testing.assert_allclose(best.sum(axis=(2, 3)), 1.0, atol=5e-6)
    # chunked (run_batch_size 10 actions = 50 sequences per launch): same bits
    pred2 = StochasticHipPredictor('', dict(hp, run_batch_size=10)).restore(weights)
    np.testing.assert_array_equal(pred2.score(ctx, {'actions': actions}, goal)[0], scores)
    assert pred.device_status() == 0


@pytest.mark.parametrize('arch', ['savp', 'savp2'])
def test_config5_full_rank_share_625_sequences_elites_match_oracle(arch):
    """(Both SAVP-class generators: ``savp`` and ``savp2`` - the conditioning vector in every conv-LSTM, published
    compositing - each against its own oracle.)
    One rank's REAL share of configs[4]: 125 actions x 5 latent draws = 625 sequences x horizon 15 x 128x128 in
    one launch (the batch size at which every conv-LSTM takes the 256-row tile plan), through
    ``StochasticHipPredictor``.  Chunked into 25-action launches (128- and 64-row plans): the same bits.  The first
    25 actions are a CEM sub-problem whose every sequence also goes through the CPU oracle: mean-over-draws scores
    to 1e-5 and the identical K = 10 elite set, with a margin assert at the K / K+1 boundary.
    Shapes: BASELINE.json configs[4]; latent repeats: reference samplers/gaussian_sampler.py:140-141."""
    from oracle.savp_predictor import OracleSavp, OracleSavp2
    from visual_foresight_amd.video_prediction.savp_arch import SavpConfig, Savp2Config
    from visual_foresight_amd.video_prediction.stochastic_predictor import StochasticHipPredictor
    torch.set_num_threads(min(32, torch.get_num_threads()))     # the oracle is fastest at 32 threads (bench.py)
    cfg_cls, oracle_cls = {'savp': (SavpConfig, OracleSavp), 'savp2': (Savp2Config, OracleSavp2)}[arch]
    H = W = 128
    T, M, nl, zd, sub = 15, 125, 5, 8, 25
    hp = dict(designated_pixel_count=1, run_batch_size=M, adim=4, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, n_latent=nl, zdim=zd, latent_seed=9, arch=arch)
    pred = StochasticHipPredictor('', hp)
    cfg = cfg_cls(height=H, width=W, adim=4 + zd, sdim=5, sequence_length=T + 2)
    weights = CdnaWeights.random(cfg, seed=6, bias_scale=0.05, ln_jitter=0.1)
    pred.restore(weights)
    rs = np.random.RandomState(13)
    ctx = {'context_frames': rs.randint(0, 256, (2, 1, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (1, 4)), 'context_states': rs.normal(0, 0.1, (2, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib([[[64, 64]]], 2, 1, H, W, 1)}
    actions = rs.normal(0, 0.1, (M, T, 4))
    actions[100:] = actions[30:55]                         # duplicates outside the oracle's sub-problem
    goal = np.array([[[32, 96]]])
    z = pred.draw_latents(T)
    scores, per_task = pred.score(ctx, {'actions': actions}, goal)
    assert scores.shape == (M,) and np.isfinite(scores).all()
    np.testing.assert_array_equal(scores[100:], scores[30:55])
    np.testing.assert_array_equal(per_task[:, 0], scores)
    best = pred.fetch_pixel_distributions(int(np.argmin(scores)))
    np.testing.assert_allclose(best.sum(axis=(2, 3)), 1.0, atol=5e-6)
    assert (best >= 0).all()
    assert pred.device_status() == 0
    # chunked: five launches of 25 actions = 125 sequences (another tile plan), same latent draws -> same bits
    chunked = StochasticHipPredictor('', dict(hp, run_batch_size=sub)).restore(weights)
    np.testing.assert_array_equal(chunked.score(ctx, {'actions': actions}, goal)[0], scores)
    # the oracle on ALL sequences of the 25-action sub-problem
    ctx_o = dict(ctx, context_actions=np.concatenate([ctx['context_actions'], np.zeros((1, zd))], axis=1))
    aug = np.concatenate([np.repeat(actions[:sub], nl, axis=0), np.tile(z, (sub, 1, 1))], axis=2)
    ora = oracle_cls(weights, torch.float32)
    want_seq = []
    for c0 in range(0, sub * nl, 25):
        _, d, _ = ora.rollout(ctx_o['context_frames'], ctx_o['context_actions'], ctx_o['context_pixel_distributions'],
                              ctx_o['context_states'], aug[c0:c0 + 25])
        want_seq.append(pixel_cost.eval_pixel_cost(d, goal, 10.)[0])
    want = np.concatenate(want_seq).reshape(sub, nl).mean(axis=1)
    got = scores[:sub]
    np.testing.assert_allclose(got, want, rtol=1e-5)
    gap = np.diff(np.sort(want))[9]
    assert gap > 4 * np.abs(got - want).max(), 'fixture seeds give an ambiguous elite boundary'
    np.testing.assert_array_equal(np.sort(np.argsort(got)[:10]), np.sort(np.argsort(want)[:10]))


def test_config4_all_1000_samples_iteration0_elites_match_oracle():
    """configs[3] at its full size: CEM iteration 0 of the planning call - all 1000 candidates the sampler draws,
    horizon 15 - through the HIP predictor (five 200-sample launches) and through the CPU oracle: scores to 1e-5,
    identical elite set of K = 10 with a margin assert (experiments/robonet/pixel_cost/hparams.py:31-42 pattern:
    nactions = T, repeat 1, rejection_sampling False)."""
    from visual_foresight_amd.policy.cem_controllers import PixelCostController
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    torch.set_num_threads(min(32, torch.get_num_threads()))     # the oracle is fastest at 32 threads (bench.py)
    M, T = 1000, 15
    ag = {'adim': 4, 'sdim': 5, 'image_height': 64, 'image_width': 64}
    pol = {'nactions': T, 'repeat': 1, 'rejection_sampling': False, 'verbose': False, 'num_samples': M,
           'iterations': 1, 'predictor_class': HipVPredEvaluation}
    frames = np.random.RandomState(1).randint(0, 256, (2, 1, 64, 64, 3)).astype(np.uint8)
    states = np.random.RandomState(2).normal(0, .1, (2, 5))
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl = PixelCostController(dict(ag), pol, 0, 1)
        ctrl.reset()
        rec = _Recorder(ctrl.predictor)
        np.random.seed(0)
        ctrl.act(t=0, i_tr=0, desig_pix=[[32, 32]], goal_pix=[[16, 48]], images=frames[:1], state=states[:1])
        ctrl.act(t=1, i_tr=0, desig_pix=[[32, 32]], goal_pix=[[16, 48]], images=frames, state=states)
    call = rec.calls[0]
    assert call['actions'].shape == (M, T, 4) and ctrl.predictor.run_batch_size == 200
    weights = ctrl.predictor.weights
    want = []
    for c0 in range(0, M, 125):
        _, d, _ = _oracle_rollout(weights, call['context'], call['actions'][c0:c0 + 125])
        want.append(pixel_cost.eval_pixel_cost(d, np.array([[[16, 48]]]), 10.)[0])
    want = np.concatenate(want)
    got = call['scores']
    np.testing.assert_allclose(got, want, rtol=1e-5)
    gap = np.diff(np.sort(want))[9]
    assert gap > 4 * np.abs(got - want).max(), 'fixture seeds give an ambiguous elite boundary'
    np.testing.assert_array_equal(np.sort(ctrl._best_indices), np.sort(np.argsort(want)[:10]))
