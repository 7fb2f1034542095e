"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (fp32, stated per north_star): the oracle is itself fp32; measured against its
float64 variant the HIP path and the fp32 oracle sit at the same distance (~5e-7 on frames).
  frames   |err| <= 1e-5              (values in [0,1])
  distrib  |err| <= 2e-5 * plane max
  states   |err| <= 1e-6
  scores   rel err <= 1e-5, elite index set identical
"""
import contextlib
import ctypes
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import pixel_cost                                           # noqa: E402
from oracle.cdna_predictor import OracleCdna                            # noqa: E402
from tests.helpers.oracle_predictor import make_oracle_predictor_class  # noqa: E402
from visual_foresight_amd import _lib                                   # noqa: E402
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights   # noqa: E402


def _predictor(H, W, T, nd, bs, seed=3, n_context=2, precision='fp32'):
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    hp = dict(designated_pixel_count=nd, run_batch_size=bs, adim=4, sdim=5, image_height=H,
              image_width=W, sequence_length=T + n_context, n_context=n_context, precision=precision)
    pred = HipVPredEvaluation('', hp)
    cfg = CdnaConfig(height=H, width=W, ndesig=nd, sequence_length=T + n_context, n_context=n_context)
    weights = CdnaWeights.random(cfg, seed=seed, bias_scale=0.05, ln_jitter=0.1)
    pred.restore(weights)
    return pred, weights


def _context(H, W, nd, rs, hist=3):
    desig = rs.randint(0, min(H, W), (1, nd, 2))
    return {'context_frames': rs.randint(0, 256, (hist, 1, H, W, 3)).astype(np.uint8),
            'context_actions': rs.normal(0, 0.05, (hist - 1, 4)),
            'context_states': rs.normal(0, 0.1, (hist, 5)),
            'context_pixel_distributions': pixel_cost.one_hot_distrib(desig, 2, 1, H, W, nd)}


def _oracle(weights, ctx, actions, dtype=torch.float32):
    return OracleCdna(weights, dtype).rollout(ctx['context_frames'], ctx['context_actions'],
                                              ctx['context_pixel_distributions'], ctx['context_states'],
                                              actions)


@pytest.mark.parametrize('H,W,T,M,nd', [(64, 64, 3, 5, 1), (48, 64, 2, 5, 2), (32, 32, 2, 9, 1),
                                        (64, 64, 2, 3, 4), (40, 56, 2, 4, 1)])
def test_rollout_matches_oracle(H, W, T, M, nd):
    pred, weights = _predictor(H, W, T, nd, bs=M)
    rs = np.random.RandomState(H + W + T + M)
    ctx = _context(H, W, nd, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = rs.randint(-2, max(H, W) + 2, (1, nd, 2))          # goals may lie off-image
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
    # the HIP cost reduction against the reference-pinned cost applied to the HIP distributions
    own, _ = pixel_cost.eval_pixel_cost(got['predicted_pixel_distributions'], goal, 10.)
    np.testing.assert_allclose(scores, own, rtol=2e-6)
    np.testing.assert_allclose(got['predicted_pixel_distributions'].sum(axis=(3, 4)), 1.0, atol=2e-6)


def test_single_context_frame():
    pred, weights = _predictor(32, 32, 2, 1, bs=4, n_context=1)
    rs = np.random.RandomState(5)
    ctx = _context(32, 32, 1, rs, hist=1)
    ctx['context_pixel_distributions'] = ctx['context_pixel_distributions'][:1]
    ctx['context_actions'] = np.zeros((0, 4))
    actions = rs.normal(0, 0.1, (4, 2, 4))
    got = pred(ctx, {'actions': actions})
    f, d, s = _oracle(weights, ctx, actions)
    assert np.abs(got['predicted_frames'] - f).max() <= 1e-5
    assert np.abs(got['predicted_states'] - s).max() <= 1e-6


def test_chunking_permutation_and_determinism():
    H = W = 32
    T, M = 2, 7
    pred_full, _ = _predictor(H, W, T, 1, bs=M)
    pred_chunk, _ = _predictor(H, W, T, 1, bs=3)          # chunks of 3, 3, 1 (ragged tail)
    rs = np.random.RandomState(9)
    ctx = _context(H, W, 1, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    actions[5] = actions[1]                                # duplicate candidate
    goal = np.array([[[3, 20]]])
    a, _ = pred_full.score(ctx, {'actions': actions}, goal)
    b, _ = pred_chunk.score(ctx, {'actions': actions}, goal)
    c, _ = pred_full.score(ctx, {'actions': actions}, goal)
    np.testing.assert_array_equal(a, b)        # samples are independent: batch layout is invisible
    np.testing.assert_array_equal(a, c)        # run-to-run bit-identical
    assert a[5] == a[1]
    perm = rs.permutation(M)
    p, _ = pred_full.score(ctx, {'actions': actions[perm]}, goal)
    np.testing.assert_array_equal(p, a[perm])
    fa = pred_full({'context_frames': ctx['context_frames'], 'context_actions': ctx['context_actions'],
                    'context_states': ctx['context_states'],
                    'context_pixel_distributions': ctx['context_pixel_distributions']}, {'actions': actions})
    fb = pred_chunk(ctx, {'actions': actions})
    np.testing.assert_array_equal(fa['predicted_frames'], fb['predicted_frames'])


def test_tile_plans_are_invisible_in_the_results():
    """The conv-LSTM tile plan follows the batch size (64 / 128 / 256 rows per workgroup: DESIGN.md 4.1).  Exact
    LayerNorm statistics and a common K order make the choice invisible: the same 160 candidates rolled as one
    batch (256-row tiles on the two widest layers), in chunks of 70 (128-row tiles) and in chunks of 30 (64-row
    tiles) give identical bits, for scores and for the materialised predictions."""
    H = W = 64
    T, M = 3, 160
    rs = np.random.RandomState(17)
    ctx = _context(H, W, 2, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = np.array([[[3, 20], [50, 9]]])
    outs = []
    for bs in (M, 70, 30):
        pred, _ = _predictor(H, W, T, 2, bs=bs)
        s, pt = pred.score(ctx, {'actions': actions}, goal)
        got = pred(ctx, {'actions': actions[:40]})
        outs.append((s, pt, got['predicted_frames'], got['predicted_pixel_distributions']))
        assert pred.device_status() == 0
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('H,W,T,M,nd', [(64, 64, 4, 25, 1), (64, 64, 3, 60, 2), (48, 64, 3, 9, 1), (64, 64, 2, 130, 1)])
def test_yielding_and_write_through_publish_are_invisible_in_the_results(H, W, T, M, nd):
    """The two timing-only mechanisms of round 5 (``vf_set_sched_option``): the cooperative CU priority - recurrent halves
    of early-started conv-LSTM items sleeping while their CU partner is on a dependency chain, any budget - and the
    write-through publish (sc1 stores, no release fence).  Every combination, small shards (where yielding is on by
    default) and a batch beyond its threshold: the same bits for scores and materialised predictions, repeated rollouts
    (cached context) included, and a healthy device status."""
    pred, _ = _predictor(H, W, T, nd, bs=M)
    rs = np.random.RandomState(H + M)
    ctx = _context(H, W, nd, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = rs.randint(0, min(H, W), (1, nd, 2))
    outs = []
    for budget, wt in ((-1, 1), (0, 0), (0, 1), (7, 1), (120, 0), (4000, 1)):
        pred.set_sched_option('yield_budget', budget)
        pred.set_sched_option('write_through', wt)
        for rep in range(2):
            s, pt = pred.score(ctx, {'actions': actions}, goal)
            assert pred.device_status() == 0
        got = pred(ctx, {'actions': actions[:24]})
        outs.append((s, pt, got['predicted_frames'], got['predicted_pixel_distributions'], got['predicted_states']))
        pred._ctx_key = None
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)
    with pytest.raises(Exception):
        pred.set_sched_option('yield_budget', -5)


@pytest.mark.parametrize('H,W,T,M,nd', [(64, 64, 3, 37, 2), (48, 64, 3, 9, 1), (32, 32, 3, 21, 1), (64, 64, 2, 160, 1)])
def test_fused_items_are_invisible_in_the_results(H, W, T, M, nd):
    """The two fusions of the persistent schedule - decoder top + compositing, and the two convolutions of the 8 x 8
    bottleneck as one item (vf_set_fuse_top; the second since round 4: several images per row tile, 16- and 32-channel
    chunks of the first conv) - against the same schedule without them and against the oracle: same bits, cached-context
    rollouts included."""
    pred, weights = _predictor(H, W, T, nd, bs=M)
    rs = np.random.RandomState(H + M)
    ctx = _context(H, W, nd, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = rs.randint(0, min(H, W), (1, nd, 2))
    outs = []
    for fuse in (1, 0, 1):
        pred.set_fuse_top(fuse)
        for rep in range(2):
            s, pt = pred.score(ctx, {'actions': actions}, goal)
            assert pred.device_status() == 0
        got = pred(ctx, {'actions': actions[:24]})
        outs.append((s, pt, got['predicted_frames'], got['predicted_pixel_distributions'], got['predicted_states']))
        pred._ctx_key = None                    # upload the context again: the next variant computes the shared units itself
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)
    idx = [0, M // 2, M - 1]
    f, d, st = _oracle(weights, ctx, actions[idx])
    want, _ = pixel_cost.eval_pixel_cost(d, goal, 10.)
    np.testing.assert_allclose(outs[0][0][idx], want, rtol=1e-5)


def test_full_size_properties():
    """BASELINE config-2 size (200 x 13 x 64x64): size-independent properties."""
    H = W = 64
    T, M = 13, 200
    pred, weights = _predictor(H, W, T, 1, bs=M)
    rs = np.random.RandomState(21)
    ctx = _context(H, W, 1, rs)
    actions = rs.normal(0, 0.08, (M, T, 4))
    actions[150:] = actions[:50]                           # a block of duplicates
    goal = np.array([[[16, 48]]])
    s10, _ = pred.score(ctx, {'actions': actions}, goal, finalweight=10.)
    d_best = pred.fetch_pixel_distributions(int(np.argmin(s10)))
    np.testing.assert_array_equal(s10[150:], s10[:50])
    assert np.isfinite(s10).all() and (s10 > 0).all() and (s10 < np.hypot(H, W)).all()
    np.testing.assert_allclose(d_best.sum(axis=(2, 3)), 1.0, atol=5e-6)
    assert (d_best >= 0).all()
    # score(fw) * (T - 1 + fw) is affine in the final weight
    s1, _ = pred.score(ctx, {'actions': actions}, goal, finalweight=1.)
    s4, _ = pred.score(ctx, {'actions': actions}, goal, finalweight=4.)
    u = lambda s, fw: s * (T - 1 + fw)
    np.testing.assert_allclose((u(s4, 4.) - u(s1, 1.)) / 3., (u(s10, 10.) - u(s1, 1.)) / 9., rtol=2e-4)
    # spot-check 3 samples of the full-size batch against the oracle
    idx = [0, 77, 199]
    f, d, s = _oracle(weights, ctx, actions[idx])
    want, _ = pixel_cost.eval_pixel_cost(d, goal, 10.)
    np.testing.assert_allclose(s10[idx], want, rtol=1e-5)


def test_controller_elites_match_oracle_config1():
    """BASELINE configs[0]: 32 samples, horizon 5, 64x64, 1 CEM iteration; elite indices bit-exact."""
    from visual_foresight_amd.policy.cem_controllers import PixelCostController
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    ag = {'adim': 4, 'sdim': 5, 'image_height': 64, 'image_width': 64}
    base = {'num_samples': 32, 'iterations': 1, 'repeat': 1, 'rejection_sampling': False, 'verbose': False}
    factory = lambda cfg: CdnaWeights.random(cfg, seed=0)
    rs = np.random.RandomState(1)
    frames = rs.randint(0, 256, (2, 1, 64, 64, 3)).astype(np.uint8)
    states = np.random.RandomState(2).normal(0, .1, (2, 5))
    results = []
    for cls in (HipVPredEvaluation, make_oracle_predictor_class(factory)):
        with contextlib.redirect_stdout(io.StringIO()):
            ctrl = PixelCostController(dict(ag), dict(base, predictor_class=cls), 0, 1)
            ctrl.reset()
            np.random.seed(0)
            ctrl.act(t=0, i_tr=0, desig_pix=[[32, 32]], goal_pix=[[16, 48]], images=frames[:1], state=states[:1])
            out = ctrl.act(t=1, i_tr=0, desig_pix=[[32, 32]], goal_pix=[[16, 48]], images=frames, state=states)
        results.append((out, ctrl._best_indices.copy()))
    (hip, hip_idx), (ora, ora_idx) = results
    s_hip, s_ora = hip['plan_stat']['scores_itr0'], ora['plan_stat']['scores_itr0']
    np.testing.assert_allclose(s_hip, s_ora, rtol=1e-5)
    gap = np.diff(np.sort(s_ora))[9]                        # margin at the K / K+1 boundary
    assert gap > 4 * np.abs(s_hip - s_ora).max(), 'fixture seeds give an ambiguous elite boundary'
    np.testing.assert_array_equal(hip_idx, ora_idx)
    np.testing.assert_array_equal(hip['actions'], ora['actions'])


def test_errors_are_loud():
    lib = _lib.load_library()
    cfg = _lib.VfConfig(64, 64, 4, 5, 1, 2, 5, 10, 4, 0)
    h = ctypes.c_void_p()
    _lib.check(lib.vf_create(ctypes.byref(cfg), ctypes.byref(h)))
    acts = torch.zeros((2, 3, 4), device='cuda')
    sc = torch.zeros(2, dtype=torch.float64, device='cuda')
    goal = (ctypes.c_int32 * 2)(0, 0)
    assert lib.vf_rollout(h, acts.data_ptr(), 2, goal, ctypes.c_float(1.), None, sc.data_ptr(), None, None) == -4
    assert b'vf_load_weights' in lib.vf_last_error()
    bad = _lib.VfConfig(60, 64, 4, 5, 1, 2, 5, 10, 4, 0)
    h2 = ctypes.c_void_p()
    assert lib.vf_create(ctypes.byref(bad), ctypes.byref(h2)) == -1
    lib.vf_destroy(h)
    pred, _ = _predictor(32, 32, 2, 1, bs=2)
    with pytest.raises(ValueError):
        pred.score(_context(32, 32, 1, np.random.RandomState(0)), {'actions': np.zeros((2, 5, 4))}, [[[0, 0]]])


def test_launch_strategies_are_bit_identical():
    """Context de-duplication, the persistent single-launch rollout, the XCD queues and the fused decoder top only
    reorganise the same per-sample arithmetic: same bits out."""
    H = W = 32
    T, M = 3, 37
    pred, _ = _predictor(H, W, T, 2, bs=M)
    rs = np.random.RandomState(31)
    ctx = _context(H, W, 2, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = np.array([[[3, 20], [30, 1]]])
    outs = []
    for dedup, persistent, xcd, fuse in ((1, 0, 1, 1), (0, 0, 1, 1), (1, 1, 1, 1), (0, 1, 1, 1), (1, 1, 0, 1),
                                         (1, 1, 1, 0), (0, 1, 0, 0)):
        pred.set_dedup(dedup)
        pred.set_persistent(persistent)
        pred.set_xcd_queues(xcd)
        pred.set_fuse_top(fuse)
        s, pt = pred.score(ctx, {'actions': actions}, goal)
        assert pred.device_status() == 0
        got = pred(ctx, {'actions': actions})
        outs.append((s, pt, got['predicted_frames'], got['predicted_pixel_distributions'], got['predicted_states']))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)


def test_two_view_predictor_matches_per_view_oracle():
    """BASELINE configs[2] shape in miniature: 2 views x 2 designated pixels, both views in one launch,
    every launch strategy."""
    from visual_foresight_amd.video_prediction.hip_predictor import MultiViewHipPredictor
    H = W = 32
    T, M, nd, ncam = 2, 6, 2, 2
    hp = dict(designated_pixel_count=nd, run_batch_size=M, adim=4, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, ncam=ncam)
    pred = MultiViewHipPredictor('', hp)
    cfg = CdnaConfig(height=H, width=W, ndesig=nd, sequence_length=T + 2)
    weights = [CdnaWeights.random(cfg, seed=10 + c, bias_scale=0.05, ln_jitter=0.1) for c in range(ncam)]
    pred.restore(weights)
    rs = np.random.RandomState(77)
    desig = rs.randint(0, H, (ncam, nd, 2))
    ctx = {'context_frames': rs.randint(0, 256, (3, ncam, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (2, 4)), 'context_states': rs.normal(0, 0.1, (3, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib(desig, 2, ncam, H, W, nd)}
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = rs.randint(0, H, (ncam, nd, 2))
    want_d = []
    oracle_frames = []
    for c in range(ncam):
        f, d, s = _oracle(weights[c], MultiViewHipPredictor.view_context(ctx, c), actions)
        oracle_frames.append(f)
        want_d.append(d)
    want_d = np.concatenate(want_d, axis=2)
    want, want_pt = pixel_cost.eval_pixel_cost(want_d, goal, 10.)
    first = None
    for persistent in (1, 0):
        pred.set_persistent(persistent)
        scores, per_task = pred.score(ctx, {'actions': actions}, goal)
        got = pred(ctx, {'actions': actions})
        assert got['predicted_frames'].shape == (M, T, ncam, H, W, 3)
        for c in range(ncam):
            assert np.abs(got['predicted_frames'][:, :, c] - oracle_frames[c][:, :, 0]).max() <= 1e-5
        dmax = want_d.max(axis=(3, 4), keepdims=True)
        assert (np.abs(got['predicted_pixel_distributions'] - want_d) / dmax).max() <= 2e-5
        np.testing.assert_allclose(per_task, want_pt, rtol=1e-5)        # camera-major task order
        np.testing.assert_allclose(scores, want, rtol=1e-5)
        best = pred.fetch_pixel_distributions(int(np.argmin(scores)))
        assert best.shape == (T, ncam, H, W, nd)
        np.testing.assert_array_equal(best, got['predicted_pixel_distributions'][int(np.argmin(scores))])
        if first is None:
            first = (scores, per_task, got['predicted_frames'])
        else:       # the launch strategies agree bit for bit
            np.testing.assert_array_equal(scores, first[0])
            np.testing.assert_array_equal(per_task, first[1])
            np.testing.assert_array_equal(got['predicted_frames'], first[2])
        assert pred.device_status() == 0


def test_shared_unit_cache_across_rollouts():
    """The context-only (batch-1) units are computed once per context and reused by later rollouts."""
    H = W = 32
    T, M = 3, 21
    rs = np.random.RandomState(41)
    ctx_a, ctx_b = _context(H, W, 1, rs), _context(H, W, 1, rs)
    acts1, acts2 = rs.normal(0, 0.1, (M, T, 4)), rs.normal(0, 0.1, (M, T, 4))
    goal = np.array([[[5, 9]]])
    for persistent in (1, 0):
        pred, _ = _predictor(H, W, T, 1, bs=M)
        pred.set_persistent(persistent)
        pred.score(ctx_a, {'actions': acts1}, goal)            # computes and caches the shared units
        cached, _ = pred.score(ctx_a, {'actions': acts2}, goal)   # reuses them
        pred.set_dedup(1)                                       # invalidates the cache
        pred._ctx_key = None
        fresh, _ = pred.score(ctx_a, {'actions': acts2}, goal)
        np.testing.assert_array_equal(cached, fresh)
        other, _ = pred.score(ctx_b, {'actions': acts2}, goal)  # new context: must not reuse
        ref, _ = _predictor(H, W, T, 1, bs=M)[0].score(ctx_b, {'actions': acts2}, goal)
        np.testing.assert_array_equal(other, ref)
        assert not np.array_equal(other, fresh)
        assert pred.device_status() == 0


@pytest.mark.parametrize('H,W,T,M,nd', [(64, 64, 3, 5, 1), (48, 64, 2, 5, 2), (32, 32, 4, 9, 1)])
def test_split_bf16_mode_has_fp32_class_accuracy(H, W, T, M, nd):
    """precision='bf16x6' (conv-LSTM GEMMs as six bf16 MFMA products per multiply) against the
    float64 oracle: it must sit within 4x of the distance the exact-fp32 path sits at, and inside
    the same absolute tolerances as the fp32 path."""
    rs = np.random.RandomState(H + T)
    ctx = _context(H, W, nd, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = rs.randint(0, min(H, W), (1, nd, 2))
    errs = {}
    for prec in ('fp32', 'bf16x6'):
        for persistent in (1, 0):
            pred, weights = _predictor(H, W, T, nd, bs=M, precision=prec)
            pred.set_persistent(persistent)
            scores, _ = pred.score(ctx, {'actions': actions}, goal)
            got = pred(ctx, {'actions': actions})
            f, d, s = _oracle(weights, ctx, actions, torch.float64)
            want, _ = pixel_cost.eval_pixel_cost(d.astype(np.float32), goal, 10.)
            dmax = d.max(axis=(3, 4), keepdims=True)
            errs[(prec, persistent)] = (np.abs(got['predicted_frames'] - f).max(),
                                        (np.abs(got['predicted_pixel_distributions'] - d) / dmax).max(),
                                        np.abs(scores / want - 1).max())
    for persistent in (1, 0):
        e32, e16 = errs[('fp32', persistent)], errs[('bf16x6', persistent)]
        assert e16[0] <= 1e-5 and e16[1] <= 2e-5 and e16[2] <= 1e-5
        assert e16[0] <= 4 * e32[0] + 1e-7 and e16[1] <= 4 * e32[1] + 1e-7, (e32, e16)
    # the two launch strategies agree bit-for-bit within a precision mode
    assert errs[('bf16x6', 1)] == errs[('bf16x6', 0)]


@pytest.mark.parametrize('H,W,adim,sdim,T,M', [(128, 128, 4, 5, 2, 2), (48, 64, 3, 3, 3, 4), (64, 96, 5, 5, 2, 3)])
def test_other_resolutions_and_action_state_dims(H, W, adim, sdim, T, M):
    """128x128 (BASELINE config 5 resolution), the sim cart-gripper's adim = sdim = 3, a wide frame."""
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    hp = dict(designated_pixel_count=1, run_batch_size=M, adim=adim, sdim=sdim, image_height=H, image_width=W,
              sequence_length=T + 2)
    pred = HipVPredEvaluation('', hp)
    cfg = CdnaConfig(height=H, width=W, adim=adim, sdim=sdim, sequence_length=T + 2)
    weights = CdnaWeights.random(cfg, seed=8, bias_scale=0.05, ln_jitter=0.1)
    pred.restore(weights)
    rs = np.random.RandomState(H + adim)
    ctx = {'context_frames': rs.randint(0, 256, (2, 1, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (1, adim)), 'context_states': rs.normal(0, 0.1, (2, sdim)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib([[[H // 2, W // 3]]], 2, 1, H, W, 1)}
    actions = rs.normal(0, 0.1, (M, T, adim))
    goal = np.array([[[H // 4, W // 2]]])
    scores, _ = pred.score(ctx, {'actions': actions}, goal)
    got = pred(ctx, {'actions': actions})
    f, d, s = _oracle(weights, ctx, actions)
    assert np.abs(got['predicted_frames'] - f).max() <= 1e-5
    assert (np.abs(got['predicted_pixel_distributions'] - d) / d.max(axis=(3, 4), keepdims=True)).max() <= 2e-5
    assert np.abs(got['predicted_states'] - s).max() <= 1e-6
    want, _ = pixel_cost.eval_pixel_cost(d, goal, 10.)
    np.testing.assert_allclose(scores, want, rtol=1e-5)


def test_config2_planning_call_elites_match_oracle():
    """BASELINE configs[1] at full size: 200 samples x horizon 13 x 64x64, 3 CEM iterations.

    The same controller is driven by the CPU oracle (host cost path of the reference) and by the
    HIP predictor in both precision modes.  Iteration i+1 samples from the elites of iteration i,
    so matching final elites means every iteration selected identically."""
    from visual_foresight_amd.policy.cem_controllers import PixelCostController
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ag = {'adim': 4, 'sdim': 5, 'image_height': 64, 'image_width': 64}
    base = {'nactions': 13, 'repeat': 1, 'rejection_sampling': False, 'verbose': False}
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

    class HipSplit(HipVPredEvaluation):
        def __init__(self, model_path, hparams, n_gpus=1, first_gpu=0):
            super(HipSplit, self).__init__(model_path, dict(hparams, precision='bf16x6'), n_gpus, first_gpu)

    ora, ora_idx = run(make_oracle_predictor_class(factory))
    for cls in (HipVPredEvaluation, HipSplit):
        hip, hip_idx = run(cls)
        for itr in range(3):
            key = 'scores_itr%d' % itr
            np.testing.assert_allclose(hip['plan_stat'][key], ora['plan_stat'][key], rtol=1e-5)
            gap = np.diff(np.sort(ora['plan_stat'][key]))[9]            # margin at the K / K+1 boundary (K = 10)
            assert gap > 4 * np.abs(hip['plan_stat'][key] - ora['plan_stat'][key]).max(), \
                'fixture seeds give an ambiguous elite boundary in iteration %d' % itr
        np.testing.assert_array_equal(hip_idx, ora_idx)
        np.testing.assert_array_equal(hip['actions'], ora['actions'])


@pytest.mark.parametrize('arch', ['cdna', 'savp'])
def test_stochastic_predictor_mean_over_latent_draws(arch):
    """BASELINE configs[4] in miniature: n_latent z-draws per action folded into the sample axis, on the plain
    CDNA network and on the SAVP-class generator (savp_arch.py)."""
    from oracle.savp_predictor import OracleSavp
    from visual_foresight_amd.video_prediction.savp_arch import SavpConfig
    from visual_foresight_amd.video_prediction.stochastic_predictor import StochasticHipPredictor
    H = W = 32
    T, M, nl, zd = 2, 4, 3, 8
    hp = dict(designated_pixel_count=1, run_batch_size=M, adim=4, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, n_latent=nl, zdim=zd, latent_seed=5, arch=arch)
    pred = StochasticHipPredictor('', hp)
    assert pred.arch == arch
    cfg = (SavpConfig if arch == 'savp' else CdnaConfig)(height=H, width=W, adim=4 + zd, sdim=5, sequence_length=T + 2)
    weights = CdnaWeights.random(cfg, seed=2, bias_scale=0.05, ln_jitter=0.1)
    pred.restore(weights)
    rs = np.random.RandomState(3)
    ctx = _context(H, W, 1, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = np.array([[[4, 20]]])
    z = pred.draw_latents(T)
    scores, _ = pred.score(ctx, {'actions': actions}, goal)
    assert scores.shape == (M,)
    # oracle: the same network with z appended to every action, mean over the draws
    ctx_o = dict(ctx, context_actions=np.concatenate([ctx['context_actions'], np.zeros((2, zd))], axis=1))
    aug = np.concatenate([np.repeat(actions, nl, axis=0), np.tile(z, (M, 1, 1))], axis=2)
    ora = (OracleSavp if arch == 'savp' else OracleCdna)(weights, torch.float32)
    _, d, _ = ora.rollout(ctx_o['context_frames'], ctx_o['context_actions'], ctx_o['context_pixel_distributions'],
                          ctx_o['context_states'], aug)
    want, _ = pixel_cost.eval_pixel_cost(d, goal, 10.)
    np.testing.assert_allclose(scores, want.reshape(M, nl).mean(axis=1), rtol=1e-5)
    assert pred.fetch_pixel_distributions(1).shape == (T, 1, H, W, 1)


def test_legacy_boundary_get_context_rollout_predictions():
    """Row a13: the legacy predictor_func boundary driven by get_context / rollout_predictions."""
    from visual_foresight_amd.video_prediction.pred_util import get_context, rollout_predictions
    H = W = 32
    T, M, bs = 2, 7, 3
    pred, weights = _predictor(H, W, T, 1, bs=bs)
    rs = np.random.RandomState(8)
    images = rs.randint(0, 256, (4, 1, H, W, 3)).astype(np.uint8)
    states = rs.normal(0, .1, (4, 5))
    frames_ctx, states_ctx = get_context(2, 3, states, images)
    one_hot = pixel_cost.one_hot_distrib([[[9, 20]]], 2, 1, H, W, 1)[None]
    executed = rs.normal(0, .1, (1, 4))
    future = rs.normal(0, .1, (M, T, 4))
    full = np.concatenate([np.tile(executed[None], (M, 1, 1)), future, np.zeros((M, 1, 4))], axis=1)  # [M, seq_len, adim]
    gi, gd, gs = rollout_predictions(pred.predictor_func(), bs, full, frames_ctx, states_ctx, one_hot)
    assert [x.shape[0] for x in gi] == [3, 3, 1]
    got_frames, got_distrib = np.concatenate(gi, 0), np.concatenate(gd, 0)
    ctx = {'context_frames': images, 'context_actions': executed, 'context_states': states,
           'context_pixel_distributions': one_hot[0]}
    want = pred(ctx, {'actions': future})
    np.testing.assert_array_equal(got_frames, want['predicted_frames'])
    np.testing.assert_array_equal(got_distrib, want['predicted_pixel_distributions'])
    f, d, _ = _oracle(weights, ctx, future)
    np.testing.assert_allclose(got_frames, f, atol=2e-5)
    # rows whose context actions differ are refused (the engine shares the context over the batch)
    bad = full.copy()
    bad[1, 0, 0] += 1.0
    with pytest.raises(ValueError):
        pred.predictor_func()(input_images=frames_ctx, input_state=states_ctx, input_actions=bad[:bs],
                              input_one_hot_images=one_hot)


def test_float16_conf_key_selects_the_reduced_precision_mode():
    """The reference's `'float16' in conf` switch (setup_predictor.py:92-95) maps to the split-bf16 mode."""
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    hp = dict(designated_pixel_count=1, run_batch_size=4, image_height=32, image_width=32, sequence_length=4)
    assert HipVPredEvaluation('', dict(hp, float16='')).precision == 1
    assert HipVPredEvaluation('', dict(hp, float16='', precision='fp32')).precision == 0
    assert HipVPredEvaluation('', hp).precision == (1 if __import__('os').environ.get('VF_PRECISION') == 'bf16x6' else 0)


@pytest.mark.parametrize('H,W,nd,M', [(64, 64, 1, 21), (64, 64, 4, 6), (32, 32, 2, 9), (48, 64, 1, 7), (40, 56, 2, 5)])
def test_fused_decoder_top_is_invisible_in_the_results(H, W, nd, M):
    """vf_set_fuse_top: the last transposed conv and the compositing as one item per tile (the decoder's top tensor
    never reaches memory) - same bits as the two-phase schedule and as the per-layer launches, several tile
    geometries (40x56 cannot be fused and falls back)."""
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    T = 3
    rs = np.random.RandomState(H + W + nd)
    ctx = _context(H, W, nd, rs)
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = rs.randint(0, min(H, W), (1, nd, 2))
    pred, weights = _predictor(H, W, T, nd, bs=M)
    pred.set_fuse_top(0)            # the two-phase schedule
    base, base_pt = pred.score(ctx, {'actions': actions}, goal)
    ref = pred(ctx, {'actions': actions})
    hp = dict(designated_pixel_count=nd, run_batch_size=M, adim=4, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, fuse_top=1)
    other = HipVPredEvaluation('', hp).restore(weights)
    for _ in range(2):              # the second call runs the cached-context schedule
        got, got_pt = other.score(ctx, {'actions': actions}, goal)
        np.testing.assert_array_equal(got, base)
        np.testing.assert_array_equal(got_pt, base_pt)
    out = other(ctx, {'actions': actions})
    np.testing.assert_array_equal(out['predicted_frames'], ref['predicted_frames'])
    np.testing.assert_array_equal(out['predicted_pixel_distributions'], ref['predicted_pixel_distributions'])
    other.set_persistent(0)         # per-layer launches (never fused)
    np.testing.assert_array_equal(other.score(ctx, {'actions': actions}, goal)[0], base)
    assert other.device_status() == 0
