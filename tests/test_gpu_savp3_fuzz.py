"""Seeded sweep of ``arch = 'savp3'`` over shapes the fixed parity cases do not visit: image sizes that are multiples of 8 between
32 and 72, non-square and - unlike every case of test_gpu_savp3.py - with widths that are NOT multiples of 16 (the fused heads
then keep plain stores and a release fence instead of the block turn-over of the write-through path, and their 8 x 16 tiles
hang over the image), all three layer tables (by the size rule and forced), batches from 1 to 40, 1 - 4 designated pixels.

Every launch strategy must give the same bits (persistent / one launch per layer / one ticket queue / plain stores; the
four-phase form of the heads runs its convs on the matrix pipe in another summation order and agrees to rounding), and the result
must match the CPU oracle (oracle/savp3_predictor.py; tolerances of test_gpu_savp3.py).  Parity
unpinned, as for every network here (``visual_mpc/video_prediction/vpred_model_interface.py:52-58`` only instantiates the class).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import pixel_cost                                           # noqa: E402
from tests.test_gpu_savp3 import _actions, _context, _oracle, _predictor  # noqa: E402


def _cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        if i % 2:       # both sides >= 64: the paper's three-scale table by the size rule, or the two-scale one forced
            H, W = int(rs.choice([64, 72, 80])), int(rs.choice([64, 72, 80, 88]))
        else:
            H, W = int(8 * rs.randint(4, 10)), int(8 * rs.randint(4, 10))
        spec = int(rs.choice([0, 0, 32])) if min(H, W) >= 64 else 0
        out.append(dict(H=H, W=W, T=int(rs.randint(1, 3)), M=int(rs.choice([1, 2, 5, 9, 23, 40])), nd=int(rs.randint(1, 5)),
                        spec=spec, seed=300 + i))
    return out


CASES = _cases(12, 606)


@pytest.mark.parametrize('case', CASES, ids=['%dx%d_M%d_T%d_nd%d_spec%d' % (c['H'], c['W'], c['M'], c['T'], c['nd'], c['spec'])
                                             for c in CASES])
def test_savp3_every_launch_strategy_gives_the_same_bits_and_matches_the_oracle(case):
    H, W, T, M, nd, spec = (case[k] for k in ('H', 'W', 'T', 'M', 'nd', 'spec'))
    pred, weights = _predictor(H, W, T, nd, bs=M, seed=case['seed'], layer_spec=spec)
    rs = np.random.RandomState(case['seed'])
    ctx = _context(H, W, nd, rs)
    actions = _actions(M, T, rs)
    goal = rs.randint(-2, max(H, W) + 2, (1, nd, 2))
    outs = []
    #            persistent xcd fuse write-through
    for knobs in ((1, 1, 1, 1), (0, 1, 1, 1), (1, 0, 1, 1), (1, 1, 0, 1), (1, 1, 1, 0)):
        pred.set_persistent(knobs[0]); pred.set_xcd_queues(knobs[1]); pred.set_fuse_top(knobs[2])
        pred.set_sched_option('write_through', knobs[3])
        pred._ctx_key = None
        s, pt = pred.score(ctx, {'actions': actions}, goal, finalweight=10.)
        assert pred.device_status() == 0, knobs
        got = pred(ctx, {'actions': actions[:min(M, 6)]})
        outs.append((s, pt, got['predicted_frames'], got['predicted_pixel_distributions'], got['predicted_states']))
    for knobs, other in zip(('per-layer', 'one queue', 'four-phase heads', 'plain stores'), outs[1:]):
        for a, b in zip(outs[0], other):
            if knobs == 'four-phase heads':     # MFMA convs instead of the fused item's VALU convs: another summation order
                np.testing.assert_allclose(a, b, rtol=1e-5, atol=3e-5, err_msg=knobs)
            else:
                np.testing.assert_array_equal(a, b, err_msg=knobs)
    n = min(M, 6)
    f, d, st = _oracle(weights, ctx, actions[:n])
    s, pt, gf, gd, gs = outs[0]
    assert np.abs(gf - f).max() <= 3e-5
    dmax = d.max(axis=(3, 4), keepdims=True)
    assert (np.abs(gd - d) / dmax).max() <= 2e-5
    assert np.abs(gs - st).max() <= 1e-6
    want, want_pt = pixel_cost.eval_pixel_cost(d, goal, 10.)
    np.testing.assert_allclose(s[:n], want, rtol=1e-5)
    np.testing.assert_allclose(pt[:n], want_pt, rtol=1e-5)
    np.testing.assert_allclose(gd.sum(axis=(3, 4)), 1.0, atol=5e-6)
