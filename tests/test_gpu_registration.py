"""GPU parity of the registration path: ``vf_register`` through the C ABI
against the loop oracle (``oracle/registration.py``; parity unpinned - the reference's registration
network and therefore its outputs do not exist in the snapshot), and ``RegisterGtruthController``
driving the device path."""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import registration as oracle_reg                                   # noqa: E402
from tests.helpers.flow_warper import make_flow_warper, synthetic_flow          # noqa: E402
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights   # noqa: E402


def _predictor(H, W, ncam, nd=1, bs=4, T=2):
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    hp = dict(designated_pixel_count=nd, run_batch_size=bs, adim=4, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2, ncam=ncam)
    return HipVPredEvaluation('', hp).restore()


@pytest.mark.parametrize('H,W,ncam', [(64, 64, 2), (48, 64, 1), (128, 128, 2)])
def test_warp_and_register_match_loop_oracle(H, W, ncam):
    pred = _predictor(H, W, ncam)
    rs = np.random.RandomState(H + ncam)
    cur, ref = (rs.uniform(0, 1, (ncam, H, W, 3)).astype(np.float32) for _ in range(2))
    flow = synthetic_flow(cur, ref, scale=3.0)
    flow[0, :3, :3] = -9.0                               # warp points left of / above the frame: clamped
    flow[-1, -3:, -3:] = 11.0
    ntask = 5
    pix = rs.randint(0, [H, W], (ncam, ntask, 2))
    pix[0, 0] = (0, 0)
    pix[0, 1] = (H - 1, W - 1)                           # windows clipped at both borders
    pix[-1, 2] = (1, W - 2)
    want_warped, want_pts = oracle_reg.bilinear_warp_loops(cur, flow)
    for region_on in (False, True):
        for clip_sub, which in ((1, 'start'), (0, 'goal')):
            region = (5 if H >= 96 else 2) if region_on else 0
            desig, err, warped, pts = pred.register(cur, ref, flow, pix, region=region, clip_sub=clip_sub,
                                                    want_warped=True)
            np.testing.assert_array_equal(pts, want_pts)                # same float32 additions
            np.testing.assert_allclose(warped, want_warped, rtol=0, atol=2e-7)
            for c in range(ncam):
                if which == 'start':
                    eo, do = oracle_reg.warp_err_loops(c, pix[c], pix[c], ref, ref, want_pts, None, want_warped,
                                                       None, ['start'], region_on)
                else:
                    eo, do = oracle_reg.warp_err_loops(c, pix[c], pix[c], ref, ref, None, want_pts, None,
                                                       want_warped, ['goal'], region_on)
                np.testing.assert_array_equal(desig[c], do[:, 0])       # medians / flips of the same float32 values
                np.testing.assert_allclose(err[c], eo[:, 0], rtol=2e-5)


def test_register_controller_device_path_matches_host_path():
    """The same controller with the registration arithmetic on the device and on the host (NumPy path on the
    plug-in's own warped / warp_pts): same tracked pixels, same trade-off, same scores and elites."""
    from visual_foresight_amd.policy.cem_controllers import RegisterGtruthController
    H = W = 64
    ncam, T, M = 2, 3, 24
    ag = {'adim': 4, 'sdim': 5, 'image_height': H, 'image_width': W, 'ncam': ncam}
    base = {'nactions': T, 'repeat': 1, 'rejection_sampling': False, 'verbose': False, 'num_samples': M,
            'designated_pixel_count': 2, 'registration_warper': make_flow_warper(), 'iterations': 2,
            'trade_off_reg': True, 'register_region': True}
    rs = np.random.RandomState(4)
    frames = rs.randint(0, 256, (2, ncam, H, W, 3)).astype(np.uint8)
    states = rs.normal(0, .1, (2, 5))
    goal_image = rs.uniform(0, 1, (1, ncam, H, W, 3)).astype(np.float32)
    outs = []
    for on_device in (True, False):
        pol = dict(base) if on_device else dict(base, registration_on_device=False)
        with contextlib.redirect_stdout(io.StringIO()):
            ctrl = RegisterGtruthController(dict(ag), pol, 0, 1)
            ctrl.reset()
            np.random.seed(0)
            kw = dict(goal_image=goal_image, i_tr=0, desig_pix=[[20, 30], [40, 12]], goal_pix=[[10, 50], [33, 33]])
            ctrl.act(t=0, images=frames[:1], state=states[:1], **kw)
            out = ctrl.act(t=1, images=frames, state=states, **kw)
        outs.append((out, ctrl._best_indices.copy(), ctrl._desig_pix.copy()))
    (dev, dev_idx, dev_pix), (host, host_idx, host_pix) = outs
    np.testing.assert_array_equal(dev_pix, host_pix)
    np.testing.assert_allclose(dev['plan_stat']['tradeoff'], host['plan_stat']['tradeoff'], rtol=2e-5)
    for k in ('scores_itr0', 'scores_itr1'):
        np.testing.assert_allclose(dev['plan_stat'][k], host['plan_stat'][k], rtol=1e-5)
    np.testing.assert_array_equal(dev_idx, host_idx)
    np.testing.assert_array_equal(dev['actions'], host['actions'])
    assert dev['plan_stat']['tradeoff'].shape == (ncam, 2)
    np.testing.assert_allclose(dev['plan_stat']['tradeoff'].sum(), 1.0, rtol=1e-6)   # one task: weights sum to 1


def test_vf_register_reproduces_the_reference_fixture(golden_dir):
    """``vf_register`` through the C ABI against ``tests/golden/registration.npz`` - outputs of the
    reference's REAL ``get_warp_err`` (``register_gtruth_controller.py:113-173``) on the same seeded images
    and flow: tracked pixels bit-exact (window medians / flips of the same float32 warp points), region
    warp errors to 2e-5 (float32 sums in another order); and the trade-off of ``register_gtruth``
    (``:88-91``) computed from the device's errors."""
    import json
    import os
    from tests.helpers.flow_warper import registration_inputs, synthetic_flow
    from visual_foresight_amd.policy.cem_controllers.registration import tradeoff_weights
    meta = json.load(open(os.path.join(golden_dir, 'registration.json')))
    arrays = np.load(os.path.join(golden_dir, 'registration.npz'))
    done = 0
    for case in meta['cases']:
        if case.get('pred_height', case['H']) != case['H']:
            continue                                    # medium-resolution images: host-only feature
        name, ncam, ntask, H, W = case['name'], case['ncam'], case['ntask'], case['H'], case['W']
        pred = _predictor(H, W, ncam)
        region = (5 if H >= 96 else 2) if case['region'] else 0
        if case.get('full'):
            rs = np.random.RandomState(case['seed'])
            start, goal, cur = (rs.uniform(0, 1, (ncam, H, W, 3)).astype(np.float32) for _ in range(3))
            pix_t0, goal_pix = arrays[name + '/pix_t0'], arrays[name + '/goal_pix']
        else:
            start, goal, cur = registration_inputs(case['seed'], ncam, H, W, case['flow_scale'])[:3]
            pix_t0, goal_pix = np.array(case['pix_t0']), np.array(case['goal_pix'])
        per_reg = []
        if 'start' in case['regs']:
            per_reg.append(pred.register(cur, start, synthetic_flow(cur, start, case['flow_scale']), pix_t0,
                                         region=region, clip_sub=1))
        if 'goal' in case['regs']:
            per_reg.append(pred.register(cur, goal, synthetic_flow(cur, goal, case['flow_scale']), goal_pix,
                                         region=region, clip_sub=0))
        desig = np.stack([d for d, _ in per_reg], axis=2)               # [ncam, ntask, nreg, 2]
        errs = np.stack([e for _, e in per_reg], axis=2)                # [ncam, ntask, nreg]
        if case.get('full'):
            np.testing.assert_array_equal(desig.reshape(ncam, -1, 2), arrays[name + '/desig_pix'])
            np.testing.assert_allclose(errs.reshape(ncam, -1), arrays[name + '/warperrs'], rtol=2e-5)
            np.testing.assert_allclose(tradeoff_weights(errs).reshape(ncam, -1), arrays[name + '/tradeoff'],
                                       rtol=4e-5)
        else:
            for c in range(ncam):
                np.testing.assert_array_equal(desig[c], arrays['%s/cam%d/desig' % (name, c)])
                if case['region']:
                    np.testing.assert_allclose(errs[c], arrays['%s/cam%d/warperrs' % (name, c)], rtol=2e-5)
        done += 1
    assert done == 7
