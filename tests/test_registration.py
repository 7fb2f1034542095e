"""Registration cost arithmetic and the multi-view controller flow on CPU."""
import contextlib
import io

import numpy as np
import pytest

from oracle import registration as oracle_reg
from tests.helpers.fake_predictor import make_fake_predictor_class
from visual_foresight_amd.policy.cem_controllers import RegisterGtruthController
from visual_foresight_amd.policy.cem_controllers.registration import get_warp_err, tradeoff_weights


def shift_warper(dy, dx):
    """Test double: pretends the reference frame is the current frame shifted by (dy, dx)."""
    def warper(current, reference):
        ncam, H, W = current.shape[:3]
        cols, rows = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        pts = np.stack([np.clip(cols + dx, 0, W - 1), np.clip(rows + dy, 0, H - 1)], -1)
        warped = np.roll(current, (-dy, -dx), axis=(1, 2))
        return warped, None, np.tile(pts[None], (ncam, 1, 1, 1))
    return warper


@pytest.mark.parametrize('region', [False, True])
@pytest.mark.parametrize('regs', [['start', 'goal'], ['start'], ['goal']])
def test_warp_err_matches_loop_oracle(region, regs):
    rs = np.random.RandomState(3)
    ncam, H, W, ntask = 2, 24, 32, 3
    start, goal, cur = (rs.uniform(0, 1, (ncam, H, W, 3)) for _ in range(3))
    ws, _, ps = shift_warper(2, -1)(cur, start)
    wg, _, pg = shift_warper(-1, 3)(cur, goal)
    pix_t0 = rs.randint(0, [H, W], (ncam, ntask, 2))
    pix_t0[0, 0] = (0, 0)
    pix_t0[0, 1] = (H - 1, W - 1)                          # windows clipped at the border
    goal_pix = rs.randint(0, [H, W], (ncam, ntask, 2))
    errs = []
    for c in range(ncam):
        e, d = get_warp_err(c, pix_t0[c], goal_pix[c], start, goal, ps, pg, ws, wg, regs, region)
        eo, do = oracle_reg.warp_err_loops(c, pix_t0[c], goal_pix[c], start, goal, ps, pg, ws, wg, regs, region)
        np.testing.assert_allclose(e, eo, rtol=1e-12)
        np.testing.assert_allclose(d, do, rtol=0, atol=0)
        errs.append(e)
    w = tradeoff_weights(np.stack(errs, 0))
    np.testing.assert_allclose(w, oracle_reg.tradeoff_loops(np.stack(errs, 0)), rtol=1e-12)
    np.testing.assert_allclose(w.sum(axis=(0, 2)), 1.0, rtol=1e-12)


def test_controller_tracks_pixels_and_weights_scores():
    H = W = 16
    T, ncam = 5, 2
    fake = make_fake_predictor_class(T, H, W, ncam=ncam)
    fake.n_cam = ncam
    pol = {'predictor_class': fake, 'verbose': False, 'rejection_sampling': False, 'repeat': 1,
           'num_samples': 20, 'designated_pixel_count': 2, 'registration_warper': shift_warper(1, 2),
           'iterations': 2, 'trade_off_reg': True}
    ag = {'adim': 4, 'sdim': 5, 'image_height': H, 'image_width': W, 'ncam': ncam}
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl = RegisterGtruthController(ag, pol, 0, 1)
        ctrl.reset()
    assert ctrl.ntask == 1 and ctrl._n_cam == 2
    rs = np.random.RandomState(0)
    images = rs.randint(0, 256, (3, ncam, H, W, 3)).astype(np.uint8)
    goal_image = rs.uniform(0, 1, (1, ncam, H, W, 3))
    states = rs.normal(size=(3, 5))
    np.random.seed(4)
    for t in range(3):
        with contextlib.redirect_stdout(io.StringIO()):
            out = ctrl.act(goal_image=goal_image, t=t, i_tr=0, desig_pix=[[5, 6], [7, 8]],
                           goal_pix=[[2, 3], [12, 13]], images=images[:t + 1], state=states[:t + 1])
    # pixel (r, c) of the start frame sits at (r + 1, c + 2) in the current frame
    np.testing.assert_array_equal(ctrl._desig_pix[0, 0], (6, 8))
    np.testing.assert_array_equal(ctrl._desig_pix[1, 0], (8, 10))
    np.testing.assert_array_equal(ctrl._goal_pix[0], [[2, 3], [2, 3]])         # tiled over registrations
    w = out['plan_stat']['tradeoff']
    assert w.shape == (ncam, 2) and np.isclose(w.sum(), 1.0)
    assert out['plan_stat']['scores_itr1'].shape == (20,)
    assert out['actions'].shape == (4,)


def test_missing_warper_is_an_error():
    fake = make_fake_predictor_class(5, 16, 16)
    pol = {'predictor_class': fake, 'verbose': False, 'designated_pixel_count': 2}
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl = RegisterGtruthController({'adim': 4, 'sdim': 5, 'image_height': 16, 'image_width': 16}, pol, 0, 1)
        ctrl.reset()
    img = np.zeros((2, 1, 16, 16, 3), np.uint8)
    with pytest.raises(ValueError), contextlib.redirect_stdout(io.StringIO()):
        ctrl.act(goal_image=np.zeros((1, 1, 16, 16, 3)), t=0, i_tr=0, desig_pix=[[1, 1]], goal_pix=[[2, 2]],
                 images=img[:1], state=np.zeros((1, 5)))
        ctrl.act(goal_image=np.zeros((1, 1, 16, 16, 3)), t=1, i_tr=0, desig_pix=[[1, 1]], goal_pix=[[2, 2]],
                 images=img, state=np.zeros((2, 5)))


# ------------------------------------------------------------------ pinned to the reference (tests/golden/registration.*)
def _reg_fixture(golden_dir):
    import json
    import os
    return (json.load(open(os.path.join(golden_dir, 'registration.json'))),
            np.load(os.path.join(golden_dir, 'registration.npz')))


def test_get_warp_err_reproduces_the_reference(golden_dir):
    """Product ``get_warp_err`` == the reference's real ``get_warp_err`` outputs, bit for bit (same NumPy
    expressions); point-mode errors are this repo's documented deviation and are not compared."""
    from tests.helpers.flow_warper import registration_inputs
    meta, arrays = _reg_fixture(golden_dir)
    for case in meta['cases']:
        if case.get('full'):
            continue
        name, ncam, H, W = case['name'], case['ncam'], case['H'], case['W']
        start, goal, cur, ws, ps, wg, pg = registration_inputs(case['seed'], ncam, H, W, case['flow_scale'])
        pix_t0 = (np.array(case['pix_t0']) * H / case['pred_height']).astype(int)
        goal_pix = (np.array(case['goal_pix']) * H / case['pred_height']).astype(int)
        for c in range(ncam):
            e, d = get_warp_err(c, pix_t0[c], goal_pix[c], start, goal, ps, pg, ws, wg, case['regs'],
                                case['region'], pred_height=case['pred_height'])
            np.testing.assert_array_equal(d, arrays['%s/cam%d/desig' % (name, c)])
            if case['region']:
                np.testing.assert_array_equal(e, arrays['%s/cam%d/warperrs' % (name, c)])
            else:
                assert np.all(arrays['%s/cam%d/warperrs' % (name, c)] == 0) and np.all(e > 0)


@pytest.mark.parametrize('name', ['full64', 'full128'])
def test_controller_register_gtruth_reproduces_the_reference(golden_dir, name):
    """``RegisterGtruthController.register_gtruth`` (host path) == the reference's ``register_gtruth``
    (``:54-111``) on the same images and warper: tracked pixels, warp errors and trade-off bit for bit."""
    from tests.helpers.flow_warper import make_flow_warper
    meta, arrays = _reg_fixture(golden_dir)
    case = [c for c in meta['cases'] if c['name'] == name][0]
    ncam, ntask, H, W = case['ncam'], case['ntask'], case['H'], case['W']
    rs = np.random.RandomState(case['seed'])
    start, goal, cur = (rs.uniform(0, 1, (ncam, H, W, 3)).astype(np.float32) for _ in range(3))
    fake = make_fake_predictor_class(5, H, W, ncam=ncam)
    fake.n_cam = ncam
    pol = {'predictor_class': fake, 'verbose': False, 'designated_pixel_count': 2 * ntask,
           'registration_warper': make_flow_warper(case['flow_scale']), 'register_region': True}
    ag = {'adim': 4, 'sdim': 5, 'image_height': H, 'image_width': W, 'ncam': ncam}
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl = RegisterGtruthController(ag, pol, 0, 1)
        ctrl.reset()
    ctrl.desig_pix_t0 = arrays[name + '/pix_t0']
    ctrl.goal_pix_sel = arrays[name + '/goal_pix']
    ctrl.goal_image = goal
    ctrl.plan_stat = {}
    desig, tradeoff = ctrl.register_gtruth(start, cur)
    np.testing.assert_array_equal(desig, arrays[name + '/desig_pix'])
    np.testing.assert_array_equal(tradeoff, arrays[name + '/tradeoff'])
    np.testing.assert_array_equal(ctrl.plan_stat['warperrs'], arrays[name + '/warperrs'])
