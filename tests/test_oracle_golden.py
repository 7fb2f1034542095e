"""Pin the oracle's cost restatement to vectors produced by the reference itself."""
import json
import os

import numpy as np

from oracle import pixel_cost


def _load(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, 'cost.json')))
    arrays = np.load(os.path.join(golden_dir, 'cost.npz'))
    return meta, arrays


def _inputs(case):
    rs = np.random.RandomState(case['seed'])
    M, T, H, W, nd = case['M'], case['T'], case['H'], case['W'], case['ndesig']
    distrib = rs.uniform(0.0, 1.0, (M, T, 1, H, W, nd)).astype(np.float32)
    goal = rs.randint(-3, max(H, W) + 3, (1, nd, 2))
    desig = rs.randint(-3, max(H, W) + 3, (1, nd, 2))
    return distrib, goal, desig


def test_fixture_numpy_version_matches(golden_dir):
    meta, _ = _load(golden_dir)
    # score dtype is float32 under numpy 1.14 and float64 under numpy >= 2 (NEP 50); the
    # fixtures record which world they were minted in
    assert meta['numpy'].split('.')[0] == np.__version__.split('.')[0]


def test_cost_matches_reference(golden_dir):
    meta, arrays = _load(golden_dir)
    for case in meta['cases']:
        name = case['name']
        distrib, goal, desig = _inputs(case)
        np.testing.assert_array_equal(goal, arrays[name + '/goal'])
        scores, per_task = pixel_cost.eval_pixel_cost(distrib, goal, case['finalweight'],
                                                      case['only_take_first_view'])
        # same NumPy, same order of operations -> bit-identical
        np.testing.assert_array_equal(scores, arrays[name + '/scores'])
        np.testing.assert_array_equal(scores.argsort(), arrays[name + '/argsort'])
        assert str(scores.dtype) == case['scores_dtype']
        full = np.stack([pixel_cost.expected_distance(
            distrib[:, :, 0, :, :, p], pixel_cost.distance_grid(goal[0, p], case['H'], case['W']),
            case['finalweight']) for p in range(case['ndesig'])], axis=1)
        np.testing.assert_array_equal(full, arrays[name + '/scores_per_task'])


def test_distance_grid_matches_reference(golden_dir):
    meta, arrays = _load(golden_dir)
    for case in meta['cases']:
        goal = arrays[case['name'] + '/goal']
        grid = pixel_cost.distance_grid(goal[0, 0], case['H'], case['W'])
        np.testing.assert_allclose(grid, arrays[case['name'] + '/grid0'], rtol=0, atol=1e-12)


def test_one_hot_matches_reference(golden_dir):
    meta, arrays = _load(golden_dir)
    for case in meta['cases']:
        desig = arrays[case['name'] + '/desig']
        oh = pixel_cost.one_hot_distrib(desig, 2, 1, case['H'], case['W'], case['ndesig'])
        assert list(oh.shape) == case['onehot_shape']
        np.testing.assert_array_equal(np.argwhere(oh != 0), arrays[case['name'] + '/onehot_nonzero'])


# ------------------------------------------------------------------ a15 / f2: registration arithmetic
def _reg_fixture(golden_dir):
    return (json.load(open(os.path.join(golden_dir, 'registration.json'))),
            np.load(os.path.join(golden_dir, 'registration.npz')))


def test_registration_oracle_matches_reference_get_warp_err(golden_dir):
    """``oracle/registration.py`` against the outputs of the reference's REAL ``get_warp_err``
    (``register_gtruth_controller.py:113-173``): region mode at 64 / 48x64 / 128 / medium-resolution
    images, windows clipped at both borders (the start window to size-1, the goal window to size),
    point mode (where the reference leaves the errors zero)."""
    from oracle import registration as oreg
    from tests.helpers.flow_warper import registration_inputs
    meta, arrays = _reg_fixture(golden_dir)
    assert meta['region_start_only'] == 'TypeError'
    for case in meta['cases']:
        if case.get('full'):
            continue
        name, ncam, H, W = case['name'], case['ncam'], case['H'], case['W']
        start, goal, cur, ws, ps, wg, pg = registration_inputs(case['seed'], ncam, H, W, case['flow_scale'])
        scale = H // case['pred_height']
        pix_t0 = (np.array(case['pix_t0']) * H / case['pred_height']).astype(int)
        goal_pix = (np.array(case['goal_pix']) * H / case['pred_height']).astype(int)
        assert scale >= 1
        for c in range(ncam):
            e, d = oreg.warp_err_loops(c, pix_t0[c], goal_pix[c], start, goal, ps, pg, ws, wg, case['regs'],
                                       case['region'], pred_height=case['pred_height'],
                                       point_errors='reference')
            np.testing.assert_array_equal(d, arrays['%s/cam%d/desig' % (name, c)])
            np.testing.assert_allclose(e, arrays['%s/cam%d/warperrs' % (name, c)], rtol=1e-6, atol=0)


def test_registration_tradeoff_matches_reference_register_gtruth(golden_dir):
    """Whole ``register_gtruth`` (``:54-111``): tracked pixels [ncam, ndesig, 2], warp errors and the
    trade-off normalised over (camera, registration) per task (``:88-91``)."""
    from oracle import registration as oreg
    from tests.helpers.flow_warper import make_flow_warper
    meta, arrays = _reg_fixture(golden_dir)
    n = 0
    for case in meta['cases']:
        if not case.get('full'):
            continue
        n += 1
        name, ncam, ntask, H, W = case['name'], case['ncam'], case['ntask'], case['H'], case['W']
        rs = np.random.RandomState(case['seed'])
        start, goal, cur = (rs.uniform(0, 1, (ncam, H, W, 3)).astype(np.float32) for _ in range(3))
        pix_t0 = rs.randint(0, [H, W], (ncam, ntask, 2))
        goal_pix = rs.randint(0, [H, W], (ncam, ntask, 2))
        np.testing.assert_array_equal(pix_t0, arrays[name + '/pix_t0'])
        warper = make_flow_warper(case['flow_scale'])
        ws, _, ps = warper(cur, start)
        wg, _, pg = warper(cur, goal)
        errs, desig = zip(*[oreg.warp_err_loops(c, pix_t0[c], goal_pix[c], start, goal, ps, pg, ws, wg,
                                                case['regs'], case['region']) for c in range(ncam)])
        errs = np.stack(errs, 0)
        np.testing.assert_array_equal(np.stack(desig, 0).reshape(ncam, -1, 2), arrays[name + '/desig_pix'])
        np.testing.assert_allclose(errs.reshape(ncam, -1), arrays[name + '/warperrs'], rtol=1e-6)
        np.testing.assert_allclose(oreg.tradeoff_loops(errs).reshape(ncam, -1), arrays[name + '/tradeoff'],
                                   rtol=1e-6)
    assert n == 2
