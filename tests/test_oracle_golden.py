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
