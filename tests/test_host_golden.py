"""This repo's host code (policy / CEM / samplers / cost) vs. vectors minted from the reference."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from tests.helpers.fake_predictor import make_fake_predictor_class
from visual_foresight_amd.hparams import HParams
from visual_foresight_amd.policy import get_policy_args
from visual_foresight_amd.policy.cem_controllers import PixelCostController
from visual_foresight_amd.policy.cem_controllers.samplers import GaussianCEMSampler, CorrelatedNoiseSampler
from visual_foresight_amd.policy.utils import controller_utils as cu

AG = {'adim': 4, 'sdim': 5, 'image_height': 16, 'image_width': 16}


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def _meta(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name + '.json')))


def _arrays(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def _decode(d):
    """json -> policy dict (lists stay lists, floats stay floats)."""
    return {k: v for k, v in d.items()}


# ----------------------------------------------------------------------------- a1
def test_get_policy_args(golden_dir):
    want = _meta(golden_dir, 'policy_args')
    pol = PixelCostController.__new__(PixelCostController)
    rs = np.random.RandomState(11)
    obs = {'images': rs.randint(0, 256, (3, 1, 8, 8, 3)).astype(np.uint8),
           'state': rs.normal(size=(3, 5))}
    agent_data = {'desig_pix': [[3, 4]], 'goal_pix': [[6, 1]], 'verbose_worker': 'queue-handle'}
    got = get_policy_args(pol, obs, 2, 7, agent_data)
    assert sorted(got.keys()) == want['keys']
    for k in ('t', 'i_tr', 'desig_pix', 'goal_pix', 'verbose_worker'):
        assert got[k] == want[k]
    assert got['images'] is obs['images'] and got['state'] is obs['state']

    class NeedsGoal(object):
        def act(self, t, goal_image):
            pass
    with pytest.raises(ValueError) as e:
        get_policy_args(NeedsGoal(), obs, 0, 0, agent_data)
    assert str(e.value) == want['missing_required']


# ----------------------------------------------------------------------------- a2
def test_experiment_policy_dicts_drop_in(golden_dir):
    want = _meta(golden_dir, 'hparams')
    fake = make_fake_predictor_class(5, 16, 16)
    for name, case in want['cases'].items():
        pdict = _decode(case['policy'])
        pdict['predictor_class'] = fake
        with quiet():
            ctrl = PixelCostController(dict(AG), pdict, 0, 1)
        vals = ctrl._hp.values()
        vals.pop('predictor_class')
        for k, v in case['values'].items():
            if k == 'sampler':
                assert 'class:' + vals[k].__name__ == v
            else:
                assert vals[k] == v, (name, k, vals[k], v)
        assert set(vals.keys()) == set(case['values'].keys())
        assert ctrl._hp.start_planning == case['start_planning_after_ctor']


def test_every_experiment_file_of_the_reference_drops_in(golden_dir):
    """ALL experiment files of the reference that configure a ``PixelCostController`` / ``Register_Gtruth_Controller`` (18),
    loaded from the reference's own sources by tools/make_golden.py - nothing hand-typed: the product's controllers give the
    same hyper-parameter values where the reference's constructor succeeds and raise the same exception type where the
    reference itself refuses its own file (keys that are no hyper-parameters, a list-valued ``num_samples``).  Two of the
    files do not even import in the reference (a syntax error; a sampler module that is gone): pinned as such.
    Reference: policy.py:51-63, cem_base_controller.py:66-76."""
    from visual_foresight_amd.policy.cem_controllers import RegisterGtruthController
    want = _meta(golden_dir, 'experiment_files')['files']
    assert len(want) == 18
    classes = {'PixelCostController': PixelCostController, 'Register_Gtruth_Controller': RegisterGtruthController}
    named = {'class:CorrelatedNoiseSampler': CorrelatedNoiseSampler, 'class:GaussianCEMSampler': GaussianCEMSampler}
    product_only = {'trade_off_reg', 'registration_warper', 'registration_on_device'}      # RegisterGtruthController's plug-in keys
    seen = {'ok': 0, 'raises': 0, 'unloadable': 0}
    for rel, case in sorted(want.items()):
        if 'unloadable' in case:
            assert case['unloadable'] in ('SyntaxError', 'ImportError'), rel
            seen['unloadable'] += 1
            continue
        pdict = {k: named.get(v, v) if isinstance(v, str) else v for k, v in case['policy'].items()}
        assert not any(isinstance(v, str) and v.startswith('class:') for v in pdict.values()), (rel, pdict)
        fake = make_fake_predictor_class(5, case['ag_params']['image_height'], case['ag_params']['image_width'])
        cls = classes[case['controller']]
        pdict['predictor_class'] = fake
        if 'raises' in case:
            with pytest.raises(Exception) as e:
                with quiet():
                    cls(dict(case['ag_params']), pdict, 0, 1)
            assert type(e.value).__name__ == case['raises'], (rel, type(e.value).__name__, case['raises'])
            seen['raises'] += 1
            continue
        with quiet():
            ctrl = cls(dict(case['ag_params']), pdict, 0, 1)
        vals = ctrl._hp.values()
        vals.pop('predictor_class')
        for k, v in case['values'].items():
            if k == 'sampler':
                assert 'class:' + vals[k].__name__ == v, rel
            else:
                assert vals[k] == v, (rel, k, vals[k], v)
        assert set(vals.keys()) - product_only == set(case['values'].keys()), rel
        assert ctrl._hp.start_planning == case['start_planning_after_ctor'], rel
        seen['ok'] += 1
    assert seen == {'ok': 6, 'raises': 10, 'unloadable': 2}


def test_override_errors(golden_dir):
    want = _meta(golden_dir, 'hparams')['errors']
    fake = make_fake_predictor_class(5, 16, 16)
    bad = {'identical_to_default': {'iterations': 3}, 'unknown_key': {'no_such_param': 1},
           'list_for_scalar': {'T': [400, 200]}, 'wrong_type': {'num_samples': 'many'}}
    for label, pdict in bad.items():
        pdict = dict(pdict, predictor_class=fake)
        try:
            with quiet():
                PixelCostController(dict(AG), pdict, 0, 1)
            got = None
        except Exception as e:  # noqa
            got = type(e).__name__
        assert got == want[label], label


# ----------------------------------------------------------------------------- a5
def _hp_for(sampler_cls, **over):
    hp = HParams(replan_interval=0)
    for k, v in sampler_cls.get_default_hparams().items():
        hp.add_hparam(k, v)
    for k, v in over.items():
        if isinstance(v, list) and k == 'mean_bias':
            v = np.array(v)
        setattr(hp, k, v)
    return hp


def test_gaussian_sampler_matches_reference(golden_dir):
    meta, arrays = _meta(golden_dir, 'sampler'), _arrays(golden_dir, 'sampler')
    for case in meta['cases']:
        if case.get('kind') == 'corr':
            continue
        name = case['name']
        hp = _hp_for(GaussianCEMSampler, **case['over'])
        with quiet():
            sigma0 = cu.construct_initial_sigma(hp, case['adim'], case['t'])
            smp = GaussianCEMSampler(hp, case['adim'], 5)
            np.random.seed(case['seed'])
            a0 = smp.sample_initial_actions(case['t'], 32, np.zeros(5))
            elites = a0[np.argsort(np.abs(a0).sum((1, 2)))[:10]].copy()
            a1 = smp.sample_next_actions(32, elites, np.arange(10.))
        np.testing.assert_array_equal(sigma0, arrays[name + '/sigma0'])
        np.testing.assert_array_equal(a0, arrays[name + '/a0'])
        np.testing.assert_array_equal(a1, arrays[name + '/a1'])
        np.testing.assert_array_equal(smp._mean, arrays[name + '/mean'])
        np.testing.assert_array_equal(smp._sigma, arrays[name + '/sigma'])


def test_gaussian_reuse_mean(golden_dir):
    arrays = _arrays(golden_dir, 'sampler')
    hp = _hp_for(GaussianCEMSampler, rejection_sampling=False, reuse_mean=True, reduce_std_dev=0.5)
    with quiet():
        smp = GaussianCEMSampler(hp, 4, 5)
        np.random.seed(7)
        a0 = smp.sample_initial_actions(1, 16, np.zeros(5))
        smp.log_best_action(a0[0, 0], a0[:5, 1:])
        b0 = smp.sample_initial_actions(2, 16, np.zeros(5))
    np.testing.assert_array_equal(a0, arrays['gauss_reuse_mean/a0'])
    np.testing.assert_array_equal(b0, arrays['gauss_reuse_mean/b0'])
    np.testing.assert_array_equal(smp._mean, arrays['gauss_reuse_mean/mean'])


def test_correlated_sampler_matches_reference(golden_dir):
    meta, arrays = _meta(golden_dir, 'sampler'), _arrays(golden_dir, 'sampler')
    n = 0
    for case in meta['cases']:
        if case.get('kind') != 'corr':
            continue
        n += 1
        hp = _hp_for(CorrelatedNoiseSampler, **case['over'])
        with quiet():
            smp = CorrelatedNoiseSampler(hp, 4, 5)
            np.random.seed(case['seed'])
            a0 = smp.sample_initial_actions(0, 24, None)
            a1 = smp.sample_next_actions(24, a0[:8].copy(), np.linspace(1., 3., 8))
        np.testing.assert_array_equal(a0, arrays[case['name'] + '/a0'])
        np.testing.assert_array_equal(a1, arrays[case['name'] + '/a1'])
    assert n == 3


def test_controller_utils_helpers(golden_dir):
    meta, arrays = _meta(golden_dir, 'sampler'), _arrays(golden_dir, 'sampler')
    hp = _hp_for(GaussianCEMSampler)
    np.testing.assert_array_equal(cu.truncate_movement(arrays['helpers/trunc3_in'].copy(), hp),
                                  arrays['helpers/trunc3_out'])
    np.testing.assert_array_equal(cu.truncate_movement(arrays['helpers/trunc2_in'].copy(), hp),
                                  arrays['helpers/trunc2_out'])
    hp_o = _hp_for(GaussianCEMSampler, action_order=['x', 'y', 'z', 'theta'])
    np.testing.assert_array_equal(cu.truncate_movement(arrays['helpers/trunc3_in'].copy(), hp_o),
                                  arrays['helpers/trunc3_order_out'])
    np.testing.assert_array_equal(cu.make_blockdiagonal(arrays['helpers/cov_in'], 5, 4),
                                  arrays['helpers/blockdiag_out'])
    np.testing.assert_array_equal(cu.discretize(arrays['helpers/disc_in'].copy(), 4, 5, [2, 3]),
                                  arrays['helpers/disc_out'])
    hp_r = _hp_for(GaussianCEMSampler, reuse_cov=0.25)
    hp_r.replan_interval = 3
    assert meta['reuse_cov_with_defaults'] == 'TypeError'
    with pytest.raises(TypeError), quiet():
        cu.reuse_cov(arrays['helpers/reuse_cov_in'], 4, hp_r)
    hp_r.del_hparam('reduce_std_dev')
    with quiet():
        out = cu.reuse_cov(arrays['helpers/reuse_cov_in'], 4, hp_r)
    np.testing.assert_array_equal(out, arrays['helpers/reuse_cov_out'])


# ----------------------------------------------------------------------------- a8-a11 (host path)
def test_host_cost_path_matches_reference(golden_dir):
    meta, arrays = _meta(golden_dir, 'cost'), _arrays(golden_dir, 'cost')
    for case in meta['cases']:
        name, H, W, nd, M, T = (case[k] for k in ('name', 'H', 'W', 'ndesig', 'M', 'T'))
        fake = make_fake_predictor_class(T, H, W)
        pol = {'predictor_class': fake, 'repeat': 1, 'rejection_sampling': False, 'verbose': False,
               'num_samples': M + 1}
        if nd != 1:
            pol['designated_pixel_count'] = nd
        if T != 5:
            pol['nactions'] = T
        if case['finalweight'] != 10.:
            pol['finalweight'] = case['finalweight']
        if case['only_take_first_view']:
            pol['only_take_first_view'] = True
        with quiet():
            ctrl = PixelCostController(dict(AG, image_height=H, image_width=W), pol, 0, 1)
            ctrl.reset()
        rs = np.random.RandomState(case['seed'])
        distrib = rs.uniform(0.0, 1.0, (M, T, 1, H, W, nd)).astype(np.float32)
        goal = rs.randint(-3, max(H, W) + 3, (1, nd, 2))
        desig = rs.randint(-3, max(H, W) + 3, (1, nd, 2))
        ctrl._goal_pix, ctrl._desig_pix = goal, desig
        with quiet():
            scores = ctrl._eval_pixel_cost(0, distrib, None)
            grid0 = ctrl._get_distancegrid(goal[0, 0])
            onehot = ctrl._switch_on_pix(desig)
        np.testing.assert_array_equal(scores, arrays[name + '/scores'])
        np.testing.assert_array_equal(scores.argsort(), arrays[name + '/argsort'])
        np.testing.assert_allclose(grid0, arrays[name + '/grid0'], rtol=0, atol=1e-12)
        assert list(onehot.shape) == case['onehot_shape']
        np.testing.assert_array_equal(np.argwhere(onehot != 0), arrays[name + '/onehot_nonzero'])


# ----------------------------------------------------------------------------- a3 a4
def test_act_traces_match_reference(golden_dir):
    meta, arrays = _meta(golden_dir, 'act'), _arrays(golden_dir, 'act')
    for case in meta['cases']:
        name, T, H, W = case['name'], case['T'], case['H'], case['W']
        fake = make_fake_predictor_class(T, H, W)
        over = dict(case['over'])
        pol = {'predictor_class': fake, 'verbose': False}
        if case['correlated']:
            pol['sampler'] = CorrelatedNoiseSampler
            pol.update(over)
            pol['nactions'] = T
        else:
            pol.update(dict(rejection_sampling=False, repeat=1))
            pol.update(over)
        ag = dict(AG, adim=case['adim'], image_height=H, image_width=W)
        with quiet():
            ctrl = PixelCostController(ag, pol, 0, 1)
            if 'append_action' in over:
                ctrl._adim = case['sampler_adim']
            ctrl.reset()
        np.random.seed(case['seed'])
        rs = np.random.RandomState(case['seed'])
        n_steps = case['n_steps']
        images = rs.randint(0, 256, (n_steps + 1, 1, H, W, 3)).astype(np.uint8)
        states = rs.normal(0, 0.1, (n_steps + 1, 5))
        trace = []
        for t in range(n_steps):
            with quiet():
                out = ctrl.act(t=t, i_tr=0, desig_pix=case['desig'], goal_pix=case['goal'],
                               images=images[:t + 1], state=states[:t + 1])
            np.testing.assert_array_equal(out['actions'], arrays['%s/t%d/action' % (name, t)])
            for k, v in out['plan_stat'].items():
                np.testing.assert_array_equal(v, arrays['%s/t%d/%s' % (name, t, k)])
            if ctrl._best_indices is not None:
                # elite index set, bit-exact
                np.testing.assert_array_equal(ctrl._best_indices,
                                              arrays['%s/t%d/best_indices' % (name, t)])
            trace.append(ctrl._t_since_replan)
        assert trace == case['t_since_replan']
        assert list(fake.calls) == case['predictor_calls']


def test_pred_util_context_and_chunking_match_reference(golden_dir):
    """Row a13: get_context / rollout_predictions against outputs of the reference's pred_util."""
    import types
    from visual_foresight_amd.video_prediction.pred_util import get_context, rollout_predictions
    g, meta = _arrays(golden_dir, 'pred_util'), _meta(golden_dir, 'pred_util')
    images, state = g['ctx/images'], g['ctx/state']
    f, s = get_context(2, 4, state, images, types.SimpleNamespace(state_append=[0.5, -1.0]))
    assert f.dtype == np.float32 and f.shape == g['ctx/frames_out'].shape
    np.testing.assert_array_equal(f, g['ctx/frames_out'])
    np.testing.assert_array_equal(s, g['ctx/states_out'])
    f3, s3 = get_context(2, 3, state, images, None)
    np.testing.assert_array_equal(f3, g['ctx/frames_out_t3'])
    np.testing.assert_array_equal(s3, g['ctx/states_out_t3'])

    seen = []

    def recording_predictor(input_images=None, input_state=None, input_actions=None, input_one_hot_images=None):
        seen.append(np.array(input_actions))
        b = input_actions.shape[0]
        tag = input_actions.sum((1, 2))
        return tag[:, None] * np.ones((b, 2)), tag[:, None] + np.ones((b, 3)), None

    gi, gd, gs = rollout_predictions(recording_predictor, 200, g['roll/actions'], f, s, None)
    assert [list(x.shape) for x in seen] == meta['chunk_shapes'] and len(seen) == meta['n_runs']
    assert np.abs(seen[-1][50:]).sum() == g['roll/last_chunk_sum_padded_rows'][0] == 0.0
    np.testing.assert_array_equal(np.concatenate(gi, 0), g['roll/gen_images'])
    np.testing.assert_array_equal(np.concatenate(gd, 0), g['roll/gen_distrib'])
    assert all(x is None for x in gs) == meta['gen_state_all_none']


def test_all_views_config_fails_like_the_reference():
    """``experiments/robonet/view_generalization/all_views.py:25-40``: a 5-dim ``initial_std`` (the correlated sampler takes
    its action dimension from it) against the 4 entries of ``context_action_weight`` - the product in the reference's
    ``cem_base_controller.py:143-144`` cannot broadcast, so the first ``act()`` of that experiment raises there; same here
    (the file's ``model_params_path`` / ``model_restore_path`` are unknown hyper-parameters on top of that)."""
    from visual_foresight_amd.policy.cem_controllers.samplers import CorrelatedNoiseSampler
    fake = make_fake_predictor_class(13, 16, 16)
    pol = {'replan_interval': 13, 'verbose_every_iter': True, 'zeros_for_start_frames': False, 'num_samples': 600,
           'selection_frac': 2. / 3, 'predictor_propagation': True, 'nactions': 13, 'sampler': CorrelatedNoiseSampler,
           'context_action_weight': [2, 2, 0.05, 2], 'initial_std': [0.05, 0.05, 0.2, np.pi / 10, 1],
           'predictor_class': fake}
    ag = dict(AG, adim=5, image_height=16, image_width=16)
    with quiet():
        ctrl = PixelCostController(ag, pol, 0, 1)
        ctrl.reset()
        with pytest.raises(ValueError, match='broadcast'):
            ctrl.act(t=0, i_tr=0, desig_pix=[[8, 8]], goal_pix=[[3, 12]],
                     images=np.zeros((1, 1, 16, 16, 3), np.uint8), state=np.zeros((1, 5)))
    with pytest.raises(Exception):
        with quiet():
            PixelCostController(dict(ag), dict(pol, model_params_path='x.json'), 0, 1)
