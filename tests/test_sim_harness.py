"""The rollout harness drives a reference-shaped config end to end (CPU: fake predictor)."""
import contextlib
import io
import os
import pickle

import numpy as np
import pytest

from tests.helpers.fake_predictor import make_fake_predictor_class
from visual_foresight_amd.policy.cem_controllers import PixelCostController
from visual_foresight_amd.sim import Sim, SyntheticAgent, SyntheticPushEnv


def _config(tmp_path, predictor_class, T_plan=5, steps=5):
    agent = {'type': SyntheticAgent, 'env': (SyntheticPushEnv, {'seed': 3}), 'data_save_dir': str(tmp_path),
             'T': steps, 'image_height': 16, 'image_width': 16}
    policy = {'type': PixelCostController, 'predictor_class': predictor_class, 'replan_interval': 3,
              'num_samples': 24, 'rejection_sampling': False, 'repeat': 1, 'verbose': False}
    return {'agent': agent, 'policy': policy, 'start_index': 0, 'end_index': 1, 'save_data': True,
            'save_raw_images': True, 'ngroup': 1000}


def test_sim_runs_and_writes_reference_layout(tmp_path):
    cfg = _config(tmp_path, make_fake_predictor_class(5, 16, 16))
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        results = Sim(cfg).run()
    assert len(results) == 2 and all('final_goal_distance' in r for r in results)
    traj = os.path.join(str(tmp_path), 'train', 'traj_group0', 'traj1')
    for name in ('agent_data.pkl', 'obs_dict.pkl', 'policy_out.pkl'):
        assert os.path.exists(os.path.join(traj, name))
    policy_out = pickle.load(open(os.path.join(traj, 'policy_out.pkl'), 'rb'))
    assert len(policy_out) == 5
    assert policy_out[0]['actions'].shape == (4,) and np.all(policy_out[0]['actions'] == 0)   # t < start_planning
    assert 'scores_itr2' in policy_out[1]['plan_stat']
    obs = pickle.load(open(os.path.join(traj, 'obs_dict.pkl'), 'rb'))
    assert obs['state'].shape == (6, 5) and 'images' not in obs
    # frames: images{c}/im_{t}.png as the reference writes them (simulator.py:82-86), lossless RGB
    from visual_foresight_amd.utils.png import read_png
    assert sorted(os.listdir(os.path.join(traj, 'images0'))) == sorted('im_%d.png' % t for t in range(6))
    blob = open(os.path.join(traj, 'images0', 'im_5.png'), 'rb').read()
    assert blob[:8] == b'\x89PNG\r\n\x1a\n' and blob[12:16] == b'IHDR' and blob[-8:-4] == b'IEND'
    frame = read_png(os.path.join(traj, 'images0', 'im_5.png'))
    assert frame.shape == (16, 16, 3) and frame.dtype == np.uint8


def test_png_roundtrip_and_header(tmp_path):
    import struct
    from visual_foresight_amd.utils.png import read_png, write_png
    rs = np.random.RandomState(0)
    for shape in ((48, 64, 3), (1, 1, 3), (7, 5, 3)):
        img = rs.randint(0, 256, shape).astype(np.uint8)
        path = os.path.join(str(tmp_path), 'x.png')
        write_png(path, img)
        np.testing.assert_array_equal(read_png(path), img)
        blob = open(path, 'rb').read()
        w, h, depth, ctype = struct.unpack('>IIBB', blob[16:26])
        assert (w, h, depth, ctype) == (shape[1], shape[0], 8, 2)          # 8-bit truecolour
    with pytest.raises(ValueError):
        write_png(path, np.zeros((4, 4), np.uint8))


@pytest.mark.gpu
def test_sim_with_hip_predictor_and_propagation(tmp_path):
    cfg = _config(tmp_path, None, steps=4)
    cfg['agent'].update(image_height=32, image_width=32)
    cfg['policy'].pop('predictor_class')
    cfg['policy'].update(predictor_propagation=True, replan_interval=2, nactions=4, iterations=2)
    cfg['end_index'] = 0
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        sim = Sim(cfg)
        sim.run()
    chosen = sim.policy._chosen_distrib
    assert chosen.shape == (4, 1, 32, 32, 1)
    np.testing.assert_allclose(chosen.sum(axis=(2, 3)), 1.0, atol=1e-5)


def test_stochastic_predictor_host_logic():
    """Latent draws fold into the sample axis draw-minor; context actions get zero latents (no GPU needed)."""
    from visual_foresight_amd.video_prediction.stochastic_predictor import StochasticHipPredictor
    cls = StochasticHipPredictor.with_options(n_latent=3, zdim=2, latent_seed=7)
    assert cls.options == {'n_latent': 3, 'zdim': 2, 'latent_seed': 7} and StochasticHipPredictor.options == {}
    p = cls.__new__(cls)            # host-side helpers only: the engine needs a GPU
    p.n_latent, p.zdim, p.adim, p.latent_seed, p._calls = 3, 2, 4, 7, 0
    z = p.draw_latents(5)
    assert z.shape == (3, 5, 2)
    np.testing.assert_array_equal(z, p.draw_latents(5))         # same planning call -> same draws
    p._calls = 1
    assert not np.array_equal(z, p.draw_latents(5))             # next planning call -> new draws
    actions = np.arange(2 * 5 * 4, dtype=np.float64).reshape(2, 5, 4)
    p._z = z
    ctx, aug = p._prepare({'context_actions': np.ones((1, 4))}, actions)
    assert aug.shape == (6, 5, 6) and ctx['context_actions'].shape == (1, 6)
    np.testing.assert_array_equal(ctx['context_actions'][0], [1, 1, 1, 1, 0, 0])
    for m in range(2):
        for d in range(3):
            np.testing.assert_array_equal(aug[m * 3 + d, :, :4], actions[m])
            np.testing.assert_array_equal(aug[m * 3 + d, :, 4:], z[d])


def test_trajectory_layout_matches_reference_golden(tmp_path, golden_dir):
    """tests/golden/traj_layout.*: what the reference's own ``Sim._save_raw_data`` (``simulator.py:64-93``) wrote
    for a synthetic trajectory (tools/make_golden.py, cv2.imwrite recorded): same tree, same file names, same
    pickles, and every PNG decodes to the frame the reference handed to OpenCV (BGR there, RGB in the file)."""
    import json
    import types
    from visual_foresight_amd.utils.png import read_png
    meta = json.load(open(os.path.join(golden_dir, 'traj_layout.json')))
    gold = np.load(os.path.join(golden_dir, 'traj_layout.npz'))
    obs = {'images': gold['images'], 'state': gold['state']}
    fake = types.SimpleNamespace(agentparams={'data_save_dir': str(tmp_path)}, _hyperparams={'ngroup': meta['ngroup']},
                                 task_mode='train')
    policy_out = [{'actions': np.zeros(4)} for _ in range(meta['n_policy_out'])]
    Sim._save_raw_data(fake, meta['itr'], dict(meta['agent_data']), obs, policy_out)
    root = str(tmp_path)
    files = sorted(os.path.relpath(os.path.join(d, f), root) for d, _, fs in os.walk(root) for f in fs)
    dirs = sorted(os.path.relpath(os.path.join(d, x), root) for d, xs, _ in os.walk(root) for x in xs)
    assert dirs == meta['dirs']
    assert files == sorted(meta['files_on_disk'] + meta['imwrite_paths'])
    traj = os.path.join(root, 'train', 'traj_group2', 'traj23')
    for rel in meta['imwrite_paths']:
        name = os.path.relpath(rel, os.path.join('train', 'traj_group2', 'traj23'))
        bgr = gold['imwrite/' + name]
        np.testing.assert_array_equal(read_png(os.path.join(root, rel)), bgr[:, :, ::-1])
    back = pickle.load(open(os.path.join(traj, 'obs_dict.pkl'), 'rb'))
    assert sorted(back.keys()) == meta['obs_dict_keys']
    np.testing.assert_array_equal(back['state'], gold['state'])
    assert pickle.load(open(os.path.join(traj, 'agent_data.pkl'), 'rb')) == meta['agent_data']
    assert len(pickle.load(open(os.path.join(traj, 'policy_out.pkl'), 'rb'))) == meta['n_policy_out']
