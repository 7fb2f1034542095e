"""The drop-in boundary is a C ABI: prove it from C.

``tools/c_host/vf_c_host.c`` includes nothing but ``include/vf_hip.h`` and the HIP runtime's C API; it is built here with
``gcc`` and drives ``vf_create -> vf_load_weights -> vf_set_context -> vf_rollout -> vf_device_status -> vf_export`` on the
inputs of a Python-side planning call.  The same engine behind the same boundary: scores, per-task scores, predicted
frames, normalised distributions and states must come back bit for bit as ``HipVPredEvaluation`` returns them."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import pixel_cost  # noqa: E402  (input construction only)
from visual_foresight_amd import _lib  # noqa: E402
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights  # noqa: E402

pytestmark = pytest.mark.gpu


def build_c_host(out_dir):
    exe = os.path.join(str(out_dir), 'vf_c_host')
    rocm = '/opt/rocm'
    cmd = ['gcc', '-O2', '-std=c11', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(REPO, 'include'),
           '-I', os.path.join(rocm, 'include'), os.path.join(REPO, 'tools', 'c_host', 'vf_c_host.c'), _lib.LIB_PATH,
           '-L' + os.path.join(rocm, 'lib'), '-lamdhip64', '-Wl,-rpath,' + os.path.join(rocm, 'lib'), '-o', exe]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    return exe


@pytest.mark.parametrize('nd,M,T', [(1, 12, 4), (2, 5, 3)])
def test_c_host_reproduces_the_python_path_bit_for_bit(tmp_path, nd, M, T):
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    H = W = 64
    nc, adim, sdim, nex = 2, 4, 5, 3
    cfg = CdnaConfig(height=H, width=W, ndesig=nd, sequence_length=T + nc)
    weights = CdnaWeights.random(cfg, seed=11, bias_scale=0.05, ln_jitter=0.1)
    rs = np.random.RandomState(7)
    desig = rs.randint(0, H, (1, nd, 2))
    ctx = {'context_frames': rs.randint(0, 256, (nc, 1, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (nc - 1, adim)), 'context_states': rs.normal(0, 0.1, (nc, sdim)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib(desig, nc, 1, H, W, nd)}
    actions = rs.normal(0, 0.1, (M, T, adim))
    goal = rs.randint(0, H, (1, nd, 2))
    fw = 7.5

    # ---- the Python host of the same library
    hp = dict(designated_pixel_count=nd, run_batch_size=M, adim=adim, sdim=sdim, image_height=H, image_width=W,
              sequence_length=T + nc)
    pred = HipVPredEvaluation('', hp)
    pred.restore(weights)
    scores, per_task = pred.score(ctx, {'actions': actions}, goal, finalweight=fw)
    out = pred(ctx, {'actions': actions[:nex]})

    # ---- the C host
    blob = np.concatenate([v.ravel() for v in weights.tensors.values()]).astype(np.float32)
    inp, outp = tmp_path / 'in.bin', tmp_path / 'out.bin'
    with open(inp, 'wb') as f:
        np.array([H, W, adim, sdim, nd, nc, T + nc, M, nex, blob.size], np.int32).tofile(f)
        blob.tofile(f)
        ctx['context_frames'].tofile(f)
        ctx['context_states'].astype(np.float32).tofile(f)
        ctx['context_actions'].astype(np.float32).tofile(f)
        ctx['context_pixel_distributions'].astype(np.float32).tofile(f)
        actions.astype(np.float32).tofile(f)
        goal.astype(np.int32).tofile(f)
        np.array([fw], np.float32).tofile(f)
    exe = build_c_host(tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(_lib.LIB_PATH) + ':' + os.environ.get('LD_LIBRARY_PATH', ''))
    proc = subprocess.run([exe, str(inp), str(outp)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env,
                          timeout=600)
    assert proc.returncode == 0, proc.stdout
    assert 'rolled through the C ABI' in proc.stdout
    raw = np.fromfile(outp, np.uint8)
    off = [0]

    def take(dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = raw[off[0]:off[0] + n].view(dtype).reshape(shape)
        off[0] += n
        return a
    c_scores, c_per_task = take(np.float64, (M,)), take(np.float64, (M, nd))
    c_frames, c_distrib = take(np.float32, (nex, T, 1, H, W, 3)), take(np.float32, (nex, T, 1, H, W, nd))
    c_states = take(np.float32, (nex, T, sdim))
    assert off[0] == raw.size
    np.testing.assert_array_equal(c_scores, scores)
    np.testing.assert_array_equal(c_per_task, per_task)
    np.testing.assert_array_equal(c_frames, out['predicted_frames'])
    np.testing.assert_array_equal(c_distrib, out['predicted_pixel_distributions'])
    np.testing.assert_array_equal(c_states, out['predicted_states'])
    assert np.isfinite(c_scores).all() and np.ptp(c_scores) > 0
