"""bench.py on the GPU box the way the driver runs it: alone, self-launched with 2 ranks, and under
``python -m torch.distributed.run`` with 2 ranks (on a one-GPU box the ranks share the GPU over gloo and the line says
so; on a multi-GPU box they are RCCL ranks).  Checks the contract fields, the `collective` evidence block, and that the
plan found does not depend on the rank count (scores are bit-identical across shardings: same best score)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--samples', '24', '--steps', '2', '--warmup', '1', '--no-alt', '--no-cpu-baseline']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _line(cmd, env=None):
    out = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line on stdout, got %d' % len(lines)
    return json.loads(lines[0])


@pytest.fixture(scope='module')
def single():
    return _line([sys.executable, 'bench.py', '--gpus', '1'] + ARGS)


def _check_common(r, n):
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in r, key
    assert r['n_gpus'] == n and r['steps'] == 2 and r['warmup'] == 1 and r['value'] > 0
    assert r['dtype'] == 'f32' and r['data'] == 'synthetic' and r['vs_baseline'] is None
    assert r['config']['num_samples'] == 24 and r['config']['samples_per_rank'] == 24 // n
    assert r['roofline']['bound'] == 'mfma' and 0 < r['roofline']['frac'] < 1.2


def test_single_rank_line(single):
    _check_common(single, 1)
    assert single['collective'] is None
    assert single['elites_identical_across_ranks'] is True and len(single['scores_sha']) == 16
    assert single['scores_sha_per_rank'] is None


@pytest.mark.parametrize('launcher', ['self', 'torchrun'])
def test_two_rank_line_and_collective_evidence(single, launcher):
    if launcher == 'self':
        env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
        r = _line([sys.executable, 'bench.py', '--gpus', '2'] + ARGS, env=env)
    else:       # exactly the driver's command line for N > 1
        r = _line([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                   '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), 'bench.py', '--gpus', '2'] + ARGS)
    _check_common(r, 2)
    c = r['collective']
    assert c['world_size'] == 2 and c['backend'] in ('nccl', 'gloo') and len(c['ranks']) == 2
    assert sorted(x['rank'] for x in c['ranks']) == [0, 1] and len({x['pid'] for x in c['ranks']}) == 2
    if c['backend'] == 'nccl':
        assert c['distinct_gpus'] == 2 and c['devices'] == [0, 1]
    else:
        assert 'gloo dry run' in r['config']['sharding']
    a = c['allgather_ms_per_cem_iter']
    assert a['calls_per_rank'] == 2 * 3, 'one all-gather per CEM iteration of every timed call'
    assert 0 < a['mean_over_ranks'] <= a['max_over_ranks'] and c['bytes_per_rank_per_allgather'] == 12 * 2 * 8
    assert all(v == '1' for v in c['host_threads_per_rank'].values())
    # the same candidates, the same bits: sharding cannot change the plan that is found
    assert r['best_score_last_plan'] == single['best_score_last_plan']
    # ... and the record proves it by itself: every rank hashed every timed call's scores / elites / action
    assert r['elites_identical_across_ranks'] is True
    assert r['scores_sha_per_rank'] == [r['scores_sha']] * 2
    assert r['scores_sha'] == single['scores_sha'], 'the 2-rank job planned something else than one GPU'
    import bench
    assert r['scores_sha_over'] == single['scores_sha_over'] and bench.compare_lines(r, single) == 'equal'
    assert c['rank_id_allgather_verified_on_every_rank'] is True
    assert (c['rccl_version'] is not None) == (c['backend'] == 'nccl')
    assert all(x['avg_launch_us'] > 0 for x in c['ranks'])
    lo, hi = c['avg_launch_us_min_max_over_ranks']
    assert 0 < lo <= hi


def test_lines_with_different_warmups_are_not_comparable(single):
    """``scores_sha`` is one string per (workload, candidates, --steps, --warmup, seed): a line run with another warm-up
    hashes other planning calls - the comparison says 'not comparable', never 'mismatch'."""
    import bench
    other = _line([sys.executable, 'bench.py', '--gpus', '1', '--samples', '24', '--steps', '2', '--warmup', '2', '--no-alt',
                   '--no-cpu-baseline'])
    assert other['scores_sha_over']['warmup'] == 2 and single['scores_sha_over']['warmup'] == 1
    assert other['scores_sha'] != single['scores_sha']
    assert bench.compare_lines(other, single) == 'not comparable'
    assert bench.compare_lines(single, dict(single)) == 'equal'
    assert bench.compare_lines(single, dict(single, scores_sha='0' * 16)) == 'mismatch'


def test_build_then_smoke_in_one_process():
    """The driver's hooks back to back in ONE fresh process: build() loads libvf_hip.so before anything has imported
    torch - the order in which the two HIP runtimes of the image (PyTorch's bundled one, /opt/rocm's) used to end up
    both loaded, the second one blind ("no ROCm-capable device"); `_lib.load_library` imports torch first now."""
    out = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.build(); g.smoke()'], cwd=REPO,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:]
    assert 'smoke: frame err' in out.stdout
