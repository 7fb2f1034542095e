"""Multi-rank planning on CPU (gloo, world_size 2 and 8): sharded scoring + all-gather gives every rank
the single-process score vector and therefore bit-identical elites and actions."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

from visual_foresight_amd.video_prediction.sharding import shard_bounds

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, 'tests', 'helpers', 'gloo_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(world, out_dir, num_samples, propagation):
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), str(out_dir),
                               str(num_samples), '1' if propagation else '0'], env=env)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    return [pickle.load(open(os.path.join(out_dir, 'rank%d_of%d.pkl' % (r, world)), 'rb')) for r in range(world)]


def test_shard_bounds_partition():
    for M in (1, 7, 200, 1000):
        for G in (1, 2, 3, 8):
            spans = [shard_bounds(M, r, G) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == M
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('world,num_samples,propagation', [(2, 24, False), (2, 23, True), (8, 203, True)])
def test_ranks_match_single_process(tmp_path, world, num_samples, propagation):
    """World 2 (even and ragged) and the 8-rank shape of the node with a ragged candidate count (203 = 3 x 26 + 5 x 25:
    the padded all-gather and the propagation fetch from whichever rank owns the winner)."""
    single = _launch(1, tmp_path, num_samples, propagation)[0]
    ranks = _launch(world, tmp_path, num_samples, propagation)
    # each rank rolled only its own shard
    for r, res in enumerate(ranks):
        assert set(res['evaluated']) == {shard_bounds(num_samples, r, world)}
    assert set(single['evaluated']) == {(0, num_samples)}
    # the self-proving fields of a scaling record (bench.py): every rank's digest of every planning call agrees, and
    # equals the single-process digest
    for res in ranks:
        assert res['identical_across_ranks'] is True
        assert res['scores_sha_per_rank'] == single['scores_sha_per_rank'] * world
    for res in ranks:
        for a, b in zip(res['log'], single['log']):
            np.testing.assert_array_equal(a['action'], b['action'])
            assert a['plan_stat'].keys() == b['plan_stat'].keys()
            for k in a['plan_stat']:
                np.testing.assert_array_equal(a['plan_stat'][k], b['plan_stat'][k])
            if b['best'] is None:
                assert a['best'] is None
            else:
                np.testing.assert_array_equal(a['best'], b['best'])
