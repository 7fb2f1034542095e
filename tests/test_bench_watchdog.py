"""bench.py's rank watchdog: one dead rank must not leave the others hanging in a collective."""
import subprocess
import sys
import time

import bench


def test_a_failing_rank_stops_the_others():
    sleeper = [sys.executable, '-c', 'import time; time.sleep(120)']
    procs = [subprocess.Popen(sleeper), subprocess.Popen([sys.executable, '-c', 'import sys; sys.exit(3)']),
             subprocess.Popen(sleeper)]
    t0 = time.monotonic()
    rc = bench.wait_ranks(procs, grace=5.0, poll=0.05)
    assert rc == 3
    assert time.monotonic() - t0 < 30
    assert all(p.poll() is not None for p in procs)


def test_all_ranks_succeeding_returns_zero():
    procs = [subprocess.Popen([sys.executable, '-c', 'pass']) for _ in range(3)]
    assert bench.wait_ranks(procs, poll=0.05) == 0


def test_metric_label_names_the_workload():
    import types
    b = bench.Bench.__new__(bench.Bench)
    b.args = types.SimpleNamespace(workload='c2', samples=0, scaling='strong')
    b.M, b.T, b.H, b.W, b.ncam, b.draws = 200, 13, 64, 64, 1, 0
    assert b.metric_label() == 'predicted frames/sec (whole node), 200-sample x 13-step x 64x64 CEM'
    b.args = types.SimpleNamespace(workload='c5', samples=125, scaling='strong')
    b.M, b.T, b.H, b.W, b.ncam, b.draws = 125, 15, 128, 128, 1, 5
    assert 'workload c5' in b.metric_label() and '125-sample' in b.metric_label() and '5 latent draws' in b.metric_label()
