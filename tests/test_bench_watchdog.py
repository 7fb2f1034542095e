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
    b.M, b.T, b.H, b.W, b.ncam, b.draws, b.ndesig = 200, 13, 64, 64, 1, 0, 1
    assert b.metric_label() == 'predicted frames/sec (whole node), 200-sample x 13-step x 64x64 CEM'
    b.args = types.SimpleNamespace(workload='c5', samples=125, scaling='strong')
    b.M, b.T, b.H, b.W, b.ncam, b.draws = 125, 15, 128, 128, 1, 5
    assert 'workload c5' in b.metric_label() and '125-sample' in b.metric_label() and '5 latent draws' in b.metric_label()


def test_launcher_env_caps_host_threads_and_never_needs_torch():
    """The 8-GPU pre-flight: the self-launcher counts GPUs in a throw-away child (its own process stays GPU-free) and
    gives every rank one host math thread."""
    import types
    args = types.SimpleNamespace(gpus=8)
    env = bench.rank_env(args, have=8, port=12345)
    assert env['WORLD_SIZE'] == '8' and env['MASTER_ADDR'] == '127.0.0.1' and env['MASTER_PORT'] == '12345'
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    for k in bench.HOST_THREAD_CAPS:
        assert env[k] == '1'
    assert 'VF_BENCH_BACKEND' not in env                       # 8 GPUs for 8 ranks: RCCL
    assert bench.rank_env(args, have=1, port=1)['VF_BENCH_BACKEND'] == 'gloo'
    assert bench.count_gpus_in_child() == 0 or bench.count_gpus_in_child() >= 1    # runs, returns an int
    import inspect
    src = inspect.getsource(bench.spawn_ranks)
    assert 'import torch' not in src and 'torch.cuda' not in src


def test_a_failed_gpu_probe_aborts_instead_of_becoming_a_gloo_dry_run(monkeypatch):
    """A probe that times out or dies is NOT "0 GPUs": the launcher must stop (non-zero exit code, the child's stderr
    in the message) rather than silently run an 8-rank scaling job as a shared-GPU gloo dry run."""
    import subprocess
    import types

    def broken(*a, **kw):
        return subprocess.CompletedProcess(a, 1, stdout='', stderr='ImportError: libamdhip64.so')
    monkeypatch.setattr(bench.subprocess, 'run', broken)
    try:
        bench.count_gpus_in_child()
        raise AssertionError('a probe without an answer must raise')
    except bench.GpuProbeError as e:
        assert 'libamdhip64' in str(e)
    started = []
    monkeypatch.setattr(bench.subprocess, 'Popen', lambda *a, **kw: started.append(a))
    # (the launcher refuses to run in a process that has imported torch; other tests of this session have)
    monkeypatch.delitem(bench.sys.modules, 'torch', raising=False)
    assert bench.spawn_ranks(types.SimpleNamespace(gpus=8)) == 3 and not started

    def timeout(*a, **kw):
        raise subprocess.TimeoutExpired(a, 1)
    monkeypatch.setattr(bench.subprocess, 'run', timeout)
    assert bench.spawn_ranks(types.SimpleNamespace(gpus=8)) == 3 and not started


def test_plan_digests_are_order_and_content_sensitive():
    """The `scores_sha` of a bench line: equal plans hash equal, one changed score / elite / action bit does not."""
    import numpy as np
    from visual_foresight_amd.video_prediction.sharding import plan_digest, run_digest, gather_plan_digests
    ps = {'scores_itr0': np.arange(5.0), 'scores_itr1': np.arange(5.0) * 2, 'scores_itr10': np.ones(5)}
    d = plan_digest(ps, [1, 2], np.zeros(4))
    assert d == plan_digest(dict(reversed(list(ps.items()))), np.array([1, 2]), np.zeros(4))
    assert d != plan_digest(ps, [2, 1], np.zeros(4))
    assert d != plan_digest(ps, [1, 2], np.array([0, 0, 0, 1e-300]))
    ps2 = dict(ps, scores_itr1=np.nextafter(ps['scores_itr1'], 9.0))
    assert d != plan_digest(ps2, [1, 2], np.zeros(4))
    assert run_digest([d, d]) != run_digest([d]) and len(run_digest([d])) == 16
    assert gather_plan_digests([d]) == (True, [run_digest([d])])


def test_traffic_is_quoted_only_from_a_profile_of_this_library():
    """`roofline.traffic` comes from the newest committed PMC profile whose source hash equals the running library's
    (the counters cannot be read in-process); any other profile, workload or precision leaves it null."""
    import glob
    import json
    import os
    import types
    b = bench.Bench.__new__(bench.Bench)
    b.args = types.SimpleNamespace(workload='c2', samples=0)
    b.world = 1
    b.ndesig = 1
    matching = [p for p in sorted(glob.glob(os.path.join(bench.REPO, 'profiles', 'r*_hbm_traffic.json')), reverse=True)
                if json.load(open(p)).get('lib_sources_sha16') == bench.library_hash()]
    roof = {'traffic': None, 'avg_launch_us': 63000.0}
    b.attach_traffic(roof, 'fp32')
    if matching:
        prof = json.load(open(matching[0]))
        assert roof['traffic'] == prof['hbm_bytes_per_launch'] and os.path.basename(matching[0]) in roof['traffic_source']
        assert 0 < roof['hbm_frac_of_peak'] < 1
    else:
        assert roof['traffic'] is None
    for kw, prec in (({'workload': 'c4', 'samples': 0}, 'fp32'), ({'workload': 'c2', 'samples': 25}, 'fp32'),
                     ({'workload': 'c2', 'samples': 0}, 'bf16x6')):
        b.args = types.SimpleNamespace(**kw)
        other = {'traffic': None, 'avg_launch_us': 1.0}
        b.attach_traffic(other, prec)
        assert other['traffic'] is None
