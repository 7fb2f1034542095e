"""GPU tests of the C-ABI contract of include/vf_hip.h: in-band failure status, no allocation after
vf_create, the RCCL all-gather entry point, and the product's multi-rank path on real hardware."""
import ctypes
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import pixel_cost                                           # noqa: E402
from visual_foresight_amd import _lib                                   # noqa: E402
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights   # noqa: E402
from visual_foresight_amd.video_prediction.sharding import shard_bounds              # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, 'tests', 'helpers', 'gpu_rank_worker.py')


def _small(bs=6, T=2, precision='fp32'):
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    hp = dict(designated_pixel_count=1, run_batch_size=bs, adim=4, sdim=5, image_height=32, image_width=32,
              sequence_length=T + 2, precision=precision)
    pred = HipVPredEvaluation('', hp).restore()
    rs = np.random.RandomState(1)
    ctx = {'context_frames': rs.randint(0, 256, (2, 1, 32, 32, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (1, 4)), 'context_states': rs.normal(0, 0.1, (2, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib([[[16, 16]]], 2, 1, 32, 32, 1)}
    return pred, ctx, rs.normal(0, 0.1, (bs, T, 4))


def test_device_failure_is_reported_in_band():
    """A raised device status word (a tile gave up waiting) must turn every score into NaN and the Python
    wrapper into an exception - never into elite candidates; reading the status re-arms the engine."""
    pred, ctx, actions = _small()
    good, _ = pred.score(ctx, {'actions': actions}, [[[3, 20]]])
    _lib.check(pred._libh.vf_debug_poison_status(pred._handle))
    with pytest.raises(_lib.VfError, match='device status 1'):
        pred.score(ctx, {'actions': actions}, [[[3, 20]]])
    # the raw C entry point: scores come back as NaN, the status stays raised until it is read
    _lib.check(pred._libh.vf_debug_poison_status(pred._handle))
    with torch.cuda.device(pred.device):
        a = torch.from_numpy(actions.astype(np.float32)).to(pred.device)
        sc = torch.zeros(len(actions), dtype=torch.float64, device=pred.device)
        pt = torch.zeros((len(actions), 1), dtype=torch.float64, device=pred.device)
        for _ in range(2):
            pred._rollout_chunk(a, [[[3, 20]]], 10., sc, pt)
            assert torch.isnan(sc).all() and torch.isnan(pt).all()
    assert pred.device_status() == 1
    assert pred.device_status() == 0
    again, _ = pred.score(ctx, {'actions': actions}, [[[3, 20]]])
    np.testing.assert_array_equal(again, good)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x6'])
def test_nothing_is_allocated_after_create(precision):
    """vf_hip.h: every device buffer is allocated in vf_create.  Repeated restore() (weight hot-swap), context
    changes, ragged batches (schedule rebuilds) and exports leave the device's free memory untouched."""
    pred, ctx, actions = _small(bs=8, precision=precision)
    cfg = pred.cfg
    pred(ctx, {'actions': actions[:3]})
    free0 = None
    for i in range(4):          # round 0 warms PyTorch's own caching allocator (tensor sizes of this loop)
        pred.restore(CdnaWeights.random(cfg, seed=40 + i))
        for n in (8, 5, 8, 1):
            pred.score(ctx, {'actions': actions[:n]}, [[[3 + i, 20]]])
        pred.fetch_pixel_distributions(0)
        if i == 0:
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info(pred.device)[0]
    pred.restore(CdnaWeights.random(cfg, seed=40))
    s, _ = pred.score(ctx, {'actions': actions}, [[[3, 20]]])
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(pred.device)[0] == free0
    # a hot-swapped weight set is really in use: same scores as a fresh engine restored from it
    fresh, _, _ = _small(bs=8, precision=precision)
    fresh.restore(CdnaWeights.random(cfg, seed=40))
    np.testing.assert_array_equal(fresh.score(ctx, {'actions': actions}, [[[3, 20]]])[0], s)


def test_allgather_scores_entry_point():
    """vf_allgather_scores with a caller-owned RCCL communicator (world size 1 on the one-GPU box: the
    collective degenerates to a copy, which still exercises the dlopen binding and the call)."""
    pred, ctx, actions = _small()
    rccl = None
    for cand in ('librccl.so.1', 'librccl.so', '/opt/rocm/lib/librccl.so'):
        try:
            rccl = ctypes.CDLL(cand)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip('no RCCL library to build a communicator with')
    class NcclUniqueId(ctypes.Structure):           # nccl.h: struct { char internal[128]; }, passed by value
        _fields_ = [('internal', ctypes.c_char * 128)]

    uid = NcclUniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, NcclUniqueId, ctypes.c_int]
    with torch.cuda.device(pred.device):
        assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
        local = torch.arange(14, dtype=torch.float64, device=pred.device) * 0.5 + 1e-12
        out = torch.zeros(14, dtype=torch.float64, device=pred.device)
        _lib.check(pred._libh.vf_allgather_scores(pred._handle, comm, local.data_ptr(), 14, out.data_ptr(),
                                                  pred._stream()))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), local.cpu().numpy())
        assert pred._libh.vf_allgather_scores(pred._handle, None, local.data_ptr(), 14, out.data_ptr(), None) == -1
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(world, out_dir, num_samples, stochastic, vpred_batch_size=0):
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), str(out_dir),
                               str(num_samples), '1' if stochastic else '0', str(vpred_batch_size)], env=env)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [pickle.load(open(os.path.join(out_dir, 'gpu_rank%d_of%d.pkl' % (r, world)), 'rb')) for r in range(world)]


@pytest.mark.parametrize('num_samples,stochastic', [(23, False), (14, True)])
def test_two_ranks_of_the_real_predictor_match_one(tmp_path, num_samples, stochastic):
    """2 ranks x the real HIP predictor (deterministic and latent-draw) with predictor_propagation on: every
    rank sees the single-process scores, elites, actions and propagated distributions bit for bit."""
    single = _launch(1, tmp_path, num_samples, stochastic)[0]
    ranks = _launch(2, tmp_path, num_samples, stochastic)
    for r, res in enumerate(ranks):
        lo, hi = shard_bounds(num_samples, r, 2)
        assert res['rolled'] == hi - lo                     # each rank rolled only its own shard
        for a, b in zip(res['log'], single['log']):
            np.testing.assert_array_equal(a['action'], b['action'])
            for k in b['plan_stat']:
                np.testing.assert_array_equal(a['plan_stat'][k], b['plan_stat'][k])
            for k in ('best', 'chosen'):
                if b[k] is None:
                    assert a[k] is None
                else:
                    np.testing.assert_array_equal(a[k], b[k])
    assert single['log'][-1]['chosen'] is not None


@pytest.mark.parametrize('stochastic', [False, True])
def test_propagation_of_a_sample_that_is_no_longer_resident(tmp_path, stochastic):
    """``num_samples > vpred_batch_size`` (the reference's 600-sample configs with a 200-sample predictor batch):
    only every rank's last chunk stays on the device, so the propagated winner usually has to be rolled again -
    alone, on every rank - and the whole closed loop still equals the unchunked single-process run bit for bit."""
    single = _launch(1, tmp_path, 14, stochastic)[0]
    chunked = _launch(1, tmp_path, 14, stochastic, 4)[0]
    ranks = _launch(2, tmp_path, 14, stochastic, 3)
    for res in [chunked] + ranks:
        for a, b in zip(res['log'], single['log']):
            np.testing.assert_array_equal(a['action'], b['action'])
            for k in ('best', 'chosen'):
                if b[k] is None:
                    assert a[k] is None
                else:
                    np.testing.assert_array_equal(a[k], b[k])
    assert single['log'][-1]['chosen'] is not None


def test_n_gpus_in_process_lanes_match_one_engine():
    """``n_gpus=2`` inside ONE process (the reference's in-graph towers, setup_predictor.py:70,117-123): two
    engines ("lanes") roll contiguous shards - on this one-GPU box both on the same device, gathered through the
    host - and scores, per-task scores, exported predictions and the propagation fetch equal the single engine bit
    for bit, ragged shards and chunked lanes included.  Without the oversubscription switch a host with fewer GPUs
    than ``n_gpus`` is refused instead of silently planning on one GPU."""
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    H = W = 32
    T, M = 3, 23
    hp = dict(designated_pixel_count=2, run_batch_size=M, adim=4, sdim=5, image_height=H, image_width=W,
              sequence_length=T + 2)
    cfg = CdnaConfig(height=H, width=W, ndesig=2, sequence_length=T + 2)
    weights = CdnaWeights.random(cfg, seed=12, bias_scale=0.05, ln_jitter=0.1)
    one = HipVPredEvaluation('', hp).restore(weights)
    rs = np.random.RandomState(8)
    ctx = {'context_frames': rs.randint(0, 256, (2, 1, H, W, 3)).astype(np.uint8),
           'context_actions': rs.normal(0, 0.05, (1, 4)), 'context_states': rs.normal(0, 0.1, (2, 5)),
           'context_pixel_distributions': pixel_cost.one_hot_distrib([[[9, 9], [20, 3]]], 2, 1, H, W, 2)}
    actions = rs.normal(0, 0.1, (M, T, 4))
    goal = np.array([[[3, 20], [30, 1]]])
    want, want_pt = one.score(ctx, {'actions': actions}, goal)
    want_out = one(ctx, {'actions': actions})
    if torch.cuda.device_count() < 2:
        with pytest.raises(ValueError):
            HipVPredEvaluation('', hp, n_gpus=2)
    with pytest.raises(ValueError):
        HipVPredEvaluation('', hp, n_gpus=0)
    for n, bs in ((2, M), (3, 5)):
        lanes = HipVPredEvaluation('', dict(hp, oversubscribe_gpus=1, run_batch_size=bs), n_gpus=n, first_gpu=0)
        lanes.restore(weights)
        assert len(lanes._lanes) == n
        if torch.cuda.device_count() < n:
            assert lanes._use_rccl is False                 # lanes share a device: the gather goes through the host
        got, got_pt = lanes.score(ctx, {'actions': actions}, goal)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(got_pt, want_pt)
        sizes = [l._last_lo for l in lanes._lanes]
        assert sizes == sorted(sizes) and sizes[0] <= sizes[-1]
        # whole shards resident (bs == M) or only every lane's last chunk (a sample nobody holds is rolled again)
        for idx in (0, M // 2, M - 1, 3, 3):
            np.testing.assert_array_equal(lanes.fetch_pixel_distributions(idx),
                                          want_out['predicted_pixel_distributions'][idx])
        with pytest.raises(IndexError):
            lanes.fetch_pixel_distributions(M)
        out = lanes(ctx, {'actions': actions})
        for k in want_out:
            np.testing.assert_array_equal(out[k], want_out[k])
        assert lanes.device_status() == 0
        # fewer candidates than lanes: the lanes left empty forget what they held (round-3 advisor: a stale range must
        # not answer a fetch with a sample of an EARLIER rollout), and an empty candidate set is not an error
        few = lanes(ctx, {'actions': actions[:n - 1]})
        for k in want_out:
            np.testing.assert_array_equal(few[k], want_out[k][:n - 1])
        assert [l._last_M for l in lanes._lanes][n - 1] == 0
        np.testing.assert_array_equal(lanes.fetch_pixel_distributions(n - 2),
                                      want_out['predicted_pixel_distributions'][n - 2])
        with pytest.raises(IndexError):
            lanes.fetch_pixel_distributions(n - 1)
        none = lanes(ctx, {'actions': actions[:0]})
        assert none['predicted_frames'].shape == (0, T, 1, H, W, 3)
        assert torch.cuda.current_device() == 0             # every entry point leaves the caller's device current
    # devices first_gpu .. first_gpu + n_gpus - 1 are a promise: no silent wrap onto GPUs below first_gpu
    n_dev = torch.cuda.device_count()
    with pytest.raises(ValueError):
        HipVPredEvaluation('', hp, n_gpus=1, first_gpu=n_dev)
    wrapped = HipVPredEvaluation('', dict(hp, oversubscribe_gpus=1), n_gpus=1, first_gpu=n_dev)    # opt-in: wraps
    assert wrapped.device_index == 0


def test_grouped_allgather_entry_points():
    """``vf_comm_init_all`` + ``vf_allgather_scores_group`` + ``vf_comm_destroy`` on the GPUs this box has (one:
    the grouped collective degenerates to a copy but runs ncclCommInitAll / ncclGroupStart / ncclGroupEnd through
    the library's own RCCL binding); a device listed twice is refused in-band."""
    pred, ctx, actions = _small()
    lib, P = pred._libh, ctypes.c_void_p
    n = max(1, min(torch.cuda.device_count(), 2))
    devs = (ctypes.c_int32 * n)(*range(n))
    comms = (P * n)()
    rc = lib.vf_comm_init_all(n, devs, comms)
    if rc != 0:
        pytest.skip('no usable RCCL: %s' % lib.vf_last_error().decode())
    dup = (ctypes.c_int32 * 2)(0, 0)
    two = (P * 2)()
    assert lib.vf_comm_init_all(2, dup, two) == -1 and b'twice' in lib.vf_last_error()
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    hp = dict(designated_pixel_count=1, run_batch_size=4, adim=4, sdim=5, image_height=32, image_width=32,
              sequence_length=5)
    engines = [pred] + [HipVPredEvaluation('', hp, first_gpu=i) for i in range(1, n)]
    local, full = [], []
    for i, e in enumerate(engines):
        with torch.cuda.device(e.device):
            local.append(torch.arange(6, dtype=torch.float64, device=e.device) + 10.0 * i + 1e-12)
            full.append(torch.zeros(6 * n, dtype=torch.float64, device=e.device))
    arr = lambda items: (P * n)(*items)
    _lib.check(lib.vf_allgather_scores_group(n, arr([e._handle for e in engines]), arr([P(c) for c in comms]),
                                             arr([P(t.data_ptr()) for t in local]), 6,
                                             arr([P(t.data_ptr()) for t in full]), arr([e._stream() for e in engines])))
    want = np.concatenate([t.cpu().numpy() for t in local])
    for e, t in zip(engines, full):
        torch.cuda.synchronize(e.device)
        np.testing.assert_array_equal(t.cpu().numpy(), want)
    assert lib.vf_allgather_scores_group(n, None, None, None, 6, None, None) == -1
    for c in comms:
        assert lib.vf_comm_destroy(P(c)) == 0


def test_failed_first_rollout_of_a_context_does_not_poison_the_shared_cache():
    """A launch that is abandoned while it computes the context-only (batch-1) part of the network leaves those
    shared buffers half-written.  Reading the status must drop that cache on both sides of the boundary, so the
    retry with the SAME context recomputes it and equals a fresh engine - not finite scores from stale state."""
    pred, ctx, actions = _small()
    rs = np.random.RandomState(77)
    new_ctx = dict(ctx, context_frames=rs.randint(0, 256, (2, 1, 32, 32, 3)).astype(np.uint8),
                   context_states=rs.normal(0, 0.1, (2, 5)))
    pred.score(ctx, {'actions': actions}, [[[3, 20]]])              # fills the cache for the OLD context
    _lib.check(pred._libh.vf_debug_poison_status(pred._handle))
    with pytest.raises(_lib.VfError, match='device status 1'):      # first rollout of the NEW context is abandoned
        pred.score(new_ctx, {'actions': actions}, [[[3, 20]]])
    assert pred.device_status() == 0                                 # (read and re-armed by the failed call)
    retry, _ = pred.score(new_ctx, {'actions': actions}, [[[3, 20]]])
    fresh, _, _ = _small()
    want, _ = fresh.score(new_ctx, {'actions': actions}, [[[3, 20]]])
    np.testing.assert_array_equal(retry, want)
    # the raw C path: poison, roll (NaN), read the status, roll again without a new vf_set_context
    _lib.check(pred._libh.vf_set_context(pred._handle, *[t.data_ptr() for t in pred._ctx], pred._stream()))
    _lib.check(pred._libh.vf_debug_poison_status(pred._handle))
    with torch.cuda.device(pred.device):
        a = torch.from_numpy(actions.astype(np.float32)).to(pred.device)
        sc = torch.zeros(len(actions), dtype=torch.float64, device=pred.device)
        pt = torch.zeros((len(actions), 1), dtype=torch.float64, device=pred.device)
        pred._rollout_chunk(a, [[[3, 20]]], 10., sc, pt)
        assert torch.isnan(sc).all()
        assert pred.device_status() == 1
        pred._rollout_chunk(a, [[[3, 20]]], 10., sc, pt)
        np.testing.assert_array_equal(sc.cpu().numpy(), want)
