#!/usr/bin/env python
"""Headline benchmark: visual-MPC CEM planning throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one planning call ``policy.act()`` = one full CEM (reference
``cem_base_controller.py:85-116``): sample -> upload -> roll every candidate through the CDNA
predictor -> pixel-distance cost on the device -> (all-gather) -> argsort -> refit, for
``iterations=3`` CEM iterations.  Workload = BASELINE.json configs[1]: 200 samples x horizon 13
x 64x64, pixel-distance cost, random-init CDNA predictor, synthetic context frames.  With
``--gpus N`` each rank rolls 200 samples (weak scaling: 200*N candidates per CEM iteration,
sharded by sample, one RCCL all-gather of the score rows per iteration).

Prints ONE JSON line on rank 0: ``value`` = predicted frames / second over the whole job
(M * T * iterations * K / wall), plus CEM iterations / second, the roofline of the dominant
kernel (fused conv-LSTM gate GEMM; fp32 MFMA bound) measured with HIP events on the launch
stream, and the CPU baseline (the oracle restatement timed on the host cores on a bounded
sample; rank 0, N=1 only).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--samples-per-gpu', type=int, default=200)
    ap.add_argument('--horizon', type=int, default=13)
    ap.add_argument('--iterations', type=int, default=3)
    ap.add_argument('--ncam', type=int, default=1, help='views (BASELINE configs[2] uses 2)')
    ap.add_argument('--ndesig', type=int, default=1, help='designated pixels per view')
    ap.add_argument('--selection-frac', type=float, default=0.0)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-samples', type=int, default=128)
    return ap.parse_args()


def cpu_baseline(weights, ctx, actions, goal):
    """Time the oracle (CPU restatement, PyTorch-CPU fp32) on a bounded sample of the workload."""
    import torch
    from oracle.cdna_predictor import OracleCdna
    from oracle import pixel_cost
    # oneDNN convs on many tiny images stop scaling (and oversubscribe badly) beyond a few dozen
    # threads, so the baseline uses at most 32 host cores and says so
    cores = min(os.cpu_count() or 1, 32)
    ora = OracleCdna(weights, torch.float32, threads=cores)
    T = actions.shape[1]
    ora.rollout(ctx['context_frames'], ctx['context_actions'], ctx['context_pixel_distributions'],
                ctx['context_states'], actions[:1, :2])                                   # warm-up
    t0 = time.perf_counter()
    _, d, _ = ora.rollout(ctx['context_frames'], ctx['context_actions'],
                          ctx['context_pixel_distributions'], ctx['context_states'], actions)
    pixel_cost.eval_pixel_cost(d, goal, 10.)
    dt = time.perf_counter() - t0
    return {'value': actions.shape[0] * T / dt, 'unit': 'predicted frames/s', 'cores': cores,
            'kind': 'port',
            'sample': '%d of the workload\'s samples x %d steps, one rollout + cost, %.1f s'
                      % (actions.shape[0], T, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # one rank per GPU over RCCL; VF_BENCH_BACKEND=gloo lets several ranks share one GPU for dry runs
    backend = os.environ.get('VF_BENCH_BACKEND', 'nccl')
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(dev_index)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)
    if world != args.gpus and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world), file=sys.stderr)
    dev = torch.device('cuda', dev_index)

    from visual_foresight_amd.policy.cem_controllers import PixelCostController
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation

    H = W = 64
    T, iters = args.horizon, args.iterations
    M = args.samples_per_gpu * (world if args.scaling == 'weak' else 1)
    ag_params = {'adim': 4, 'sdim': 5, 'image_height': H, 'image_width': W}
    if args.ncam != 1:
        ag_params['ncam'] = args.ncam
    # overrides equal to a default raise (reference policy.py:57-58), hence the conditionals
    policy = {'type': PixelCostController, 'repeat': 1, 'rejection_sampling': False, 'verbose': False,
              'vpred_batch_size': max(args.samples_per_gpu, 1)}
    if args.ncam == 1:
        policy['predictor_class'] = HipVPredEvaluation      # (ncam > 1: the multi-view default)
    if args.ndesig != 1:
        policy['designated_pixel_count'] = args.ndesig
    if args.selection_frac:
        policy['selection_frac'] = args.selection_frac
    if T != 5:
        policy['nactions'] = T
    if M != 200:
        policy['num_samples'] = M
    if iters != 3:
        policy['iterations'] = iters
    if policy['vpred_batch_size'] == 200:
        policy.pop('vpred_batch_size')
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl = PixelCostController(ag_params, policy, 0, 1)
        ctrl.reset()

    # synthetic inputs: identical on every rank
    np.random.seed(0)
    frames = np.random.RandomState(1).randint(0, 256, (2, args.ncam, H, W, 3)).astype(np.uint8)
    states = np.random.RandomState(2).normal(0, .1, (2, 5))
    npix = args.ncam * args.ndesig
    desig = [[32 - 3 * i, 32 + 2 * i] for i in range(npix)]
    goal = [[16 + 2 * i, 48 - 3 * i] for i in range(npix)]

    def plan():
        return ctrl.act(t=1, i_tr=0, desig_pix=desig, goal_pix=goal, images=frames, state=states)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The sampler's 52x52 SVD / covariance refits are tiny: BLAS worker threads only add wake-up
    # latency there (several ms per CEM iteration on a 256-core host), so host math runs on 1 thread.
    try:
        from threadpoolctl import threadpool_limits
        blas_guard = threadpool_limits(limits=1, user_api='blas')
    except ImportError:
        blas_guard = contextlib.nullcontext()

    score_time = [0.0]
    inner_score = ctrl.predictor.score

    def timed_score(*a, **k):
        t = time.perf_counter()
        out = inner_score(*a, **k)
        score_time[0] += time.perf_counter() - t
        return out
    ctrl.predictor.score = timed_score

    with contextlib.redirect_stdout(io.StringIO()), blas_guard:
        ctrl.act(t=0, i_tr=0, desig_pix=desig, goal_pix=goal, images=frames[:1], state=states[:1])
        for _ in range(args.warmup):
            plan()
        score_time[0] = 0.0
        prof_pred = ctrl.predictor.views[0] if hasattr(ctrl.predictor, 'views') else ctrl.predictor
        prof_pred.set_profiling(True)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = plan()
        sync()
        elapsed = time.perf_counter() - t0
        kernel_ms, launches, flops, busy_ms = prof_pred.get_profile()
        prof_pred.set_profiling(False)

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    frames_per_s = M * T * args.ncam * iters * args.steps / elapsed
    result = {
        'metric': 'predicted frames/sec (whole node), 200-sample x 13-step x 64x64 CEM',
        'value': frames_per_s, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'cem_iters_per_sec': iters * args.steps / elapsed,
        'config': {'workload': 'BASELINE configs[1]: CDNA predictor, %d samples/GPU x horizon %d x %dx%d, '
                               '%d CEM iters, pixel-distance cost, random-init weights' %
                               (args.samples_per_gpu, T, H, W, iters),
                   'num_samples': M, 'horizon': T, 'iterations': iters, 'views': args.ncam,
                   'designated_pixels_per_view': args.ndesig, 'sharding': 'samples over %d rank(s)' % world},
        'roofline': {'bound': 'mfma',
                     'kernel': ('rollout_persistent_kernel (one launch per rollout: every conv-LSTM / conv / '
                                'transposed-conv / FC tile of all steps; FLOPs = algorithmic MFMA work of the launch)'
                                if getattr(prof_pred, 'persistent', False) else
                                'conv_mfma_kernel<4,EPI_LSTM> (fused conv-LSTM gate GEMM, one launch per layer per step)'),
                     'achieved': flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else None,
                     'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': (flops / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS) if kernel_ms > 0 else None,
                     'traffic': None, 'launches': launches,
                     'avg_launch_us': 1e3 * kernel_ms / max(launches, 1),
                     'busy_ms': busy_ms, 'achieved_while_busy': flops / (busy_ms * 1e-3) / 1e12 if busy_ms > 0 else None,
                     'substreams': prof_pred.substreams,
                     'kernel_time_share': busy_ms * 1e-3 / elapsed},
        'host_ms_per_step_outside_predictor': 1e3 * (elapsed - score_time[0]) / args.steps,
        'best_score_last_plan': float(np.min(out['plan_stat']['scores_itr%d' % (iters - 1)])),
    }
    # HBM traffic cannot be counted from inside the process; it is taken from the committed rocprofv3
    # PMC run of this same command (tools/pmc_hbm.sh), when one exists for the default workload
    traffic_file = os.path.join(REPO, 'profiles', 'r01_c_hbm_traffic.json')
    if (os.path.exists(traffic_file) and getattr(prof_pred, 'persistent', False) and M == 200 and T == 13
            and iters == 3 and npix == 1):
        with open(traffic_file) as f:
            result['roofline']['traffic'] = json.load(f)['hbm_bytes_per_launch']
        result['roofline']['traffic_source'] = 'profiles/r01_c_hbm_traffic.json (rocprofv3 PMC, offline)'
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ctx = {'context_frames': frames[:, :1], 'context_actions': np.zeros((1, 4)),
               'context_states': states,
               'context_pixel_distributions': ctrl._switch_on_pix(
                   np.array(desig).reshape(args.ncam, args.ndesig, 2))[:, :1]}
        acts = np.random.RandomState(3).normal(0, 0.05, (args.cpu_samples, T, 4))
        result['cpu_baseline'] = cpu_baseline(prof_pred.weights, ctx, acts, np.array(goal[:args.ndesig]).reshape(1, -1, 2))
    elif rank == 0:
        result['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
