#!/usr/bin/env python
"""Headline benchmark: visual-MPC CEM planning throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3|c4|c5] [--scaling strong|weak]

With ``--gpus N > 1`` and no ``WORLD_SIZE`` in the environment the script launches its N ranks
itself (child processes started BEFORE this process touches a GPU; RCCL when the box has N GPUs,
gloo with the ranks sharing the GPUs it has otherwise) and rank 0 prints the line; under
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

One "step" = one planning call ``policy.act()`` = one full CEM (reference
``cem_base_controller.py:85-116``): sample -> upload -> roll every candidate through the CDNA
predictor -> pixel-distance cost on the device -> (all-gather) -> argsort -> refit, for
``iterations=3`` CEM iterations.  Every timed call sees a NEW context frame (a growing history, as
in a real MPC loop), so the first rollout of each call pays for the context-only part of the
network.  Workloads are BASELINE.json's configs:

    c2 (default)  200 samples x horizon 13 x 64x64, pixel-distance cost           configs[1]
    c1            32 samples x horizon 5, 1 CEM iteration (plumbing)               configs[0]
    c3            2 views x 600 samples x horizon 13, flow-registration controller configs[2]
    c4            1000 samples x horizon 15                                        configs[3]
    c5            5 latent draws x 1000 actions x horizon 15 x 128x128             configs[4]

``--scaling strong`` (default) keeps the workload's candidate count and shards it over the ranks
(c2 on 8 GPUs = 25 samples per rank, the north-star target; c4 = 125 per rank); ``weak`` gives
every rank the full count.  One all-gather of ``[M, 1 + tasks]`` score rows per CEM iteration.

Prints ONE JSON line on rank 0: ``value`` = predicted frames / second over the whole job
(M * T * views * draws * iterations * K / wall, max over ranks) in exact fp32, CEM iterations /
second, the roofline of the dominant kernel measured with HIP events on the launch stream, and the
CPU baseline (the oracle restatement timed on the host cores, rank 0 at N=1 only).
``alt_precision`` repeats the timed loop with the conv-LSTM GEMMs in the split-bf16 mode.
"""
import argparse
import contextlib
import hashlib
import io
import json
import os
import subprocess
import sys
import time

HOST_THREAD_CAPS = ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS')
if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    # N ranks on one host: the sampler's 52x52 refits must not start N x (all cores) BLAS / OpenMP workers.  Set before
    # numpy / torch load their runtimes, whoever launched the ranks (spawn_ranks below or torch.distributed.run).
    for _k in HOST_THREAD_CAPS:
        os.environ.setdefault(_k, '1')

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E, ~8 TB/s
PEAK_BF16_MFMA_TFLOPS = 2500.0      # same guide: dense bf16 MFMA

WORKLOADS = {
    #      M     T  iters ncam ndesig size draws selection_frac  BASELINE.json config
    'c1': (32, 5, 1, 1, 1, 64, 0, 0.0, 'configs[0]: sim cartgripper pixel-distance CEM plumbing case'),
    'c2': (200, 13, 3, 1, 1, 64, 0, 0.0, 'configs[1]: CDNA predictor, pixel-distance cost'),
    'c3': (600, 13, 3, 2, 2, 64, 0, 0.05, 'configs[2]: 2-view, flow-registration controller'),
    'c4': (1000, 15, 3, 1, 1, 64, 0, 0.0, 'configs[3]: samples sharded across the GPUs, RCCL cost all-gather'),
    'c5': (1000, 15, 3, 1, 1, 128, 5, 0.0, 'configs[4]: stochastic predictor, latent draws'),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='c2')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='strong')
    ap.add_argument('--samples', type=int, default=0, help='override the workload\'s candidate count')
    ap.add_argument('--ndesig', type=int, default=0, choices=(0, 1, 2, 3, 4),
                    help='override the workload\'s designated pixels per view (reference hparam designated_pixel_count, '
                         'pixel_cost_controller.py:57)')
    ap.add_argument('--precision', choices=('fp32', 'bf16x6'), default=os.environ.get('VF_PRECISION', 'fp32'),
                    help='primary precision mode (the other one is reported as alt_precision)')
    ap.add_argument('--network', choices=('savp', 'savp2', 'savp3'), default='savp',
                    help='generator of the c5 workload (savp3: of any workload): savp (vf_config.arch 1), savp2 (arch 2: the conditioning vector in '
                         'every conv-LSTM, published seven-layer compositing; exact fp32 only) or savp3 (arch 3: the published '
                         'SAVP generator - instance norm, conv + pool / up-sampling + conv, dependent masks; exact fp32 only)')
    ap.add_argument('--layer-spec', type=int, default=0, choices=(0, 32, 64, 128),
                    help='savp3: 0 = the layer table the public code selects by image size (128 x 128: six conv-LSTMs up to '
                         '256 channels); 64 = the paper\'s five-cell table')
    ap.add_argument('--no-alt', action='store_true', help='skip the alt_precision measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    return ap.parse_args()


# ----------------------------------------------------------------------------- self-launch
class GpuProbeError(RuntimeError):
    pass


def count_gpus_in_child(timeout=600):
    """Number of GPUs this host exposes, asked of a throw-away child process: the launcher itself never imports
    torch nor makes any HIP call, so it provably cannot have initialised a GPU before it starts the ranks (and the
    ranks are fresh children - nothing is ever re-exec'ed).  A probe that FAILS (time-out, import error, no answer)
    raises with the child's stderr: it must never be mistaken for "this host has 0 GPUs" - that would quietly turn an
    RCCL scaling run on a real multi-GPU node into a shared-GPU gloo dry run."""
    code = 'import torch; print("VF_GPU_COUNT", torch.cuda.device_count())'
    try:
        proc = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, timeout=timeout)
    except (OSError, subprocess.TimeoutExpired) as e:
        raise GpuProbeError('GPU-count probe did not finish: %r' % (e,))
    for line in proc.stdout.splitlines():
        if line.startswith('VF_GPU_COUNT'):
            return int(line.split()[1])
    raise GpuProbeError('GPU-count probe gave no answer (exit code %d): %s' % (proc.returncode, proc.stderr[-2000:]))


def rank_env(args, have, port):
    """Environment of the self-launched ranks."""
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in HOST_THREAD_CAPS:          # N ranks share the host cores: one math thread each (not left to threadpoolctl)
        env[k] = '1'
    if have < args.gpus:
        # the probe POSITIVELY reported fewer GPUs than ranks (a 1-GPU box): the ranks share them and talk over
        # gloo - a dry run of the sharding and the collective, not a scaling measurement; the line says so
        env['VF_BENCH_BACKEND'] = 'gloo'
    return env


def spawn_ranks(args):
    """Start one child per rank (this process has not initialised any GPU and never will: no torch import, the
    device count comes from a throw-away child)."""
    import socket
    assert 'torch' not in sys.modules, 'the launcher must stay GPU-free'
    try:
        have = count_gpus_in_child()
    except GpuProbeError as e:      # never fall back to a gloo dry run on a failed probe: abort, loudly
        print('bench: %s' % e, file=sys.stderr)
        return 3
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = rank_env(args, have, port)
    procs = []
    for r in range(args.gpus):
        renv = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=renv,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    return wait_ranks(procs)


def wait_ranks(procs, grace=10.0, poll=0.2):
    """Watchdog of the self-launched ranks: when any child exits non-zero the others are terminated (they would
    otherwise sit in a collective that can never complete until the driver's timeout) and the launcher returns
    that exit code.  Children are fresh processes; this parent never touches a GPU and never re-execs."""
    alive, rc = list(procs), 0
    while alive:
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = abs(r) or 1
                print('bench: a rank exited with code %d; stopping the other %d' % (r, len(alive)), file=sys.stderr)
                for q in alive:
                    q.terminate()
                deadline = time.monotonic() + grace
                for q in alive:
                    try:
                        q.wait(timeout=max(0.0, deadline - time.monotonic()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                        q.wait()
                alive = []
                break
        if alive:
            time.sleep(poll)
    return rc


# ----------------------------------------------------------------------------- CPU baseline
def physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except ImportError:
        return os.cpu_count() or 1


def cpu_baseline():
    """The CPU restatement (oracle, PyTorch-CPU fp32) behind the SAME controller on the host cores
    (BASELINE.md section 3): one full C1 planning call, and one full CEM iteration of C2 with all 200
    candidates (sample -> rollout -> cost -> argsort -> refit).  A bounded sample: the other two C2
    iterations repeat the same work."""
    import torch
    from tests.helpers.oracle_predictor import make_oracle_predictor_class
    from visual_foresight_amd.policy.cem_controllers import PixelCostController
    from visual_foresight_amd.video_prediction.cdna_arch import CdnaWeights
    cores = physical_cores()
    # oneDNN convolutions over a few hundred small images stop scaling long before a 128-core host is full (and
    # get SLOWER beyond a few dozen threads): the full sample runs at the thread count that is fastest, the
    # all-physical-cores figure BASELINE.md asks for is measured beside it on the same full sample
    used = min(cores, 32)
    torch.set_num_threads(used)
    factory = lambda cfg: CdnaWeights.random(cfg, seed=0)
    frames = np.random.RandomState(1).randint(0, 256, (2, 1, 64, 64, 3)).astype(np.uint8)
    states = np.random.RandomState(2).normal(0, .1, (2, 5))

    def plan(M, T, iters):
        pol = {'predictor_class': make_oracle_predictor_class(factory), 'repeat': 1, 'rejection_sampling': False,
               'verbose': False}
        if T != 5:                                      # an override equal to the default raises
            pol['nactions'] = T
        if M != 200:
            pol['num_samples'] = M
        if iters != 3:
            pol['iterations'] = iters
        with contextlib.redirect_stdout(io.StringIO()):
            ctrl = PixelCostController({'adim': 4, 'sdim': 5, 'image_height': 64, 'image_width': 64}, pol, 0, 1)
            ctrl.reset()
            np.random.seed(0)
            ctrl.act(t=0, i_tr=0, desig_pix=[[32, 32]], goal_pix=[[16, 48]], images=frames[:1], state=states[:1])
            t0 = time.perf_counter()
            ctrl.act(t=1, i_tr=0, desig_pix=[[32, 32]], goal_pix=[[16, 48]], images=frames, state=states)
            return time.perf_counter() - t0

    plan(12, 2, 1)                                          # warm-up (thread pool, oneDNN primitives)
    c1_calls = sorted(plan(32, 5, 1) for _ in range(5))    # median / p10 / p90 over repeated calls
    t_c1 = c1_calls[len(c1_calls) // 2]
    t_c2 = plan(200, 13, 1)
    all_cores = None
    if cores > used:
        torch.set_num_threads(cores)
        plan(12, 2, 1)
        t_all = plan(200, 13, 1)        # BASELINE.md section 3 literally: N = all physical cores, the whole config
        all_cores = {'threads': cores, 'value': 200 * 13 / t_all, 'unit': 'predicted frames/s',
                     'cem_iters_per_sec': 1.0 / t_all,
                     'sample': 'C2: one full CEM iteration with all 200 samples on all %d physical cores, %.1f s' % (cores, t_all)}
        torch.set_num_threads(used)
    return {'value': 200 * 13 / t_c2, 'unit': 'predicted frames/s', 'cores': used, 'kind': 'port',
            'physical_cores_of_host': cores,
            'sample': 'C2: one full CEM iteration of the workload (all 200 samples x 13 steps: sample, rollout, '
                      'cost, argsort, refit) through the controller, %.1f s, %d torch threads (the fastest setting '
                      'on this host; all-cores figure beside it)' % (t_c2, used),
            'cem_iters_per_sec': 1.0 / t_c2, 'at_all_physical_cores': all_cores,
            'c1': {'value': 32 * 5 / t_c1, 'unit': 'predicted frames/s', 'cem_iters_per_sec': 1.0 / t_c1,
                   'calls': len(c1_calls), 'seconds_median_p10_p90': [t_c1, c1_calls[0], c1_calls[-1]],
                   'sample': 'C1: the whole planning call (32 samples x horizon 5, 1 iteration), median of %d calls '
                             '%.2f s (min %.2f, max %.2f)' % (len(c1_calls), t_c1, c1_calls[0], c1_calls[-1])},
            'note': 'CPU restatement baseline (oracle/), reported, not the optimisation target; the literal '
                    'TF1-CPU reference cannot run anywhere in this project (no TF, no video_prediction)'}


# ----------------------------------------------------------------------------- the benchmark
def library_hash():
    """sha256 over the kernel sources: ties an offline PMC profile to the library it was taken from."""
    from visual_foresight_amd import _lib
    h = hashlib.sha256()
    for path in _lib.SOURCES:
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


_FLOW_CACHE = {}


def smooth_flow_warper(current, reference):
    """Plug-in for the registration controller in the c3 workload (the registration network is not part of the
    reference snapshot): a fixed smooth flow field of +-2.5 pixels with fractional parts, so the device side -
    bilinear warp, window medians, warp error - samples between pixels and near the borders instead of running its
    best case (an identity flow hits every pixel centre)."""
    ncam, H, W = current.shape[:3]
    key = (ncam, H, W)
    if key not in _FLOW_CACHE:
        rows, cols = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
        flow = np.zeros((ncam, H, W, 2), np.float32)
        for c in range(ncam):
            flow[c, :, :, 0] = 2.5 * np.sin(rows / 7.0 + c) + 0.37
            flow[c, :, :, 1] = 2.5 * np.cos(cols / 5.0 - c) - 0.21
        _FLOW_CACHE[key] = flow
    return None, _FLOW_CACHE[key], None


class Bench(object):
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        # one rank per GPU over RCCL; VF_BENCH_BACKEND=gloo lets several ranks share one GPU for dry runs
        self.backend = os.environ.get('VF_BENCH_BACKEND', 'nccl')
        n_dev = max(torch.cuda.device_count(), 1)
        if self.world > n_dev and 'VF_BENCH_BACKEND' not in os.environ:
            # more ranks than GPUs (a launcher started N ranks on a smaller box): RCCL refuses two ranks on one device,
            # so the ranks share GPUs over gloo - a dry run of the sharding and the collective, and the line says so
            self.backend = 'gloo'
            if self.rank == 0:
                print('note: %d ranks on %d GPU(s): gloo dry run, not a scaling measurement' % (self.world, n_dev),
                      file=sys.stderr)
        dev_index = local_rank % n_dev
        if self.world > 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            torch.cuda.set_device(dev_index)
            if self.backend == 'nccl':
                dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
            else:
                dist.init_process_group(self.backend)
        if self.world != args.gpus and self.rank == 0:
            print('warning: --gpus %d but WORLD_SIZE=%d' % (args.gpus, self.world), file=sys.stderr)
        self.dev = torch.device('cuda', dev_index)
        self.dev_index = dev_index
        (M, self.T, self.iters, self.ncam, self.ndesig, size, self.draws, self.sel_frac,
         self.workload_name) = WORKLOADS[args.workload]
        if args.samples:
            M = args.samples
        if args.ndesig and args.workload != 'c3':       # (c3's pixels are (start, goal) registrations of one task)
            self.ndesig = args.ndesig
        self.H = self.W = size
        self.M = M * (self.world if args.scaling == 'weak' else 1)
        self.per_rank = -(-self.M // self.world)
        # synthetic inputs, identical on every rank; the history grows by one frame per call
        n_calls = 1 + args.warmup + args.steps
        self.frames = np.random.RandomState(1).randint(0, 256, (1 + n_calls, self.ncam, self.H, self.W, 3)
                                                       ).astype(np.uint8)
        self.states = np.random.RandomState(2).normal(0, .1, (1 + n_calls, 5))
        k = self.H // 64 or 1
        ntask = self.ndesig // 2 if args.workload == 'c3' else self.ndesig      # c3: 1 task x (start, goal)
        self.desig = [[k * (32 - 3 * i), k * (32 + 2 * i)] for i in range(self.ncam * ntask)]
        self.goal = [[k * (16 + 2 * i), k * (48 - 3 * i)] for i in range(self.ncam * ntask)]
        self.goal_image = np.random.RandomState(3).uniform(0, 1, (1, self.ncam, self.H, self.W, 3)).astype(np.float32)

    @staticmethod
    def blas_guard():
        # The sampler's 52x52 SVD / covariance refits are tiny: BLAS worker threads only add wake-up latency
        # there (several ms per CEM iteration on a 256-core host), so host math runs on 1 thread.
        try:
            from threadpoolctl import threadpool_limits
            return threadpool_limits(limits=1, user_api='blas')
        except ImportError:
            return contextlib.nullcontext()

    def sync(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def build_controller(self, precision):
        from visual_foresight_amd.policy.cem_controllers import PixelCostController, RegisterGtruthController
        from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
        ag_params = {'adim': 4, 'sdim': 5, 'image_height': self.H, 'image_width': self.W}
        if self.ncam != 1:
            ag_params['ncam'] = self.ncam
        # overrides equal to a default raise (reference policy.py:57-58), hence the conditionals
        cls = RegisterGtruthController if self.args.workload == 'c3' else PixelCostController
        policy = {'type': cls, 'repeat': 1, 'rejection_sampling': False, 'verbose': False,
                  'predictor_class': HipVPredEvaluation}
        if self.args.workload == 'c3':
            policy.update(registration_warper=smooth_flow_warper, register_region=True)
        if self.draws or self.args.network == 'savp3':
            # (savp3 on a deterministic workload - the shape the reference's RoboNet-era configs run their SAVP-architecture
            # models at: 64 x 64, a few hundred candidates - is the stochastic predictor with ONE latent draw per action)
            from visual_foresight_amd.video_prediction.stochastic_predictor import StochasticHipPredictor
            opts = dict(n_latent=max(self.draws, 1), arch=self.args.network)
            if self.args.network == 'savp3' and self.args.layer_spec:
                opts['layer_spec'] = self.args.layer_spec
            policy['predictor_class'] = StochasticHipPredictor.with_options(**opts)
        if self.per_rank != 200:
            policy['vpred_batch_size'] = self.per_rank      # engine buffers sized for one rank's shard
        if self.ndesig != 1:
            policy['designated_pixel_count'] = self.ndesig
        if self.sel_frac:
            policy['selection_frac'] = self.sel_frac
        if self.T != 5:
            policy['nactions'] = self.T
        if self.M != 200:
            policy['num_samples'] = self.M
        if self.iters != 3:
            policy['iterations'] = self.iters
        os.environ['VF_PRECISION'] = precision      # read by HipVPredEvaluation
        with contextlib.redirect_stdout(io.StringIO()):
            ctrl = cls(ag_params, policy, 0, 1)      # the predictor adds LOCAL_RANK itself
            ctrl.reset()
        return ctrl

    def measure(self, precision):
        """Warm up, then time exactly --steps planning calls.  Returns the raw measurements."""
        a, torch = self.args, self.torch
        ctrl = self.build_controller(precision)
        pred = ctrl.predictor
        for opt in ('yield_budget', 'write_through'):   # A/B runs only: timing-only scheduler options
            if os.environ.get('VF_BENCH_' + opt.upper()) is not None:
                (getattr(pred, 'predictor', pred)).set_sched_option(opt, int(os.environ['VF_BENCH_' + opt.upper()]))
        score_time = [0.0]
        inner_score = pred.score

        def timed_score(*args_, **kw):
            t = time.perf_counter()
            out = inner_score(*args_, **kw)
            score_time[0] += time.perf_counter() - t
            return out
        pred.score = timed_score
        extra = {'goal_image': self.goal_image} if a.workload == 'c3' else {}

        def plan(i):
            # call i sees frames[0 .. i+1]: a new last frame (and state) every call, like a real MPC loop
            return ctrl.act(t=i, i_tr=0, desig_pix=self.desig, goal_pix=self.goal, images=self.frames[:i + 1],
                            state=self.states[:i + 1], **extra)

        np.random.seed(0)       # same candidate stream for every rank and every precision mode
        with contextlib.redirect_stdout(io.StringIO()), self.blas_guard():
            plan(0)                             # t = 0 < start_planning: no rollout (one context frame only)
            for i in range(a.warmup):
                plan(1 + i)
            score_time[0] = 0.0
            pred.set_profiling(True)
            if self.world > 1:
                pred.set_collective_timing(True)
            self.sync()
            t0 = time.perf_counter()
            marks = [t0]
            decided = []                        # what every timed call decided: hashed AFTER the timed region
            for i in range(a.steps):
                out = plan(1 + a.warmup + i)    # synchronous: returns after the scores are back on the host
                marks.append(time.perf_counter())
                decided.append(({k: v for k, v in out['plan_stat'].items()}, ctrl._best_indices, out['actions']))
            self.sync()
            elapsed = time.perf_counter() - t0
            kernel_ms, launches, flops, busy_ms = pred.get_profile()
            pred.set_profiling(False)
            avg_launch_us = 1e3 * kernel_ms / max(launches, 1)
            collective = self.collective_report(pred, avg_launch_us) if self.world > 1 else None
            # G-invariance, proven by the record itself: every rank hashes the score vectors of every CEM iteration,
            # the elite indices and the executed action of every timed call; the hashes are compared across ranks, and
            # `scores_sha` is the same string at N = 1 and at any N under strong scaling (same candidates, same bits)
            from visual_foresight_amd.video_prediction.sharding import plan_digest, gather_plan_digests
            same, shas = gather_plan_digests([plan_digest(ps, bi, act) for ps, bi, act in decided])
        if self.world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if self.backend == 'nccl' else 'cpu')
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        per_call = 1e3 * np.diff(marks)
        lo, hi = 0, self.M
        if self.world > 1:
            from visual_foresight_amd.video_prediction.sharding import shard_bounds
            lo, hi = shard_bounds(self.M, self.rank, self.world)
        return dict(ctrl=ctrl, pred=pred, elapsed=elapsed, collective=collective,
                    call_ms=[float(np.percentile(per_call, q)) for q in (50, 10, 90)], kernel_ms=kernel_ms,
                    launches=launches, flops=flops, busy_ms=busy_ms,
                    host_ms=1e3 * (elapsed - score_time[0]) / a.steps,
                    rollouts=(hi - lo) * max(self.draws, 1) * self.ncam * self.iters * a.steps,
                    elites=[int(i) for i in ctrl._best_indices],
                    elites_identical_across_ranks=bool(same), scores_sha=shas[0], scores_sha_per_rank=shas,
                    best=float(np.min(out['plan_stat']['scores_itr%d' % (self.iters - 1)])))

    def collective_report(self, pred, avg_launch_us=None):
        """What the process group actually was - backend, world size, every rank's device - and what the one
        all-gather of score rows per CEM iteration cost (HIP events around the collective on every rank; mean over the
        timed calls, then mean / max over ranks).  Evidence in a SCALE record that RCCL saw N ranks on N GPUs."""
        torch, dist = self.torch, self.dist
        st = pred.collective_stats()
        pred.set_collective_timing(False)
        props = torch.cuda.get_device_properties(self.dev)
        mine = {'rank': self.rank, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'device': self.dev_index,
                'name': props.name, 'uuid': str(getattr(props, 'uuid', '')), 'pid': os.getpid(),
                'allgather_calls': st['calls'], 'allgather_mean_ms': st['mean_ms'], 'allgather_max_ms': st['max_ms'],
                'bytes_per_rank': st['bytes_per_rank'],
                'avg_launch_us': avg_launch_us}      # this rank's rollout kernel: a straggler GPU shows here
        ranks = [None] * self.world
        dist.all_gather_object(ranks, mine)
        means = [r['allgather_mean_ms'] for r in ranks if r['allgather_mean_ms'] is not None]
        # proof of transport that needs no NCCL_DEBUG log: a device all-gather of the rank ids through the SAME process group
        # the score rows travel on must come back as 0 .. N-1 on every rank; the RCCL version and the peer-access matrix of
        # the devices say what the ranks could have talked over
        on_gpu = dist.get_backend() == 'nccl'
        ids = torch.full((1,), self.rank, dtype=torch.int64, device=self.dev if on_gpu else 'cpu')
        got = [torch.empty_like(ids) for _ in range(self.world)]
        dist.all_gather(got, ids)
        ids_ok = [int(t.item()) for t in got] == list(range(self.world))
        all_ok = [None] * self.world
        dist.all_gather_object(all_ok, bool(ids_ok))
        try:
            rccl_version = '.'.join(str(v) for v in torch.cuda.nccl.version()) if on_gpu else None
        except Exception as e:       # (informational only)
            rccl_version = 'unavailable: %r' % (e,)
        devs = sorted({r['device'] for r in ranks})
        peer = None
        if on_gpu and len(devs) > 1:
            peer = {'%d->%d' % (i, j): bool(torch.cuda.can_device_access_peer(i, j)) for i in devs for j in devs if i != j}
        return {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                'rccl_version': rccl_version,
                'rank_id_allgather_verified_on_every_rank': bool(all(all_ok)),
                'peer_access': peer,
                'devices': [r['device'] for r in ranks],
                'distinct_gpus': len({(r['uuid'] or r['device']) for r in ranks}),
                'allgather_ms_per_cem_iter': {'mean_over_ranks': float(np.mean(means)) if means else None,
                                              'max_over_ranks': float(np.max(means)) if means else None,
                                              'worst_single_call': max([r['allgather_max_ms'] or 0.0 for r in ranks]),
                                              'calls_per_rank': ranks[0]['allgather_calls'],
                                              'how': 'HIP events on the stream the collective is ordered on, around '
                                                     'every all-gather of the timed region (includes waiting for '
                                                     'the slowest rank to arrive)'},
                'bytes_per_rank_per_allgather': ranks[0]['bytes_per_rank'],
                'host_threads_per_rank': {k: os.environ.get(k) for k in HOST_THREAD_CAPS},
                'avg_launch_us_min_max_over_ranks': [min(r['avg_launch_us'] for r in ranks),
                                                     max(r['avg_launch_us'] for r in ranks)]
                if all(r.get('avg_launch_us') is not None for r in ranks) else None,
                'ranks': ranks}

    def survey_rate(self, m):
        """SURVEY.md 8(d) accounting: every sample charged with all seq_len - 1 cell evaluations at the layer
        table's MAC count, including the context step the engine runs once and shares (so >= `achieved`)."""
        pred = m['pred']
        steps = pred.sequence_length - 1
        if not m['launches'] or m['kernel_ms'] <= 0:
            return None
        rollouts_per_launch = m['rollouts'] / m['launches']
        macs = float(sum(pred.cfg.macs_per_sample_step().values()))
        # networks whose checkpoint describes more MACs than the matrix pipe runs (savp2 / savp3: the conditioning channels are
        # border-class bias tables; savp3: conv + pool is one stride-2 convolution) are priced on the EXECUTED count; the
        # checkpoint's own count is reported beside it and kept away from the MFMA peak
        executed = getattr(pred.cfg, 'executed_macs_per_sample_step', None)
        priced = float(sum(executed().values())) if executed else macs
        flops = 2.0 * priced * steps * rollouts_per_launch
        tf = flops / (1e-3 * m['kernel_ms'] / m['launches']) / 1e12
        out = {'flops_per_launch': flops, 'tflops': tf, 'frac_of_fp32_mfma_peak': tf / PEAK_FP32_MFMA_TFLOPS,
               'what': '2 x %.3f GMAC per sample-step x %d steps x %d single-view rollouts per launch / avg launch '
                       'duration' % (priced / 1e9, steps, rollouts_per_launch)}
        if executed:
            out['what'] += ' (MACs the matrix pipe executes; the checkpoint describes %.3f GMAC per sample-step: conditioning ' \
                           'channels as convolution rows%s - those run as scalar-FMA bias tables / one fused stride-2 conv)' % (
                               macs / 1e9, ', stride-1 convs in front of the pools' if pred.cfg.arch == 'savp3' else '')
            out['checkpoint_gmacs_per_sample_step'] = macs / 1e9
        return out

    def roofline(self, m, precision):
        tf = m['flops'] / (m['kernel_ms'] * 1e-3) / 1e12 if m['kernel_ms'] > 0 else None
        r = {'bound': 'mfma',
             'kernel': 'rollout_persistent_kernel (one launch per rollout of one rank\'s shard: every conv-LSTM / '
                       'conv / transposed-conv / FC tile of all steps and views; FLOPs = the MFMA work the launch '
                       'executes, context step counted once; the 3-channel first conv runs on the vector ALUs and is not counted)',
             'achieved': tf, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
             'frac': tf / PEAK_FP32_MFMA_TFLOPS if tf else None, 'traffic': None,
             'traffic_measured_in_run': False,
             'launches': m['launches'], 'avg_launch_us': 1e3 * m['kernel_ms'] / max(m['launches'], 1),
             'survey_8d': self.survey_rate(m),
             'kernel_time_share': m['busy_ms'] * 1e-3 / m['elapsed']}
        if precision == 'bf16x6':
            # the same algorithmic (fp32-equivalent) FLOPs cost six bf16 MFMA FLOPs each in the
            # conv-LSTM tiles; quote that rate against the bf16 MFMA peak as well
            r['note'] = ('achieved/peak/frac are fp32-EQUIVALENT algorithmic FLOP/s against the fp32 MFMA peak; '
                         'the conv-LSTM tiles execute 6 bf16 MFMA FLOPs per algorithmic FLOP')
            r['bf16_mfma_tflops'] = 6.0 * tf if tf else None
            r['bf16_mfma_frac_of_peak'] = 6.0 * tf / PEAK_BF16_MFMA_TFLOPS if tf else None
        return r

    def attach_traffic(self, roof, precision):
        """HBM bytes per launch cannot be counted from inside the process: they come from the committed
        rocprofv3 PMC passes of this same command (tools/pmc_hbm.sh) and are only quoted when that profile
        was taken from this very library (source hash) on this workload."""
        import glob
        prof = name = None
        for path in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_hbm_traffic.json')), reverse=True):
            with open(path) as f:
                cand = json.load(f)
            if cand.get('lib_sources_sha16') == library_hash():     # the newest round's profile of THIS library
                prof, name = cand, os.path.basename(path)
                break
        if prof is None:
            return
        if (prof.get('workload') == self.args.workload
                and prof.get('precision') == precision and self.world == 1 and not self.args.samples
                and self.ndesig == WORKLOADS[self.args.workload][4]):
            roof['traffic'] = prof['hbm_bytes_per_launch']
            roof['traffic_source'] = 'profiles/%s (rocprofv3 PMC passes of this library, offline)' % name
            roof['traffic_vs_algorithmic'] = prof.get('ratio_to_algorithmic')
            if roof.get('avg_launch_us'):       # HBM-side rate of the launch against the 8 TB/s peak (north_star: GB/s vs peak)
                roof['hbm_gbps'] = prof['hbm_bytes_per_launch'] / (roof['avg_launch_us'] * 1e-6) / 1e9
                roof['hbm_frac_of_peak'] = roof['hbm_gbps'] / PEAK_HBM_GBPS

    def metric_label(self):
        """BASELINE.json's metric string for the configuration it is quoted on (c2 at its own size); any other
        workload or --samples override names what it actually counts, so a c3/c5 or shard line cannot be read as
        the headline (frames of every view and latent draw are counted)."""
        if (self.args.workload == 'c2' and not self.args.samples and self.args.scaling == 'strong'
                and self.ndesig == WORKLOADS['c2'][4]):
            return 'predicted frames/sec (whole node), 200-sample x 13-step x 64x64 CEM'
        extra = ''.join([' x %d views' % self.ncam if self.ncam > 1 else '',
                         ' x %d latent draws' % self.draws if self.draws else ''])
        if self.ndesig != WORKLOADS[self.args.workload][4]:
            extra += ', %d designated pixels per view' % self.ndesig
        return 'predicted frames/sec (whole node), workload %s: %d-sample x %d-step x %dx%d%s CEM' % (
            self.args.workload, self.M, self.T, self.H, self.W, extra)

    def run(self):
        a = self.args
        T, iters = self.T, self.iters
        primary = a.precision
        m = self.measure(primary)
        frames_total = self.M * max(self.draws, 1) * T * self.ncam * iters * a.steps
        shared_gpus = self.world > 1 and self.backend != 'nccl'
        result = {
            'metric': self.metric_label(),
            'value': frames_total / m['elapsed'], 'unit': 'frames/s', 'n_gpus': self.world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': 1e3 * m['elapsed'] / a.steps,
            'ms_per_step_median_p10_p90': m['call_ms'],
            'higher_is_better': True, 'scaling': a.scaling, 'vs_baseline': None,
            'dtype': 'f32' if primary == 'fp32' else 'f32 emulated by 6 bf16 MFMA products (fp32 accumulate)',
            'data': 'synthetic',
            'cem_iters_per_sec': iters * a.steps / m['elapsed'],
            'config': {'workload': 'BASELINE %s: %d samples (%d per rank) x horizon %d x %dx%d, %d CEM iters, '
                                   'random-init weights, a new context frame per planning call' %
                                   (self.workload_name, self.M, self.per_rank, T, self.H, self.W, iters),
                       'workload_id': a.workload, 'num_samples': self.M, 'samples_per_rank': self.per_rank,
                       'horizon': T, 'iterations': iters, 'views': self.ncam,
                       'designated_pixels_per_view': self.ndesig, 'precision': primary,
                       'latent_draws_per_action': self.draws,
                       'network': getattr(m['pred'], 'arch', 'cdna'),
                       **({'registration_flow': 'synthetic smooth flow of +-2.5 px (plug-in; the registration '
                                                'network is absent from the reference)'} if a.workload == 'c3' else {}),
                       'sharding': 'samples over %d rank(s), one all-gather of score rows per CEM iteration%s' %
                                   (self.world, ' (gloo dry run: ranks share a GPU, not a scaling measurement)'
                                    if shared_gpus else ' over RCCL' if self.world > 1 else '')},
            'roofline': self.roofline(m, primary),
            'collective': m['collective'],
            'host_ms_per_step_outside_predictor': m['host_ms'],
            'best_score_last_plan': m['best'],
            # sha256 over (scores of every CEM iteration, elite indices, executed action) of every timed call: equal on
            # every rank of this run (`elites_identical_across_ranks`) and, under strong scaling, equal to the N = 1
            # line's string - the sharded job planned bit for bit what one GPU plans
            'elites_identical_across_ranks': m['elites_identical_across_ranks'],
            'scores_sha': m['scores_sha'],
            # what the digest was taken over: two lines are only comparable when these agree (the frame history and the
            # position in the candidate stream depend on --steps / --warmup) - see compare_lines()
            'scores_sha_over': {'workload': a.workload, 'samples': self.M, 'horizon': T, 'iterations': iters,
                                'steps': a.steps, 'warmup': a.warmup, 'seed': 0, 'precision': primary,
                                'network': getattr(m['pred'], 'arch', 'cdna'), 'layer_spec': a.layer_spec,
                                'views': self.ncam, 'designated_pixels_per_view': self.ndesig,
                                'latent_draws_per_action': self.draws},
            'scores_sha_per_rank': m['scores_sha_per_rank'] if self.world > 1 else None,
        }
        self.attach_traffic(result['roofline'], primary)

        if not a.no_alt and not ((self.draws and a.network == 'savp2') or a.network == 'savp3'):   # (arch 2 / 3 are built for exact fp32 only)
            other = 'bf16x6' if primary == 'fp32' else 'fp32'
            am = self.measure(other)
            result['alt_precision'] = {
                'precision': other,
                'what': ('conv-LSTM gate GEMMs as six bf16 MFMA products per multiply (3-way exact operand split, '
                         'fp32 accumulate); everything else fp32' if other == 'bf16x6' else 'exact fp32 MFMA'),
                'value': frames_total / am['elapsed'], 'unit': 'frames/s',
                'ms_per_step': 1e3 * am['elapsed'] / a.steps,
                'ms_per_step_median_p10_p90': am['call_ms'],
                'cem_iters_per_sec': iters * a.steps / am['elapsed'],
                'roofline': self.roofline(am, other),
                'elites_identical_to_primary': am['elites'] == m['elites'],
                'best_score_last_plan': am['best'],
            }

        if self.rank == 0 and self.world == 1 and not a.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline()
        elif self.rank == 0:
            result['cpu_baseline'] = None
        if self.rank == 0:
            print(json.dumps(result), flush=True)
        if self.world > 1:
            self.dist.destroy_process_group()


def compare_lines(a, b):
    """Do two bench lines (dicts) show the same plans?  'equal' / 'mismatch' when both hashed the same job (``scores_sha_over``
    agrees: same workload, candidates, --steps, --warmup, seed, precision, network), else 'not comparable' - a different
    warm-up moves the frame history and the position in the candidate stream, so different strings then prove nothing."""
    oa, ob = a.get('scores_sha_over'), b.get('scores_sha_over')
    if not oa or not ob or oa != ob:
        return 'not comparable'
    return 'equal' if a.get('scores_sha') == b.get('scores_sha') else 'mismatch'


if __name__ == '__main__':
    _args = parse()
    if _args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(_args))
    Bench(_args).run()
