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
(M * T * views * iterations * K / wall) in the PRIMARY precision mode - exact fp32 MFMA - plus CEM
iterations / second, the roofline of the dominant kernel measured with HIP events on the launch
stream, and the CPU baseline (the oracle restatement timed on the host cores on a bounded sample;
rank 0, N=1 only).  ``alt_precision`` repeats the timed loop with the conv-LSTM GEMMs in the
split-bf16 mode (six bf16 MFMA products per multiply, fp32 accumulate, fp32-class accuracy; see
csrc/vf_conv_bf16x6.h) and says whether it selected the same elites.
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
PEAK_BF16_MFMA_TFLOPS = 2500.0      # same guide: dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--samples-per-gpu', type=int, default=200)
    ap.add_argument('--horizon', type=int, default=13)
    ap.add_argument('--iterations', type=int, default=3)
    ap.add_argument('--ncam', type=int, default=1, help='views (BASELINE configs[2] uses 2)')
    ap.add_argument('--ndesig', type=int, default=1, help='designated pixels per view')
    ap.add_argument('--selection-frac', type=float, default=0.0)
    ap.add_argument('--image-size', type=int, default=64, help='square frame size (BASELINE configs[4] uses 128)')
    ap.add_argument('--latent-draws', type=int, default=0,
                    help='stochastic predictor: z-draws per action (BASELINE configs[4] uses 5)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--precision', choices=('fp32', 'bf16x6'), default=os.environ.get('VF_PRECISION', 'fp32'),
                    help='primary precision mode (the other one is reported as alt_precision)')
    ap.add_argument('--no-alt', action='store_true', help='skip the alt_precision measurement')
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
        dev_index = local_rank % max(torch.cuda.device_count(), 1)
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
        self.H = self.W = args.image_size
        self.M = args.samples_per_gpu * (self.world if args.scaling == 'weak' else 1)
        # synthetic inputs: identical on every rank
        self.frames = np.random.RandomState(1).randint(0, 256, (2, args.ncam, self.H, self.W, 3)).astype(np.uint8)
        self.states = np.random.RandomState(2).normal(0, .1, (2, 5))
        npix = args.ncam * args.ndesig
        k = self.H // 64 or 1
        self.desig = [[k * (32 - 3 * i), k * (32 + 2 * i)] for i in range(npix)]
        self.goal = [[k * (16 + 2 * i), k * (48 - 3 * i)] for i in range(npix)]
        # The sampler's 52x52 SVD / covariance refits are tiny: BLAS worker threads only add wake-up
        # latency there (several ms per CEM iteration on a 256-core host), so host math runs on 1 thread.
        # (a fresh guard per measurement: a threadpool_limits object restores the old limits on exit)

    @staticmethod
    def blas_guard():
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
        from visual_foresight_amd.policy.cem_controllers import PixelCostController
        from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
        a = self.args
        ag_params = {'adim': 4, 'sdim': 5, 'image_height': self.H, 'image_width': self.W}
        if a.ncam != 1:
            ag_params['ncam'] = a.ncam
        # overrides equal to a default raise (reference policy.py:57-58), hence the conditionals
        policy = {'type': PixelCostController, 'repeat': 1, 'rejection_sampling': False, 'verbose': False}
        if a.latent_draws:
            from visual_foresight_amd.video_prediction.stochastic_predictor import StochasticHipPredictor
            policy['predictor_class'] = StochasticHipPredictor.with_options(n_latent=a.latent_draws)
        elif a.ncam == 1:
            policy['predictor_class'] = HipVPredEvaluation      # (ncam > 1: the multi-view default)
        if a.samples_per_gpu != 200:
            policy['vpred_batch_size'] = max(a.samples_per_gpu, 1)
        if a.ndesig != 1:
            policy['designated_pixel_count'] = a.ndesig
        if a.selection_frac:
            policy['selection_frac'] = a.selection_frac
        if a.horizon != 5:
            policy['nactions'] = a.horizon
        if self.M != 200:
            policy['num_samples'] = self.M
        if a.iterations != 3:
            policy['iterations'] = a.iterations
        os.environ['VF_PRECISION'] = precision      # read by HipVPredEvaluation (also inside multi-view)
        with contextlib.redirect_stdout(io.StringIO()):
            ctrl = PixelCostController(ag_params, policy, 0, 1)
            ctrl.reset()
        return ctrl

    def measure(self, precision):
        """Warm up, then time exactly --steps planning calls.  Returns the raw measurements."""
        a, torch = self.args, self.torch
        ctrl = self.build_controller(precision)
        prof_pred = ctrl.predictor.views[0] if hasattr(ctrl.predictor, 'views') else ctrl.predictor
        prof_pred = getattr(prof_pred, 'engine', prof_pred)     # stochastic wrapper -> its engine
        score_time = [0.0]
        inner_score = ctrl.predictor.score

        def timed_score(*args_, **kw):
            t = time.perf_counter()
            out = inner_score(*args_, **kw)
            score_time[0] += time.perf_counter() - t
            return out
        ctrl.predictor.score = timed_score

        def plan():
            return ctrl.act(t=1, i_tr=0, desig_pix=self.desig, goal_pix=self.goal, images=self.frames,
                            state=self.states)

        np.random.seed(0)       # same candidate stream for every rank and every precision mode
        with contextlib.redirect_stdout(io.StringIO()), self.blas_guard():
            ctrl.act(t=0, i_tr=0, desig_pix=self.desig, goal_pix=self.goal, images=self.frames[:1],
                     state=self.states[:1])
            for _ in range(a.warmup):
                plan()
            score_time[0] = 0.0
            prof_pred.set_profiling(True)
            self.sync()
            t0 = time.perf_counter()
            marks = [t0]
            for _ in range(a.steps):
                out = plan()                    # synchronous: returns after the scores are back on the host
                marks.append(time.perf_counter())
            self.sync()
            elapsed = time.perf_counter() - t0
            kernel_ms, launches, flops, busy_ms = prof_pred.get_profile()
            prof_pred.set_profiling(False)
        if self.world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if self.backend == 'nccl' else 'cpu')
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        per_call = 1e3 * np.diff(marks)
        return dict(ctrl=ctrl, prof_pred=prof_pred, elapsed=elapsed,
                    call_ms=[float(np.percentile(per_call, q)) for q in (50, 10, 90)], kernel_ms=kernel_ms, launches=launches,
                    flops=flops, busy_ms=busy_ms, host_ms=1e3 * (elapsed - score_time[0]) / a.steps,
                    rollouts=self.args.samples_per_gpu * max(a.latent_draws, 1) * a.iterations * a.steps,
                    elites=[int(i) for i in ctrl._best_indices],
                    best=float(np.min(out['plan_stat']['scores_itr%d' % (a.iterations - 1)])))

    def survey_rate(self, m):
        """SURVEY.md 8(d) accounting: every sample charged with all seq_len - 1 cell evaluations at the layer
        table's MAC count, including the context step the engine runs once and shares (so >= `achieved`)."""
        from visual_foresight_amd.video_prediction.cdna_arch import macs_per_sample_step
        pred = m['prof_pred']
        steps = pred.sequence_length - 1
        if not m['launches'] or m['kernel_ms'] <= 0:
            return None
        rollouts_per_launch = m['rollouts'] / m['launches']
        macs = float(sum(macs_per_sample_step(pred.cfg).values()))
        flops = 2.0 * macs * steps * rollouts_per_launch
        tf = flops / (1e-3 * m['kernel_ms'] / m['launches']) / 1e12
        return {'flops_per_launch': flops, 'tflops': tf, 'frac_of_fp32_mfma_peak': tf / PEAK_FP32_MFMA_TFLOPS,
                'what': '2 x %.3f GMAC per sample-step x %d steps x %d rollouts per launch / avg launch duration'
                        % (macs / 1e9, steps, rollouts_per_launch)}

    def roofline(self, m, precision):
        persistent = getattr(m['prof_pred'], 'persistent', False)
        tf = m['flops'] / (m['kernel_ms'] * 1e-3) / 1e12 if m['kernel_ms'] > 0 else None
        r = {'bound': 'mfma',
             'kernel': ('rollout_persistent_kernel (one launch per rollout: every conv-LSTM / conv / '
                        'transposed-conv / FC tile of all steps; FLOPs = algorithmic MFMA work of the launch)'
                        if persistent else
                        'conv_mfma_kernel<4,EPI_LSTM> (fused conv-LSTM gate GEMM, one launch per layer per step)'),
             'achieved': tf, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
             'frac': tf / PEAK_FP32_MFMA_TFLOPS if tf else None, 'traffic': None,
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

    def run(self):
        a = self.args
        T, iters = a.horizon, a.iterations
        primary = a.precision
        m = self.measure(primary)
        frames_total = self.M * max(a.latent_draws, 1) * T * a.ncam * iters * a.steps
        result = {
            'metric': 'predicted frames/sec (whole node), 200-sample x 13-step x 64x64 CEM',
            'value': frames_total / m['elapsed'], 'unit': 'frames/s', 'n_gpus': self.world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': 1e3 * m['elapsed'] / a.steps,
            'ms_per_step_median_p10_p90': m['call_ms'],
            'higher_is_better': True, 'scaling': a.scaling, 'vs_baseline': None,
            'dtype': 'f32' if primary == 'fp32' else 'f32 emulated by 6 bf16 MFMA products (fp32 accumulate)',
            'data': 'synthetic',
            'cem_iters_per_sec': iters * a.steps / m['elapsed'],
            'config': {'workload': 'BASELINE configs[1]: CDNA predictor, %d samples/GPU x horizon %d x %dx%d, '
                                   '%d CEM iters, pixel-distance cost, random-init weights' %
                                   (a.samples_per_gpu, T, self.H, self.W, iters),
                       'num_samples': self.M, 'horizon': T, 'iterations': iters, 'views': a.ncam,
                       'designated_pixels_per_view': a.ndesig, 'precision': primary,
                       'latent_draws_per_action': a.latent_draws,
                       'sharding': 'samples over %d rank(s)' % self.world},
            'roofline': self.roofline(m, primary),
            'host_ms_per_step_outside_predictor': m['host_ms'],
            'best_score_last_plan': m['best'],
        }
        # HBM traffic cannot be counted from inside the process; it is taken from the committed rocprofv3
        # PMC run of this same command (tools/pmc_hbm.sh), when one exists for the default workload
        traffic_file = os.path.join(REPO, 'profiles', 'r01_h_hbm_traffic.json')
        if (os.path.exists(traffic_file) and getattr(m['prof_pred'], 'persistent', False) and self.M == 200
                and T == 13 and iters == 3 and a.ncam * a.ndesig == 1 and primary == 'fp32'
                and self.H == 64 and not a.latent_draws):
            with open(traffic_file) as f:
                result['roofline']['traffic'] = json.load(f)['hbm_bytes_per_launch']
            result['roofline']['traffic_source'] = 'profiles/r01_h_hbm_traffic.json (rocprofv3 PMC, offline)'

        if not a.no_alt:
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
            ctrl = m['ctrl']
            adim = m['prof_pred'].cfg.adim          # 4 (+ zdim for the stochastic predictor)
            ctx = {'context_frames': self.frames[:, :1], 'context_actions': np.zeros((1, adim)),
                   'context_states': self.states,
                   'context_pixel_distributions': ctrl._switch_on_pix(
                       np.array(self.desig).reshape(a.ncam, a.ndesig, 2))[:, :1]}
            acts = np.random.RandomState(3).normal(0, 0.05, (a.cpu_samples, T, adim))
            result['cpu_baseline'] = cpu_baseline(m['prof_pred'].weights, ctx, acts,
                                                  np.array(self.goal[:a.ndesig]).reshape(1, -1, 2))
        elif self.rank == 0:
            result['cpu_baseline'] = None
        if self.rank == 0:
            print(json.dumps(result))
        if self.world > 1:
            self.dist.destroy_process_group()


if __name__ == '__main__':
    Bench(parse()).run()
