"""MI355X-native predictor with the ``VPredEvaluation`` duck-type.

Plugs into ``PixelCostController`` where the reference plugs
``robonet.video_prediction.testing.VPredEvaluation`` (reference
``visual_mpc/policy/cem_controllers/pixel_cost_controller.py:11-12,29-36,83-84,175``): ctor
``(model_path, hparams_dict, n_gpus=, first_gpu=)``, ``restore()``, ``n_context``,
``sequence_length``, ``__call__(context, {'actions'}) -> {'predicted_frames',
'predicted_pixel_distributions'}``.  Like the legacy adapter it takes the *whole* history and
slices the last ``n_context`` frames itself (reference ``video_prediction/pred_util.py:4-13``)
and chunks the sample batch by ``run_batch_size`` (``pred_util.py:21-48``; the last chunk is
simply run ragged instead of zero-padded - samples are independent).

Multi-view models (the reference's ``IndepMultiSAVP...`` family: one network per view sharing
actions and states, outputs stacked on a camera axis, ``vpred_model_interface.py:60-88``) are
one engine with ``ncam`` weight sets: every view of every sample is rolled by the same launch.

On top of that contract it offers the fused fast path ``score()``: rollouts are reduced to
per-sample costs on the GPU, sharded over the ranks of ``torch.distributed`` when that is
initialised (rank r evaluates samples ``[r*M/G, (r+1)*M/G)``; reference tower slicing
``video_prediction/setup_predictor.py:34-39``), and only the ``M`` scores are all-gathered
(RCCL) - the reference instead concatenates full predicted videos on the host
(``setup_predictor.py:155-162``).

``n_gpus`` / ``first_gpu`` mean what they mean in the reference (``setup_predictor.py:70,117-123``: ``ngpu``
towers on devices ``gpu_id ..`` inside the ONE policy process that ``sim/run.py:79-82`` creates): outside
``torch.distributed``, ``n_gpus > 1`` builds one engine per device ``first_gpu .. first_gpu + n_gpus - 1``
("lanes"), one host thread enqueues every lane's contiguous shard on its device's stream - rank-LOCAL slice
offsets, not the reference's ``gpu_id * nsmp_per_gpu`` (``setup_predictor.py:38``), which over-runs the batch
when ``first_gpu != 0`` - and the score rows are gathered with one grouped RCCL all-gather
(``vf_comm_init_all`` + ``vf_allgather_scores_group``) or, when the lanes share a device (hyper-parameter
``oversubscribe_gpus``, tests on a one-GPU box), through the host.  Under ``torch.distributed`` every rank
drives one GPU and ``n_gpus`` must be 1 or the world size.

PyTorch is the buffer carrier only: tensors own the device memory whose pointers cross the
C ABI of ``include/vf_hip.h``; all arithmetic runs in ``libvf_hip.so``.
"""
import ctypes
import os

import numpy as np

from visual_foresight_amd import _lib
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights
from visual_foresight_amd.video_prediction.savp_arch import SavpConfig, Savp2Config
from visual_foresight_amd.video_prediction.savp3_arch import Savp3Config
from visual_foresight_amd.video_prediction.sharding import dist_info as _dist_info, shard_bounds, all_gather_rows


class HipVPredEvaluation(object):
    wants_agent_params = True       # PixelCostController passes adim/sdim/size/sequence_length
    supports_task_weights = True    # score(..., task_weights=) applies trade-off weights on the device
    n_context_default = 2

    def __init__(self, model_path, hparams, n_gpus=1, first_gpu=0):
        import torch
        self._torch = torch
        self._lanes = self._comms = None        # in-process multi-GPU lanes (below), their RCCL communicators
        hp = dict(hparams)
        if isinstance(model_path, (list, tuple)):
            self.model_path = [os.path.expanduser(p) for p in model_path]
        else:
            self.model_path = os.path.expanduser(model_path) if model_path else ''
        self.n_context = int(hp.get('n_context', self.n_context_default))
        self.sequence_length = int(hp.get('sequence_length', 15))
        self.n_cam = int(hp.get('ncam', 1))
        self.n_draws = int(hp.get('n_draws', 1))        # latent draws per action (stochastic_predictor.py)
        self.run_batch_size = int(hp.get('run_batch_size', 200)) * self.n_draws
        self.seed = int(hp.get('seed', 0))
        # 'arch': 'cdna' (cdna_arch.py, default), 'savp' (savp_arch.py: four scales, first-frame compositing), 'savp2'
        # (savp + the conditioning vector in every conv-LSTM + the published seven-layer compositing) or 'savp3' (savp3_arch.py:
        # the published SAVP generator; 'zdim' of the 'adim' channels are the latent, 'layer_spec' overrides the size rule)
        self.arch = str(hp.get('arch', 'cdna'))
        if self.arch not in ('cdna', 'savp', 'savp2', 'savp3'):
            raise ValueError("arch must be 'cdna', 'savp', 'savp2' or 'savp3', got %r" % (self.arch,))
        cfg_cls = {'cdna': CdnaConfig, 'savp': SavpConfig, 'savp2': Savp2Config, 'savp3': Savp3Config}[self.arch]
        extra = dict(zdim=int(hp.get('zdim', 8)), layer_spec=int(hp.get('layer_spec', 0))) if self.arch == 'savp3' else {}
        if self.arch == 'cdna' and hp.get('decoder', 'survey') != 'survey':
            extra = dict(decoder=hp['decoder'])     # 'public': the decoder widths of the public CDNA code (cdna_arch.py)
        self.cfg = cfg_cls(height=hp.get('image_height', 64), width=hp.get('image_width', 64),
                           adim=hp.get('adim', 4), sdim=hp.get('sdim', 5),
                           ndesig=hp.get('designated_pixel_count', 1), n_context=self.n_context,
                           sequence_length=self.sequence_length, **extra)
        if not torch.cuda.is_available():
            raise _lib.VfError('HipVPredEvaluation needs a ROCm GPU (no CPU fallback)')
        world = _dist_info()[1]
        self.n_gpus = int(n_gpus)
        n_dev = max(torch.cuda.device_count(), 1)
        if self.n_gpus < 1:
            raise ValueError('n_gpus must be >= 1, got %r' % (n_gpus,))
        if world > 1 and self.n_gpus not in (1, world):
            raise ValueError('under torch.distributed every rank drives one GPU: n_gpus must be 1 or the world size '
                             '(%d), got %d' % (world, self.n_gpus))
        oversubscribe = bool(int(hp.get('oversubscribe_gpus', os.environ.get('VF_OVERSUBSCRIBE_GPUS', 0))))
        if world == 1 and self.n_gpus > n_dev and not oversubscribe:
            raise ValueError('n_gpus=%d but this host exposes %d GPU(s) (set the predictor hyper-parameter '
                             "'oversubscribe_gpus' to let the lanes share devices)" % (self.n_gpus, n_dev))
        if world == 1 and not oversubscribe and (int(first_gpu) < 0 or int(first_gpu) + self.n_gpus > n_dev):
            # devices first_gpu .. first_gpu + n_gpus - 1 are promised: never wrap onto GPUs below first_gpu,
            # which another policy process may own (reference sim/run.py hands out disjoint gpu_id ranges)
            raise ValueError('first_gpu=%d + n_gpus=%d exceeds the %d GPU(s) of this host (set the predictor '
                             "hyper-parameter 'oversubscribe_gpus' to wrap around)"
                             % (int(first_gpu), self.n_gpus, n_dev))
        # under torchrun LOCAL_RANK picks this rank's GPU; the lanes of the in-process mode follow first_gpu
        local_rank = int(os.environ.get('LOCAL_RANK', 0)) if world > 1 else 0
        shared_dry_run = oversubscribe or os.environ.get('VF_BENCH_BACKEND') == 'gloo'
        if world > 1 and not shared_dry_run and (int(first_gpu) < 0 or int(first_gpu) + local_rank >= n_dev):
            import torch.distributed as dist
            if dist.get_backend() != 'gloo':
                # one process per GPU: rank r owns device first_gpu + LOCAL_RANK - never wrap onto a GPU below first_gpu
                # (another job's), and never put two RCCL ranks on one device
                raise ValueError('first_gpu=%d + LOCAL_RANK=%d is outside the %d GPU(s) of this host (ranks sharing a '
                                 "GPU need the gloo backend or the hyper-parameter 'oversubscribe_gpus')"
                                 % (int(first_gpu), local_rank, n_dev))
        self.device_index = (int(first_gpu) + local_rank) % n_dev
        self.device = torch.device('cuda', self.device_index)
        self._libh = _lib.load_library()
        c = self.cfg
        # 'fp32' (default): exact fp32 MFMA.  'bf16x6': fp32 emulated by six bf16 MFMA products in the
        # conv-LSTM gate GEMMs (fp32-class accuracy, not bit-identical to 'fp32').
        # The reference's reduced-precision switch is the bare key 'float16' in the predictor conf
        # (video_prediction/setup_predictor.py:92-95: placeholders and model in tf.float16).  Its counterpart here is
        # the split-bf16 mode: the same kind of opt-in for speed, but with fp32-class accuracy.
        default_precision = 'bf16x6' if 'float16' in hp else os.environ.get('VF_PRECISION', 'fp32')
        precision = hp.get('precision', default_precision)
        self.precision = {'fp32': 0, '0': 0, 0: 0, 'bf16x6': 1, '1': 1, 1: 1}[precision]
        self._c_cfg = _lib.VfConfig(c.height, c.width, c.adim, c.sdim, c.ndesig, c.n_context,
                                    c.sequence_length, c.num_masks, self.run_batch_size,
                                    self.device_index, self.precision, self.n_cam, self.n_draws, c.arch_id,
                                    getattr(c, 'zdim', 0), getattr(c, 'layer_spec', 0))
        self._handle = ctypes.c_void_p()
        _lib.check(self._libh.vf_create(ctypes.byref(self._c_cfg), ctypes.byref(self._handle)))
        self.set_dedup(int(hp.get('dedup', os.environ.get('VF_DEDUP', 1))))
        self.set_persistent(int(hp.get('persistent', os.environ.get('VF_PERSISTENT', 1))))
        self.set_xcd_queues(int(hp.get('xcd_queues', os.environ.get('VF_XCD_QUEUES', 1))))
        self.set_fuse_top(int(hp.get('fuse_top', os.environ.get('VF_FUSE_TOP', 1))))
        self.weights = None
        self._ctx_key = None
        self._last_M = 0
        self._last_lo = 0
        self._last_prepared = None      # (engine context, sequences, M) of the last score() / __call__
        # in-process multi-GPU: this object is lane 0, the others are plain engines on the following devices
        self.gather = str(hp.get('gather', 'auto'))         # 'auto' | 'rccl' | 'host'
        if self.gather not in ('auto', 'rccl', 'host'):
            raise ValueError("gather must be 'auto', 'rccl' or 'host'")
        if world == 1 and self.n_gpus > 1 and not hp.get('_lane'):
            lane_hp = dict(hp, _lane=True, oversubscribe_gpus=1)
            self._lanes = [self] + [HipVPredEvaluation(model_path, lane_hp, n_gpus=1, first_gpu=int(first_gpu) + i)
                                    for i in range(1, self.n_gpus)]
            distinct = len({l.device_index for l in self._lanes}) == self.n_gpus
            if self.gather == 'rccl' and not distinct:
                raise ValueError("gather='rccl' needs one distinct GPU per lane")
            self._use_rccl = distinct if self.gather == 'auto' else self.gather == 'rccl'

    def _all_lanes(self):
        return self._lanes if self._lanes else [self]

    def __del__(self):
        try:
            for comm in (getattr(self, '_comms', None) or []):
                self._libh.vf_comm_destroy(comm)
            self._comms = None
            if getattr(self, '_handle', None) and self._handle.value:
                self._libh.vf_destroy(self._handle)
                self._handle = ctypes.c_void_p()
        except Exception:   # interpreter shutdown
            pass

    # ------------------------------------------------------------------ measurement hooks
    def set_profiling(self, enable):
        _lib.check(self._libh.vf_set_profiling(self._handle, int(bool(enable))))

    def get_profile(self):
        """-> (kernel_ms, launches, flops, busy_ms) of the conv-LSTM kernel since the last call."""
        ms, n, fl, busy = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        _lib.check(self._libh.vf_get_profile(self._handle, ctypes.byref(ms), ctypes.byref(n),
                                             ctypes.byref(fl), ctypes.byref(busy)))
        return ms.value, n.value, fl.value, busy.value

    def set_persistent(self, enable):
        """Run each rollout as one persistent launch (bit-identical results; see vf_persistent.h)."""
        for lane in self._all_lanes():
            with self._torch.cuda.device(lane.device):
                _lib.check(lane._libh.vf_set_persistent(lane._handle, int(bool(enable))))
            lane.persistent = bool(enable)

    def set_xcd_queues(self, enable):
        """One ticket queue per XCD (default) or plain phase order; placement only, bit-identical results."""
        for lane in self._all_lanes():
            with self._torch.cuda.device(lane.device):
                _lib.check(lane._libh.vf_set_xcd_queues(lane._handle, int(bool(enable))))
            lane.xcd_queues = bool(enable)

    def set_fuse_top(self, enable):
        """Top transposed conv + compositing as one item per tile (vf_set_fuse_top); bit-identical results."""
        for lane in self._all_lanes():
            with self._torch.cuda.device(lane.device):
                _lib.check(lane._libh.vf_set_fuse_top(lane._handle, int(bool(enable))))
            lane.fuse_top = bool(enable)

    def set_sched_option(self, option, value):
        """Timing-only options of the persistent launch (``vf_set_sched_option``): ``'yield_budget'`` (0 = off, -1 = automatic)
        and ``'write_through'`` (0 / 1).  Results are bit-identical for every setting."""
        code = {'yield_budget': 0, 'write_through': 1}[option]
        for lane in self._all_lanes():
            with self._torch.cuda.device(lane.device):
                _lib.check(lane._libh.vf_set_sched_option(lane._handle, code, int(value)))

    def device_status(self):
        """Synchronise, return the sticky failure word of the persistent kernel (0 = healthy), re-arm it."""
        worst = 0
        for lane in self._all_lanes():
            st = ctypes.c_int32()
            with self._torch.cuda.device(lane.device):
                _lib.check(lane._libh.vf_device_status(lane._handle, ctypes.byref(st)))
            worst = max(worst, st.value)
        return worst

    def _check_scores(self, scores_np):
        """A rollout whose tiles gave up waiting poisons its scores with NaN (vf_hip.h): never hand
        them to the elite selection."""
        if np.isnan(scores_np).any():
            for lane in self._all_lanes():      # the engine drops its context-only cache with the status:
                lane._ctx_key = None            # upload the context again
            raise _lib.VfError('the persistent rollout kernel reported a failure (device status %d): a tile gave '
                               'up waiting for its producers; scores are invalid' % self.device_status())

    def set_dedup(self, enable):
        """Switch context de-duplication (bit-identical results either way; for A/B timing)."""
        for lane in self._all_lanes():
            with self._torch.cuda.device(lane.device):
                _lib.check(lane._libh.vf_set_dedup(lane._handle, int(bool(enable))))

    # ------------------------------------------------------------------ weights
    def restore(self, weights=None):
        """Load ``model_path`` (manifest.json + weights.bin; ``view%d/`` sub-directories or a list of paths for
        several views) or, with no path, seeded random weights (seed + view)."""
        if weights is None:
            weights = []
            for v in range(self.n_cam):
                path = self.model_path[v] if isinstance(self.model_path, (list, tuple)) else self.model_path
                if path and self.n_cam > 1 and os.path.isdir(os.path.join(path, 'view%d' % v)):
                    path = os.path.join(path, 'view%d' % v)
                weights.append(CdnaWeights.load(path, self.cfg) if path else
                               CdnaWeights.random(self.cfg, seed=self.seed + v))
        elif not isinstance(weights, (list, tuple)):
            weights = [weights]
        if len(weights) != self.n_cam:
            raise ValueError('need one weight set per view (%d), got %d' % (self.n_cam, len(weights)))
        self.weights = weights[0] if self.n_cam == 1 else list(weights)
        self._ctx_key = None
        blob = np.concatenate([v.ravel() for w in weights for v in w.tensors.values()]).astype(np.float32)
        want = self._libh.vf_weight_count(ctypes.byref(self._c_cfg)) * self.n_cam
        if blob.size != want:
            raise _lib.VfError('weight blob has %d floats, library expects %d' % (blob.size, want))
        with self._torch.cuda.device(self.device):      # the library selects the engine's device: put the caller's back
            _lib.check(self._libh.vf_load_weights(self._handle, blob.ctypes.data_as(ctypes.c_void_p),
                                                  blob.size))
        for lane in (self._lanes or [])[1:]:        # the weights are replicated on every lane's device
            lane.restore(list(weights))
        return self

    # ------------------------------------------------------------------ context
    def _stream(self):
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _set_context(self, context):
        torch, nc, c = self._torch, self.n_context, self.cfg
        ncam = self.n_cam
        frames = np.ascontiguousarray(np.asarray(context['context_frames'])[-nc:, :ncam])
        if frames.dtype != np.uint8 or frames.shape != (nc, ncam, c.height, c.width, 3):
            raise ValueError('context_frames must be uint8 [>=%d, %d, %d, %d, 3], got %s %s'
                             % (nc, ncam, c.height, c.width, frames.dtype, frames.shape))
        distrib = np.ascontiguousarray(
            np.asarray(context['context_pixel_distributions'], dtype=np.float32)[-nc:, :ncam])
        states = np.ascontiguousarray(np.asarray(context['context_states'], dtype=np.float32)[-nc:])
        if states.shape != (nc, c.sdim) or distrib.shape != (nc, ncam, c.height, c.width, c.ndesig):
            raise ValueError('bad context shapes: states %s distrib %s' % (states.shape, distrib.shape))
        if nc > 1:
            acts = np.asarray(context['context_actions'], dtype=np.float32).reshape(-1, c.adim)[-(nc - 1):]
            if acts.shape[0] != nc - 1:
                raise ValueError('need at least %d executed actions as context' % (nc - 1))
            acts = np.ascontiguousarray(acts)
        else:
            acts = np.zeros((1, c.adim), np.float32)
        # the CEM iterations of one planning call pass the same context: upload it (and let the
        # engine recompute the context-only part of the network) only when it actually changed
        key = (frames, states, acts, distrib)
        if self._ctx_key is not None and all(np.array_equal(a, b) for a, b in zip(key, self._ctx_key)):
            return
        self._ctx_key = tuple(a.copy() for a in key)
        dev = self.device
        self._ctx = [torch.from_numpy(a).to(dev) for a in (frames, states, acts, distrib)]
        f, s, a, d = self._ctx
        _lib.check(self._libh.vf_set_context(self._handle, f.data_ptr(), s.data_ptr(), a.data_ptr(),
                                             d.data_ptr(), self._stream()))

    # ------------------------------------------------------------------ rollouts
    def _rollout_chunk(self, actions_dev, goal_pix, finalweight, scores_dev, per_task_dev, task_weights=None):
        B = actions_dev.shape[0]
        ntask = self.n_cam * self.cfg.ndesig
        goal = np.asarray(goal_pix).reshape(-1)
        if goal.size != 2 * ntask:
            raise ValueError('goal_pix must hold [ncam=%d][ndesig=%d][2] values, got %d'
                             % (self.n_cam, self.cfg.ndesig, goal.size))
        goal_c = (ctypes.c_int32 * (2 * ntask))(*[int(v) for v in goal])
        tw = None
        if task_weights is not None:
            w = np.asarray(task_weights, dtype=np.float64).reshape(-1)
            if w.size != ntask:
                raise ValueError('task_weights must hold ncam*ndesig = %d values' % ntask)
            tw = (ctypes.c_float * ntask)(*[float(v) for v in w])
        _lib.check(self._libh.vf_rollout(self._handle, actions_dev.data_ptr(), B, goal_c,
                                         ctypes.c_float(finalweight), tw, scores_dev.data_ptr(),
                                         per_task_dev.data_ptr(), self._stream()))

    def _check_actions(self, actions):
        actions = np.asarray(actions)
        T = self.sequence_length - self.n_context
        if actions.ndim != 3 or actions.shape[1] != T or actions.shape[2] != self._adim_in():
            raise ValueError('actions must be [M, %d, %d], got %s' % (T, self._adim_in(), actions.shape))
        return actions

    # hooks of the stochastic predictor: the caller's actions -> the sequences the engine rolls
    def _adim_in(self):
        return self.cfg.adim

    def _prepare(self, context, actions):
        """(context, actions[n]) -> (engine context, sequences[n * n_draws])."""
        return context, actions

    def _score_prepared(self, context, seqs, n, goal_pix, finalweight, index_base=0, task_weights=None):
        """Roll ``seqs [n * n_draws, T, engine adim]`` (as ``_prepare`` returns them) on this engine -> device tensors
        (scores[n], per_task[n, ncam*nd]).  ``index_base`` is the global index of the first action (what
        ``fetch_pixel_distributions`` is asked for).  Only enqueues work on this engine's device (the uploads of
        pageable host arrays aside)."""
        torch = self._torch
        ntask = self.n_cam * self.cfg.ndesig
        nd = self.n_draws
        self._set_context(context)
        local = torch.from_numpy(np.ascontiguousarray(seqs, dtype=np.float32)).to(self.device)
        scores = torch.empty(n, dtype=torch.float64, device=self.device)
        per_task = torch.empty((n, ntask), dtype=torch.float64, device=self.device)
        bs = self.run_batch_size // nd          # actions per chunk
        for c0 in range(0, n, bs):
            c1 = min(c0 + bs, n)
            self._rollout_chunk(local[c0 * nd:c1 * nd], goal_pix, finalweight, scores[c0:c1], per_task[c0:c1],
                                task_weights)
            self._last_lo, self._last_M = index_base + c0, c1 - c0
        return scores, per_task

    def score(self, context, inputs, goal_pix, finalweight=10., only_take_first_view=False, task_weights=None):
        """Fused rollout + expected-pixel-distance cost.  Returns (scores[M], scores_per_task[M, ncam*nd]) float64.

        ``task_weights`` ([ncam, ndesig] trade-off weights, reference ``register_gtruth_controller.py:88-94``)
        replace the plain mean over tasks.  Only the chunk evaluated last stays resident for
        ``fetch_pixel_distributions``; with ``M <= run_batch_size`` (the reference default,
        ``pixel_cost_controller.py:31``) that is the whole local shard.
        """
        actions = self._check_actions(inputs['actions'])
        M = actions.shape[0]
        rank, world = _dist_info()
        # every rank / lane slices the SAME prepared sequences (latent draws included); they are also what a
        # propagation fetch of a sample that is no longer resident is re-rolled from
        ctx_p, seqs = self._prepare(context, actions)
        self._last_prepared = (ctx_p, seqs, M)
        nd = self.n_draws
        if self._lanes:
            scores_np, per_task_np = self._score_lanes(ctx_p, seqs, M, goal_pix, finalweight, task_weights)
            self._check_scores(scores_np)
            if only_take_first_view:
                per_task_np = per_task_np[:, :1]
                scores_np = per_task_np[:, 0].copy()
            return scores_np, per_task_np
        lo, hi = shard_bounds(M, rank, world)
        with self._torch.cuda.device(self.device):
            scores, per_task = self._score_prepared(ctx_p, seqs[lo * nd:hi * nd], hi - lo, goal_pix, finalweight,
                                                    index_base=lo, task_weights=task_weights)
            if world > 1:
                scores, per_task = self._all_gather(scores, per_task, M, world)
            scores_np = scores.cpu().numpy()
            per_task_np = per_task.cpu().numpy()
        self._check_scores(scores_np)
        if only_take_first_view:
            per_task_np = per_task_np[:, :1]
            scores_np = per_task_np[:, 0].copy()
        return scores_np, per_task_np

    # ------------------------------------------------------------------ in-process multi-GPU (n_gpus > 1)
    def _score_lanes(self, context, seqs, M, goal_pix, finalweight, task_weights):
        """Lane i rolls the contiguous shard ``shard_bounds(M, i, n_gpus)`` of the prepared sequences on its own
        device; every lane's work is enqueued before anything is waited for, then one gather of the
        ``[M, 1 + tasks]`` score rows."""
        torch = self._torch
        lanes, nd = self._lanes, self.n_draws
        packed = []
        for i, lane in enumerate(lanes):
            lo, hi = shard_bounds(M, i, len(lanes))
            with torch.cuda.device(lane.device):
                if hi > lo:
                    s, pt = lane._score_prepared(context, seqs[lo * nd:hi * nd], hi - lo, goal_pix, finalweight,
                                                 index_base=lo, task_weights=task_weights)
                    packed.append(torch.cat([s[:, None], pt], dim=1).contiguous())
                else:
                    lane._last_lo, lane._last_M = lo, 0
                    packed.append(torch.empty((0, 1 + self.n_cam * self.cfg.ndesig), dtype=torch.float64,
                                              device=lane.device))
        rows = None
        if self._use_rccl:
            try:
                rows = self._gather_rccl(packed, M)
            except _lib.VfError as e:
                if self.gather != 'auto':
                    raise
                # 'auto' promises scores, not a transport: say so once and gather through the host from now on
                print('HipVPredEvaluation: grouped RCCL all-gather unavailable (%s); gathering score rows through '
                      'the host' % e)
                self._use_rccl = False
        if rows is None:
            rows = np.concatenate([p.cpu().numpy() for p in packed], axis=0)
        return np.ascontiguousarray(rows[:, 0]), np.ascontiguousarray(rows[:, 1:])

    def _gather_rccl(self, packed, M):
        """One grouped RCCL all-gather over the lanes' devices (padded to equal rows); lane 0's copy goes to the host."""
        torch, n = self._torch, len(self._lanes)
        P = ctypes.c_void_p
        if self._comms is None:
            devs = (ctypes.c_int32 * n)(*[l.device_index for l in self._lanes])
            comms = (P * n)()
            _lib.check(self._libh.vf_comm_init_all(n, devs, comms))
            self._comms = [P(c) for c in comms]
        sizes = [p.shape[0] for p in packed]
        width, cols = max(sizes), packed[0].shape[1]
        local, full = [], []
        for lane, p in zip(self._lanes, packed):
            with torch.cuda.device(lane.device):
                buf = torch.zeros((width, cols), dtype=torch.float64, device=lane.device)
                buf[:p.shape[0]] = p
                local.append(buf)
                full.append(torch.empty((n * width, cols), dtype=torch.float64, device=lane.device))
        arr = lambda items: (P * n)(*items)
        with torch.cuda.device(self.device):    # (the library also puts the calling thread's device back itself)
            _lib.check(self._libh.vf_allgather_scores_group(
                n, arr([l._handle for l in self._lanes]), arr(self._comms), arr([P(t.data_ptr()) for t in local]),
                width * cols, arr([P(t.data_ptr()) for t in full]), arr([l._stream() for l in self._lanes])))
        with torch.cuda.device(self.device):
            out = full[0].cpu().numpy().reshape(n, width, cols)
        for lane in self._lanes[1:]:        # every lane's collective has completed before its buffers are released
            torch.cuda.synchronize(lane.device)
        return np.concatenate([out[i, :sizes[i]] for i in range(n)], axis=0)

    def set_collective_timing(self, enable):
        """Bracket every score all-gather with HIP events on the stream the collective is ordered on (the current
        stream of this rank's device: c10d makes it wait for the RCCL stream before the call returns) so a bench line
        can show what the one collective of a CEM iteration costs.  ``collective_stats()`` reads them."""
        self._coll_events = [] if enable else None

    def collective_stats(self):
        """-> {'calls', 'mean_ms', 'max_ms', 'bytes_per_rank'} of the all-gathers since ``set_collective_timing(True)``."""
        ev = getattr(self, '_coll_events', None) or []
        if not ev:
            return {'calls': 0, 'mean_ms': None, 'max_ms': None, 'bytes_per_rank': None}
        self._torch.cuda.synchronize(self.device)
        ms = [a.elapsed_time(b) for a, b, _ in ev]
        return {'calls': len(ms), 'mean_ms': float(np.mean(ms)), 'max_ms': float(np.max(ms)),
                'bytes_per_rank': int(ev[-1][2])}

    def _all_gather(self, scores, per_task, M, world):
        """One collective: every rank's [score | per-task scores] rows -> all M rows on every rank."""
        torch = self._torch
        packed = torch.cat([scores[:, None], per_task], dim=1).contiguous()
        timing = getattr(self, '_coll_events', None)
        if timing is not None and len(timing) < 4096:
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record(torch.cuda.current_stream(self.device))
            out = all_gather_rows(packed, M)
            stop.record(torch.cuda.current_stream(self.device))
            timing.append((start, stop, packed.numel() * packed.element_size()))
        else:
            out = all_gather_rows(packed, M)
        return out[:, 0].contiguous(), out[:, 1:].contiguous()

    def fetch_pixel_distributions(self, sample_index):
        """Normalised distributions ``[T, ncam, H, W, ndesig]`` of one action of the last ``score()`` call (its first
        latent draw) - what ``predictor_propagation`` feeds back as the next context (reference
        ``pixel_cost_controller.py:161-165``, where all predicted distributions sit on the host).

        Here only the chunk rolled last stays resident per engine.  The engine that still holds the sample exports
        it (under ``torch.distributed`` a small all-reduce hands it to the other ranks).  A sample nobody holds any
        more - ``num_samples > vpred_batch_size``, e.g. the reference's 600-sample RoboNet configs - is simply rolled
        again, alone: a sample's arithmetic does not depend on the batch it is rolled in, so the result is
        bit-identical to the first pass, and every rank can do it locally without a collective."""
        torch, c = self._torch, self.cfg
        T = self.sequence_length - self.n_context
        rank, world = _dist_info()
        prepared = getattr(self, '_last_prepared', None)
        if prepared is not None and not 0 <= sample_index < prepared[2]:
            raise IndexError('sample %d outside the last scoring call (%d actions)' % (sample_index, prepared[2]))
        for lane in (self._lanes or [])[1:]:        # in-process multi-GPU: the lane that rolled it exports it
            if 0 <= sample_index - lane._last_lo < lane._last_M:
                return lane.fetch_pixel_distributions(sample_index)
        local = sample_index - self._last_lo
        have = 0 <= local < self._last_M
        out = torch.zeros((T, self.n_cam, c.height, c.width, c.ndesig), dtype=torch.float32, device=self.device)
        if have:
            with torch.cuda.device(self.device):
                _lib.check(self._libh.vf_export(self._handle, int(local) * self.n_draws, 1, None, out.data_ptr(),
                                                None, self._stream()))
        if world > 1:
            import torch.distributed as dist
            # the "somebody holds it" flag travels with the data (one extra element)
            flat = torch.cat([out.reshape(-1), torch.tensor([1.0 if have else 0.0], device=self.device)])
            if dist.get_backend() == 'gloo':
                host = flat.cpu()
                dist.all_reduce(host)
                flat = host.to(self.device)
            else:
                dist.all_reduce(flat)   # exactly one rank holds the sample, the others add zeros
            if float(flat[-1].item()) >= 0.5:
                return flat[:-1].reshape(out.shape).cpu().numpy()
            have = False                # nobody holds it: every rank rolls it again below (identical results)
        if have:
            return out.cpu().numpy()
        if prepared is None:
            raise IndexError('sample %d is not resident (last chunk holds [%d, %d))'
                             % (sample_index, self._last_lo, self._last_lo + self._last_M))
        return self._reroll_one(sample_index, out)

    def _reroll_one(self, sample_index, out):
        """Roll action ``sample_index`` of the last scoring call again (its ``n_draws`` sequences, alone) and export
        the first draw's distributions into ``out``."""
        torch, nd = self._torch, self.n_draws
        ctx_p, seqs, _ = self._last_prepared
        ntask = self.n_cam * self.cfg.ndesig
        with torch.cuda.device(self.device):
            self._set_context(ctx_p)
            seq = torch.from_numpy(np.ascontiguousarray(seqs[sample_index * nd:(sample_index + 1) * nd],
                                                        dtype=np.float32)).to(self.device)
            scores = torch.empty(1, dtype=torch.float64, device=self.device)
            per_task = torch.empty((1, ntask), dtype=torch.float64, device=self.device)
            self._rollout_chunk(seq, np.zeros((self.n_cam, self.cfg.ndesig, 2), np.int32), 1.0, scores, per_task)
            # under torch.distributed EVERY rank has just rolled it: nobody may claim to be "the" holder afterwards
            # (the fetch's all-reduce adds the holders' copies), so the ranks keep nothing resident
            self._last_lo, self._last_M = sample_index, (1 if _dist_info()[1] == 1 else 0)
            _lib.check(self._libh.vf_export(self._handle, 0, 1, None, out.data_ptr(), None, self._stream()))
            self._check_scores(scores.cpu().numpy())
            return out.cpu().numpy()

    # ------------------------------------------------------------------ registration
    def register(self, current, reference, flow, pix, region=0, clip_sub=1, want_warped=False):
        """Device side of ``get_warp_err`` (reference ``register_gtruth_controller.py:113-173``).

        current, reference ``[ncam, H, W, 3]`` float images, flow ``[ncam, H, W, 2]`` (dx, dy) of the plug-in
        registration network, pix ``[ncam, ntask, 2]`` (row, col) in the reference image.  Returns
        ``desig [ncam, ntask, 2]`` (row, col in the current frame), ``err [ncam, ntask]`` and, on request,
        the warped frame ``[ncam, H, W, 3]`` and the warp points ``[ncam, H, W, 2]`` (x, y).
        """
        torch, c = self._torch, self.cfg
        cur = np.ascontiguousarray(current, dtype=np.float32)
        ref = np.ascontiguousarray(reference, dtype=np.float32)
        fl = np.ascontiguousarray(flow, dtype=np.float32)
        px = np.ascontiguousarray(pix, dtype=np.int32).reshape(self.n_cam, -1, 2)
        if cur.shape != (self.n_cam, c.height, c.width, 3) or ref.shape != cur.shape or \
                fl.shape != (self.n_cam, c.height, c.width, 2):
            raise ValueError('register: need [ncam, H, W, 3] images and a [ncam, H, W, 2] flow field')
        ntask = px.shape[1]
        with torch.cuda.device(self.device):
            d_cur, d_ref, d_fl, d_px = (torch.from_numpy(a).to(self.device) for a in (cur, ref, fl, px))
            desig = torch.empty((self.n_cam, ntask, 2), dtype=torch.float32, device=self.device)
            err = torch.empty((self.n_cam, ntask), dtype=torch.float32, device=self.device)
            warped = torch.empty_like(d_cur) if want_warped else None
            pts = torch.empty_like(d_fl) if want_warped else None
            _lib.check(self._libh.vf_register(
                self._handle, d_cur.data_ptr(), d_ref.data_ptr(), d_fl.data_ptr(), d_px.data_ptr(), ntask,
                int(region), int(clip_sub), warped.data_ptr() if want_warped else None,
                pts.data_ptr() if want_warped else None, desig.data_ptr(), err.data_ptr(), self._stream()))
            out = desig.cpu().numpy().astype(np.float64), err.cpu().numpy().astype(np.float64)
            if want_warped:
                out = out + (warped.cpu().numpy(), pts.cpu().numpy())
        return out

    def predictor_func(self):
        """The legacy boundary (reference ``video_prediction/setup_predictor.py:164-200``): a callable

            ``f(input_images=, input_one_hot_images=, input_state=, input_actions=) -> (gen_images, gen_distrib, gen_states)``

        as ``rollout_predictions`` (``pred_util.py:21-48``) drives it.  ``input_images`` is the batch-1
        float context ``[1, n_context, 1, H, W, 3]`` in [0, 1] (``get_context``), ``input_state``
        ``[1, n_context, sdim]``, ``input_one_hot_images`` ``[1, n_context, 1, H, W, ndesig]`` or None and
        ``input_actions`` ``[B, sequence_length (or sequence_length - 1), adim]`` whose first
        ``n_context - 1`` steps are the executed actions (identical for every row, as the reference
        tiles them).  Returns the ``sequence_length - n_context`` future frames
        ``[B, T, 1, H, W, 3]``, distributions ``[B, T, 1, H, W, ndesig]`` (None without input
        distributions) and states ``[B, T, sdim]``.
        """
        nc, c = self.n_context, self.cfg
        T = self.sequence_length - nc
        if self.n_cam != 1 or self.n_draws != 1:
            raise NotImplementedError('the legacy predictor_func boundary is single-view, deterministic')

        def predictor_func(input_images=None, input_one_hot_images=None, input_state=None, input_actions=None):
            acts = np.asarray(input_actions, dtype=np.float64)
            if acts.ndim != 3 or acts.shape[1] not in (self.sequence_length, self.sequence_length - 1):
                raise ValueError('input_actions must be [B, %d, %d], got %s'
                                 % (self.sequence_length, c.adim, acts.shape))
            ctx_actions, future = acts[:, :nc - 1], acts[:, nc - 1:nc - 1 + T]
            padding = ~acts.reshape(acts.shape[0], -1).any(axis=1)      # zero rows of a padded last chunk
            if nc > 1 and not np.all((ctx_actions == ctx_actions[:1]).all(axis=(1, 2)) | padding):
                raise ValueError('the first n_context-1 actions are context and must be the same for every row')
            frames = np.asarray(input_images)[0]
            if frames.dtype != np.uint8:        # get_context hands over float frames in [0, 1]
                frames = np.rint(frames * 255.).astype(np.uint8)
            if input_one_hot_images is None:
                distrib = np.zeros((nc, 1, c.height, c.width, c.ndesig), np.float32)
                distrib[:, :, 0, 0, :] = 1.0
            else:
                distrib = np.asarray(input_one_hot_images, dtype=np.float32)[0]
            context = {'context_frames': frames, 'context_states': np.asarray(input_state)[0],
                       'context_actions': ctx_actions[0] if nc > 1 else np.zeros((1, c.adim)),
                       'context_pixel_distributions': distrib}
            out = self(context, {'actions': future})
            return (out['predicted_frames'],
                    None if input_one_hot_images is None else out['predicted_pixel_distributions'],
                    out['predicted_states'])

        return predictor_func

    def __call__(self, context, inputs):
        """Reference-compatible path: materialise all predicted frames and distributions on the host
        (``[M, T, ncam, H, W, C]``; with latent draws, the first draw of every action)."""
        actions = self._check_actions(inputs['actions'])
        context, seqs = self._prepare(context, actions)
        self._last_prepared = (context, seqs, actions.shape[0])
        if self._lanes:     # lane i materialises its contiguous shard (one lane after the other: the D2H dominates)
            M, n, nd = actions.shape[0], len(self._lanes), self.n_draws
            parts = []
            for i, lane in enumerate(self._lanes):
                lo, hi = shard_bounds(M, i, n)
                if hi > lo:
                    parts.append(lane._materialise(context, seqs[lo * nd:hi * nd], hi - lo, lo))
                else:           # nothing of THIS call is resident on the lane (a stale range must not match a fetch)
                    lane._last_lo, lane._last_M = lo, 0
            if not parts:       # M == 0: empty arrays of the right shapes
                return self._materialise(context, seqs, 0, 0)
            return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}
        return self._materialise(context, seqs, actions.shape[0], 0)

    def _materialise(self, context, seqs, M, index_base):
        torch, c = self._torch, self.cfg
        T = self.sequence_length - self.n_context
        ncam, nd = self.n_cam, self.n_draws
        frames = np.empty((M, T, ncam, c.height, c.width, 3), np.float32)
        distrib = np.empty((M, T, ncam, c.height, c.width, c.ndesig), np.float32)
        states = np.empty((M, T, c.sdim), np.float32)
        zero_goal = np.zeros((ncam, c.ndesig, 2), np.int32)
        bs = self.run_batch_size // nd
        with torch.cuda.device(self.device):
            self._set_context(context)
            acts = torch.from_numpy(np.ascontiguousarray(seqs, dtype=np.float32)).to(self.device)
            scores = torch.empty(bs, dtype=torch.float64, device=self.device)
            per_task = torch.empty((bs, ncam * c.ndesig), dtype=torch.float64, device=self.device)
            for c0 in range(0, M, bs):
                c1 = min(c0 + bs, M)
                n = c1 - c0
                self._rollout_chunk(acts[c0 * nd:c1 * nd], zero_goal, 1.0, scores[:n], per_task[:n])
                self._last_lo, self._last_M = index_base + c0, n
                f = torch.empty((n * nd, T, ncam, c.height, c.width, 3), dtype=torch.float32, device=self.device)
                d = torch.empty((n * nd, T, ncam, c.height, c.width, c.ndesig), dtype=torch.float32,
                                device=self.device)
                s = torch.empty((n * nd, T, c.sdim), dtype=torch.float32, device=self.device)
                _lib.check(self._libh.vf_export(self._handle, 0, n * nd, f.data_ptr(), d.data_ptr(), s.data_ptr(),
                                                self._stream()))
                frames[c0:c1] = f[::nd].cpu().numpy()
                distrib[c0:c1] = d[::nd].cpu().numpy()
                states[c0:c1] = s[::nd].cpu().numpy()
            if M > 0:
                self._check_scores(scores[:n].cpu().numpy())
            else:
                self._last_lo, self._last_M = index_base, 0
        return {'predicted_frames': frames, 'predicted_pixel_distributions': distrib,
                'predicted_states': states}


class MultiViewHipPredictor(HipVPredEvaluation):
    """``ncam`` independent single-view networks behind the ``VPredEvaluation`` duck-type (the reference's
    ``IndepMultiSAVPVideoPredictionModel``: one network per view sharing actions and states, outputs stacked on a camera
    axis, ``visual_mpc/video_prediction/vpred_model_interface.py:60-88``).  The engine is the base class's - every view
    of every sample rolls in the SAME persistent launch, per-task scores come back camera-major - this class only fixes
    the construction conventions: ``ncam`` defaults to 2, ``model_path`` is a list of per-view directories or a directory
    with ``view0/``, ``view1/`` ... inside, random-init seeds are ``seed + view``."""

    def __init__(self, model_path, hparams, n_gpus=1, first_gpu=0):
        hp = dict(hparams)
        hp.setdefault('ncam', 2)
        super(MultiViewHipPredictor, self).__init__(model_path, hp, n_gpus=n_gpus, first_gpu=first_gpu)

    @staticmethod
    def view_context(context, c):
        """The single-view slice of a multi-view context (what one view's network sees)."""
        return {'context_frames': np.asarray(context['context_frames'])[:, c:c + 1],
                'context_pixel_distributions': np.asarray(context['context_pixel_distributions'])[:, c:c + 1],
                'context_actions': context['context_actions'], 'context_states': context['context_states']}
