#!/usr/bin/env python
"""GPU box diagnostic: the per-step dependency chain of a rollout, phase by phase.

Needs the event-log build (see tools/trace_cu.py):
    VF_LIBRARY=build/ab/trace.so python tools/trace_chain.py [M] [first_step] [n_steps]
Every item logs the index of the phase it belongs to (TR_PHASE); this script lists, for the phases of a few steps in
the middle of the launch, when the first item of the phase started running, when its last item finished, how long an
item ran on average (prologue to publish, waits inside the item included) and how long its items waited in front for
their producers - i.e. where the ~0.9 ms of a chain-bound step (25 samples) or the step-boundary window of a full batch
(200 samples) goes.  Times are relative to the first event of the first listed phase.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from visual_foresight_amd import _lib  # noqa: E402
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation  # noqa: E402

M, T = int(sys.argv[1]) if len(sys.argv) > 1 else 25, 13
step0 = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
WGS, MAXE, TICK_US = 512, 8192, 0.01
TR_TICKET, TR_DONE, TR_PHASE, TR_RUN = 1, 3, 21, 32
TR_STAGE, TR_KLOOP, TR_LATE, TR_LATE_END, TR_EPI = 10, 11, 12, 13, 14
NAMES = ['LSTM', 'CONV_RELU', 'CONV_RAW', 'CONVT_RELU', 'CONVT_RAW', 'FC', 'SA', 'FIN', 'COMPOSITE', 'TOP_FUSED', 'CONV_PAIR']

pred = HipVPredEvaluation('', dict(designated_pixel_count=1, run_batch_size=M, sequence_length=T + 2)).restore()
rs = np.random.RandomState(0)
d = np.zeros((2, 1, 64, 64, 1), np.float32)
d[:, 0, 32, 32, 0] = 1
ctx = {'context_frames': rs.randint(0, 256, (2, 1, 64, 64, 3)).astype(np.uint8), 'context_actions': np.zeros((1, 4)),
       'context_states': np.zeros((2, 5)), 'context_pixel_distributions': d}
acts = rs.normal(0, 0.05, (M, T, 4))
lib = _lib.load_library()
pred.score(ctx, {'actions': acts}, [[[16, 48]]])
_lib.check(lib.vf_set_phase_stats(pred._handle, 1))
pred.score(ctx, {'actions': acts}, [[[16, 48]]])        # the cached-context schedule: what a CEM iteration runs
N = 4096
types, items = (ctypes.c_int32 * N)(), (ctypes.c_int32 * N)()
wr = (ctypes.c_uint64 * (2 * N))()
n_ph = lib.vf_debug_phase_stats(pred._handle, N, types, items, wr)
ev = np.zeros(WGS * MAXE, np.uint64)
cnt = np.zeros(WGS, np.uint32)
lib.vf_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_lib.check(lib.vf_debug_trace(ev.ctypes.data, cnt.ctypes.data))
ev = ev.reshape(WGS, MAXE)

first = np.full(n_ph, np.inf)
last = np.zeros(n_ph)
run_sum, wait_sum, seen = np.zeros(n_ph), np.zeros(n_ph), np.zeros(n_ph, int)
# the part of an early-started item behind its mid-item wait - what sits on the sample's dependency chain:
# [late wait, wait end -> first staging (LayerNorm table), staging, K loops, epilogue + publish], n items with a late wait
tail = np.zeros((n_ph, 5))
n_late = np.zeros(n_ph, int)
for w in range(WGS):
    n = int(cnt[w])
    codes = (ev[w, :n] & np.uint64(255)).astype(np.int64)
    vals = (ev[w, :n] >> np.uint64(8)).astype(np.int64)
    t_ticket = t_run = None
    ph = -1
    seg = []            # (code, time) of the item's events behind TR_LATE
    t_late = None
    for c, v in zip(codes, vals):
        if c == TR_TICKET:
            t_ticket = v
        elif c >= TR_RUN:
            t_run = v
            seg, t_late = [], None
        elif c == TR_PHASE:
            ph = int(v)
        elif c == TR_LATE:
            t_late = v
            seg = []
        elif c in (TR_LATE_END, TR_STAGE, TR_KLOOP, TR_EPI) and t_late is not None:
            seg.append((int(c), v))
        elif c == TR_DONE and ph >= 0 and t_run is not None:
            if ph < n_ph and t_late is not None and seg and seg[0][0] == TR_LATE_END:
                n_late[ph] += 1
                tail[ph, 0] += seg[0][1] - t_late
                cur, t_cur = 1, seg[0][1]       # 1 = LayerNorm table, 2 = staging, 3 = K, 4 = epilogue + publish
                for code, t in seg[1:] + [(TR_DONE, v)]:
                    tail[ph, cur] += t - t_cur
                    t_cur = t
                    cur = {TR_STAGE: 2, TR_KLOOP: 3, TR_EPI: 4}.get(code, cur)
            if ph < n_ph:
                first[ph] = min(first[ph], t_run)
                last[ph] = max(last[ph], v)
                run_sum[ph] += v - t_run
                wait_sum[ph] += (t_run - t_ticket) if t_ticket is not None else 0
                seen[ph] += 1
            ph = -1

# steps: a step starts at every phase of type SA (state FC) in the cached-context schedule
starts = [i for i in range(n_ph) if types[i] == NAMES.index('SA')]
starts.append(n_ph)
lo = starts[min(step0, len(starts) - 2)]
hi = starts[min(step0 + nsteps, len(starts) - 1)]
t0 = min(first[lo:hi][np.isfinite(first[lo:hi])])
print('M = %d, %d phases, %d steps; phases %d..%d (steps %d..%d); times in us relative to the first listed item' % (
    M, n_ph, len(starts) - 1, lo, hi - 1, step0, step0 + nsteps - 1))
print('%5s %-11s %6s %9s %9s %9s %10s %10s' % ('phase', 'type', 'items', 'first run', 'last done', 'span', 'run/item', 'wait/item'))
prev_done = None
for i in range(lo, hi):
    if not seen[i]:
        continue
    f, l = (first[i] - t0) * TICK_US, (last[i] - t0) * TICK_US
    print('%5d %-11s %6d %9.1f %9.1f %9.1f %10.1f %10.1f' % (i, NAMES[types[i]], items[i], f, l, l - f,
                                                         run_sum[i] / seen[i] * TICK_US, wait_sum[i] / seen[i] * TICK_US))
print('behind the mid-item wait (us per item): phase type  late-wait | LN table  staging  K loops  epilogue+publish')
for i in range(lo, hi):
    if n_late[i]:
        t = tail[i] / n_late[i] * TICK_US
        print('%5d %-11s %8.1f | %7.1f %7.1f %7.1f %7.1f   = %.1f on the chain' % (i, NAMES[types[i]], t[0], t[1], t[2], t[3], t[4], t[1:].sum()))
print('step length: %.1f us' % ((min(first[hi:hi + 3][np.isfinite(first[hi:hi + 3])]) - t0) * TICK_US / nsteps
                                if hi + 3 <= n_ph else float('nan')))
