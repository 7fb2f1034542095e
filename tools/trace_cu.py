#!/usr/bin/env python
"""GPU box diagnostic: what the two resident workgroups of a CU are doing, and how fast the matrix pipe runs meanwhile.

Needs the event-log build of the library:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -DVF_TRACE -o build/ab/trace.so \
          visual_foresight_amd/csrc/vf_engine.hip
    VF_LIBRARY=build/ab/trace.so python tools/trace_cu.py [M] [precision]
Every workgroup of the persistent rollout logs (time, event) pairs; this script turns each log into state intervals
(K loop / staging / mid-item wait / prologue / epilogue / light item / dependency wait), pairs the workgroups that share
a CU, and reports (a) the share of the launch each JOINT state takes, (b) the MFMA issue rate of a K loop while the
other workgroup is in its own K loop and while it is not (least squares over all K-loop intervals).
"""
import ctypes
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from visual_foresight_amd import _lib  # noqa: E402
from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation  # noqa: E402

M, T = int(sys.argv[1]) if len(sys.argv) > 1 else 200, 13
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
# optional: network, image size, horizon, layer_spec, events per workgroup of the trace build (-DVF_TRACE_MAX)
#   VF_LIBRARY=build/ab/trace64k.so python tools/trace_cu.py 625 fp32 savp3 128 2 0 65536
ARCH = sys.argv[3] if len(sys.argv) > 3 else 'cdna'
HH = int(sys.argv[4]) if len(sys.argv) > 4 else 64
T = int(sys.argv[5]) if len(sys.argv) > 5 else T
SPEC = int(sys.argv[6]) if len(sys.argv) > 6 else 0
WGS, MAXE = 512, int(sys.argv[7]) if len(sys.argv) > 7 else 8192
TICK_US = 0.01
TR_TICKET, TR_DONE, TR_STAGE, TR_KLOOP, TR_LATE, TR_LATE_END, TR_EPI, TR_MFMAS, TR_HWID, TR_RUN = 1, 3, 10, 11, 12, 13, 14, 15, 20, 32
TR_ST_LOADED, TR_ST_WRITTEN = 16, 17
TR_YIELD, TR_YIELD_END = 18, 19    # a recurrent half asleep for its CU partner (round 5), inside a K loop
PH_LSTM = 0
PH_NAMES = ['LSTM', 'CONV_RELU', 'CONV_RAW', 'CONVT_RELU', 'CONVT_RAW', 'FC', 'SA', 'FIN', 'COMPOSITE', 'TOP_FUSED', 'CONV_PAIR',
            'COND', 'CONV_RAW3', 'GATES_RAW', 'EW', 'CONV_RAW3G2', 'CONV_RAW3G4']
PH_GATES_RAW = 13       # arch 3's gate GEMM: the gate-split K loop with the raw epilogue - a conv-LSTM K loop for this analysis

adim = 12 if ARCH.startswith('savp') else 4
extra = dict(arch=ARCH, adim=adim, image_height=HH, image_width=HH) if ARCH != 'cdna' or HH != 64 else {}
if ARCH == 'savp3':
    extra.update(zdim=8, layer_spec=SPEC)
pred = HipVPredEvaluation('', dict(designated_pixel_count=1, run_batch_size=M, sequence_length=T + 2, precision=prec, **extra)).restore()
rs = np.random.RandomState(0)
d = np.zeros((2, 1, HH, HH, 1), np.float32)
d[:, 0, HH // 2, HH // 2, 0] = 1
ctx = {'context_frames': rs.randint(0, 256, (2, 1, HH, HH, 3)).astype(np.uint8), 'context_actions': np.zeros((1, adim)),
       'context_states': np.zeros((2, 5)), 'context_pixel_distributions': d}
acts = rs.normal(0, 0.05, (M, T, adim))
lib = _lib.load_library()
pred.score(ctx, {'actions': acts}, [[[16, 48]]])
pred.set_profiling(True)
pred.score(ctx, {'actions': acts}, [[[16, 48]]])
k_ms = pred.get_profile()[0]
ev = np.zeros(WGS * MAXE, np.uint64)
cnt = np.zeros(WGS, np.uint32)
lib.vf_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_lib.check(lib.vf_debug_trace(ev.ctypes.data, cnt.ctypes.data))
ev = ev.reshape(WGS, MAXE)
print('kernel %.2f ms (event-log build), M = %d, %s; events per workgroup: mean %.0f max %d' % (k_ms, M, prec, cnt.mean(), cnt.max()))

# ---- per workgroup: events -> intervals (t0, t1, state, mfmas)
wg_cu, wg_iv = {}, {}
t_min, t_max = None, None
for w in range(WGS):
    n = int(cnt[w])
    if n == 0:
        continue
    codes = (ev[w, :n] & np.uint64(255)).astype(np.int64)
    vals = (ev[w, :n] >> np.uint64(8)).astype(np.int64)
    iv = []
    state, t_state, lstm, mf, cid = None, None, False, 0, 0

    def close(t, new_state):
        global state, t_state
        if state is not None and t > t_state:
            iv.append((t_state, t, state, mf if state == 'K' else 0, cid))
        state, t_state = new_state, t

    for c, v in zip(codes, vals):
        if c == TR_HWID:
            wg_cu[w] = (int(v >> 32), int((v >> 8) & 0xFF))
        elif c == TR_MFMAS:
            mf = int(v)
        elif c == TR_TICKET:
            close(v, 'wait')
        elif c >= TR_RUN:
            lstm = (c - TR_RUN) in (PH_LSTM, PH_GATES_RAW)
            if state == 'wait':
                state = 'wait:' + PH_NAMES[c - TR_RUN]      # the wait in front of an item belongs to that item's type
            close(v, 'pro' if lstm else 'light:' + PH_NAMES[c - TR_RUN])
        elif c == TR_DONE:
            close(v, 'sched')
        elif not lstm and state is not None and state.startswith('light:'):
            base = ':'.join(state.split(':')[:2])
            sub = {TR_STAGE: 'stage', TR_KLOOP: 'K', TR_EPI: 'epi', TR_LATE: 'late', TR_LATE_END: 'stage',
                   24: 'epi:halo', 25: 'epi:mates', 26: 'epi:compose'}.get(int(c))
            if sub:
                close(v, base + ':' + sub)
        elif lstm:
            if c == TR_STAGE:
                close(v, 'stage')
            elif c == TR_KLOOP:
                close(v, 'K')
                cid += 1                # the K loop of one chunk: one sample of the rate regression, yields cut out
            elif c == TR_YIELD:
                close(v, 'yield')
            elif c == TR_YIELD_END:
                close(v, 'K')
            elif c == TR_ST_LOADED:
                close(v, 'stage:ln+write')
            elif c == TR_ST_WRITTEN:
                close(v, 'stage:barrier')
            elif c == TR_LATE:
                close(v, 'late')
            elif c == TR_LATE_END:
                close(v, 'stage')
            elif c == TR_EPI:
                close(v, 'epi')
    wg_iv[w] = iv
    if iv:
        t_min = iv[0][0] if t_min is None else min(t_min, iv[0][0])
        t_max = iv[-1][1] if t_max is None else max(t_max, iv[-1][1])

span = (t_max - t_min) * TICK_US
print('traced span %.2f ms' % (span / 1e3))
tot = defaultdict(float)
for w, iv in wg_iv.items():
    for t0, t1, s, _, _c in iv:
        tot[s.split(':')[0] if s.startswith(('light', 'wait')) else s] += (t1 - t0) * TICK_US
nw = len(wg_iv)
print('per workgroup slot (mean over %d), ms: ' % nw + '  '.join('%s %.2f' % (k, v / nw / 1e3) for k, v in sorted(tot.items())))
light, lightn = defaultdict(float), defaultdict(int)
for w, iv in wg_iv.items():
    for t0, t1, s, _, _c in iv:
        if s.startswith('light:'):
            parts = s.split(':')
            light[(parts[1], ':'.join(parts[2:]) if len(parts) > 2 else 'pro')] += (t1 - t0) * TICK_US
            if len(parts) == 2:
                lightn[parts[1]] += 1
print('   light items: type, items per slot, ms per slot (us per item) by part')
for ty in sorted(set(k[0] for k in light)):
    n = max(lightn[ty], 1)
    tot_ty = sum(v for k, v in light.items() if k[0] == ty)
    print('   %-11s %6.1f items  %5.2f ms (%6.1f us):  ' % (ty, n / nw, tot_ty / nw / 1e3, tot_ty / n) +
          '  '.join('%s %.1f' % (part, light[(ty, part)] / n) for part in ('pro', 'stage', 'K', 'late', 'epi', 'epi:halo', 'epi:mates', 'epi:compose') if (ty, part) in light))

# ---- pair the workgroups of a CU
by_cu = defaultdict(list)
for w, cu in wg_cu.items():
    by_cu[cu].append(w)
pairs = [ws for ws in by_cu.values() if len(ws) == 2]
print('%d CUs host exactly two workgroups (%d CUs seen)' % (len(pairs), len(by_cu)))


def simplify(s):
    return 'K' if s == 'K' else ('idle' if s in ('wait', 'late', 'sched', 'stage:barrier', 'yield') or s.startswith('wait:') else 'other')


joint = defaultdict(float)
unc = defaultdict(lambda: [0.0, 0.0])     # waiting state -> [time, time with no K loop on the CU's other workgroup]
starve = defaultdict(list)   # non-K phase class -> [(duration, time the partner spent in a K loop during it)]
X, Y = [], []       # per K interval: (overlap with the partner's K, rest), MFMAs
for a, b in pairs:
    for me, other in ((a, b), (b, a)):
        oiv = wg_iv[other]
        j = 0
        chunk = {}      # chunk id -> [overlap with the partner's K, rest, MFMAs]
        for t0, t1, s, mf, cid in wg_iv[me]:
            # walk the partner's intervals overlapping [t0, t1)
            while j < len(oiv) and oiv[j][1] <= t0:
                j += 1
            k = j
            ovK = 0
            cover = 0
            while k < len(oiv) and oiv[k][0] < t1:
                lo, hi = max(t0, oiv[k][0]), min(t1, oiv[k][1])
                if hi > lo:
                    if me == a:
                        joint[(simplify(s), simplify(oiv[k][2]))] += (hi - lo) * TICK_US
                    if oiv[k][2] == 'K':
                        ovK += hi - lo
                    cover += hi - lo
                k += 1
            if s == 'K' and mf > 0:
                acc = chunk.setdefault(cid, [0.0, 0.0, mf])
                acc[0] += ovK * TICK_US
                acc[1] += (t1 - t0 - ovK) * TICK_US
            elif s.startswith('wait') or s in ('late', 'yield') or s.endswith(':late') or s.endswith(':mates'):
                key = s if s.startswith('wait') else ('late (conv-LSTM)' if s == 'late' else
                                                      'yield (recurrent half)' if s == 'yield' else s.replace('light:', ''))
                unc[key][0] += (t1 - t0) * TICK_US
                unc[key][1] += (t1 - t0 - ovK) * TICK_US
            elif s in ('epi', 'pro', 'stage') or s.startswith('light:'):
                cls = s if not s.startswith('light:') else ':'.join(s.split(':')[:2])
                if s.startswith('light:') and (s.endswith(':late') or s.endswith(':mates')):
                    continue
                starve[cls].append(((t1 - t0) * TICK_US, ovK * TICK_US))
        for ov_us, rest_us, mf in chunk.values():
            X.append((ov_us, rest_us))
            Y.append(mf)
npair = len(pairs)
print('joint state of a CU, %% of the traced span (K = issuing MFMAs of a conv-LSTM K loop, idle = waiting for a ticket / '
      'dependency / late input or asleep for the CU partner, other = prologue, staging, epilogue, light items):')
keys = ['K', 'other', 'idle']
sym = defaultdict(float)
for (s1, s2), v in joint.items():
    sym[tuple(sorted((s1, s2)))] += v
for k, v in sorted(sym.items(), key=lambda kv: -kv[1]):
    print('   %-14s %5.1f %%' % ('%s + %s' % k, 100.0 * v / npair / span))
X, Y = np.array(X), np.array(Y, dtype=np.float64)
if len(Y) == 0:
    sys.exit(0)         # (the split-bf16 tile logs no K-loop events)
for mfc in sorted(set(Y.tolist())):
    sel = Y == mfc
    rr, *_ = np.linalg.lstsq(X[sel], Y[sel], rcond=None)
    print('   tiles with %5d MFMAs per wave and chunk (%6d chunks): both in K %.3f of the pipe, alone %.3f' % (
        int(mfc), int(sel.sum()), 2 * rr[0] * 64 / 2380.0, rr[1] * 64 / 2380.0))
r, *_ = np.linalg.lstsq(X, Y, rcond=None)
GHZ = 2.38
print('MFMA issue rate of one wave in its K loop (least squares over %d K-loop intervals):' % len(Y))
print('   partner also in its K loop: %.2f MFMAs/us per wave -> two waves use %.3f of the SIMD\'s matrix pipe' % (r[0], 2 * r[0] * 64 / (GHZ * 1e3)))
print('   partner NOT in a K loop:    %.2f MFMAs/us per wave -> one wave uses  %.3f of the SIMD\'s matrix pipe' % (r[1], r[1] * 64 / (GHZ * 1e3)))
print('   (fp32 32x32x2 MFMA = 64 cycles; %.2f GHz -> %.1f MFMAs/us per SIMD at most)' % (GHZ, GHZ * 1e3 / 64))
tK2 = sym[('K', 'K')] / npair
tK1 = (sym[('K', 'other')] + sym[('K', 'idle')]) / npair
print('   pipe time accounted: both-K %.1f ms x %.3f + one-K %.1f ms x %.3f = %.1f ms of MFMA work per SIMD' % (
    tK2 / 1e3, 2 * r[0] * 64 / (GHZ * 1e3), tK1 / 1e3, r[1] * 64 / (GHZ * 1e3),
    (tK2 * 2 * r[0] + tK1 * r[1]) * 64 / (GHZ * 1e3) / 1e3))

# ---- how much does a non-K phase advance while the other workgroup of the CU is in a K loop?  duration = a + b x (partner's
# K time during the phase): b = 1 - the phase stands still next to a K loop and `a` is its real cost; b = 0 - it does not care
print('non-K phases next to a K loop: duration = a + b x (partner K-loop time inside the phase), least squares')
for cls in sorted(starve):
    d = np.array(starve[cls])
    if len(d) < 50:
        continue
    A = np.stack([np.ones(len(d)), d[:, 1]], axis=1)
    (a_, b_), *_ = np.linalg.lstsq(A, d[:, 0], rcond=None)
    print('   %-18s n %7d  mean %6.1f us (partner in K %4.1f us of it)  a %6.1f us  b %.2f' % (
        cls, len(d), d[:, 0].mean(), d[:, 1].mean(), a_, b_))

# ---- waits: which ones fall next to a K loop (free: the matrix pipe is busy anyway) and which do not
print('waiting, ms per slot [of it with NO K loop next to it]:')
for k, (t_all, t_unc) in sorted(unc.items(), key=lambda kv: -kv[1][1]):
    print('   %-24s %6.2f  [%5.2f]' % (k, t_all / (2 * npair) / 1e3, t_unc / (2 * npair) / 1e3))

# ---- time-resolved: share of the workgroups in each state, in bins over the middle of the launch
if os.environ.get('VF_TRACE_TIMELINE'):
    nb = 400
    lo_t, hi_t = t_min + (t_max - t_min) * 0.40, t_min + (t_max - t_min) * 0.56       # ~ two steps
    edges = np.linspace(lo_t, hi_t, nb + 1)
    occ = defaultdict(lambda: np.zeros(nb))
    for w, iv in wg_iv.items():
        for t0, t1, s, _, _c in iv:
            if t1 <= lo_t or t0 >= hi_t:
                continue
            key = 'K' if s == 'K' else ('idle' if s in ('wait', 'late', 'sched', 'yield') or s.startswith('wait:') or s.endswith(':late') else
                                        (s.split(':')[1] if s.startswith('light') else 'lstm-other'))
            a, b = max(t0, lo_t), min(t1, hi_t)
            i0, i1 = int((a - lo_t) / (hi_t - lo_t) * nb), min(nb - 1, int((b - lo_t) / (hi_t - lo_t) * nb))
            for i in range(i0, i1 + 1):
                occ[key][i] += (min(b, edges[i + 1]) - max(a, edges[i])) / (edges[i + 1] - edges[i])
    keys = sorted(occ, key=lambda k: -occ[k].sum())
    print('timeline: %% of the %d workgroups per state, bins of %.1f us' % (nw, (hi_t - lo_t) / nb * TICK_US))
    print('  t_us ' + ' '.join('%10s' % k[:10] for k in keys))
    for i in range(nb):
        print('%6.0f ' % ((edges[i] - lo_t) * TICK_US) + ' '.join('%10.1f' % (100.0 * occ[k][i] / nw) for k in keys))
