#!/usr/bin/env python
"""Lists the `s_barrier`s of the gfx950 code object that sit in a loop-header block without an `s_waitcnt lgkmcnt(0)`
in front of them.

Why: hipcc 7.2 does not wait for an LDS store that is still pending on the BACK EDGE of a loop when the loop body starts
with `__syncthreads()` - the other waves pass the barrier and read the old LDS contents (round 3: a scheduler-loop
variant whose last statement stored the next ticket to LDS ran items twice, profiles/r03_tile_plan_sweep.txt).  Every
site listed here must have no LDS store between the last LDS wait of the loop body and the back edge; the listing shows
the LDS instructions after the last `lgkmcnt(0)` / barrier / call of the function's textual tail for a quick look.

    python tools/lint_barriers.py            # device-only compile of csrc/vf_engine.hip to assembly in /tmp (~50 s)

Exit code 1 when a loop-head barrier has an LDS store pending on its back edge, or when the scheduler loop of the
persistent kernel lost its explicit wait; tests/test_kernel_lint.py runs this in the CPU suite.
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_assembly(extra_flags=()):
    """gfx950 assembly of the engine (device-only compile, ~50 s)."""
    tmp = tempfile.mkdtemp(prefix='vf_lint_')
    src = os.path.join(REPO, 'visual_foresight_amd', 'csrc', 'vf_engine.hip')
    out = os.path.join(tmp, 'vf_engine_gfx950.s')
    hipcc = '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else 'hipcc'
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only',
           '--no-gpu-bundle-output', '-S', '-o', out, src] + list(extra_flags)
    subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(out) as f:
        return f.read().split('\n')


def lint(lines):
    """-> list of findings {'func', 'line', 'label', 'tail', 'pending': [LDS stores behind the last LDS wait of the loop
    tail]} for every loop-head `s_barrier` without an `s_waitcnt lgkmcnt(0)` in its block."""
    findings = []
    func, label, label_line = None, None, 0
    for k, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            func = m.group(1)
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            label, label_line = m.group(1), k
        if 's_barrier' not in l or label is None:
            continue
        block = lines[label_line:k]
        if any('lgkmcnt(0)' in x for x in block) or not any('Loop Header' in x for x in lines[label_line:label_line + 12]):
            continue
        # the loop's textual extent: up to the last branch back to a label at or before this header
        hdr_no = int(label.split('_')[1])
        end = k
        for j in range(k, len(lines)):
            if re.match(r'^_Z\w+:', lines[j]) or '.Lfunc_end' in lines[j]:
                break
            mm = re.search(r's_c?branch\w*\s+\.LBB\d+_(\d+)', lines[j])
            if mm and int(mm.group(1)) <= hdr_no:
                end = j
        pending = []
        for q in range(end, k, -1):
            if 'lgkmcnt(0)' in lines[q] or 's_barrier' in lines[q] or 's_swappc' in lines[q]:
                break
            if re.search(r'\bds_(write|store|add|min|max|or|and|xor)', lines[q]):
                pending.append(lines[q].strip())
        findings.append({'func': func, 'line': k, 'label': label, 'tail': end, 'pending': pending})
    return findings


SCHED_MARKER = 'vf_sched_loop_head'


def scheduler_barrier_is_guarded(lines):
    """The scheduler loop of every rollout_persistent_kernel instance must reach its loop-head barrier through the
    explicit `s_waitcnt lgkmcnt(0)` of vf_persistent.h (not through whatever the compiler decides to emit).  The inline
    asm carries the marker comment `vf_sched_loop_head`: every instance must contain it exactly once, on an
    `s_waitcnt lgkmcnt(0)`, inside a loop-header block, and the next instruction that is not another wait must be the
    `s_barrier` - so the check cannot land on a prologue barrier or on a block the compiler moved.  -> names of the
    instances that fail."""
    bad = []
    func, start = None, 0
    spans = []
    for k, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            if func:
                spans.append((func, start, k))
            func, start = m.group(1), k
    if func:
        spans.append((func, start, len(lines)))
    for name, a, b in spans:
        if not name.startswith('_ZN2vf25rollout_persistent_kernel'):
            continue
        marks = [k for k in range(a, b) if SCHED_MARKER in lines[k]]
        ok = len(marks) == 1 and 's_waitcnt' in lines[marks[0]] and 'lgkmcnt(0)' in lines[marks[0]]
        if ok:
            k = marks[0]
            # the block the wait sits in must be a loop header
            hdr = next((j for j in range(k, a, -1) if re.match(r'^\.LBB\d+_\d+:', lines[j])), None)
            ok = hdr is not None and any('Loop Header' in x for x in lines[hdr:hdr + 12])
            nxt = [x.strip() for x in lines[k + 1:k + 12]
                   if x.strip() and not x.strip().startswith(';') and not x.strip().startswith('.')]
            nxt = [x for x in nxt if not x.startswith('s_waitcnt') and not x.startswith('s_nop')]
            ok = ok and bool(nxt) and nxt[0].startswith('s_barrier')
        if not ok:
            bad.append(name)
    return bad


def main():
    lines = device_assembly()
    findings = lint(lines)
    for f in findings:
        print('%s\n   loop-head barrier at line %d (%s), loop tail line %d: %s' % (
            f['func'], f['line'], f['label'], f['tail'],
            'LDS STORES PENDING ON THE BACK EDGE: ' + '; '.join(f['pending'][:4]) if f['pending']
            else 'no LDS store behind the last LDS wait of the tail'))
    print('%d loop-head barriers without an LDS wait in their block' % len(findings))
    unguarded = scheduler_barrier_is_guarded(lines)
    for name in unguarded:
        print('scheduler loop of %s: no explicit lgkmcnt(0) in front of its loop-head barrier' % name)
    return 1 if unguarded or any(f['pending'] for f in findings) else 0


if __name__ == '__main__':
    sys.exit(main())
