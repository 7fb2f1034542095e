#!/usr/bin/env python
"""Lists the `s_barrier`s of the gfx950 code object that sit in a loop-header block without an `s_waitcnt lgkmcnt(0)`
in front of them.

Why: hipcc 7.2 does not wait for an LDS store that is still pending on the BACK EDGE of a loop when the loop body starts
with `__syncthreads()` - the other waves pass the barrier and read the old LDS contents (round 3: a scheduler-loop
variant whose last statement stored the next ticket to LDS ran items twice, profiles/r03_tile_plan_sweep.txt).  Every
site listed here must have no LDS store between the last LDS wait of the loop body and the back edge; the listing shows
the LDS instructions after the last `lgkmcnt(0)` / barrier / call of the function's textual tail for a quick look.

    python tools/lint_barriers.py            # compiles csrc/vf_engine.hip with -save-temps into /tmp (~70 s)
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tmp = tempfile.mkdtemp(prefix='vf_lint_')
    src = os.path.join(REPO, 'visual_foresight_amd', 'csrc', 'vf_engine.hip')
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC',
           '-save-temps=obj', '-o', os.path.join(tmp, 'lint.so'), src]
    subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')]
    lines = open(os.path.join(tmp, asm[0])).read().split('\n')
    func, label, label_line, n = None, None, 0, 0
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
        n += 1
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
            if re.search(r'\bds_(write|add|min|max|or|and|xor)', lines[q]):
                pending.append(lines[q].strip())
        print('%s\n   loop-head barrier at line %d (%s), loop tail line %d: %s' % (
            func, k, label, end, 'LDS STORES PENDING ON THE BACK EDGE: ' + '; '.join(pending[:4]) if pending
            else 'no LDS store behind the last LDS wait of the tail'))
    print('%d loop-head barriers without an LDS wait in their block' % n)
    return 0


if __name__ == '__main__':
    sys.exit(main())
