#!/usr/bin/env python
"""GPU box: where the host side of a planning call goes (cProfile over bench.py's own loop).
    python tools/host_profile.py [--samples 25] [--steps 10]"""
import cProfile
import contextlib
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402

sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['--samples', '25']) + ['--no-alt', '--no-cpu-baseline']
args = bench.parse()
b = bench.Bench(args)
ctrl = b.build_controller('fp32')
extra = {}


def plan(i):
    return ctrl.act(t=i, i_tr=0, desig_pix=b.desig, goal_pix=b.goal, images=b.frames[:i + 1], state=b.states[:i + 1], **extra)


np.random.seed(0)
with contextlib.redirect_stdout(io.StringIO()), b.blas_guard():
    plan(0)
    for i in range(3):
        plan(1 + i)
    b.sync()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(args.steps):
        plan(4 + i)
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
st.sort_stats('tottime').print_stats(18)
