#!/usr/bin/env python
"""Build container only (it imports /root/reference): wall time of the reference's own NumPy cost path on a C2-sized
prediction - `_eval_pixel_cost` = `_get_distancegrid` (Python double loop, pixel_cost_controller.py:189-197) +
`_expected_distance` (three passes over [M, T, H, W], :168-187) - the figure SURVEY.md 8(d) asks to be reported beside
the device numbers.  The device does the same reduction inside the rollout (block sums in the compositing epilogue +
`scores_kernel`, 13 us per launch): it never sees a [200, 13, 64, 64] tensor in memory.

    python tools/time_reference_cost.py [M T H W]      -> one line per repetition and the median
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import make_golden as mg  # noqa: E402  (the stub-import harness of the golden fixtures)


def main():
    M, T, H, W = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (200, 13, 64, 64)
    mg.install_stubs()
    from visual_mpc.policy.cem_controllers import PixelCostController
    mg.reference_predictor(mg.make_fake_predictor_class(T, H, W))
    pol = {'nactions': T, 'repeat': 1, 'rejection_sampling': False, 'verbose': False}
    if M != 200:
        pol['num_samples'] = M
    with mg.quiet():
        ctrl = PixelCostController(dict(mg.AG, image_height=H, image_width=W), pol, 0, 1)
        ctrl.reset()
    rs = np.random.RandomState(0)
    ctrl._goal_pix, ctrl._desig_pix = np.array([[[16, 48]]]), np.array([[[32, 32]]])
    times = []
    for rep in range(7):
        distrib = rs.uniform(0.0, 1.0, (M, T, 1, H, W, 1)).astype(np.float32)   # (the reference normalises in place)
        t0 = time.perf_counter()
        with mg.quiet():
            grid = ctrl._get_distancegrid(ctrl._goal_pix[0, 0])
        t1 = time.perf_counter()
        with mg.quiet():
            ctrl._expected_distance(0, 0, distrib[:, :, 0, :, :, 0], grid)
        t2 = time.perf_counter()
        with mg.quiet():
            ctrl._eval_pixel_cost(0, distrib, None)
        t3 = time.perf_counter()
        times.append((t1 - t0, t2 - t1, t3 - t2))
        print('rep %d: distance grid %.1f ms, expected distance %.1f ms, whole _eval_pixel_cost %.1f ms' % (
            rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    med = np.median(np.array(times), axis=0) * 1e3
    print('median over %d: distance grid %.1f ms + expected distance %.1f ms; _eval_pixel_cost %.1f ms per CEM iteration '
          '(M %d, T %d, %dx%d, numpy %s, %d host cores, 1 thread of NumPy)' % (
              len(times), med[0], med[1], med[2], M, T, H, W, np.__version__, os.cpu_count()))


if __name__ == '__main__':
    main()
