"""ORACLE (test infrastructure): loop restatement of the reference's registration arithmetic.

Follows ``/root/reference/visual_mpc/policy/cem_controllers/register_gtruth_controller.py``
``get_warp_err`` :113-173 and the trade-off normalisation of ``register_gtruth`` :88-91, written
with explicit loops.  PINNED: ``tools/make_golden.py::golden_registration`` imports that file (with
three more stub modules for the imports of ``:4,5,7``), builds the controller with
``object.__new__`` and runs the REAL ``get_warp_err`` and ``register_gtruth`` on seeded images and a
seeded fake warper; ``tests/test_oracle_golden.py`` requires this restatement to reproduce
``tests/golden/registration.npz`` (tracked pixels bit-exact; warp errors / trade-off to 1e-6,
because the reference takes ``np.mean`` of the float32 squares IN float32 while the loops here add
them in float64 - the product's NumPy version, same expression as the reference, is bit-exact).  What stays unpinned is the
registration NETWORK (absent from the snapshot): ``bilinear_warp_loops`` below is this repo's
definition of its output convention.

One documented deviation from the text of the reference: in point mode
(``register_region=False``) the reference leaves ``region_tradeoff = True`` (``:118``), never
fills the warp errors (``:163-170`` is dead; the fixture records the zeros) and every trade-off
becomes NaN; ``point_errors='l2'`` (what the product uses) takes the pixel-L2 error of
``:165-170``, which is evidently what was meant, ``point_errors='reference'`` leaves the zeros.
In region mode with only ``'start'`` registered the reference raises TypeError (the goal half of
the branch is unconditional, ``:152-160``); the oracle and the product handle that case.
"""
import numpy as np


def warp_err_loops(icam, pix_t0, goal_pix, start_image, goal_image, start_warp_pts, goal_warp_pts,
                   warped_start, warped_goal, register_gtruth, register_region, pred_height=None,
                   point_errors='l2'):
    """``pix_t0`` / ``goal_pix`` [ntask, 2] (row, col) at the resolution of the images handed in (the
    reference picks its ``*_med`` arrays when that differs from the predictor's, ``:121-128``);
    ``pred_height``: the predictor's image height, tracked pixels are rescaled to it (``:172``)."""
    H, W = start_image.shape[1:3]
    nreg, ntask = len(register_gtruth), len(pix_t0)
    errs = np.zeros((ntask, nreg))
    desig = np.zeros((ntask, nreg, 2))
    width = 5 if H >= 96 else 2
    for p in range(ntask):
        r = 0
        for name in ('start', 'goal'):
            if name not in register_gtruth:
                continue
            if name == 'start':
                pix, ref, pts, warped, hi_r, hi_c = pix_t0[p], start_image, start_warp_pts, warped_start, H - 1, W - 1
            else:
                pix, ref, pts, warped, hi_r, hi_c = goal_pix[p], goal_image, goal_warp_pts, warped_goal, H, W
            if register_region:
                r0, r1 = min(max(pix[0] - width, 0), hi_r), min(max(pix[0] + width + 1, 0), hi_r)
                c0, c1 = min(max(pix[1] - width, 0), hi_c), min(max(pix[1] + width + 1, 0), hi_c)
                acc, xs, ys = [], [], []
                for y in range(r0, r1):
                    for x in range(c0, c1):
                        acc.extend(((ref[icam][y, x] - warped[icam][y, x]) ** 2).tolist())
                        xs.append(pts[icam][y, x, 0]); ys.append(pts[icam][y, x, 1])
                errs[p, r] = sum(acc) / len(acc)
                desig[p, r] = (np.median(ys), np.median(xs))
            else:
                d = ref[icam][pix[0], pix[1]] - warped[icam][pix[0], pix[1]]
                errs[p, r] = np.sqrt(np.sum(d * d)) if point_errors == 'l2' else 0.0
                desig[p, r] = (pts[icam][pix[0], pix[1], 1], pts[icam][pix[0], pix[1], 0])
            r += 1
    if pred_height is not None:
        desig = desig * pred_height / H
    return errs, desig


def tradeoff_loops(warperrs):
    ncam, ntask, nreg = warperrs.shape
    out = np.zeros_like(warperrs)
    for p in range(ntask):
        total = sum(1.0 / warperrs[c, p, r] for c in range(ncam) for r in range(nreg))
        for c in range(ncam):
            for r in range(nreg):
                out[c, p, r] = (1.0 / warperrs[c, p, r]) / total
    return out


def bilinear_warp_loops(current, flow):
    """warped[c, r, col] = bilinear sample of current[c] at pts[c, r, col] = (col + dx, row + dy), border-clamped.

    The reference obtains ``warped`` and ``warp_pts`` from its registration network
    (``register_gtruth_controller.py:64-66``, module absent from the snapshot); the network's
    OUTPUT convention assumed here - a flow field in pixels, sampled bilinearly with clamped
    coordinates - is this repo's definition (parity unpinned).  float32 arithmetic in the same
    operation order as ``csrc/vf_small_kernels.h`` (``bilinear_clamped``), so the warp points and
    hence the tracked pixels agree bit for bit.
    """
    cur = np.asarray(current, dtype=np.float32)
    fl = np.asarray(flow, dtype=np.float32)
    ncam, H, W = cur.shape[:3]
    warped = np.zeros_like(cur)
    pts = np.zeros((ncam, H, W, 2), np.float32)
    f32 = np.float32
    for c in range(ncam):
        for r in range(H):
            for col in range(W):
                x = f32(col) + fl[c, r, col, 0]
                y = f32(r) + fl[c, r, col, 1]
                pts[c, r, col] = (x, y)
                xc = min(max(x, f32(0)), f32(W - 1))
                yc = min(max(y, f32(0)), f32(H - 1))
                x0, y0 = int(np.floor(xc)), int(np.floor(yc))
                x1, y1 = min(x0 + 1, W - 1), min(y0 + 1, H - 1)
                fx, fy = f32(xc - f32(x0)), f32(yc - f32(y0))
                a, b = cur[c, y0, x0].astype(np.float64), cur[c, y0, x1].astype(np.float64)
                cc, d = cur[c, y1, x0].astype(np.float64), cur[c, y1, x1].astype(np.float64)
                # fmaf(fx, b - a, a): the difference is rounded to float32, the fma is exact then rounded
                top = (np.float64(fx) * np.float32(b - a).astype(np.float64) + a).astype(np.float32).astype(np.float64)
                bot = (np.float64(fx) * np.float32(d - cc).astype(np.float64) + cc).astype(np.float32).astype(np.float64)
                warped[c, r, col] = (np.float64(fy) * (bot - top).astype(np.float32).astype(np.float64) + top).astype(np.float32)
    return warped, pts
