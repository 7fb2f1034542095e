"""Designated-pixel registration arithmetic (host NumPy).

Restates the cost-side math of the reference's
``visual_mpc/policy/cem_controllers/register_gtruth_controller.py`` (``register_gtruth`` :54-111,
``get_warp_err`` :113-173): given a warp field that maps the current frame onto a reference
frame (the first frame of the episode and/or the goal image), re-localise every designated pixel
in the current frame and derive per-(camera, registration) trade-off weights from the warp error.

The registration *network* (``visual_mpc.registration_network``, ``:7,32``) is not part of the
reference snapshot, so the warper is a plug-in:

    warper(current [ncam,H,W,3] float32, reference [ncam,H,W,3] float32)
        -> warped [ncam,H,W,3], flow (ignored), warp_pts [ncam,H,W,2]

``warp_pts[c, r, col]`` holds the (x, y) = (col, row) position in the *current* frame that
corresponds to reference pixel (r, col) - the reference flips it to (row, col) (``:132-135``).
"""
import numpy as np


def region_bounds(center, width, limit_rows, limit_cols, inclusive_limit):
    """[lo, hi) row and column ranges of the (2*width+1)^2 window around ``center`` (row, col).

    The reference clips the start window to ``size - 1`` (``:142-143``) and the goal window to
    ``size`` (``:152-153``); ``inclusive_limit`` selects which.
    """
    hi_r = limit_rows - 1 if inclusive_limit else limit_rows
    hi_c = limit_cols - 1 if inclusive_limit else limit_cols
    r = np.clip(np.array((center[0] - width, center[0] + width + 1)), 0, hi_r)
    c = np.clip(np.array((center[1] - width, center[1] + width + 1)), 0, hi_c)
    return r, c


def get_warp_err(icam, pix_t0, goal_pix, start_image, goal_image, start_warp_pts, goal_warp_pts,
                 warped_image_start, warped_image_goal, register_gtruth=('start', 'goal'),
                 register_region=False, pred_height=None):
    """Tracked pixel and warp error of every task for one camera.

    pix_t0, goal_pix: [ntask, 2] (row, col) of the designated pixel in the first frame / of the
    goal pixel in the goal image.  Returns ``warperrs [ntask, nreg]`` and ``desig [ntask, nreg, 2]``
    (row, col) in the current frame.  ``register_region`` takes the median flow and the mean
    squared photometric error over a window (half-width 2 below 96 rows, else 5, ``:139-141``);
    otherwise the flow and the L2 photometric error at the single pixel (``:129-135,163-170`` - dead code
    in the reference, whose point mode leaves the errors zero; see ``oracle/registration.py``).
    ``pred_height``: the predictor's image height when it differs from the images' (``:172``).
    Pinned to the reference's own outputs by ``tests/golden/registration.npz``.
    """
    H, W = start_image.shape[1:3]
    nreg = len(register_gtruth)
    ntask = len(pix_t0)
    warperrs = np.zeros((ntask, nreg))
    desig = np.zeros((ntask, nreg, 2))
    refs = []
    if 'start' in register_gtruth:
        refs.append((pix_t0, start_image, start_warp_pts, warped_image_start, True))
    if 'goal' in register_gtruth:
        refs.append((goal_pix, goal_image, goal_warp_pts, warped_image_goal, False))
    width = 5 if H >= 96 else 2
    for p in range(ntask):
        for r, (pix, ref_img, warp_pts, warped, is_start) in enumerate(refs):
            pr, pc = int(pix[p][0]), int(pix[p][1])
            if register_region:
                rr, cc = region_bounds((pr, pc), width, H, W, inclusive_limit=is_start)
                win = (slice(rr[0], rr[1]), slice(cc[0], cc[1]))
                warperrs[p, r] = np.mean(np.square(ref_img[icam][win] - warped[icam][win]))
                field = warp_pts[icam][win]
                desig[p, r] = (np.median(field[:, :, 1]), np.median(field[:, :, 0]))      # (x,y) -> (row,col)
            else:
                warperrs[p, r] = np.linalg.norm(ref_img[icam][pr, pc] - warped[icam][pr, pc])
                desig[p, r] = np.flip(warp_pts[icam][pr, pc], 0)
    if pred_height is not None:
        desig = desig * pred_height / H
    return warperrs, desig


def tradeoff_weights(warperrs):
    """``warperrs [ncam, ntask, nreg]`` -> weights of the same shape, each task's summing to 1.

    weight = (1/err) / sum over (camera, registration) of (1/err)   (reference ``:88-91``).
    """
    inv = 1.0 / warperrs
    return inv / np.sum(np.sum(inv, 0, keepdims=True), 2, keepdims=True)
