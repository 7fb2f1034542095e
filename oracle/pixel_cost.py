"""ORACLE (test infrastructure): NumPy restatement of the reference's pixel-distance cost.

Follows ``/root/reference/visual_mpc/policy/cem_controllers/pixel_cost_controller.py``:
  * ``distance_grid``      <- ``_get_distancegrid``  :189-197
  * ``expected_distance``  <- ``_expected_distance`` :168-187
  * ``eval_pixel_cost``    <- ``_eval_pixel_cost``   :135-166 (scoring part)
  * ``one_hot_distrib``    <- ``_switch_on_pix``     :206-215

PINNED: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/cost.npz``, which ``tools/make_golden.py`` produced by running the reference's
own methods (stub-imported) on seeded inputs.  This file is the checker for the HIP cost
reduction; it is never on the product path.
"""
import numpy as np


def distance_grid(goal_pix, height, width):
    """D[i, j] = euclidean distance of pixel (i, j) to goal_pix (row, col); float64."""
    grid = np.empty((height, width))
    goal = np.asarray(goal_pix, dtype=np.float64)
    for i in range(height):
        for j in range(width):
            grid[i, j] = np.sqrt((goal[0] - i) ** 2 + (goal[1] - j) ** 2)
    return grid


def expected_distance(gen_distrib, grid, finalweight, normalize=True):
    """gen_distrib [M, T, H, W] float32 -> scores [M].

    Keeps the reference's order of operations: in-place float32 normalisation, in-place
    float32 multiply by the float64 grid, two nested sums, time weights (1,..,1,finalweight).
    """
    assert gen_distrib.ndim == 4
    T = gen_distrib.shape[1]
    t_mult = np.ones([T])
    t_mult[-1] = finalweight
    p = gen_distrib.copy()
    if normalize:
        p /= np.sum(np.sum(p, axis=2), 2)[:, :, None, None]
    p *= grid[None, None]
    scores = np.sum(np.sum(p, axis=2), 2)
    scores *= t_mult[None]
    return np.sum(scores, axis=1) / np.sum(t_mult)


def eval_pixel_cost(gen_distrib, goal_pix, finalweight, only_take_first_view=False):
    """gen_distrib [M, T, ncam, H, W, ndesig], goal_pix [ncam, ndesig, 2] -> (scores, per_task)."""
    M, T, ncam, H, W, nd = gen_distrib.shape
    per_task = []
    for c in range(ncam):
        for p in range(nd):
            grid = distance_grid(goal_pix[c, p], H, W)
            per_task.append(expected_distance(gen_distrib[:, :, c, :, :, p], grid, finalweight))
    per_task = np.stack(per_task, axis=1)
    if only_take_first_view:
        per_task = per_task[:, 0][:, None]
    return np.mean(per_task, axis=1), per_task


def one_hot_distrib(desig_pix, n_context, ncam, height, width, ndesig):
    """float32 [n_context, ncam, H, W, ndesig] with 1 at every (clipped) designated pixel."""
    out = np.zeros((n_context, ncam, height, width, ndesig), dtype=np.float32)
    d = np.clip(np.asarray(desig_pix), [[0, 0]], [[height - 1, width - 1]]).astype(int)
    for c in range(ncam):
        for p in range(ndesig):
            out[:, c, d[c, p, 0], d[c, p, 1], p] = 1.
    return out
