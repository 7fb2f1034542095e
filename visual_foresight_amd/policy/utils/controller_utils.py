"""Numerical helpers shared by the CEM samplers.

Behavioural restatement of the reference's ``visual_mpc/policy/utils/controller_utils.py``
(``truncate_movement`` :6-44, ``construct_initial_sigma`` :47-84, ``reuse_cov`` :87-96,
``make_blockdiagonal`` :99-104, ``discretize`` :107-117).  All of it is tiny float64 host
math on ``[M, nactions, adim]`` arrays and stays on the host.
"""
import numpy as np

_MAX_ROT = np.pi / 4


def truncate_movement(actions, hp):
    """Clip translation (and rotation, when present) of sampled actions in place: xy to 2*initial_std, theta to
    pi/4.  Without ``action_order`` the first two dims are xy and dim 3 (if it exists) the rotation."""
    if actions.ndim not in (2, 3):
        raise NotImplementedError
    adim = actions.shape[-1]
    xy_limit = hp.initial_std * 2
    if hp.action_order is None:
        limits = [xy_limit, xy_limit, None, _MAX_ROT][:adim]
    else:
        table = {'x': xy_limit, 'y': xy_limit, 'theta': _MAX_ROT}
        limits = [table.get(a) for a in hp.action_order]
    for i, lim in enumerate(limits):
        if lim is not None:
            actions[..., i] = np.clip(actions[..., i], -lim, lim)
    return actions


def construct_initial_sigma(hp, adim, t=None):
    """Diagonal covariance of the initial proposal, one block of variances per action step."""
    if hp.action_order is not None:
        std_of = {'x': hp.initial_std, 'y': hp.initial_std, 'z': hp.initial_std_lift,
                  'theta': hp.initial_std_rot, 'grasp': hp.initial_std_grasp}
        for a in hp.action_order:
            if a not in std_of:
                raise NotImplementedError
        stds = [std_of[a] for a in hp.action_order]
    else:
        stds = [hp.initial_std, hp.initial_std]
        if adim >= 3:
            stds.append(hp.initial_std_lift)
        if adim >= 4:
            stds.append(hp.initial_std_rot)
        if adim == 5:
            stds.append(hp.initial_std_grasp)
    per_step = np.array([s ** 2 for s in stds])
    block = len(per_step)
    diag = np.tile(per_step, hp.nactions)

    if 'reduce_std_dev' in hp:
        assert 'reuse_mean' in hp
        if t >= 2:      # (t=None raises TypeError exactly as the reference does)
            print('reducing std dev by factor', hp.reduce_std_dev)
            # all but the last action step: that one cannot have been planned before
            diag[:(hp.nactions - 1) * block] *= hp.reduce_std_dev
    return np.diag(diag)


def reuse_cov(sigma, adim, hp):
    """Shift the previous plan's covariance one action step forward and re-inflate it."""
    assert hp.replan_interval == 3
    print('reusing cov form last MPC step...')
    fresh = construct_initial_sigma(hp, adim)
    shifted = np.zeros_like(sigma)
    shifted[:-adim, :-adim] = sigma[adim:, adim:] + fresh[:-adim, :-adim] * hp.reuse_cov
    shifted[-adim:, -adim:] = fresh[:adim, :adim]
    return shifted


def make_blockdiagonal(cov, nactions, adim):
    """Keep only the covariance between neighbouring action steps."""
    keep = np.zeros_like(cov)
    for i in range(nactions - 1):
        keep[i * adim:(i + 2) * adim, i * adim:(i + 2) * adim] = 1.
    return cov * keep


def discretize(actions, M, naction_steps, discrete_ind):
    """floor + clip to {0..4} on the listed action dimensions."""
    for ind in discrete_ind:
        actions[:M, :naction_steps, ind] = np.clip(np.floor(actions[:M, :naction_steps, ind]), 0, 4)
    return actions
