"""Context slicing and batch chunking of the legacy predictor boundary.

Work-alikes of the reference's ``visual_mpc/video_prediction/pred_util.py`` (``get_context`` :4-13,
``rollout_predictions`` :21-48), pinned against outputs of the reference itself
(tests/golden/pred_util.npz, made by tools/make_golden.py).  They define what the boundary means:
the predictor sees only the last ``n_context`` frames/states, as one batch-1 float context, and the
candidate actions arrive in chunks of exactly ``b_size`` rows (the last one zero-padded, its
padding rows dropped again from the outputs).  ``HipVPredEvaluation.predictor_func()`` is a
callable these functions can drive.
"""
import numpy as np


def get_context(n_context, t, state, images, hp=None):
    """-> (frames float32 ``[1, n_context, ncam, H, W, 3]`` in [0, 1], states ``[1, n_context, sdim(+k)]``).

    ``hp.state_append`` (a list of k constants), when set, is appended to every context state.
    """
    window = slice(t + 1 - n_context, t + 1)
    frames = (images[window].astype(np.float32, copy=False) / 255.)[None]
    states = state[window][None]
    extra = getattr(hp, 'state_append', None) if hp else None
    if extra:
        tail = np.broadcast_to(np.asarray(extra), (1, n_context, len(extra)))
        states = np.concatenate((states, tail), axis=-1)
    return frames, states


def _head(arr, n):
    return None if arr is None else arr[:n]


def rollout_predictions(predictor, b_size, actions, context_frames, context_states=None, input_distribs=None,
                        logger=None):
    """Feed ``actions [M, T, adim]`` to ``predictor`` in chunks of ``b_size`` rows.

    Returns three lists (images, distributions, states) with one entry per chunk, each cut back to
    the number of real rows of that chunk; entries are None where the predictor returned None.
    """
    n = actions.shape[0]
    n_chunks = max(1, (n + b_size - 1) // b_size)
    images, distribs, states = [], [], []
    for k in range(n_chunks):
        rows = actions[k * b_size:(k + 1) * b_size]
        real = rows.shape[0]
        if k == n_chunks - 1:               # the final chunk is always rebuilt at full size
            batch = np.zeros((b_size,) + rows.shape[1:])
            batch[:real] = rows
        else:
            batch = rows
        if logger:
            logger.log("Vpred run: {} with {} actions".format(k, real))
        out = predictor(input_images=context_frames, input_state=context_states, input_actions=batch,
                        input_one_hot_images=input_distribs)
        for sink, value in zip((images, distribs, states), out):
            sink.append(_head(value, real))
    return images, distribs, states
