"""Sample sharding of the CEM candidates over the ranks of one node.

Rank r evaluates the contiguous sample range ``shard_bounds(M, r, G)`` (same slicing as the
reference's per-GPU towers, ``visual_mpc/video_prediction/setup_predictor.py:34-39``, but by
*rank* rather than absolute device id - the reference's ``gpu_id * nsmp_per_gpu`` over-runs
the batch when ``first_gpu != 0``).  The only exchange is one all-gather of the per-sample
score rows per CEM iteration; the reference instead concatenates whole predicted videos
(``setup_predictor.py:155-162``).  Backend: RCCL ("nccl") on GPUs, gloo in the CPU tests.
"""
import torch


def dist_info():
    """(rank, world) of the default process group; (0, 1) when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(M, rank, world):
    """[lo, hi) of rank's samples; the ranges partition [0, M) in rank order, sizes differ by <= 1."""
    base, extra = divmod(M, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local, M):
    """Gather every rank's ``[hi-lo, C]`` rows into the full ``[M, C]`` matrix on every rank."""
    import torch.distributed as dist
    rank, world = dist_info()
    if world == 1:
        return local
    sizes = [b - a for a, b in (shard_bounds(M, r, world) for r in range(world))]
    assert local.shape[0] == sizes[rank], 'local rows do not match this rank\'s shard'
    width = max(sizes)
    if min(sizes) == width and dist.get_backend() == 'nccl':
        out = torch.empty((M,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    # gloo gathers host tensors only: stage device rows through the host (tests / single-GPU dry runs)
    via_host = dist.get_backend() == 'gloo' and local.is_cuda
    src = local.cpu() if via_host else local
    padded = torch.zeros((width,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    padded[:src.shape[0]] = src
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    out = torch.cat([bufs[r][:sizes[r]] for r in range(world)], dim=0)
    return out.to(local.device) if via_host else out


def plan_digest(plan_stat, best_indices, action=None):
    """sha256 (hex) of what one planning call decided: every iteration's score vector (float64 bytes, in iteration
    order), the elite indices of the last iteration and, optionally, the action handed to the agent.  Two ranks - or an
    N-rank job and the single-GPU job on the same candidates - planned identically iff their digests are equal: the
    reference's towers can only be compared by eye (``setup_predictor.py:155-162`` concatenates whole videos), here a
    scaling record proves G-invariance by itself."""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for key in sorted((k for k in plan_stat if k.startswith('scores_itr')), key=lambda k: int(k[len('scores_itr'):])):
        h.update(key.encode())
        h.update(np.ascontiguousarray(plan_stat[key], dtype=np.float64).tobytes())
    h.update(b'elites')
    h.update(np.ascontiguousarray(best_indices if best_indices is not None else [], dtype=np.int64).tobytes())
    if action is not None:
        h.update(b'action')
        h.update(np.ascontiguousarray(action, dtype=np.float64).tobytes())
    return h.hexdigest()


def run_digest(call_digests):
    """One sha256 over the per-call digests of a run (what a bench line prints as ``scores_sha``)."""
    import hashlib
    h = hashlib.sha256()
    for d in call_digests:
        h.update(d.encode())
    return h.hexdigest()[:16]


def gather_plan_digests(call_digests):
    """Every rank's list of per-call digests -> (identical_across_ranks, [run digest of every rank]).  One object
    all-gather AFTER the timed region (tens of bytes per call); world 1: trivially identical."""
    import torch.distributed as dist
    rank, world = dist_info()
    if world == 1:
        return True, [run_digest(call_digests)]
    everyone = [None] * world
    dist.all_gather_object(everyone, list(call_digests))
    same = all(d == everyone[0] for d in everyone)
    return same, [run_digest(d) for d in everyone]
