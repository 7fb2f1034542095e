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
