"""One process per GPU: shard independent ME units across ranks and gather the MV /
cost records back to the rank that feeds the host encoder.

The units (lowres fields; (frame, reference) job lists; CTU-row bands) never exchange
data while they run, so the only collective is the final gather of result records
(KB..MB, latency-bound on xGMI) -- RCCL when the tensors live on the GPU, gloo on CPU.
"""
import numpy as np


def shard_range(n_units, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_units for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n_units, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_rows(n_ctu_rows, rank, world, merange, ctu=64, apron=8):
    """CTU-row band of `rank` plus the reference rows it needs: band +- (merange + apron) pixels."""
    lo, hi = shard_range(n_ctu_rows, rank, world)
    return lo, hi, lo * ctu - merange - apron, hi * ctu + merange + apron


def gather_records(local, dist=None, dst=0):
    """Gather equally-shaped record tensors to `dst`. `local` is a torch tensor (uint8 view of the
    records); returns the list of per-rank tensors on dst, None elsewhere."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    import torch
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        bufs = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, bufs, dst=dst)
        return bufs
    dist.gather(local, None, dst=dst)
    return None
