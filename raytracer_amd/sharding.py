"""Multi-GPU partition of the render path: pixels are independent given a replicated scene.

Two partitions, both without any collective inside the frame:
  * row slabs of one frame (rank r renders rows [r*H/N, (r+1)*H/N)), then ONE gather of the
    final frame on the destination rank -- the only exchange of the whole path;
  * whole frames of a camera sweep, frame f on rank f mod N (no exchange at all).
Works on any torch.distributed backend (nccl = RCCL on the GPU box, gloo in CPU tests).
"""
from __future__ import annotations

from typing import List, Tuple


def slab_rows(height: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous row slabs [begin, end) covering 0..height, sizes differing by at most one row."""
    if world < 1 or height < 1:
        raise ValueError("world and height must be positive")
    edges = [(height * i) // world for i in range(world + 1)]
    return [(edges[i], edges[i + 1]) for i in range(world)]


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Round-robin frame assignment of a camera sweep: frame f goes to rank f mod world."""
    return list(range(rank, n_frames, world))


def gather_frame(local_frame, rank: int, world: int, dst: int = 0):
    """Assemble the final frame on `dst` from every rank's row slab.

    local_frame: a (H, W, 3) uint8 torch tensor whose rows slab_rows(H, world)[rank] are valid.
    Returns the full (H, W, 3) tensor on `dst`, None elsewhere.  Slabs are padded to the largest
    slab height so that one gather with equal-sized buffers suffices.
    """
    import torch
    import torch.distributed as dist

    H = local_frame.shape[0]
    rows = slab_rows(H, world)
    if world == 1:
        return local_frame
    pad = max(e - b for b, e in rows)
    b, e = rows[rank]
    # gloo moves host memory: with that backend a slab on the GPU is staged through the host (RCCL takes it as it is)
    via_host = local_frame.is_cuda and dist.get_backend() == "gloo"
    send = torch.zeros((pad,) + tuple(local_frame.shape[1:]), dtype=local_frame.dtype,
                       device="cpu" if via_host else local_frame.device)
    send[: e - b].copy_(local_frame[b:e])
    bufs = [torch.zeros_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.empty_like(local_frame)
    for r, (rb, re) in enumerate(rows):
        out[rb:re].copy_(bufs[r][: re - rb])
    return out
