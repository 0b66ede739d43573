"""Multi-GPU partition of the render path: pixels are independent given a replicated scene.

Three partitions, none with a collective inside the frame:
  * row slabs of one frame (rank r renders rows [r*H/N, (r+1)*H/N)),
  * interleaved row tiles of one frame (tile k of `tile_rows` rows goes to rank k mod N: the partition for
    frames whose cost is uneven from top to bottom -- sky above, geometry below),
    each followed by ONE gather of the final frame on the destination rank, the only exchange of the path;
  * whole frames of a camera sweep, frame f on rank f mod N (no exchange at all).
Works on any torch.distributed backend (nccl = RCCL on the GPU box, gloo in CPU tests).
"""
from __future__ import annotations

from typing import List, Tuple

TILE_ROWS = 32


def slab_rows(height: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous row slabs [begin, end) covering 0..height, sizes differing by at most one row.  Every rank
    must get at least one row (the C ABI reads the empty slab 0, 0 as "the whole frame")."""
    if world < 1 or height < 1:
        raise ValueError("world and height must be positive")
    if world > height:
        raise ValueError(f"{world} ranks for {height} rows: a slab would be empty")
    edges = [(height * i) // world for i in range(world + 1)]
    return [(edges[i], edges[i + 1]) for i in range(world)]


def tile_rows_of(height: int, world: int, rank: int, tile_rows: int = TILE_ROWS) -> List[Tuple[int, int]]:
    """The row ranges [begin, end) of rank's share of the interleaved row tiles: what rt355_opts
    (tile_rows, tile_count = world, tile_phase = rank) makes a call render.  May be empty."""
    if world < 1 or height < 1 or tile_rows < 1 or not 0 <= rank < world:
        raise ValueError("bad tile partition")
    return [(b, min(height, b + tile_rows)) for b in range(rank * tile_rows, height, world * tile_rows)]


def rows_of(height: int, world: int, rank: int, shard: str, tile_rows: int = TILE_ROWS) -> List[Tuple[int, int]]:
    """Row ranges of `rank` under the partition `shard` ("slabs" or "tiles")."""
    if shard == "tiles":
        return tile_rows_of(height, world, rank, tile_rows)
    return [slab_rows(height, world)[rank]]


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Round-robin frame assignment of a camera sweep: frame f goes to rank f mod world."""
    return list(range(rank, n_frames, world))


def gather_frame(local_frame, rank: int, world: int, dst: int = 0, shard: str = "slabs", tile_rows: int = TILE_ROWS):
    """Assemble the final frame on `dst` from every rank's rows.

    local_frame: a (H, W, 3) uint8 torch tensor whose rows rows_of(H, world, rank, shard) are valid.
    Returns the full (H, W, 3) tensor on `dst`, None elsewhere.  Every rank sends its rows packed
    together, padded to the largest share so that one gather with equal-sized buffers suffices.
    """
    import torch
    import torch.distributed as dist

    H = local_frame.shape[0]
    if world == 1:
        return local_frame
    shares = [rows_of(H, world, r, shard, tile_rows) for r in range(world)]
    pad = max(sum(e - b for b, e in s) for s in shares)
    # gloo moves host memory: with that backend rows on the GPU are staged through the host (RCCL takes them as they are)
    via_host = local_frame.is_cuda and dist.get_backend() == "gloo"
    send = torch.zeros((pad,) + tuple(local_frame.shape[1:]), dtype=local_frame.dtype,
                       device="cpu" if via_host else local_frame.device)
    at = 0
    for b, e in shares[rank]:
        send[at: at + e - b].copy_(local_frame[b:e])
        at += e - b
    bufs = [torch.zeros_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.empty_like(local_frame)
    for r, share in enumerate(shares):
        at = 0
        for b, e in share:
            out[b:e].copy_(bufs[r][at: at + e - b])
            at += e - b
    return out
