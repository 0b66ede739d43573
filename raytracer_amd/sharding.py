"""Multi-GPU partition of the render path: pixels are independent given a replicated scene.

Three partitions, none with a collective inside the frame:
  * row slabs of one frame (rank r renders rows [r*H/N, (r+1)*H/N)),
  * interleaved row tiles of one frame (tile k of `tile_rows` rows goes to rank k mod N: the partition for
    frames whose cost is uneven from top to bottom -- sky above, geometry below),
    each followed by the final frame coming together in ONE place, the only exchange of the path: every rank's
    rt355_render writes its rows over its own GPU's link straight into one pixmap in host memory that the ranks share
    (SharedPixmap; a barrier tells when the frame is complete), or the rows are gathered on one GPU (gather_frame:
    over xGMI with RCCL) and read back from there;
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


class SlabBalancer:
    """Row cuts [0, c1, ..., height] of contiguous slabs, moved from frame to frame so that the slabs take the same
    time.  A slab's time is read as a part every share pays whatever its size (`fixed` of the fastest share's time:
    launches, fills, the searches' depth) plus the sum of a cost per run of `align` rows (the packets of the search
    are 16 pixels high); after a frame the costs of a slab's runs are scaled so that they add up to what the slab
    took -- so the profile keeps what earlier cuts have shown of it, and sharpens where the cuts move -- and the new
    cuts divide the summed cost evenly (the scaling is damped, to the power `gain`: a share's time also moves with what
    the GPU did just before, and a full correction swings to and fro).  No slab is left without rows.  Deterministic: every rank computes the same
    cuts from the same gathered times."""

    def __init__(self, height: int, world: int, align: int = 16, fixed: float = 0.5, gain: float = 0.6):
        if world < 1 or height < world:
            raise ValueError(f"{world} ranks for {height} rows: a slab would be empty")
        self.height, self.world, self.fixed, self.gain = height, world, fixed, gain
        self.align = align if height >= world * align else 1
        self.n_runs = (height + self.align - 1) // self.align
        self.cost = [1.0] * self.n_runs          # per run of rows, in arbitrary units until the first update
        self.cuts = [b for b, _ in slab_rows(height, world)] + [height]
        if self.align > 1:                       # (cuts on whole runs, every slab at least one)
            self.cuts = self._cuts_of()

    def slabs(self) -> List[Tuple[int, int]]:
        return [(self.cuts[k], self.cuts[k + 1]) for k in range(self.world)]

    def _cuts_of(self) -> List[int]:
        total, n, cuts = sum(self.cost), self.world, [0]
        run, done = 0, 0.0                       # runs before `run` add up to `done`
        for j in range(1, n):
            want = total * j / n
            while run < self.n_runs - 1 and done + self.cost[run] <= want:
                done += self.cost[run]
                run += 1
            # the cut goes to the nearer end of the run the target falls in
            at = run + (1 if want - done > 0.5 * self.cost[run] else 0)
            lo, hi = cuts[-1] // self.align + 1, self.n_runs - (n - j)
            cuts.append(max(lo, min(hi, at)) * self.align)
        cuts.append(self.height)
        return cuts

    def update(self, times: List[float]) -> List[int]:
        """Take the times the slabs of the current cuts took (one per rank, any unit); returns the new cuts."""
        if len(times) != self.world:
            raise ValueError("one time per slab")
        floor = self.fixed * min(times)
        for k, t in enumerate(times):
            first, last = self.cuts[k] // self.align, (self.cuts[k + 1] + self.align - 1) // self.align
            have = sum(self.cost[first:last])
            scale = (max(t - floor, 1e-9) / max(have, 1e-30)) ** self.gain
            for r in range(first, last):
                self.cost[r] *= scale
        self.cuts = self._cuts_of()
        return self.cuts


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


def host_barrier(dist):
    """A barrier over host memory (an all-reduce of one CPU number): with the control plane on gloo it involves no GPU
    collective, whatever backend moves GPU tensors."""
    import torch
    t = torch.zeros(1, dtype=torch.float64)
    dist.all_reduce(t)


class SharedPixmap:
    """One (height, width, 3) uint8 pixmap in host memory shared by every rank of the job (a file in /dev/shm that
    all of them map), page-locked by each rank for its own GPU where the runtime allows it: rank r's rt355_render writes r's rows of the frame
    straight into it, so the frame comes together in the caller's memory over N links at once and nothing passes
    through another GPU.  `dist` is torch.distributed (initialised); register=False leaves the memory pageable
    (CPU tests)."""

    def __init__(self, width: int, height: int, rank: int, world: int, dist, register: bool = True):
        import mmap
        import os
        import numpy as np
        n = width * height * 3
        flags_at = (n + 4095) // 4096 * 4096  # behind the pixels, on a page of their own: one arrival counter per rank
        size = flags_at + 4096
        name = [None]
        self.array = None
        self.registered = False
        self.rank, self.world, self._epoch = rank, world, 0
        try:
            if rank == 0:
                name[0] = f"/dev/shm/rt355_pixmap_{os.getpid()}_{id(self):x}"
                with open(name[0], "wb") as f:
                    f.truncate(size)
            if world > 1:
                import torch
                dist.broadcast_object_list(name, src=0, device=torch.device("cpu"))
            fd = os.open(name[0], os.O_RDWR)
            try:
                self._map = mmap.mmap(fd, size)
            finally:
                os.close(fd)
            if world > 1:
                host_barrier(dist)  # every rank has mapped it: the name can go
        finally:
            # whatever happened on the way (a rank that failed leaves the others in the collective until the launcher
            # ends them), the file's name does not outlive this constructor on rank 0
            if rank == 0 and name[0] and os.path.exists(name[0]):
                os.unlink(name[0])
        self.array = np.frombuffer(self._map, dtype=np.uint8, count=n).reshape(height, width, 3)
        self._arrived = np.frombuffer(self._map, dtype=np.int64, count=world, offset=flags_at)
        if world > 128:
            raise ValueError("the shared page holds the counters and times of at most 128 ranks")
        # ... and, in the page's second half, two rows of one time per rank (see barrier_with_times)
        self._times = np.frombuffer(self._map, dtype=np.float64, count=2 * world, offset=flags_at + 2048).reshape(2, world)
        # The HIP runtime of this image refuses to page-lock a mapping that a file backs (/dev/shm and memfd alike):
        # `registered` is then False, the pixmap stays pageable, and rt355_render passes a rank's rows through the
        # handle's own page-locked staging memory and a host copy, part by part as they arrive.  The registered
        # direct-write path across processes has therefore never run here; one process driving several GPUs
        # (rt355_scene_create_multi) has every GPU write into one page-locked pixmap without any of this.
        if register:
            from .api import lib
            self.registered = lib().rt355_host_register(self.array.ctypes.data, n) == 0

    def barrier(self, timeout_s: float = 120.0):
        """Every rank's rows of the current frame have been written: each rank publishes how many frames it has
        finished in its own word of the shared page and waits until no rank is behind it.  Plain loads and stores of
        the one mapping (x86 keeps their order; each word has one writer), a few microseconds after the last rank
        arrives -- the frame-complete signal of a frame shared between processes, where a collective over sockets
        would cost a good part of a share's render time."""
        import time
        self._epoch += 1
        self._arrived[self.rank] = self._epoch
        t0 = time.perf_counter()
        spins = 0
        while int(self._arrived.min()) < self._epoch:
            spins += 1
            if spins % 4096 == 0 and time.perf_counter() - t0 > timeout_s:
                raise TimeoutError(f"rank {self.rank}: the other ranks did not finish frame {self._epoch} within {timeout_s} s")

    def barrier_with_times(self, mine: float, timeout_s: float = 120.0) -> List[float]:
        """barrier(), with one number per rank exchanged on the way: what this rank's share of the frame took, for the
        balancing of the next frame's slabs (SlabBalancer).  A rank writes its number into the row of the current
        frame's parity before it arrives and reads the whole row after the barrier; the other row is the next
        frame's, which no rank can be done with before every rank has passed this barrier."""
        row = self._times[(self._epoch + 1) & 1]
        row[self.rank] = mine
        self.barrier(timeout_s)
        return [float(v) for v in row]

    def close(self):
        if self.array is not None:
            if self.registered:
                from .api import lib
                lib().rt355_host_unregister(self.array.ctypes.data)
                self.registered = False
            self.array = None
            self._arrived = None
            self._times = None
            try:
                self._map.close()
            except BufferError:  # (a view of the array is still alive somewhere: the mapping goes with it)
                pass


def gather_frame(local_frame, rank: int, world: int, dst: int = 0, shard: str = "slabs", tile_rows: int = TILE_ROWS, group=None):
    """Assemble the final frame on `dst` from every rank's rows.

    local_frame: a (H, W, 3) uint8 torch tensor whose rows rows_of(H, world, rank, shard) are valid.
    Returns the full (H, W, 3) tensor on `dst`, None elsewhere.  Every rank sends its rows packed
    together, padded to the largest share so that one gather with equal-sized buffers suffices.  `group`: the process
    group that carries the rows (an RCCL group for rows in HBM; None = the default group).
    """
    import torch
    import torch.distributed as dist

    H = local_frame.shape[0]
    if world == 1:
        return local_frame
    shares = [rows_of(H, world, r, shard, tile_rows) for r in range(world)]
    pad = max(sum(e - b for b, e in s) for s in shares)
    # gloo moves host memory: with that backend rows on the GPU are staged through the host (RCCL takes them as they are)
    via_host = local_frame.is_cuda and dist.get_backend(group) == "gloo"
    send = torch.zeros((pad,) + tuple(local_frame.shape[1:]), dtype=local_frame.dtype,
                       device="cpu" if via_host else local_frame.device)
    at = 0
    for b, e in shares[rank]:
        send[at: at + e - b].copy_(local_frame[b:e])
        at += e - b
    bufs = [torch.zeros_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty_like(local_frame)
    for r, share in enumerate(shares):
        at = 0
        for b, e in share:
            out[b:e].copy_(bufs[r][at: at + e - b])
            at += e - b
    return out
