"""CPU, world size 2 over gloo: the slab partition + final-frame gather and the round-robin
frame partition reproduce the single-process frame.  Slabs are rendered by the CPU checker here
(the GPU test test_slabs_equal_whole_frame covers the HIP side of the same partition)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle.oracle_py import Oracle
    from raytracer_amd import binding as B
    from raytracer_amd import scenes as S
    from raytracer_amd.sharding import SharedPixmap, frames_for_rank, gather_frame, slab_rows, tile_rows_of

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O = Oracle()
    sc, sh, cam, w, h = S.small_case("default_2lights")
    p = sc.pack()
    b, e = slab_rows(h, world)[rank]
    local = np.zeros((h, w, 3), dtype=np.uint8)
    O.render(p, cam, sh, B.opts(w, h, row_begin=b, row_end=e), threads=2, out=local)
    full = gather_frame(torch.from_numpy(local), rank, world, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "slabs.npy"), full.numpy())
    else:
        assert full is None
    # the same slabs into one pixmap that the ranks share in host memory (pageable here: no GPU): every rank writes its
    # rows, a barrier, and the frame is whole on every rank
    shared = SharedPixmap(w, h, rank, world, dist, register=False)
    O.render(p, cam, sh, B.opts(w, h, row_begin=b, row_end=e), threads=2, out=shared.array)
    shared.barrier()  # the frame-complete signal: arrival counters in the mapping itself
    for _ in range(50):  # (and again and again: a rank may be a whole frame ahead of a slower one, never two)
        shared.barrier()
    # the barrier that carries one number per rank (what its share took): every rank reads every rank's, frame after
    # frame, and balancers fed with them stay the same on all ranks
    from raytracer_amd.sharding import SlabBalancer
    bal = SlabBalancer(h, world, align=4)
    for it in range(20):
        times = shared.barrier_with_times(100.0 * it + rank + 1.0)
        assert times == [100.0 * it + r_ + 1.0 for r_ in range(world)], times
        bal.update(times)
    cuts = torch.tensor(bal.cuts, dtype=torch.int64)
    both = [torch.zeros_like(cuts) for _ in range(world)]
    dist.all_gather(both, cuts)
    assert all(torch.equal(c, cuts) for c in both)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "shared.npy"), shared.array.copy())
    dist.barrier()
    shared.close()
    # interleaved row tiles of 16 rows: the call renders this rank's share in place, one gather assembles the frame
    local = np.zeros((h, w, 3), dtype=np.uint8)
    O.render(p, cam, sh, B.opts(w, h, tile_rows=16, tile_count=world, tile_phase=rank), threads=2, out=local)
    mine = np.zeros(h, dtype=bool)
    for tb, te in tile_rows_of(h, world, rank, 16):
        mine[tb:te] = True
    assert not local[~mine].any() and local[mine].any()
    full = gather_frame(torch.from_numpy(local), rank, world, dst=0, shard="tiles", tile_rows=16)
    if rank == 0:
        np.save(os.path.join(out_dir, "tiles.npy"), full.numpy())
    # camera sweep: frame f on rank f mod world, hashes exchanged only to check the partition
    mine = frames_for_rank(5, rank, world)
    hashes = torch.zeros(5, dtype=torch.int64)
    for f in mine:
        frame, _ = O.render(p, S.orbit_camera(f, 48, 32, n_frames=5, center=(0, 0, 0), radius=30, eye_z=8), sh,
                            B.opts(48, 32), threads=2)
        hashes[f] = O.fnv(frame) & 0x7FFFFFFFFFFFFFFF
    dist.all_reduce(hashes)
    if rank == 0:
        np.save(os.path.join(out_dir, "sweep.npy"), hashes.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_slabs_and_sweep_world2(tmp_path, oracle):
    import torch.multiprocessing as mp
    from raytracer_amd import binding as B
    from raytracer_amd import scenes as S
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sc, sh, cam, w, h = S.small_case("default_2lights")
    want, _ = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=2)
    got = np.load(tmp_path / "slabs.npy")
    assert np.array_equal(got, want)
    assert np.array_equal(np.load(tmp_path / "tiles.npy"), want)
    assert np.array_equal(np.load(tmp_path / "shared.npy"), want)
    sweep = np.load(tmp_path / "sweep.npy")
    for f in range(5):
        frame, _ = oracle.render(sc.pack(), S.orbit_camera(f, 48, 32, n_frames=5, center=(0, 0, 0), radius=30, eye_z=8),
                                 sh, B.opts(48, 32), threads=2)
        assert int(sweep[f]) == oracle.fnv(frame) & 0x7FFFFFFFFFFFFFFF


def test_slab_rows_cover_frame():
    from raytracer_amd.sharding import frames_for_rank, slab_rows
    for h in (1, 7, 600, 2160, 4320):
        for n in (1, 2, 3, 4, 8):
            if n > h:
                continue
            rows = slab_rows(h, n)
            assert rows[0][0] == 0 and rows[-1][1] == h
            assert all(rows[i][1] == rows[i + 1][0] for i in range(n - 1))
            sizes = [e - b for b, e in rows]
            assert max(sizes) - min(sizes) <= 1
    assert sorted(sum((frames_for_rank(60, r, 8) for r in range(8)), [])) == list(range(60))
    with pytest.raises(ValueError):
        slab_rows(0, 2)
    with pytest.raises(ValueError):   # an empty slab would read as "the whole frame" at the C ABI
        slab_rows(3, 8)


def test_slab_balancer():
    """Cuts that follow the shares' times: they cover the frame, leave no slab empty, stay on the alignment, and settle
    where the slabs of a frame with a costly band take the same time."""
    from raytracer_amd.sharding import SlabBalancer
    for h, n in ((4320, 8), (2160, 8), (600, 3), (64, 4), (10, 8), (8, 8), (1, 1)):
        bal = SlabBalancer(h, n)
        for it in range(6):
            slabs = bal.slabs()
            assert slabs[0][0] == 0 and slabs[-1][1] == h and len(slabs) == n
            assert all(slabs[i][1] == slabs[i + 1][0] for i in range(n - 1)) and all(e > b for b, e in slabs)
            if h >= 16 * n:
                assert all(b % 16 == 0 for b, _ in slabs)
            bal.update([1.0 + ((7 * k + it) % 5) for k in range(n)])
    with pytest.raises(ValueError):
        SlabBalancer(3, 8)
    h, n = 4320, 8

    def cost(r):  # per row: a costly band at the horizon, a cheap sky below
        return 0.02 + (3.0 if 1700 < r < 2100 else 0.3 if r <= 1700 else 0.01)
    bal = SlabBalancer(h, n)
    first = None
    for it in range(8):
        times = [0.25 + sum(cost(r) for r in range(b, e)) / 1000 for b, e in bal.slabs()]
        first = first or max(times) / (sum(times) / n)
        bal.update(times)
    times = [0.25 + sum(cost(r) for r in range(b, e)) / 1000 for b, e in bal.slabs()]
    assert first > 2.5 and max(times) / (sum(times) / n) < 1.1
    with pytest.raises(ValueError):
        bal.update([1.0] * (n - 1))


def test_tile_shares_cover_frame():
    from raytracer_amd.sharding import rows_of, tile_rows_of
    for h in (1, 31, 32, 33, 600, 2160, 4320):
        for n in (1, 2, 3, 8):
            for t in (1, 7, 32):
                seen = np.zeros(h, dtype=int)
                for r in range(n):
                    for b, e in tile_rows_of(h, n, r, t):
                        assert 0 <= b < e <= h and b % t == 0 and (b // t) % n == r
                        seen[b:e] += 1
                assert (seen == 1).all()
    assert rows_of(2160, 8, 3, "slabs") == [(810, 1080)]
    assert rows_of(2160, 8, 3, "tiles")[:2] == [(96, 128), (352, 384)]


def test_checker_renders_tile_shares(oracle):
    """The checker's side of rt355_opts.tile_*: the shares together are the frame, each leaves the other rows alone."""
    from raytracer_amd import binding as B
    from raytracer_amd import scenes as S
    sc, sh, cam, w, h = S.small_case("default_2lights")
    want, _ = oracle.render(sc.pack(), cam, sh, B.opts(w, h, alias=2, seed=3), threads=2)
    got = np.zeros_like(want)
    rays = 0
    for g in range(3):
        _, st = oracle.render(sc.pack(), cam, sh, B.opts(w, h, alias=2, seed=3, tile_rows=20, tile_count=3, tile_phase=g),
                              threads=2, out=got)
        rays += st["primary_rays"]
    assert np.array_equal(got, want) and rays == w * h * 4


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the script spawns the ranks itself (before it touches a GPU)
    and rank 0 prints the one line.  --check-launch stops after the ranks have met, so this runs without a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"launch": "ok", "n_gpus": 2, "rank_sum": 3.0}
