"""GPU box: what each rank of an N-way split of ONE frame would have to render, timed one share at a time on this
box's GPU (framebuffer left in HBM): row slabs against interleaved 32-row tiles.  The slowest share is the frame time
N GPUs cannot beat; the ratio of the slowest to the mean share is the partition's imbalance.
Usage: python tools/share_times.py WORKLOAD [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import binding as B
from raytracer_amd.api import Renderer
from raytracer_amd.sharding import TILE_ROWS, slab_rows

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene, sh, cam, w, h = bench.build_workload(name)
r = Renderer(scene)
fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")


def ms_of(o, reps=5):
    for _ in range(2):
        r.render_device(cam, sh, o, fb.data_ptr(), 0)
        r.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r.render_device(cam, sh, o, fb.data_ptr(), 0)
        st = r.sync()
    return (time.perf_counter() - t0) / reps * 1e3, st["primary_rays"] + st["shadow_rays"]


whole, rays = ms_of(B.opts(w, h))
print(f"{name}: whole frame {whole:.3f} ms, {rays} rays", flush=True)
for shard in ("slabs", "tiles"):
    per = []
    for k in range(n):
        if shard == "slabs":
            b, e = slab_rows(h, n)[k]
            o = B.opts(w, h, row_begin=b, row_end=e)
        else:
            o = B.opts(w, h, tile_rows=TILE_ROWS, tile_count=n, tile_phase=k)
        per.append(ms_of(o))
    ms = [p[0] for p in per]
    print(f"  {shard} x{n}: shares {' '.join(f'{m:.3f}' for m in ms)} ms; slowest {max(ms):.3f}, mean {sum(ms) / n:.3f}, "
          f"slowest/mean {max(ms) * n / sum(ms):.2f}; rays of the shares {min(p[1] for p in per)} .. {max(p[1] for p in per)}; "
          f"whole frame / slowest share = {whole / max(ms):.2f} of {n}", flush=True)
r.close()
