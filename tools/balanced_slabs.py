"""GPU box: row slabs of ONE frame cut so that every rank's share takes the same time.  The shares are timed one at a
time on this box's GPU (framebuffer left in HBM), the cuts are moved by the measured times
(raytracer_amd.sharding.SlabBalancer) and the round is repeated; printed per round: the shares' times, the slowest, and
whole frame / slowest share.  Usage: python tools/balanced_slabs.py WORKLOAD [N] [ROUNDS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import binding as B
from raytracer_amd.api import Renderer
from raytracer_amd.sharding import SlabBalancer

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
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
        r.sync()
    return (time.perf_counter() - t0) / reps * 1e3


whole = ms_of(B.opts(w, h))
print(f"{name}: whole frame {whole:.3f} ms", flush=True)
bal = SlabBalancer(h, n)
for it in range(rounds):
    cuts = bal.cuts
    ms = [ms_of(B.opts(w, h, row_begin=b, row_end=e)) for b, e in bal.slabs()]
    print(f"  round {it}: cuts {cuts[1:-1]}; shares {' '.join(f'{m:.3f}' for m in ms)} ms; slowest {max(ms):.3f}, "
          f"mean {sum(ms) / n:.3f}; whole frame / slowest share = {whole / max(ms):.2f} of {n}", flush=True)
    bal.update(ms)
r.close()
