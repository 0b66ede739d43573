"""GPU box: one share of an N-way tile split of a frame rendered over and over, framebuffer in HBM -- to be run under
`rocprofv3 --kernel-trace --stats`, whose per-kernel averages then say where a share's time goes (against the whole
frame's trace: what does not shrink with the share is the split's fixed cost).
Usage: python tools/share_trace.py WORKLOAD [N [SHARE [FRAMES]]]   (N = 1: the whole frame)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import binding as B
from raytracer_amd.api import Renderer
from raytracer_amd.sharding import TILE_ROWS

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 10
scene, sh, cam, w, h = bench.build_workload(name)
r = Renderer(scene)
for opt in sys.argv[5:]:  # name=value options of the handle
    key, val = opt.split("=")
    r.set_option(key, int(val))
fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
o = B.opts(w, h) if n == 1 else B.opts(w, h, tile_rows=TILE_ROWS, tile_count=n, tile_phase=k)
for _ in range(4):
    r.render_device(cam, sh, o, fb.data_ptr(), 0)
    r.sync()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    r.render_device(cam, sh, o, fb.data_ptr(), 0)
    st = r.sync()
ms = (time.perf_counter() - t0) / frames * 1e3
print(f"{name} share {k} of {n}: {ms:.3f} ms per frame, search launches {[round(v, 3) for v in r.search_launch_ms()]}, "
      f"resolve/shade launches {[round(v, 3) for v in r.resolve_launch_ms()]}, {st['primary_rays']} + {st['shadow_rays']} rays", flush=True)
r.close()
