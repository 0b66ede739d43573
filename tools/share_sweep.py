"""GPU box: one tile share of a frame under several settings of the handle's options, wall time per frame and the
search launches' times.  Usage: python tools/share_sweep.py WORKLOAD N "opt=val,opt=val" ["opt=val,..." ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import binding as B
from raytracer_amd.api import Renderer
from raytracer_amd.sharding import TILE_ROWS

name, n = sys.argv[1], int(sys.argv[2])
scene, sh, cam, w, h = bench.build_workload(name)
fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
o = B.opts(w, h) if n == 1 else B.opts(w, h, tile_rows=TILE_ROWS, tile_count=n, tile_phase=0)
for setting in sys.argv[3:] or [""]:
    r = Renderer(scene)
    for opt in filter(None, setting.split(",")):
        key, val = opt.split("=")
        r.set_option(key, int(val))
    for _ in range(5):
        r.render_device(cam, sh, o, fb.data_ptr(), 0)
        r.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        r.render_device(cam, sh, o, fb.data_ptr(), 0)
        st = r.sync()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name} share 0 of {n} [{setting}]: {ms:.3f} ms per frame, device {st['ms_device']:.3f}; search {[round(v, 3) for v in r.search_launch_ms()]}, "
          f"resolve/shade {[round(v, 3) for v in r.resolve_launch_ms()]}", flush=True)
    r.close()
