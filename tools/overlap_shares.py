"""GPU box: does one GPU render a frame faster as several tile shares in flight at once (several scenes on the same
device, each on its own thread: rt355_scene_create_multi with the device named more than once) than as one frame?
Usage: python tools/overlap_shares.py WORKLOAD [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd.api import HostPixmap, Renderer

name = sys.argv[1]
scene, sh, cam, w, h = bench.build_workload(name)
pix = HostPixmap(w, h)
for n in [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]:
    r = Renderer(scene, devices=[0] * n) if n > 1 else Renderer(scene)
    for _ in range(8):
        r.render(cam, sh, w, h, out=pix.array)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        _, st = r.render(cam, sh, w, h, out=pix.array)
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name}: {n} share(s) in flight on one GPU: {ms:.3f} ms per frame (rt355_render into a page-locked pixmap), device time of the longest share {st['ms_device']:.3f}", flush=True)
    r.close()
