"""GPU box: how long rt355_render_device takes to return (the host side of a frame: its launches are enqueued, not
awaited) against how long the frame takes on the device.  Usage: python tools/enqueue_time.py WORKLOAD..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import binding as B
from raytracer_amd.api import Renderer

for name in sys.argv[1:]:
    scene, sh, cam, w, h = bench.build_workload(name)
    cams = bench.sweep_of(name, w, h) or [cam]
    r = Renderer(scene)
    fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    o = B.opts(w, h)
    for c in cams[:3]:
        r.render_device(c, sh, o, fb.data_ptr(), 0)
        r.sync()
    enq = dev = 0.0
    n = 0
    for c in (cams * 20)[:40]:
        t0 = time.perf_counter()
        r.render_device(c, sh, o, fb.data_ptr(), 0)
        t1 = time.perf_counter()
        r.sync()
        t2 = time.perf_counter()
        enq += t1 - t0
        dev += t2 - t0
        n += 1
    print(f"{name}: enqueue {enq / n * 1e3:.3f} ms, frame {dev / n * 1e3:.3f} ms per frame", flush=True)
    r.close()
