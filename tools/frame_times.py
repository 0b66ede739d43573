"""GPU box: device-resident ms per frame and the kernel-class breakdown (HIP events) of bench workloads.
Usage: python tools/frame_times.py WORKLOAD..."""
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
    shading, dirs = bench.gather_of(name)
    r = Renderer(scene)
    if dirs is not None:
        r.set_directions(dirs)
    fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    o = B.opts(w, h, shading=shading)
    for _ in range(4):
        r.render_device(cam, sh, o, fb.data_ptr(), 0)
        r.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    search = primary = shade = 0.0
    for _ in range(n):
        r.render_device(cam, sh, o, fb.data_ptr(), 0)
        stats = r.sync()
        search += r.search_kernel_ms()[0]
        per = r.resolve_launch_ms()
        primary += per[0]
        shade += sum(per[1:])
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"{name:32s} frame {dt:7.3f} ms   search {search / n:6.3f}  primary/direct {primary / n:6.3f}  shade {shade / n:6.3f}", flush=True)
    if os.environ.get("FRAME_STATS"):
        print("   ", {k: v for k, v in stats.items() if k != "ms_device"}, flush=True)
    r.close()
