"""GPU box: ms of each of the first N calls of rt355_render (page-locked pixmap) in a fresh process.
Usage: [RT355_READBACK_COPY=1 ...] python tools/frame_series.py WORKLOAD [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd.api import HostPixmap, Renderer

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
scene, sh, cam, w, h = bench.build_workload(name)
shading, dirs = bench.gather_of(name)
r = Renderer(scene)
if dirs is not None:
    r.set_directions(dirs)
host = HostPixmap(w, h)
if os.environ.get("PRIME"):  # device-to-host copies of another buffer first: is the slow start the copy engine's or this pixmap's?
    src = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda")
    dst = torch.zeros(w * h * 3, dtype=torch.uint8).pin_memory()
    for _ in range(int(os.environ["PRIME"])):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
ms = []
for _ in range(n):
    t0 = time.perf_counter()
    r.render(cam, sh, w, h, shading=shading, out=host.array)
    ms.append((time.perf_counter() - t0) * 1e3)
print(name, " ".join(f"{m:.2f}" for m in ms), flush=True)
r.close()
host.close()
