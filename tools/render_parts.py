"""GPU box: ms per frame of rt355_render (page-locked pixmap) with the pixels written straight into the mapped pixmap
("direct"), with a readback by copy in 1 .. 8 parts ("4", "6s", "6sd": RT355_READBACK_COPY, RT355_RENDER_PARTS) and
as the library chooses by itself ("auto"), beside rt355_render_device ("dev").  Usage: python tools/render_parts.py WORKLOAD..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import binding as B
from raytracer_amd.api import HostPixmap, Renderer

for name in sys.argv[1:]:
    scene, sh, cam, w, h = bench.build_workload(name)
    shading, dirs = bench.gather_of(name)
    r = Renderer(scene)
    if dirs is not None:
        r.set_directions(dirs)
    host = HostPixmap(w, h)
    fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    line = [name]
    for parts in os.environ.get("PARTS", "dev auto direct 1 4 6 6sd").split():
        for k in ("RT355_READBACK_COPY", "RT355_READBACK_DIRECT", "RT355_COPY_STREAMS", "RT355_LAST_COPIED", "RT355_RENDER_PARTS", "RT355_TAIL_PARTS", "RT355_TAIL_BOTTOM_FIRST"):
            os.environ.pop(k, None)
        if parts == "direct":
            os.environ["RT355_READBACK_DIRECT"] = "1"
        elif parts not in ("dev", "auto"):  # "4": four parts, one copy stream, all copied; "6s": two copy streams; "6sd": last part written directly
            os.environ["RT355_READBACK_COPY"] = "1"
            os.environ["RT355_RENDER_PARTS"] = parts.rstrip("sdb")  # a one-kernel frame: rendered in that many parts
            os.environ["RT355_TAIL_PARTS"] = parts.rstrip("sdb")    # a searched frame: its last stage in that many parts
            if "s" not in parts:
                os.environ["RT355_COPY_STREAMS"] = "1"
            if "d" not in parts:
                os.environ["RT355_LAST_COPIED"] = "1"
            if "b" in parts:  # a searched frame's parts from the bottom row up (the library's order is from the top down)
                os.environ["RT355_TAIL_BOTTOM_FIRST"] = "1"
        def frame():
            if parts == "dev":
                r.render_device(cam, sh, B.opts(w, h, shading=shading), fb.data_ptr(), 0)
                r.sync()
            else:
                r.render(cam, sh, w, h, shading=shading, out=host.array)
        for _ in range(3):
            frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            frame()
        torch.cuda.synchronize()
        line.append(f"{parts}: {(time.perf_counter() - t0) * 50:.3f}")
    print("  ".join(line), flush=True)
    r.close()
    host.close()
