#!/usr/bin/env python
"""Print per-launch search-kernel times and statistics of one frame for a workload and mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_workload
from raytracer_amd import binding as B
from raytracer_amd.api import Renderer

wl = sys.argv[1]
mode = {"allpairs": B.MODE_ALLPAIRS, "fast": B.MODE_ALLPAIRS, "packet": B.MODE_PACKET}[sys.argv[2]]
scene, shader, cam, w, h = build_workload(wl)
r = Renderer(scene)
fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
o = B.opts(w, h, mode=mode)
for _ in range(3):
    r.render_device(cam, shader, o, fb.data_ptr(), 0)
    st = r.sync()
print(wl, sys.argv[2], "frame ms", round(st["ms_device"], 3), "launch ms", [round(x, 3) for x in r.search_launch_ms()])
print({k: v for k, v in st.items()})
