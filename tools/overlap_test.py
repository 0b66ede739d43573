#!/usr/bin/env python
"""Throughput of N frames in flight: N scene contexts, each on its own HIP stream, frames dealt round-robin.
Usage: python tools/overlap_test.py <workload> <frames> <contexts...>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_workload
from raytracer_amd import binding as B
from raytracer_amd.api import Renderer

wl, frames = sys.argv[1], int(sys.argv[2])
scene, shader, cam, w, h = build_workload(wl)
o = B.opts(w, h)
for n_ctx in [int(a) for a in sys.argv[3:]]:
    rs = [Renderer(scene) for _ in range(n_ctx)]
    fbs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(n_ctx)]
    streams = [torch.cuda.Stream() for _ in range(n_ctx)]
    def run(k):
        for i in range(k):
            c = i % n_ctx
            rs[c].render_device(cam, shader, o, fbs[c].data_ptr(), streams[c].cuda_stream)  # syncs that context's previous frame
        for r in rs:
            st = r.sync()
        return st
    run(2 * n_ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = run(frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rays = (st["primary_rays"] + st["shadow_rays"]) * frames
    print(f"{wl} contexts {n_ctx}: {dt*1e3/frames:.3f} ms/frame  {rays/dt/1e6:.1f} Mrays/s", flush=True)
    same = all(torch.equal(fbs[0], f) for f in fbs[1:])
    print("  frames identical across contexts:", same)
    for r in rs:
        r.close()
