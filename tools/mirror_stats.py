"""Debug aid: search statistics of the mirror-ray workload (tests per mirror ray, candidates per ray)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from raytracer_amd import binding as B
from raytracer_amd.api import Renderer
name = sys.argv[1] if len(sys.argv) > 1 else "spheres10k_mirrors_8lights_4k"
scene, shader, cam, w, h = bench.build_workload(name)
r = Renderer(scene)
fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
for mode in (B.MODE_PACKET,):
    o = B.opts(w, h, mode=mode)
    for _ in range(2):
        r.render_device(cam, shader, o, fb.data_ptr(), torch.cuda.current_stream().cuda_stream)
        st = r.sync()
    print({k: st[k] for k in st})
    print("sphere tests per mirror ray:", st["pair_tests"] / max(1, st["secondary_rays"]))
    print("search launches ms:", [round(x, 3) for x in r.search_launch_ms()])
    print("resolve launches ms:", [round(x, 3) for x in r.resolve_launch_ms()])
