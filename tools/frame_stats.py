import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from raytracer_amd import binding as B
from raytracer_amd.api import Renderer
for name in sys.argv[1:]:
    scene, sh, cam, w, h = bench.build_workload(name)
    r = Renderer(scene)
    fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        r.render_device(cam, sh, B.opts(w, h), fb.data_ptr(), 0); st = r.sync()
    print(name, st, flush=True)
    r.close()
