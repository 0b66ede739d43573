"""Debug aid: statistics of the gather shaders on a searched scene at a reduced frame size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from raytracer_amd import binding as B, scenes as S
from raytracer_amd.api import Renderer
name, w, h, level = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
scene, shader, _, _, _ = bench.build_workload(name)
cam = S.default_camera(w, h)
r = Renderer(scene)
r.set_directions(B.geodesic(B.icosphere(level)))
print("scene up", flush=True)
for it in range(2):
    t = time.time()
    _, st = r.render(cam, shader, w, h, shading=B.SHADING_BLEED)
    print(round(time.time() - t, 3), "s", st, flush=True)
    print("candidates per gather ray", st["exact_tests"] / max(1, st["gather_rays"]), "tests per ray", st["pair_tests"] / max(1, st["gather_rays"]), flush=True)
