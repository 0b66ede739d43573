"""Debug aid: where do the gather frames of the HIP library and the CPU checker differ?"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import binding as B, scenes as S, api
from oracle.oracle_py import Oracle
O = Oracle()
meta = json.load(open(os.path.join(ROOT, "tests/golden/meta.json")))
import math
rng = np.random.default_rng(357)
xs = rng.random(20000) * 2 * 3.1415927
dev = api.probe_sincos(xs)
host = np.array([api.sincos(float(x)) for x in xs])
print("sincos device != host:", int((dev != host).any(axis=1).sum()))
for c in meta["gather_cases"]:
    sc, sh, cam, w, h = S.small_case(c["case"])
    w, h = c["width"], c["height"]
    cam = S.default_camera(w, h)
    r = api.Renderer(sc)
    dirs = B.geodesic(B.icosphere(c["level"]))
    r.set_directions(dirs)
    got, st = r.render(cam, sh, w, h, alias=c["alias"], seed=c["seed"], shading=c["shading"], mode=B.MODE_BRUTE)
    want, wst = O.render(sc.pack(), cam, sh, B.opts(w, h, alias=c["alias"], seed=c["seed"], shading=c["shading"]), dirs=dirs)
    bad = np.argwhere((got != want).any(axis=2))
    print(c["file"], "differ:", len(bad), "gather rays", st["gather_rays"], wst["gather_rays"])
    for y, x in bad[:10]:
        t1 = B.rotation(c["seed"], int(x), int(y), 0, 0) * 2 * 3.1415927
        t2 = B.rotation(c["seed"], int(x), int(y), 0, 1) * 2 * 3.1415927
        print("  ", x, y, got[y, x], want[y, x], "theta", t1, t2, api.sincos(t1), O.sincos(t1), api.sincos(t2), O.sincos(t2))
    r.close()
