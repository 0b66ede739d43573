#!/usr/bin/env python
"""Debug: per-workgroup durations of the camera-ray packet search (needs a -DRT355_DEBUG_TIMES build)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_workload
from raytracer_amd import binding as B
from raytracer_amd import api
from raytracer_amd.api import Renderer

wl = sys.argv[1]
scene, shader, cam, w, h = build_workload(wl)
shader = B.Shader([], 0, 0xDDDDDD, 0)  # camera rays only
r = Renderer(scene)
fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
o = B.opts(w, h, mode=B.MODE_PACKET)
nblk = 20000
W = int(os.environ.get('PACKET_WAVES', '8'))
buf = torch.zeros(nblk * 4 * W, dtype=torch.int64, device="cuda")
L = api.lib()
r.render_device(cam, shader, o, fb.data_ptr(), 0); r.sync()
L.rt355_debug_set_times_buffer(C.c_void_p(buf.data_ptr()))
r.render_device(cam, shader, o, fb.data_ptr(), 0); st = r.sync()
L.rt355_debug_set_times_buffer(C.c_void_p(0))
a = buf.cpu().numpy().reshape(nblk, W, 4)
used = a[:, 0, 1] > 0
a = a[used]
t0 = a[:, :, 0].min()
start = (a[:, :, 0] - t0) * 0.01   # us
end = (a[:, :, 1] - t0) * 0.01
dur = end - start
surv = a[:, :, 2]
print(wl, "launch ms", r.search_launch_ms(), "blocks", len(a))
print("span us", end.max(), "wave dur us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("survivors/wave: mean %.1f p99 %.0f max %d; corr(dur,surv) %.3f" % (surv.mean(), np.percentile(surv, 99), surv.max(), np.corrcoef(dur.ravel(), surv.ravel())[0, 1]))
# occupancy over time: number of waves running at sampled times
ts = np.linspace(0, end.max(), 21)
print("running waves at t:", [int(((start <= t) & (end > t)).sum()) for t in ts])
# fit dur ~ a + b*surv
A = np.vstack([np.ones(dur.size), surv.ravel()]).T
coef = np.linalg.lstsq(A, dur.ravel(), rcond=None)[0]
print("fit: dur = %.1f us + %.4f us * survivors" % (coef[0], coef[1]))
