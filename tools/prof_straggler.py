#!/usr/bin/env python3
"""Profiling target: ONE wavefront of 64 straggler rays (512-iteration Mandelbulb rays) through
rm_march_rays -- isolates the speed of a single wave's instruction stream (the 1080p frame time is
the latency of such rays).  Run under rocprofv3 --kernel-trace / --pmc."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
W, H = 960, 540
scene = registry.SCENES[10]
cam = Camera(scene.camera_position, scene.camera_target, (0.0, 1.0, 0.0), 60.0, W, H)
out = _native.render(_native.make_desc(10, 0, cam.params14(), W, H))
ys, xs = np.nonzero(out["iters"] >= 512)
print("stragglers:", len(ys), "of", W * H)
c = cam.params14()
pos, fwd, right, up, hw, hh = c[0:3], c[3:6], c[6:9], c[9:12], c[12], c[13]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sel = np.arange(n) % len(ys)
u = (2.0 * (xs[sel] + 0.5) / W - 1.0) * hw
v = (1.0 - 2.0 * (ys[sel] + 0.5) / H) * hh
dirs = fwd[None, :] + right[None, :] * u[:, None] + up[None, :] * v[:, None]
orig = np.repeat(pos[None, :], n, 0)
ref = None
for team in (False, True, False, True):
    for rep in range(3):
        t0 = time.perf_counter()
        hit, t, iters, fs = _native.march_rays(10, 0, orig, dirs, team=team)
        dt = time.perf_counter() - t0
    print(json.dumps({"team": team, "n": n, "wall_ms": round(dt * 1e3, 3), "iters_min": int(iters.min()), "iters_mean": float(iters.mean())}))
    cur = (hit.tobytes(), t.tobytes(), iters.tobytes(), fs.tobytes())
    if ref is None: ref = cur
    print("identical to first:", cur == ref)
# a mixed batch: all rays of a small frame through both forms, every strategy
W2, H2 = 96, 54
cam2 = Camera(scene.camera_position, scene.camera_target, (0.0, 1.0, 0.0), 60.0, W2, H2)
c2 = cam2.params14()
px, py = np.meshgrid(np.arange(W2), np.arange(H2))
u2 = (2.0 * (px.ravel() + 0.5) / W2 - 1.0) * c2[12]
v2 = (1.0 - 2.0 * (py.ravel() + 0.5) / H2) * c2[13]
d2 = c2[3:6][None, :] + c2[6:9][None, :] * u2[:, None] + c2[9:12][None, :] * v2[:, None]
o2 = np.repeat(c2[0:3][None, :], len(d2), 0)
for kid in range(11):
    a = _native.march_rays(10, kid, o2, d2)
    b = _native.march_rays(10, kid, o2, d2, team=True)
    print("strategy", kid, "team == single:", all(x.tobytes() == y.tobytes() for x, y in zip(a, b)))
