#!/usr/bin/env python3
"""Developer experiment (GPU box): rm_render_batch of N Mandelbulb 384x384 viewpoints under team shares; Sphere batch."""
import sys, os, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raymarch_algo_compare_amd import _native
from raymarch_algo_compare_amd.camera import Camera
w = h = 384
for n in (8, 64):
    cams = np.array([Camera((3 * math.sin(2 * math.pi * i / n), 0.3, 3 * math.cos(2 * math.pi * i / n)), (0, 0, 0), (0, 1, 0), 60.0, w, h).params14() for i in range(n)])
    for tg in (0, 64, 96, 128, 192):
        shape = _native.make_desc(10, 0, cams[0], w, h, team_grid=tg)
        ms = [_native.render_batch(shape, cams)["ms_total"] for _ in range(4)]
        print(json.dumps({"scene": "Mandelbulb", "frames": n, "team_grid": tg, "ms": sorted(round(m, 2) for m in ms)}), flush=True)
    shape = _native.make_desc(10, 0, cams[0], w, h, pipeline=1)
    ms = [_native.render_batch(shape, cams)["ms_total"] for _ in range(4)]
    print(json.dumps({"scene": "Mandelbulb", "frames": n, "pipeline": 1, "ms": sorted(round(m, 2) for m in ms)}), flush=True)
cams = np.array([Camera((5 * math.sin(2 * math.pi * i / 64), 0.3, 5 * math.cos(2 * math.pi * i / 64)), (0, 0, 0), (0, 1, 0), 60.0, w, h).params14() for i in range(64)])
shape = _native.make_desc(0, 0, cams[0], w, h)
ms = [_native.render_batch(shape, cams)["ms_total"] for _ in range(4)]
print(json.dumps({"scene": "Sphere", "frames": 64, "ms": sorted(round(m, 2) for m in ms)}), flush=True)
