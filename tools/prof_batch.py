#!/usr/bin/env python3
"""Profiling target: one rm_render_batch call (8 Mandelbulb 384x384 viewpoints)."""
import sys, os, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raymarch_algo_compare_amd import _native
from raymarch_algo_compare_amd.camera import Camera
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = h = 384
cams = [Camera((3 * math.sin(2 * math.pi * i / n), 0.3, 3 * math.cos(2 * math.pi * i / n)), (0, 0, 0), (0, 1, 0), 60.0, w, h).params14() for i in range(n)]
shape = _native.make_desc(10, 0, cams[0], w, h)
out = _native.render_batch(shape, np.array(cams))
print(json.dumps({"frames": n, "ms_total": out["ms_total"]}))
