#!/usr/bin/env python3
"""Are the single launch's trip budgets viewpoint-robust?  Every curated viewpoint of the Mandelbulb (viewpoints.py) at
1920x1080, Standard / Enhanced / Adaptive-Hybrid: device time per frame for a few (strike, hand-over) budgets
(RmFrameDesc.suspend_after; 0:0 = the library's choice).   python tools/viewpoint_budgets.py [scene_id]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
from raymarch_algo_compare_amd.viewpoints import viewpoints_for

W, H = 1920, 1080
BUDGETS = [(0, 0), (16, 48), (16, 40), (24, 40), (24, 48), (32, 48), (24, 56)]
_native.init()
sc = registry.SCENES[int(sys.argv[1]) if len(sys.argv) > 1 else 10]
total = {}
for vp in viewpoints_for(sc):
    cam = Camera(vp.position, vp.target, vp.up, 60.0, W, H).params14()
    for kid in (0, 4, 9):
        ms = {}
        for b in BUDGETS:
            out = _native.render(_native.make_desc(sc.id, kid, cam, W, H, suspend_after=b), warmup=2, repeats=5)
            ms[f"{b[0]}:{b[1]}"] = round(out["timing"]["ms_median"], 3)
            total[(kid, b)] = total.get((kid, b), 0.0) + out["timing"]["ms_median"]
        print(json.dumps({"viewpoint": vp.name, "strategy": kid, "iter_max": int(out["stats"]["iter_max"]), "ms": ms}), flush=True)
print(json.dumps({"sum_over_viewpoints_ms": {f"strategy {k} budgets {b[0]}:{b[1]}": round(v, 2) for (k, b), v in sorted(total.items())}}), flush=True)
