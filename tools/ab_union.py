#!/usr/bin/env python3
"""A/B helper: union scenes (Sphere Cloud 14, Bumpy Sphere 15) of one checkout.  python tools/ab_union.py <checkout>"""
import json, os, sys
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
_native.init()
for sid in (14, 15, 16):
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 1920, 1080).params14()
    for kw in (dict(), dict(suspend_after=(-1, -1)), dict(resume_mode=1)):
        out = _native.render(_native.make_desc(sid, 0, cam, 1920, 1080, **kw), warmup=2, repeats=7)
        print(json.dumps({"root": os.path.basename(root), "scene": sc.name, **kw, "ms": round(out["timing"]["ms_median"], 3)}), flush=True)
