#!/usr/bin/env python3
"""A/B helper: every scene x Standard of one checkout, default schedule.  python tools/ab_all.py <checkout> [order]"""
import json, os, sys
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
_native.init()
kw = dict(tile_order_mode=int(sys.argv[2])) if len(sys.argv) > 2 else {}
for sid in range(20):
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 1920, 1080).params14()
    out = _native.render(_native.make_desc(sid, 0, cam, 1920, 1080, **kw), warmup=2, repeats=9)
    print(json.dumps({"root": os.path.basename(root), "sid": sid, "scene": sc.name, "ms": round(out["timing"]["ms_median"], 4), "min": round(out["timing"]["ms_min"], 4)}), flush=True)
