#!/usr/bin/env python3
"""Developer experiment (GPU box): cheap scenes under tile orders x age priority.  python tools/dev_cheap.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera


def run(sid, kid=0, W=1920, H=1080, **kw):
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
    desc = _native.make_desc(sid, kid, cam, W, H, **kw)
    out = _native.render(desc, warmup=3, repeats=9)
    t = out["timing"]
    print(json.dumps({"scene": sc.name, **kw, "ms": round(t["ms_median"], 4), "ms_min": round(t["ms_min"], 4),
                      "iter_max": int(out["stats"]["iter_max"])}), flush=True)


_native.init()
if len(sys.argv) > 1 and sys.argv[1] == "orders":
    for sid in range(20):
        for kid in (0, 4):
            for tom in (3, 2):
                run(sid, kid, tile_order_mode=tom)
    sys.exit(0)
scenes = [int(v) for v in sys.argv[1:]] or [0, 2, 9, 12, 3, 5]
for sid in scenes:
    for tom in (0, 2, 1):
        for ap in (0, 8, 32, 64):
            run(sid, tile_order_mode=tom, age_priority=ap)
