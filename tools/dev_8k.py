#!/usr/bin/env python3
"""Developer experiment (GPU box): 7680x4320 frames under the tile orders.  python tools/dev_8k.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
_native.init()
for sid in (10, 0, 2, 9):
    sc = registry.SCENES[sid]
    for W, H in ((7680, 4320), (3840, 2160)):
        cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
        for tom in (3, 2, 0):
            if sid == 10 and (W, H) == (3840, 2160) and tom != 0:
                pass
            out = _native.render(_native.make_desc(sid, 0, cam, W, H, tile_order_mode=tom), warmup=1, repeats=3)
            print(json.dumps({"scene": sc.name, "WxH": f"{W}x{H}", "tile_order_mode": tom, "ms": round(out["timing"]["ms_median"], 3)}), flush=True)
