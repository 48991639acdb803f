#!/usr/bin/env python3
"""Developer experiment (GPU box): tile orders natural / centre-out / middle-rows-first / previous frame for the scenes whose
geometry runs to the horizon, and a few others."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
_native.init()
for sid in (1, 12, 13, 0, 2, 9, 16, 18):
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 1920, 1080).params14()
    for kid in (0, 4, 10):
        row = {"scene": sc.name, "strategy": registry.list_strategies()[kid]}
        lip = (sc.lipschitz or 1.0) if kid == 10 else 1.0
        for tom in (3, 2, 4, 1):
            out = _native.render(_native.make_desc(sid, kid, cam, 1920, 1080, lipschitz=lip, tile_order_mode=tom), warmup=3, repeats=9)
            row[f"order_{tom}"] = round(out["timing"]["ms_median"], 4)
        print(json.dumps(row), flush=True)
