#!/usr/bin/env python3
"""Developer experiment (GPU box): refill threshold under the default tile orders."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
_native.init()
for sid in (0, 2, 4, 6, 9, 12, 18, 19, 3, 8):
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 1920, 1080).params14()
    row = {"scene": sc.name}
    for rm in (2, 4, 8, 12, 16, 24):
        out = _native.render(_native.make_desc(sid, 0, cam, 1920, 1080, refill_min=rm), warmup=2, repeats=9)
        row[f"refill_{rm}"] = round(out["timing"]["ms_median"], 4)
    print(json.dumps(row), flush=True)
