#!/usr/bin/env python3
"""Developer experiment (GPU box): persistent grid size (waves) for the cheap scenes under the centre-out order."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
_native.init()
for sid in (0, 2, 4, 9, 3, 5, 8, 12, 18):
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 1920, 1080).params14()
    row = {"scene": sc.name}
    for gw in (0, 1024, 1536, 2048, 2560, 4096):
        out = _native.render(_native.make_desc(sid, 0, cam, 1920, 1080, grid_waves=gw), warmup=2, repeats=9)
        row[f"waves_{gw}"] = round(out["timing"]["ms_median"], 4)
    print(json.dumps(row), flush=True)
