#!/usr/bin/env python3
"""Developer experiment (GPU box): 7680x4320 Mandelbulb/Standard under schedule knobs.  python tools/dev_8k_knobs.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
_native.init()
sc = registry.SCENES[10]
W, H = 7680, 4320
cam = Camera(sc.camera_position, sc.camera_target, (0.0, 1.0, 0.0), 60.0, W, H).params14()
for kw in (dict(), dict(refill_min=4), dict(refill_min=16), dict(refill_min=32), dict(eval_mode=1), dict(grid_waves=1536), dict(grid_waves=2560),
           dict(tile_order_mode=1), dict(tile_order_mode=1), dict(suspend_after=(64, 0)), dict(suspend_after=(128, 0)), dict(suspend_after=(48, 192)),
           dict(pipeline=2, suspend_after=(48, 96), team_grid=32), dict(pipeline=2, suspend_after=(64, 128), team_grid=48)):
    out = _native.render(_native.make_desc(10, 0, cam, W, H, **kw), warmup=1, repeats=3)
    print(json.dumps({**{k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}, "ms": round(out["timing"]["ms_median"], 2),
                      "mrays": round(W * H / out["timing"]["ms_median"] / 1e3, 1)}), flush=True)
