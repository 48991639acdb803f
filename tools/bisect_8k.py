#!/usr/bin/env python3
"""Throughput-regime timing of one checkout of this repository (its own package + library): Mandelbulb/Standard at
7680x4320 and 1920x1080 without suspension, Pillar Forest and Cube at 1080p.  Run once per checkout in a process of its own:
    python tools/bisect_8k.py <path to a checkout with a built librm_hip.so>"""
import json
import os
import sys

root = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from raymarch_algo_compare_amd import _native, registry           # noqa: E402
from raymarch_algo_compare_amd.camera import Camera               # noqa: E402


def run(sid, W, H, **kw):
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
    desc = _native.make_desc(sid, 0, cam, W, H, **kw)
    out = _native.render(desc, warmup=2, repeats=7)
    t = out["timing"]
    print(json.dumps({"root": os.path.basename(root), "scene": sc.name, "WxH": f"{W}x{H}", **kw, "ms": round(t["ms_median"], 3),
                      "ms_min": round(t["ms_min"], 3), "mrays": round(W * H / t["ms_median"] / 1e3, 1)}), flush=True)


_native.init()
run(10, 7680, 4320)
run(10, 7680, 4320, suspend_after=(-1, -1))
run(10, 1920, 1080, suspend_after=(-1, -1))
run(10, 1920, 1080, suspend_after=(-1, -1), eval_mode=1)
run(12, 1920, 1080)
run(2, 1920, 1080)
run(0, 1920, 1080)
run(10, 1920, 1080)
run(10, 1920, 1080, tile_order_mode=1)
run(9, 1920, 1080)
