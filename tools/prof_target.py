#!/usr/bin/env python3
"""Profiling target for rocprofv3: N launches of one (scene, strategy) kernel at WxH.
usage: python3 tools/prof_target.py [scene_id strategy_id W H launches]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
a = [int(v) for v in sys.argv[1:]] + [None] * 5
sid, kid, W, H, n = (a[0] if a[0] is not None else 10, a[1] or 0, a[2] or 1920, a[3] or 1080, a[4] or 5)
scene = registry.SCENES[sid]
cam = Camera(scene.camera_position or (0.0, 0.0, 5.0), scene.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H)
tuning = json.loads(os.environ.get("RM_TUNING", "{}"))     # e.g. {"suspend_after": [32, 128], "eval_mode": 2}
desc = _native.make_desc(sid, kid, cam.params14(), W, H, **tuning)
out = _native.render(desc, warmup=1, repeats=n)
print(json.dumps({"scene": scene.name, "WxH": f"{W}x{H}", "ms_each": out["timing"]["ms_each"], "rays": W * H}))
