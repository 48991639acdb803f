#!/usr/bin/env python3
"""Is a schedule default viewpoint-robust?  Every curated viewpoint of one scene (viewpoints.py) at 1920x1080, a few
strategies: device time per frame over the values of ONE RmFrameDesc knob (pairs as a:b; 0 or 0:0 = the library's choice).
  python tools/viewpoint_knob.py <scene_id> <knob> "<v0,v1,...>" ["<strategy ids>" [WxH]]
  python tools/viewpoint_knob.py 10 suspend_after "0:0,16:48,16:40,24:40,24:48,32:48,24:56" "0,4,9"    # profiles/r03/viewpoint_budgets.jsonl
  python tools/viewpoint_knob.py 10 tile_order_mode "0,2,3,4" "0,4,9,6"
(timing only: every value yields the same frames)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
from raymarch_algo_compare_amd.viewpoints import viewpoints_for

W, H = 1920, 1080


def val(v):
    return tuple(int(x) for x in v.split(":")) if ":" in v else int(v)


sc = registry.SCENES[int(sys.argv[1])]
knob, values = sys.argv[2], [val(v) for v in sys.argv[3].split(",")]
strategies = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 4, 9]
if len(sys.argv) > 5:
    W, H = (int(v) for v in sys.argv[5].split("x"))
_native.init()
total = {}
for vp in viewpoints_for(sc):
    cam = Camera(vp.position, vp.target, vp.up, 60.0, W, H).params14()
    for kid in strategies:
        ms = {}
        for v in values:
            out = _native.render(_native.make_desc(sc.id, kid, cam, W, H, **{knob: v}), warmup=2, repeats=5)
            name = ":".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
            ms[name] = round(out["timing"]["ms_median"], 3)
            total[(kid, name)] = total.get((kid, name), 0.0) + out["timing"]["ms_median"]
        print(json.dumps({"viewpoint": vp.name, "strategy": kid, "knob": knob, "iter_max": int(out["stats"]["iter_max"]), "ms": ms}), flush=True)
print(json.dumps({"sum_over_viewpoints_ms": {f"strategy {k} {knob} {n}": round(v, 2) for (k, n), v in sorted(total.items())}}), flush=True)
