#!/usr/bin/env python3
"""Are the library's schedule defaults viewpoint-robust?  Every curated viewpoint of the reference (viewpoints.py:41-123,
raymarch_algo_compare_amd/viewpoints.py) at 1920x1080, Standard strategy: device time per frame with the library's own
choice (tile_order_mode 0) and with each static tile order (2 centre-out, 3 natural, 4 middle rows first).
    python tools/viewpoint_orders.py > profiles/r03/viewpoint_orders.jsonl
One JSON line per (scene, viewpoint); the last line summarises: cells where the default is more than 5 % behind the best."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera
from raymarch_algo_compare_amd.viewpoints import viewpoints_for

W, H = 1920, 1080
_native.init()
worst, rows = [], 0
for sc in registry.SCENES:
    for vp in viewpoints_for(sc):
        cam = Camera(vp.position, vp.target, vp.up, 60.0, W, H).params14()
        ms = {}
        for order in (0, 2, 3, 4):
            out = _native.render(_native.make_desc(sc.id, 0, cam, W, H, tile_order_mode=order), warmup=2, repeats=5)
            ms[order] = round(out["timing"]["ms_median"], 4)
        best = min(ms[2], ms[3], ms[4])
        row = {"scene": sc.name, "viewpoint": vp.name, "category": vp.category, "default_ms": ms[0], "centre_out_ms": ms[2],
               "natural_ms": ms[3], "middle_rows_ms": ms[4], "default_over_best": round(ms[0] / best, 3),
               "iter_max": int(out["stats"]["iter_max"])}
        rows += 1
        if ms[0] > 1.05 * best:
            worst.append((sc.name, vp.name, row["default_over_best"]))
        print(json.dumps(row), flush=True)
print(json.dumps({"cells": rows, "default_more_than_5pct_behind_best": worst}), flush=True)
