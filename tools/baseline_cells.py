#!/usr/bin/env python3
"""The nine published cells of BASELINE.md section 1b (1920x1080, the reference's GLSL fp32 column) on this engine:
median device time of 9 frames per cell, default schedule and with the previous frame's tile costs.
    python tools/baseline_cells.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera

CELLS = [("Sphere", "Standard", 0.389), ("Cube", "Standard", 0.653), ("Menger", "Standard", 3.73), ("Menger", "Enhanced", 2.83),
         ("Menger", "Overstep-Bisect", 3.19), ("Mandelbulb", "Standard", 11.92), ("Mandelbulb", "Enhanced", 5.54),
         ("Mandelbulb", "Overstep-Bisect", 12.85), ("Pillar Forest", "Standard", 3.34)]
_native.init()
for scene_name, strat, ref_ms in CELLS:
    sc = registry.get_scene_by_name(scene_name)
    st = registry.get_strategy_by_name(strat)
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 1920, 1080).params14()
    row = {"scene": sc.name, "strategy": st.key, "reference_glsl_fp32_ms": ref_ms}
    for name, kw in (("ms", {}), ("ms_temporal_order", dict(tile_order_mode=1))):
        out = _native.render(_native.make_desc(sc.id, st.id, cam, 1920, 1080, **kw), warmup=3, repeats=9)
        row[name] = round(out["timing"]["ms_median"], 3)
    print(json.dumps(row), flush=True)
# the graded 14 x 9 matrix (BASELINE config 4): total device time of one frame per cell
total = 0.0
for sid in registry.GRADED_SCENE_IDS:
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 1920, 1080).params14()
    for key in registry.GRADED_STRATEGY_KEYS:
        st = registry.get_strategy_by_name(key)
        lip = (sc.lipschitz or 1.0) if st.has_lipschitz else 1.0
        out = _native.render(_native.make_desc(sc.id, st.id, cam, 1920, 1080, lipschitz=lip), warmup=1, repeats=3)
        total += out["timing"]["ms_median"]
print(json.dumps({"graded_matrix_14x9_1080p_total_ms": round(total, 1)}), flush=True)
