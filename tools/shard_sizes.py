#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: device time of a band-cyclic row shard of the 7680x4320 frame for 1/8, 1/4, 1/2 of the
rows and the whole frame -- what one rank of an N-GPU run renders (no exchange; the gather is separate) -- Mandelbulb
and Pillar Forest, Standard.  The ratio whole / shard is the speed-up N ranks could reach before the gather: PROJECTED,
not measured on N GPUs.   python tools/shard_sizes.py [knob=value ...]   (knobs go into every descriptor)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera

W, H = 7680, 4320
knobs = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:])}
_native.init()
for sid in (10, 12):
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
    whole = None
    for n, offset in ((1, 0), (2, 1), (4, 1), (8, 0), (8, 5)):
        kw = dict(knobs)
        if n > 1:
            kw.update(rows=H // n, band_rows=4, band_stride=n, band_offset=offset)
        out = _native.render(_native.make_desc(sid, 0, cam, W, H, **kw), warmup=1, repeats=3)
        ms = out["timing"]["ms_median"]
        whole = ms if n == 1 else whole
        print(json.dumps({"scene": sc.name, "ranks": n, "band_offset": offset, "shard_ms": round(ms, 3), "rays": int(out["stats"]["total_rays"]),
                          "whole_over_shard": round(whole / ms, 2), **knobs}), flush=True)
