#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box (box-to-box variance is +-5 %): every cell is timed with each library in
a process of its own.   python tools/ab_lib.py <libA.so> <libB.so>
(timing only: parity of a development library is what tests/ and tests/fuzz_parity.py check, with RM_HIP_LIB set)
Cells: Mandelbulb / Standard and Enhanced 1920x1080 and Standard 7680x4320, Sphere and Cube / Standard 1920x1080."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CELLS = [(10, 0, 1920, 1080), (10, 4, 1920, 1080), (10, 0, 7680, 4320), (0, 0, 1920, 1080), (2, 0, 1920, 1080)]


def child():
    sys.path.insert(0, ROOT)
    from raymarch_algo_compare_amd import _native, registry
    from raymarch_algo_compare_amd.camera import Camera
    _native.init()
    for sid, kid, W, H in CELLS:
        sc = registry.SCENES[sid]
        cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
        try:
            out = _native.render(_native.make_desc(sid, kid, cam, W, H), warmup=2, repeats=7 if W < 4000 else 3)
        except _native.RmError as e:
            print(json.dumps({"cell": [sid, kid, W, H], "error": str(e)}), flush=True)
            continue
        row = {"cell": [sid, kid, W, H], "ms": round(out["timing"]["ms_median"], 4), "ms_min": round(out["timing"]["ms_min"], 4)}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child()
    else:
        libs = [a for a in sys.argv[1:] if not a.startswith("--")]
        for rnd in range(2):                       # A B A B: drift of the box shows as a difference between the rounds
            for lib in libs:
                print(json.dumps({"lib": lib, "round": rnd}), flush=True)
                env = dict(os.environ, RM_HIP_LIB=os.path.abspath(lib))
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False, timeout=400)
