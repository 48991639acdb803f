#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box (box-to-box variance is +-5 %): every cell is timed with each library in
a process of its own.   python tools/ab_lib.py <libA.so> <libB.so>
(timing only: parity of a development library is what tests/ and tests/fuzz_parity.py check, with RM_HIP_LIB set)
--cells "12,0,1920,1080;12,4,7680,4320" replaces the default cells; --same also compares the two libraries' frames bit
for bit (iterations, hits, raw fp64 t, final_sdf: what a development build must not change).  Default cells: Mandelbulb / Standard and Enhanced 1920x1080 and Standard 7680x4320, Sphere and Cube / Standard 1920x1080."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CELLS = [(10, 0, 1920, 1080), (10, 4, 1920, 1080), (10, 0, 7680, 4320), (0, 0, 1920, 1080), (2, 0, 1920, 1080)]


def child():
    global CELLS
    if os.environ.get("RM_AB_CELLS"):
        CELLS = [tuple(int(v) for v in c.split(",")) for c in os.environ["RM_AB_CELLS"].split(";")]
    sys.path.insert(0, ROOT)
    from raymarch_algo_compare_amd import _native, registry
    from raymarch_algo_compare_amd.camera import Camera
    _native.init()
    for cell in CELLS:
        sid, kid, W, H = cell[:4]
        shard = dict(row0=cell[4], rows=cell[5]) if len(cell) >= 6 else {}          # "sid,kid,W,H,row0,rows": a row shard
        sc = registry.SCENES[sid]
        cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
        try:
            out = _native.render(_native.make_desc(sid, kid, cam, W, H, **shard), warmup=2, repeats=7 if W < 4000 else 3)
        except _native.RmError as e:
            print(json.dumps({"cell": list(cell), "error": str(e)}), flush=True)
            continue
        row = {"cell": list(cell), "ms": round(out["timing"]["ms_median"], 4), "ms_min": round(out["timing"]["ms_min"], 4)}
        if os.environ.get("RM_AB_DUMP"):
            import hashlib
            full = _native.render(_native.make_desc(sid, kid, cam, W, H, full=True, **shard), want_t_raw=True, want_final_sdf=True)
            row["sha"] = {k: hashlib.sha256(full[k].tobytes()).hexdigest()[:16] for k in ("iters", "hit", "depth", "t_raw", "final_sdf")}
            row["iters_total"] = int(full["iters"].sum())
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child()
    else:
        args = sys.argv[1:]
        if "--cells" in args:
            os.environ["RM_AB_CELLS"] = args.pop(args.index("--cells") + 1)
            args.remove("--cells")
        if "--same" in args:
            os.environ["RM_AB_DUMP"] = "1"
        libs = [a for a in args if not a.startswith("--")]
        seen = {}
        for rnd in range(2):                       # A B A B: drift of the box shows as a difference between the rounds
            for lib in libs:
                print(json.dumps({"lib": lib, "round": rnd}), flush=True)
                env = dict(os.environ, RM_HIP_LIB=os.path.abspath(lib))
                res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False, timeout=400,
                                     capture_output=True, text=True)
                sys.stdout.write(res.stdout)
                sys.stderr.write(res.stderr[-2000:])
                for line in res.stdout.splitlines():
                    row = json.loads(line)
                    if "sha" in row:
                        seen.setdefault(tuple(row["cell"]), set()).add(json.dumps(row["sha"], sort_keys=True))
        if seen:
            bad = [c for c, v in seen.items() if len(v) != 1]
            print(json.dumps({"same": not bad, "cells": len(seen), "differ": bad}), flush=True)
            sys.exit(1 if bad else 0)
