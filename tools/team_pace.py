#!/usr/bin/env python3
"""Pace of a team chain: the longest rays of the 1920x1080 Mandelbulb / Standard frame marched again by rm_march_rays_team
(three waves per 64 rays; since round 3 with filler workgroups that keep the rest of the chip busy): microseconds per
evaluation from the difference of two iteration budgets.   python tools/team_pace.py [--lib path]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    args = sys.argv[1:]
    if "--lib" in args:
        os.environ["RM_HIP_LIB"] = os.path.abspath(args.pop(args.index("--lib") + 1))
    from raymarch_algo_compare_amd import _native, registry
    from raymarch_algo_compare_amd.camera import Camera
    W, H, sid = 1920, 1080, 10
    _native.init()
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
    it = _native.render(_native.make_desc(sid, 0, cam, W, H))["iters"]
    ys, xs = np.nonzero(it >= 500)
    ys, xs = ys[:128], xs[:128]
    pos, fwd, right, up = (np.asarray(cam[i:i + 3], dtype=np.float64) for i in (0, 3, 6, 9))
    u = (2.0 * (xs + 0.5) / W - 1.0) * float(cam[12])
    v = (1.0 - 2.0 * (ys + 0.5) / H) * float(cam[13])
    dirs = (fwd[None, :] + right[None, :] * u[:, None]) + up[None, :] * v[:, None]
    origins = np.repeat(pos[None, :], len(xs), axis=0)
    wall = {}
    for budget in (64, 512):
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            _native.march_rays(sid, 0, origins, dirs, max_iterations=budget, team=True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        wall[budget] = best
    print(json.dumps({"rays": int(len(xs)), "call_ms": {str(k): round(v * 1e3, 3) for k, v in wall.items()},
                      "team_us_per_evaluation": round((wall[512] - wall[64]) * 1e6 / 448, 3)}), flush=True)


if __name__ == "__main__":
    main()
