#!/usr/bin/env python3
"""Is a frame bound by its longest ray?  For (scene, strategy) at 1920x1080: the frame's device time at iteration budgets
64 / 128 / 512 (the dense part against the tail), and the pace of the frame's own longest rays marched alone by
rm_march_rays (one wave, idle device): microseconds per evaluation from the difference of two budgets.
  python tools/chain_pace.py [scene_id ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry           # noqa: E402
from raymarch_algo_compare_amd.camera import Camera               # noqa: E402

W, H = 1920, 1080


def main():
    _native.init()
    for sid in [int(a) for a in sys.argv[1:]] or [0, 2, 12]:
        sc = registry.SCENES[sid]
        cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
        row = {"scene": sc.name}
        for budget in (32, 64, 128, 512):
            out = _native.render(_native.make_desc(sid, 0, cam, W, H, max_iterations=budget), warmup=2, repeats=7)
            row[f"frame_ms_budget_{budget}"] = round(out["timing"]["ms_median"], 4)
        it = out["iters"]
        order = np.argsort(it.ravel())[::-1][:32]
        ys, xs = np.unravel_index(order, it.shape)
        pos, fwd, right, up = (np.asarray(cam[i:i + 3], dtype=np.float64) for i in (0, 3, 6, 9))
        u = (2.0 * (xs + 0.5) / W - 1.0) * float(cam[12])
        v = (1.0 - 2.0 * (ys + 0.5) / H) * float(cam[13])
        dirs = (fwd[None, :] + right[None, :] * u[:, None]) + up[None, :] * v[:, None]
        origins = np.repeat(pos[None, :], len(xs), axis=0)
        top = int(it.max())
        wall = {}
        for budget in (top // 8, top):
            best = None
            for _ in range(5):
                t0 = time.perf_counter()
                _native.march_rays(sid, 0, origins, dirs, max_iterations=budget)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            wall[budget] = best
        row["longest_ray_iterations"] = top
        row["lone_wave_us_per_evaluation"] = round((wall[top] - wall[top // 8]) * 1e6 / (top - top // 8), 4)
        row["lone_chain_ms"] = round(row["lone_wave_us_per_evaluation"] * top / 1e3, 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
