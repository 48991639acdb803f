#!/usr/bin/env python3
"""When does a frame's kernel start and finish each ray?  Needs a development library built with time stamps
(make -C raymarch_algo_compare_amd/csrc DEV=1 DEVSCENES="0 12" EXTRA=-DRM_DEV_STAMP: `evals` carries the finish time and
`final_sdf` the start time of every ray, 100 MHz device clock; one launch per frame, no parked rays).
  python tools/ray_times.py <sid> [kid] [--knob name=value] --lib raymarch_algo_compare_amd/librm_hip_dev.so"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    args = sys.argv[1:]
    if "--lib" in args:
        os.environ["RM_HIP_LIB"] = os.path.abspath(args.pop(args.index("--lib") + 1))
        args.remove("--lib")
    from raymarch_algo_compare_amd import _native, registry
    from raymarch_algo_compare_amd.camera import Camera
    knobs = {}
    while "--knob" in args:                               # --knob name=value (repeatable)
        k, v = args.pop(args.index("--knob") + 1).split("=")
        knobs[k] = int(v)
        args.remove("--knob")
    sid, kid = int(args[0]), int(args[1]) if len(args) > 1 else 0
    W, H = 1920, 1080
    _native.init()
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
    desc = _native.make_desc(sid, kid, cam, W, H, full=True, suspend_after=(-1, 0), **knobs)
    for _ in range(3):                                    # warm
        out = _native.render(desc, want_final_sdf=True, want_evals=True)
    it = out["iters"].ravel()
    fin = out["evals"].ravel().astype(np.int64)
    sta = out["final_sdf"].ravel().astype(np.int64)
    t0 = int(sta.min())
    fin_us, sta_us = (fin - t0) / 100.0, (sta - t0) / 100.0
    row = {"scene": sc.name, "strategy": kid, "frame_span_us": round(float(fin_us.max()), 1),
           "last_start_us": round(float(sta_us.max()), 1), "start_percentiles_us": [round(float(v), 1) for v in np.percentile(sta_us, [10, 50, 90, 99])]}
    print(json.dumps(row), flush=True)
    # by iteration class: when they start, when they finish, pace
    edges = [1, 8, 16, 32, 64, 128, 256, 513]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (it >= lo) & (it < hi)
        if not m.any():
            continue
        life = fin_us[m] - sta_us[m]
        print(json.dumps({"iterations": [lo, hi - 1], "rays": int(m.sum()), "start_us_median": round(float(np.median(sta_us[m])), 1),
                          "start_us_max": round(float(sta_us[m].max()), 1), "finish_us_median": round(float(np.median(fin_us[m])), 1),
                          "finish_us_max": round(float(fin_us[m].max()), 1),
                          "us_per_iteration_median": round(float(np.median(life / np.maximum(it[m], 1))), 3)}), flush=True)
    order = np.argsort(fin_us)[::-1][:12]
    print(json.dumps({"last_finishers": [{"iters": int(it[i]), "start_us": round(float(sta_us[i]), 1), "finish_us": round(float(fin_us[i]), 1),
                                          "x": int(i % W), "y": int(i // W)} for i in order]}), flush=True)
    top = np.argsort(it)[::-1][:12]
    print(json.dumps({"longest_rays": [{"iters": int(it[i]), "start_us": round(float(sta_us[i]), 1), "finish_us": round(float(fin_us[i]), 1),
                                        "x": int(i % W), "y": int(i // W)} for i in top]}), flush=True)


if __name__ == "__main__":
    main()
