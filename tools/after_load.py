#!/usr/bin/env python3
"""Does a short frame run faster right after the chip has been busy?  Times a cheap frame (a) repeated on its own, (b) each
time right after a long dense frame (Mandelbulb 3840x2160 without suspension: ~20 ms of full waves on every SIMD), (c) on its
own again, and (d) right after a host-side pause of 50 ms (an idle chip).  Device time of the single launch, events on its stream.
  python tools/after_load.py [scene_id ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry           # noqa: E402
from raymarch_algo_compare_amd.camera import Camera               # noqa: E402


def cam_of(sid, W, H):
    sc = registry.SCENES[sid]
    return Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()


def main():
    _native.init()
    heavy = _native.make_desc(10, 0, cam_of(10, 3840, 2160), 3840, 2160, suspend_after=(-1, 0), pipeline=1)
    for sid in [int(a) for a in sys.argv[1:]] or [0, 2, 12]:
        W, H = 1920, 1080
        desc = _native.make_desc(sid, 0, cam_of(sid, W, H), W, H)
        row = {"scene": registry.SCENES[sid].name}
        row["alone_ms"] = [round(v, 4) for v in _native.render(desc, warmup=3, repeats=8)["timing"]["ms_each"]]
        after = []
        for _ in range(6):
            _native.render(heavy)
            after.append(round(_native.render(desc, warmup=0, repeats=3)["timing"]["ms_each"][0], 4))
        row["first_frame_after_a_dense_20ms_frame_ms"] = after
        row["three_frames_after_it_ms"] = [round(v, 4) for v in _native.render(desc, warmup=0, repeats=3)["timing"]["ms_each"]]
        idle = []
        for _ in range(6):
            time.sleep(0.05)
            idle.append(round(_native.render(desc, warmup=0, repeats=1)["timing"]["ms_each"][0], 4))
        row["first_frame_after_50ms_idle_ms"] = idle
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
