#!/usr/bin/env python3
"""Developer sweep on a GPU box: kernel ms / Mrays/s for a few (scene, strategy) cells and
schedule parameters at 1920x1080.  Not part of the product or the tests."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera

def cam_for(scene, W, H):
    pos = scene.camera_position or (0.0, 0.0, 5.0)
    tgt = scene.camera_target or (0.0, 0.0, 0.0)
    return Camera(pos, tgt, (0.0, 1.0, 0.0), 60.0, W, H)

def run(sid, kid, W=1920, H=1080, repeats=5, warmup=2, **tuning):
    scene = registry.SCENES[sid]
    lip = scene.lipschitz if (kid == 10 and scene.lipschitz) else 1.0
    desc = _native.make_desc(sid, kid, cam_for(scene, W, H).params14(), W, H, lipschitz=lip, **tuning)
    out = _native.render(desc, warmup=warmup, repeats=repeats)
    ms = out["timing"]["ms_median"]
    st = out["stats"]
    return dict(scene=scene.name, strategy=registry.list_strategies()[kid], ms=round(ms, 4),
                mrays=round(W * H / ms / 1e3, 1), mean_iters=round(st["sum_iters"] / st["total_rays"], 3),
                max_it=st["iter_max"], hits=st["hit_count"], **tuning)

if __name__ == "__main__":
    print(json.dumps(_native.device_info()))
    cells = [(0, 0), (2, 0), (9, 0), (10, 0), (10, 4), (10, 6), (12, 0), (13, 0)]
    for sid, kid in cells:
        print(json.dumps(run(sid, kid)), flush=True)
    for tuning in [dict(refill_min=1), dict(refill_min=8), dict(refill_min=48), dict(refill_min=64),
                   dict(grid_waves=1024), dict(grid_waves=2048), dict(grid_waves=4096), dict(grid_waves=8192)]:
        for sid, kid in [(10, 0), (0, 0)]:
            print(json.dumps(run(sid, kid, **tuning)), flush=True)
