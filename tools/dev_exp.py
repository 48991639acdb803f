#!/usr/bin/env python3
"""Developer experiments on a GPU box (not product, not tests): the sweeps behind the scheduling defaults
(DESIGN.md section 3).  Usage: python tools/dev_exp.py <exp> ...   exp = budget | interleave | suspend | team |
grid | big | seg | planes | survey | refill | union | one <scene> <strategy> [repeats] | matrix | batch"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera

def cam_for(scene, W, H):
    return Camera(scene.camera_position or (0.0, 0.0, 5.0), scene.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H)

def run(sid, kid, W=1920, H=1080, repeats=5, warmup=2, max_iterations=512, **tuning):
    scene = registry.SCENES[sid]
    desc = _native.make_desc(sid, kid, cam_for(scene, W, H).params14(), W, H, max_iterations=max_iterations, **tuning)
    out = _native.render(desc, warmup=warmup, repeats=repeats)
    st = out["stats"]
    r = dict(scene=scene.name, strat=registry.list_strategies()[kid], WxH=f"{W}x{H}", maxit=max_iterations,
             ms=round(out["timing"]["ms_median"], 4), ms_min=round(out["timing"]["ms_min"], 4),
             mrays=round(W * H / out["timing"]["ms_median"] / 1e3, 1),
             mean_iters=round(st["sum_iters"] / max(st["total_rays"], 1), 3), max_it=st["iter_max"], **tuning)
    print(json.dumps(r), flush=True)
    return r

exp = sys.argv[1] if len(sys.argv) > 1 else "budget"
if exp == "budget":
    for mi in (16, 32, 64, 128, 256, 512):
        run(10, 0, max_iterations=mi)
    for gw in (256, 512, 1024, 2048, 4096):
        run(10, 0, grid_waves=gw)
    for gw in (512, 1024, 2048):
        run(10, 0, grid_waves=gw, refill_min=8)
    for W, H in ((960, 540), (3840, 2160), (7680, 4320)):
        run(10, 0, W=W, H=H, repeats=3, warmup=1)
elif exp == "interleave":
    for em in (1, 2):
        run(10, 0, repeats=7, warmup=2, eval_mode=em)
        run(10, 0, repeats=7, warmup=2, eval_mode=em, tile_order_mode=1, grid_waves=512)
        run(10, 0, repeats=7, warmup=2, eval_mode=em, tile_order_mode=1, grid_waves=1024)
        run(10, 0, repeats=7, warmup=2, eval_mode=em, tile_order_mode=1)
        for mi in (32, 64):
            run(10, 0, repeats=5, warmup=2, eval_mode=em, max_iterations=mi)
        for kid in (4, 6, 10):
            run(10, kid, repeats=5, warmup=2, eval_mode=em)
        run(10, 0, W=3840, H=2160, repeats=3, warmup=1, eval_mode=em)
        run(10, 0, W=7680, H=4320, repeats=3, warmup=1, eval_mode=em)
    for rmn in (1, 4, 16, 32):
        run(10, 0, repeats=5, warmup=2, eval_mode=2, refill_min=rmn)
        run(10, 0, W=3840, H=2160, repeats=3, warmup=1, eval_mode=2, refill_min=rmn)
elif exp == "suspend":
    import numpy as np
    # correctness: parked / resumed rays reproduce the unsuspended frame bit for bit
    for sid, kid, W, H in ((10, 0, 640, 360), (10, 5, 320, 200), (10, 9, 320, 200), (10, 10, 320, 200), (12, 0, 640, 360),
                           (13, 3, 333, 121), (9, 6, 640, 360), (0, 8, 640, 360)):
        scene = registry.SCENES[sid]
        outs = []
        for sa in ((-1, -1), (8, 0), (8, 40), (32, 128)):
            for em in ((1, 2) if sid == 10 else (1,)):
                desc = _native.make_desc(sid, kid, cam_for(scene, W, H).params14(), W, H, full=True, suspend_after=sa, eval_mode=em)
                o = _native.render(desc, want_t_raw=True, want_final_sdf=True, want_block_var=(H % 4 == 0 and W % 8 == 0))
                outs.append(o)
        ref = outs[0]
        ok = True
        for o in outs[1:]:
            for key in ("depth", "iters", "hit", "t_raw", "final_sdf", "block_var"):
                if ref[key] is not None and not np.array_equal(ref[key].view(np.uint8), o[key].view(np.uint8)):
                    ok = False; print("MISMATCH", sid, kid, key, int((ref[key] != o[key]).sum()))
            for key in ("total_rays", "hit_count", "sum_iters", "iter_max", "iter_min"):
                if ref["stats"][key] != o["stats"][key]:
                    ok = False; print("STATS MISMATCH", sid, kid, key, ref["stats"][key], o["stats"][key])
            if not np.array_equal(ref["stats"]["iter_hist"], o["stats"]["iter_hist"]):
                ok = False; print("HIST MISMATCH", sid, kid)
        print("suspend parity", scene.name, registry.list_strategies()[kid], "OK" if ok else "FAILED", flush=True)
    for sa in ((-1, -1), (16, 0), (32, 0), (48, 0), (64, 0), (16, 64), (32, 96), (32, 128), (48, 128), (24, 64), (32, 192)):
        for em in (1, 2):
            run(10, 0, repeats=7, warmup=2, eval_mode=em, suspend_after=sa)
    for sa in ((-1, -1), (32, 128)):
        for sid, kid in ((10, 4), (10, 6), (10, 10), (12, 0), (13, 0), (0, 0), (9, 0)):
            run(sid, kid, repeats=5, warmup=2, suspend_after=sa)
        run(10, 0, W=3840, H=2160, repeats=3, warmup=1, suspend_after=sa)
        run(10, 0, W=7680, H=4320, repeats=3, warmup=1, suspend_after=sa)
elif exp == "team":
    import numpy as np
    scene = registry.SCENES[10]
    for kid in (0, 5, 9, 10):
        W, H = 320, 200
        ref = _native.render(_native.make_desc(10, kid, cam_for(scene, W, H).params14(), W, H, full=True, suspend_after=(-1, -1)), want_t_raw=True, want_final_sdf=True, want_block_var=True)
        for sa, rv in (((8, 0), 2), ((8, 40), 2), ((32, 128), 2)):
            o = _native.render(_native.make_desc(10, kid, cam_for(scene, W, H).params14(), W, H, full=True, suspend_after=sa, resume_mode=rv), want_t_raw=True, want_final_sdf=True, want_block_var=True)
            ok = all(np.array_equal(ref[k].view(np.uint8), o[k].view(np.uint8)) for k in ("depth", "iters", "hit", "t_raw", "final_sdf", "block_var"))
            ok = ok and all(ref["stats"][k] == o["stats"][k] for k in ("total_rays", "hit_count", "sum_iters", "iter_max", "iter_min")) and np.array_equal(ref["stats"]["iter_hist"], o["stats"]["iter_hist"])
            print("team resume parity", kid, sa, "OK" if ok else "FAILED", flush=True)
    for kid in range(11):
        run(10, kid, repeats=5, warmup=2, suspend_after=(-1, -1))
        run(10, kid, repeats=5, warmup=2)
    for W, H in ((960, 540), (2560, 1440), (3840, 2160)):
        run(10, 0, W=W, H=H, repeats=3, warmup=1, suspend_after=(-1, -1))
        run(10, 0, W=W, H=H, repeats=3, warmup=1, suspend_after=(32, 128))
    for sa in ((24, 96), (32, 160), (40, 128), (32, 128)):
        run(10, 0, repeats=7, warmup=2, suspend_after=sa)
        run(10, 0, repeats=7, warmup=2, suspend_after=sa, tile_order_mode=1)
elif exp == "grid":
    for sid, kid in ((0, 0), (2, 0), (9, 0), (12, 0), (13, 0), (1, 5), (3, 10), (7, 2), (14, 0), (16, 0), (19, 0), (0, 6)):
        for gw in (0, 512, 1024, 1536, 2048, 3072, 4096):
            run(sid, kid, repeats=9, warmup=2, grid_waves=gw)
elif exp == "big":
    for W, H in ((3840, 2160), (5120, 2880), (7680, 4320)):
        for sa in ((-1, -1), (32, 128), (48, 192), (64, 0)):
            run(10, 0, W=W, H=H, repeats=3, warmup=1, suspend_after=sa)
elif exp == "seg":
    for kid in (10, 8, 7, 6):
        for sa in ((-1, -1), (16, 64), (16, 48), (24, 96), (32, 128), (8, 32)):
            run(10, kid, repeats=5, warmup=2, suspend_after=sa)
elif exp == "planes":
    for sid in (1, 13):
        for kid in range(11):
            for sa in ((-1, -1), (128, 0), (192, 0), (256, 0)):
                run(sid, kid, repeats=7, warmup=2, suspend_after=sa)
elif exp == "survey":
    for sid in (0, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 14, 15, 16, 17, 18, 19):
        for kid in (0, 5, 10, 9):
            for sa in ((-1, -1), (128, 0)):
                run(sid, kid, repeats=7, warmup=2, suspend_after=sa)
elif exp == "refill":
    for sid, kid in ((0, 0), (2, 0), (9, 0), (12, 0), (13, 0), (1, 5), (7, 2), (16, 0), (19, 0), (5, 0)):
        for rm_ in (4, 8, 16, 24, 32, 48):
            run(sid, kid, repeats=9, warmup=2, refill_min=rm_)
elif exp == "union":
    import numpy as np
    for sid in (14, 15):
        scene = registry.SCENES[sid]
        W, H = 320, 200
        for kid in (0, 5, 10):
            ref = _native.render(_native.make_desc(sid, kid, cam_for(scene, W, H).params14(), W, H, full=True, suspend_after=(-1, -1), eval_mode=1), want_t_raw=True, want_final_sdf=True, want_block_var=True)
            for kw in (dict(suspend_after=(8, 0), resume_mode=2), dict(suspend_after=(8, 40), resume_mode=2), dict(suspend_after=(8, 40), resume_mode=1), dict(eval_mode=2, suspend_after=(-1, -1)), dict(suspend_after=(6, 30), resume_mode=3)):
                o = _native.render(_native.make_desc(sid, kid, cam_for(scene, W, H).params14(), W, H, full=True, **kw), want_t_raw=True, want_final_sdf=True, want_block_var=True)
                ok = all(np.array_equal(ref[k].view(np.uint8), o[k].view(np.uint8)) for k in ("depth", "iters", "hit", "t_raw", "final_sdf", "block_var"))
                print("union team parity", sid, kid, kw, "OK" if ok else "FAILED", flush=True)
    for sid in (14, 15):
        for kid in (0, 5):
            for sa in ((-1, -1), (16, 0), (24, 0), (32, 0), (48, 0), (32, 128), (24, 96)):
                run(sid, kid, repeats=5, warmup=2, suspend_after=sa)
elif exp == "one":
    run(int(sys.argv[2]), int(sys.argv[3]), repeats=int(sys.argv[4]) if len(sys.argv) > 4 else 5)
elif exp == "matrix":
    for sid in range(14):
        for key in registry.GRADED_STRATEGY_KEYS:
            run(sid, registry.STRATEGIES[key], repeats=3, warmup=1)
elif exp == "batch":
    import math, time
    import numpy as np
    w, h = 384, 384
    for sid, kid in ((10, 0), (0, 0), (12, 0)):
        scene = registry.SCENES[sid]
        rad = np.linalg.norm(scene.camera_position or (0, 0, 5))
        for n in (1, 8, 32, 64):
            cams = [Camera((rad * math.sin(2 * math.pi * i / n), 0.3, rad * math.cos(2 * math.pi * i / n)), (0, 0, 0), (0, 1, 0), 60.0, w, h).params14() for i in range(n)]
            shape = _native.make_desc(sid, kid, cams[0], w, h)
            _native.render_batch(shape, np.array(cams))
            out = _native.render_batch(shape, np.array(cams))
            t0 = time.perf_counter()
            for c in cams:
                _native.render(_native.make_desc(sid, kid, c, w, h))
            seq = (time.perf_counter() - t0) * 1e3
            print(json.dumps(dict(scene=scene.name, frames=n, WxH=f"{w}x{h}", batch_ms=round(out["ms_total"], 3), per_frame_ms=round(out["ms_total"] / n, 3),
                                  mrays=round(n * w * h / out["ms_total"] / 1e3, 1), sequential_wall_ms=round(seq, 2))), flush=True)
