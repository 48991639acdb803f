#!/usr/bin/env python3
"""Developer experiments for the single-launch pipeline (RmFrameDesc.pipeline = 2) on a GPU box -- not product,
not tests.  Usage: python tools/dev_pipe.py <exp>    exp = first | teams | budgets | sizes | strategies"""
import ctypes, sys, os, json, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native, registry
from raymarch_algo_compare_amd.camera import Camera


def cam_for(scene, W, H):
    return Camera(scene.camera_position or (0.0, 0.0, 5.0), scene.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H)


def run(sid=10, kid=0, W=1920, H=1080, repeats=7, warmup=2, **tuning):
    scene = registry.SCENES[sid]
    L = _native.init()
    desc = _native.make_desc(sid, kid, cam_for(scene, W, H).params14(), W, H, **tuning)
    _native.check(L.rm_set_pass_timing(1))
    out = _native.render(desc, warmup=warmup, repeats=repeats)
    n, ms = ctypes.c_int32(0), (ctypes.c_float * 4)()
    _native.check(L.rm_get_pass_ms(None, ctypes.byref(n), ms))
    _native.check(L.rm_set_pass_timing(0))
    lp, lq = ctypes.c_float(0), ctypes.c_float(0)
    _native.check(L.rm_last_queue_marks(ctypes.byref(lp), ctypes.byref(lq)))
    lm = (ctypes.c_float * 4)()
    _native.check(L.rm_long_ray_marks(lm))
    st = out["stats"]
    r = dict(ms=round(out["timing"]["ms_median"], 3), ms_min=round(out["timing"]["ms_min"], 3),
             mrays=round(W * H / out["timing"]["ms_median"] / 1e3, 1), passes=[round(ms[i], 3) for i in range(n.value)],
             q1_last_push_pop=[round(lp.value, 2), round(lq.value, 2)], long_push_min_max_team_min_max=[round(lm[i], 2) for i in range(4)], **{k: v for k, v in tuning.items()})
    if (W, H) != (1920, 1080): r["WxH"] = f"{W}x{H}"
    if (sid, kid) != (10, 0): r["cell"] = f"{scene.name}/{registry.list_strategies()[kid]}"
    print(json.dumps(r), flush=True)
    return r


exp = sys.argv[1] if len(sys.argv) > 1 else "first"
if exp == "retune4":
    for kid in range(11):
        for tg in (96, 128, 192):
            run(kid=kid, pipeline=2, team_grid=tg)
        for tg in (128, 224):
            run(kid=kid, pipeline=2, team_grid=tg, tile_order_mode=1)
elif exp == "retune3":
    for tg in (224, 240, 256):
        run(pipeline=2, tile_order_mode=1, team_grid=tg)
        run(pipeline=2, tile_order_mode=1, team_grid=tg, suspend_after=(24, 56))
    for W, H in ((2560, 1440), (3840, 2160), (5120, 2880)):
        for tg in (32, 64, 96, 128, 192):
            run(W=W, H=H, pipeline=2, team_grid=tg)
    for W, H in ((960, 540), (1280, 720)):
        for tg in (128, 192, 224):
            run(W=W, H=H, pipeline=2, team_grid=tg)
elif exp == "retune2":
    for rep in range(2):
        for tg in (128, 144, 160, 176, 192, 224):
            run(pipeline=2, suspend_after=(16, 48), team_grid=tg)
            run(pipeline=2, suspend_after=(24, 48), team_grid=tg)
            run(pipeline=2, suspend_after=(32, 64), tile_order_mode=1, team_grid=tg)
elif exp == "retune":
    # after the guarded square root made the teams faster: budgets x team share, stateless and with the previous frame's costs
    for tg in (112, 128, 160):
        for b in ((16, 48), (16, 40), (16, 32), (24, 48), (12, 40)):
            run(pipeline=2, suspend_after=b, team_grid=tg)
    for b in ((32, 64), (16, 48), (24, 56), (32, 48)):
        run(pipeline=2, suspend_after=b, tile_order_mode=1)
        run(pipeline=2, suspend_after=b, tile_order_mode=1, team_grid=160)
elif exp == "misc":
    # host-buffer boundary (PCIe copy back included), a band-cyclic 1/8 shard of the 8K frame, the unsharded 8K frame
    import time
    scene = registry.SCENES[10]
    _native.init()
    desc = _native.make_desc(10, 0, cam_for(scene, 1920, 1080).params14(), 1920, 1080)
    _native.render(desc)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); _native.render(desc); ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"rm_render_host_wall_ms": sorted(round(t, 2) for t in ts)}), flush=True)
    run(W=7680, H=4320, rows=540, band_rows=4, band_stride=8, band_offset=0)
    run(W=7680, H=4320, rows=540, band_rows=4, band_stride=8, band_offset=5)
    run(W=7680, H=4320, rows=1080, band_rows=4, band_stride=4, band_offset=1)
    run(W=7680, H=4320, rows=2160, band_rows=4, band_stride=2, band_offset=1)
    run(W=7680, H=4320)
elif exp == "q0":
    run(pipeline=2)
    for qf in (1, 2):
        for b in ((16, 48), (32, 64), (24, 48), (32, 96), (16, 32)):
            for tg in (64, 128):
                run(pipeline=2, suspend_after=b, queue_first=qf, team_grid=tg)
    for kw in (dict(queue_refill_min=4), dict(queue_refill_min=32), dict(team_steal=2), dict(queue_retry=4), dict(queue_retry=64)):
        run(pipeline=2, suspend_after=(16, 48), queue_first=1, **kw)
elif exp == "retry":
    for tr in (2, 4, 8, 16, 32, 64):
        run(pipeline=2, team_retry=tr)
        run(pipeline=2, team_retry=tr, tile_order_mode=1)
    for tg in (160, 192):
        for tr in (8, 16, 32):
            run(pipeline=2, team_retry=tr, team_grid=tg)
elif exp == "long":
    for tr in (4, 0):
        for tom in (0, 1):
            for b in ((16, 48), (32, 64)):
                run(pipeline=2, suspend_after=b, tile_rows=tr, tile_order_mode=tom)
elif exp == "th1":
    for tr in (4, 0):
        for b in ((16, 48), (32, 48), (32, 64), (24, 40)):
            for tg in (96, 128, 160):
                run(pipeline=2, suspend_after=b, team_grid=tg, tile_rows=tr)
    run(pipeline=2, tile_order_mode=1)
    run(pipeline=2, tile_order_mode=1, suspend_after=(32, 48))
    run(pipeline=2, tile_order_mode=2)
elif exp == "marks":
    for tg in (64, 128, 192):
        for b in ((16, 48), (32, 64), (32, 96), (32, 128)):
            run(pipeline=2, suspend_after=b, team_grid=tg)
elif exp == "share":
    run(pipeline=1, tile_order_mode=2)
    for tg in (64, 96, 128, 160, 192, 256):
        for b in ((16, 48), (32, 48), (32, 64)):
            run(pipeline=2, suspend_after=b, team_grid=tg)
    run(pipeline=2)
    run(pipeline=2, tile_order_mode=1)
elif exp == "cheap":
    for sid in (0, 2, 9, 12, 1, 13, 16, 5, 3):
        for kid in (0, 4):
            run(sid=sid, kid=kid)
            for ap in (8, 16, 32, 64):
                run(sid=sid, kid=kid, age_priority=ap)
elif exp == "tune":
    run(pipeline=1, tile_order_mode=2)
    base = dict(pipeline=2, tile_order_mode=2, team_grid=128)
    for b in ((16, 48), (32, 48), (24, 56), (32, 64)):
        run(suspend_after=b, **base)
        for ap in (16, 24, 32):
            run(suspend_after=b, age_priority=ap, **base)
    for tg in (96, 112, 144, 160):
        run(suspend_after=(16, 48), **dict(base, team_grid=tg))
    for kw in (dict(team_retry=1), dict(team_retry=2), dict(team_retry=8), dict(refill_min=4), dict(refill_min=16), dict(refill_min=24)):
        run(suspend_after=(16, 48), **base, **kw)
    run(suspend_after=(16, 48), **dict(base, tile_order_mode=1))
    run(suspend_after=(16, 48), **dict(base, tile_order_mode=0))
elif exp == "cu":
    run(pipeline=1, tile_order_mode=2)
    for tg in (16, 32, 48, 64, 96):
        for b in ((32, 64), (32, 48), (16, 32), (24, 40), (32, 128)):
            run(pipeline=2, tile_order_mode=2, suspend_after=b, team_grid=tg)
    for b in ((32, 64), (32, 48)):
        run(pipeline=2, tile_order_mode=1, suspend_after=b, team_grid=48)
        run(pipeline=2, tile_order_mode=0, suspend_after=b, team_grid=48)
elif exp == "poll":
    run(pipeline=1, tile_order_mode=2)
    for tg in (32, 64, 128, 256):
        for b in ((32, 64), (32, 48), (16, 48), (32, 128)):
            run(pipeline=2, tile_order_mode=2, suspend_after=b, team_grid=tg)
    for b in ((32, 64), (32, 48)):
        run(pipeline=2, tile_order_mode=1, suspend_after=b, team_grid=128)
        run(pipeline=2, tile_order_mode=0, suspend_after=b, team_grid=128)
elif exp == "detach":
    run(pipeline=1)
    run(pipeline=1, tile_order_mode=2)
    for b in ((32, 64), (32, 128), (16, 64), (32, 48), (32, 96), (24, 48)):
        for tg in (64, 128, 192):
            run(pipeline=2, tile_order_mode=2, suspend_after=b, team_grid=tg)
    run(pipeline=2, suspend_after=(32, 64), team_grid=128)
    run(pipeline=2, tile_order_mode=1, suspend_after=(32, 64), team_grid=128)
    for tr in (1, 2, 8):
        run(pipeline=2, tile_order_mode=2, suspend_after=(32, 64), team_grid=128, team_retry=tr)
elif exp == "first":
    run(pipeline=1)
    run(pipeline=1, tile_order_mode=2)
    run(pipeline=2)
    run(pipeline=2, tile_order_mode=2)
    for tg in (32, 64, 96, 128, 192, 256):
        run(pipeline=2, tile_order_mode=2, team_grid=tg)
    for b in ((32, 64), (32, 96), (32, 128), (16, 64), (24, 64), (48, 96), (32, 48)):
        run(pipeline=2, tile_order_mode=2, suspend_after=b)
        run(pipeline=2, tile_order_mode=2, suspend_after=b, team_grid=64)
    for qf, ts in itertools.product((1, 2), (1, 2)):
        run(pipeline=2, tile_order_mode=2, queue_first=qf, team_steal=ts)
        run(pipeline=2, tile_order_mode=2, queue_first=qf, team_steal=ts, suspend_after=(32, 64))
elif exp == "teams":
    for tg in (16, 32, 48, 64, 96, 128, 160, 192, 256, 320):
        for b in ((32, 64), (32, 128)):
            run(pipeline=2, tile_order_mode=2, team_grid=tg, suspend_after=b)
elif exp == "knobs":
    for kw in (dict(queue_refill_min=8), dict(queue_refill_min=32), dict(queue_retry=4), dict(queue_retry=64), dict(team_retry=1),
               dict(team_retry=16), dict(refill_min=4), dict(refill_min=16)):
        run(pipeline=2, tile_order_mode=2, suspend_after=(32, 64), **kw)
elif exp == "sizes":
    for W, H in ((640, 360), (960, 540), (2560, 1440), (3840, 2160), (5120, 2880), (7680, 4320)):
        run(W=W, H=H, repeats=3, warmup=1, pipeline=1)
        run(W=W, H=H, repeats=3, warmup=1, pipeline=1, suspend_after=(-1, -1))
        for b in ((16, 48), (32, 64), (48, 96)):
            run(W=W, H=H, repeats=3, warmup=1, pipeline=2, tile_order_mode=2, suspend_after=b)
elif exp == "strategies":
    for kid in range(11):
        run(kid=kid, repeats=3, warmup=1, pipeline=1)
        for b in ((16, 48), (8, 24), (32, 64)):
            run(kid=kid, repeats=3, warmup=1, pipeline=2, tile_order_mode=2, suspend_after=b)
