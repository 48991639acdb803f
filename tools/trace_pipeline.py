#!/usr/bin/env python3
"""Time line of a single-launch Mandelbulb frame from the development trace (rm_debug_set_trace / rm_debug_get_trace).

For every ray a wavefront team finished: when its ray started, was struck from its tile, entered queue 1, left it and
ended, how many evaluations it did with the team and how many rays the team carried.  Prints the aggregate picture the
schedule is tuned by and writes the raw join to gpurun_out/ as .npz.

  python tools/trace_pipeline.py [--width 1920 --height 1080 --strategy 0 --frames 3] [desc knobs: --team-grid N ...]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--strategy", type=int, default=0)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--team-grid", type=int, default=0)
    ap.add_argument("--suspend", type=int, nargs=2, default=(0, 0))
    ap.add_argument("--tile-order-mode", type=int, default=0)
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--tag", default="default")
    ap.add_argument("--late-teams", type=int, default=0)
    ap.add_argument("--exit-backlog", type=int, default=0)
    ap.add_argument("--brief", action="store_true")
    args = ap.parse_args()
    from raymarch_algo_compare_amd import _native, registry
    from raymarch_algo_compare_amd.camera import Camera
    L = _native.init(0)
    W, H = args.width, args.height
    sc = registry.SCENES[10]
    cam = Camera(sc.camera_position, sc.camera_target, (0.0, 1.0, 0.0), 60.0, W, H).params14()
    desc = _native.make_desc(10, args.strategy, cam, W, H, team_grid=args.team_grid, suspend_after=tuple(args.suspend),
                             tile_order_mode=args.tile_order_mode, tile_rows=args.tile_rows, late_teams=args.late_teams,
                             exit_backlog=args.exit_backlog)
    vp = ctypes.c_void_p
    p = [vp(), vp(), vp()]
    _native.check(L.rm_alloc_frame(W, H, *[ctypes.byref(q) for q in p]))
    _native.check(L.rm_debug_set_trace(1))
    _native.check(L.rm_set_pass_timing(1))
    tm = _native.RmTiming()
    for f in range(args.frames):
        _native.check(L.rm_render_device(ctypes.byref(desc), p[0], p[1], p[2], None, None))
        npass, pms = ctypes.c_int32(0), (ctypes.c_float * 4)()
        _native.check(L.rm_get_pass_ms(None, ctypes.byref(npass), pms))
    spans = [float(pms[i]) for i in range(npass.value)]
    iters = np.empty((H, W), np.int32)
    _native.check(L.rm_copy_frame_to_host(W, H, p[0], p[1], p[2], None, iters.ctypes.data_as(vp), None))
    rec = np.empty((1 << 20, 8), np.uint32)
    n = ctypes.c_int64(0)
    start, detach = np.empty(W * H, np.uint32), np.empty(W * H, np.uint32)
    t0 = ctypes.c_uint32(0)
    _native.check(L.rm_debug_get_trace(rec.ctypes.data_as(vp), len(rec), ctypes.byref(n), start.ctypes.data_as(vp),
                                       detach.ctypes.data_as(vp), W * H, ctypes.byref(t0)))
    _native.check(L.rm_debug_set_trace(0))
    _native.check(L.rm_set_pass_timing(0))
    rec = rec[: n.value]
    us = 0.01                                              # 100 MHz ticks -> microseconds
    gi = rec[:, 0].astype(np.int64)
    it = rec[:, 1].astype(np.int64)
    push, pop, end = rec[:, 2] * us, rec[:, 3] * us, rec[:, 4] * us
    nev_pop, nev_end = rec[:, 5].astype(np.int64), rec[:, 6].astype(np.int64)
    team, live = rec[:, 7] & 0xffff, rec[:, 7] >> 16
    t_start = ((start[gi] - np.uint32(t0.value)).astype(np.uint32)) * us
    t_det = ((detach[gi] - np.uint32(t0.value)).astype(np.uint32)) * us
    out = {"tag": args.tag, "frame": f"{W}x{H}", "strategy": args.strategy, "spans_ms": spans, "frame_ms": sum(spans),
           "rays_through_teams": int(n.value), "mean_iters": float(iters.mean())}

    def q(x, ps=(0, 10, 50, 90, 99, 100)):
        return [round(float(v), 1) for v in np.percentile(x, ps)] if len(x) else []

    evals_team = np.maximum(nev_end - nev_pop, 1)
    pace = (end - pop) / evals_team                        # us per evaluation with the team
    out["all_rays"] = {"queue_wait_us p0/10/50/90/99/100": q(pop - push), "start->push_us": q(push - t_start),
                       "start->detach_us": q(t_det - t_start), "team_pace_us_per_eval": q(pace), "evals_with_team": q(evals_team),
                       "push_time_ms": q(push / 1000.0)}
    lg = it >= 500
    out["long_rays(>=500)"] = {"count": int(lg.sum()), "start_ms": q(t_start[lg] / 1000), "push_ms": q(push[lg] / 1000),
                               "queue_wait_us": q((pop - push)[lg]), "start->push_us": q((push - t_start)[lg]),
                               "team_ms": q((end - pop)[lg] / 1000), "team_pace_us_per_eval": q(pace[lg]),
                               "end_ms": q(end[lg] / 1000), "live_rays_in_team_at_end": q(live[lg])}
    # the ray that ends last: its whole history
    k = int(np.argmax(end))
    out["last_ray"] = {"pixel": [int(gi[k] % W), int(gi[k] // W)], "iters": int(it[k]), "start_ms": float(t_start[k] / 1000),
                       "detach_ms": float(t_det[k] / 1000), "push_ms": float(push[k] / 1000), "pop_ms": float(pop[k] / 1000),
                       "end_ms": float(end[k] / 1000), "evals_with_team": int(evals_team[k]), "pace_us": float(pace[k])}
    # queue wait and team pace by push time (1 ms bins): when do the teams fall behind?
    bins = {}
    for b in range(int(push.max() // 1000) + 1):
        m = (push >= b * 1000) & (push < (b + 1) * 1000)
        if m.any():
            bins[f"{b}-{b + 1}ms"] = {"pushed": int(m.sum()), "wait_us_p50/p99": [round(float(np.percentile((pop - push)[m], 50)), 1),
                                                                                   round(float(np.percentile((pop - push)[m], 99)), 1)],
                                      "pace_p50": round(float(np.percentile(pace[m], 50)), 1)}
    out["by_push_time"] = bins
    # pace against the number of rays the team still carries when the ray ends
    byl = {}
    for lo, hi in ((1, 1), (2, 4), (5, 16), (17, 64)):
        m = (live >= lo) & (live <= hi) & (evals_team >= 20)
        if m.any():
            byl[f"live {lo}-{hi}"] = {"rays": int(m.sum()), "pace_p50": round(float(np.percentile(pace[m], 50)), 1)}
    out["pace_by_team_load_at_end"] = byl
    if args.brief:
        lr = out["long_rays(>=500)"]
        print(json.dumps({"tag": args.tag, "frame_ms": round(out["frame_ms"], 3), "spans_ms": [round(v, 2) for v in spans],
                          "teams_rays": out["rays_through_teams"], "long_push_ms": lr["push_ms"], "long_wait_us": lr["queue_wait_us"],
                          "long_start_push_us": lr["start->push_us"], "long_team_ms": lr["team_ms"], "long_end_ms": lr["end_ms"],
                          "pace_all": out["all_rays"]["team_pace_us_per_eval"], "last_ray": out["last_ray"]}))
    else:
        print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"trace_{args.tag}.npz"), rec=rec, start=start, detach=detach,
                        t0=np.uint32(t0.value), iters=iters)
    _native.check(L.rm_free_frame(*p))


if __name__ == "__main__":
    main()
