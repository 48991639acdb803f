#!/usr/bin/env python3
"""Randomised parity campaign on a GPU box: random (scene, strategy), frame size, camera, march configuration and
SCHEDULE knobs (evaluation mode, suspension budgets, resume mode, refill threshold, grid size, tile order, row
shards, one launch per pass or the single-launch pipeline with its team / queue knobs) against the CPU oracle, bit for bit (iterations, hits, raw fp64 t, final_sdf, frame totals).
usage: python tests/fuzz_parity.py [cases] [seed]      (test infrastructure, lives under tests/: it loads oracle/)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle                                         # noqa: E402
from raymarch_algo_compare_amd import _native, registry           # noqa: E402
from raymarch_algo_compare_amd.camera import Camera               # noqa: E402


def one_case(rng):
    sid = int(rng.choice([10, 10, 10, 14, 15, 1, 13, 12, 0, 9, 16, int(rng.integers(0, 20))]))
    kid = int(rng.integers(0, 13))                     # 11, 12: the shader-only strategies (oracle = the same shader text)
    sc = registry.SCENES[sid]
    w, h = int(rng.integers(16, 201)), int(rng.integers(16, 141))
    base = np.array(sc.camera_position or (0.0, 0.0, 5.0))
    pos = tuple(float(v) for v in base + rng.normal(size=3) * 0.4)
    tgt = tuple(float(v) for v in np.array(sc.camera_target or (0.0, 0.0, 0.0)) + rng.normal(size=3) * 0.2)
    cam = Camera(pos, tgt, (0.0, 1.0, 0.0), float(rng.uniform(40, 80)), w, h).params14()
    mi = int(rng.choice([8, 17, 33, 64, 130, 200, 512]))
    thr = float(rng.choice([1e-3, 1e-4, 1e-5]))
    far = float(rng.choice([100.0, 30.0, 9.0]))
    lip = (sc.lipschitz or 1.0) if kid == 10 else 1.0
    row0 = 0 if rng.random() < 0.7 else int(rng.integers(0, h // 8)) * 4
    rows = h - row0 if rng.random() < 0.7 else max(4, int(rng.integers(1, (h - row0) // 4 + 1)) * 4)
    rows = min(rows, h - row0)
    b0 = int(rng.choice([-1, 0, 1, 2, 5, 9, 16, 40]))
    b1 = 0 if b0 <= 0 else int(rng.choice([0, 0, b0 + 1, b0 * 3, 100]))
    sched = dict(eval_mode=int(rng.integers(0, 3)), suspend_after=(b0, b1), resume_mode=int(rng.integers(0, 4)),
                 refill_min=int(rng.choice([0, 1, 8, 33, 64])), grid_waves=int(rng.choice([0, 0, 4, 64, 1000])),
                 tile_order_mode=int(rng.choice([0, 0, 1, 2, 3, 4])), resume_grid=int(rng.choice([0, 0, 1, 7, 300])),
                 # launch structure: passes / single launch, and the single launch's knobs
                 pipeline=int(rng.choice([0, 1, 2, 2])), team_grid=int(rng.choice([0, 0, 1, 3, 40, 700])),
                 queue_first=int(rng.integers(0, 4)), team_steal=int(rng.integers(0, 3)),
                 queue_refill_min=int(rng.choice([0, 1, 16, 64])), queue_retry=int(rng.choice([0, 1, 5, 100])),
                 team_retry=int(rng.choice([0, 1, 3, 50])), age_priority=int(rng.choice([0, 0, 1, 16, 40])),
                 late_teams=int(rng.choice([0, 0, 1, 3, 40])), exit_backlog=int(rng.choice([0, 1, 8, 500])),
                 keep_busy=int(rng.choice([0, 0, -1, 1, 300])), early_handover=int(rng.choice([0, 0, -1, 1, 7, 60])),
                 early_trips=int(rng.choice([0, 0, 1, 3, 6, 8])))
    prm = None
    if rng.random() < 0.3:                             # strategy parameters, the shader-only uniforms included
        prm = dict(omega=float(rng.choice([1.0, 1.2, 1.5, 1.9])), step_scale=float(rng.choice([1.0, 0.6, 0.5])),
                   dense_min_step=float(rng.choice([1e-4, 0.002, 0.02])), beta=float(rng.choice([0.3, 0.6])),
                   overstep_bisection_steps=int(rng.choice([16, 5])), margin=float(rng.choice([0.05, 0.15])))
    # one case in four takes the PRODUCT path (march.full = 0: no final_sdf, the strategy's result record is dead until the ray
    # finishes and is not carried through the queues): depth instead of the raw fp64 t
    full = bool(rng.random() < 0.75)
    desc = _native.make_desc(sid, kid, cam, w, h, row0, rows, mi, thr, far, lip, full, params=prm, **sched)
    out = _native.render(desc, want_t_raw=full, want_final_sdf=full)
    ref = oracle.render(sid, kid, cam, w, h, row0=row0, rows=rows, max_iterations=mi, hit_threshold=thr, max_distance=far, lipschitz=lip, params=prm)
    if full:
        same_t = ((out["t_raw"].view(np.uint64) == ref.t.view(np.uint64)).all()
                  and (out["final_sdf"].view(np.uint64) == ref.final_sdf.view(np.uint64)).all())
    else:
        same_t = (out["depth"].view(np.uint32) == np.where(ref.hit > 0, ref.t, 0.0).astype(np.float32).view(np.uint32)).all()   # types.py:93
    ok = ((out["iters"] == ref.iters).all() and (out["hit"] == ref.hit).all()
          and same_t
          and out["stats"]["sum_iters"] == int(ref.iters.sum()) and out["stats"]["hit_count"] == int(ref.hit.sum())
          and out["stats"]["total_rays"] == ref.iters.size
          and (out["stats"]["iter_hist"] == np.bincount(ref.iters.ravel(), minlength=len(out["stats"]["iter_hist"]))).all())
    return ok, dict(sid=sid, kid=kid, w=w, h=h, row0=row0, rows=rows, mi=mi, thr=thr, far=far, pos=pos, params=prm, full=full, **sched)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n):
        ok, info = one_case(rng)
        if not ok:
            bad += 1
            print("MISMATCH", json.dumps(info), flush=True)
        if (i + 1) % 50 == 0:
            print(f"{i + 1} cases, {bad} mismatches", flush=True)
    print(f"done: {n} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
