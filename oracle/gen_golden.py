#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (imports /root/reference, which never travels
to the GPU box).  The outputs are data only -- inputs (camera, config, ids) and
expected outputs (iteration maps, hit masks, depths, stats scalars).

For every (scene, strategy) the scene/strategy/camera/Lipschitz wiring follows
run_once (reference main.py:39-74) with a FRESH RenderConfig, and the pixel loop
follows MetricsCollector.benchmark_strategy (metrics/collector.py:40-44); the
stats scalars come from the reference's own RayMarchStats.compute (core/types.py:77-137).

Layout of each frames_*.npz (keys prefixed "s{scene_id}_k{strategy_id}_"):
  iters   int16 (H,W)         exact iteration counts (max possible 521)
  hitbits uint8 packbits(H*W) exact hit mask
  t_hit   float64 (n_hits,)   raw t of the hit rays, row-major order
  sha_t / sha_fs              sha256 over the little-endian float64 bytes of t / final_sdf
                              of ALL rays (pins the oracle bit-for-bit without storing them)
  cam     float64 (14,)       position, forward, right, up, half_width, half_height
  meta    float64 (8,)        W, H, row0, rows, max_iterations, hit_threshold, max_distance, lipschitz

Usage:  python oracle/gen_golden.py [--only frames64|frames160|rows1080|sdf|stats|leak|leakseq|params|schema|viewpoints|evals|analytic]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import random
import sys
import time

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from raymarching_benchmark.config import MarchConfig, RenderConfig  # noqa: E402
from raymarching_benchmark.core.camera import Camera  # noqa: E402
from raymarching_benchmark.core.types import RayMarchStats  # noqa: E402
from raymarching_benchmark.core.vec3 import Vec3  # noqa: E402
from raymarching_benchmark.scenes.catalog import get_all_scenes  # noqa: E402
from raymarching_benchmark.strategies import STRATEGIES  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SCENES = get_all_scenes()
STRAT_KEYS = list(STRATEGIES.keys())


def wire(scene_id: int, strat_id: int, width: int, height: int, render: RenderConfig | None = None):
    """run_once's wiring (main.py:39-74) for registry ids."""
    scene = get_all_scenes()[scene_id]
    strategy = STRATEGIES[STRAT_KEYS[strat_id]]()
    render = render or RenderConfig(width=width, height=height)
    sug = scene.suggested_camera()
    if sug:
        render.camera_position = sug.camera_position
        render.camera_target = sug.camera_target
        render.camera_up = sug.camera_up
        render.fov_degrees = sug.fov_degrees
    lipschitz = 1.0
    if hasattr(strategy, "lipschitz"):
        bound = scene.known_lipschitz_bound()
        if bound is not None:
            strategy.lipschitz = bound
        lipschitz = float(strategy.lipschitz)
    cam = Camera(position=Vec3(*render.camera_position), target=Vec3(*render.camera_target),
                 up=Vec3(*render.camera_up), fov_degrees=render.fov_degrees,
                 width=render.width, height=render.height)
    return scene, strategy, cam, lipschitz, render


def cam14(cam: Camera) -> np.ndarray:
    return np.array([*cam.position.to_tuple(), *cam.forward.to_tuple(), *cam.right.to_tuple(),
                     *cam.up.to_tuple(), cam.half_width, cam.half_height], dtype=np.float64)


def march_rows(scene, strategy, cam, mc, row0, rows):
    results = []
    for py in range(row0, row0 + rows):
        for px in range(cam.width):
            results.append(strategy.march(cam.get_ray(px, py), scene.sdf, mc))
    return results


def pack(prefix, results, cam, mc, lipschitz, row0, rows, store):
    W = cam.width
    iters = np.array([r.iterations for r in results], dtype=np.int32).reshape(rows, W)
    hit = np.array([bool(r.hit) for r in results], dtype=bool)
    t = np.array([float(r.t) for r in results], dtype="<f8")
    fs = np.array([float(r.final_sdf) for r in results], dtype="<f8")
    assert iters.max() < 32767
    store[prefix + "iters"] = iters.astype(np.int16)
    store[prefix + "hitbits"] = np.packbits(hit)
    store[prefix + "t_hit"] = t[hit]
    store[prefix + "sha_t"] = np.frombuffer(hashlib.sha256(t.tobytes()).digest(), dtype=np.uint8)
    store[prefix + "sha_fs"] = np.frombuffer(hashlib.sha256(fs.tobytes()).digest(), dtype=np.uint8)
    store[prefix + "cam"] = cam14(cam)
    store[prefix + "meta"] = np.array([W, cam.height, row0, rows, mc.max_iterations, mc.hit_threshold,
                                       mc.max_distance, lipschitz], dtype=np.float64)


def stats_dict(scene, strategy, results, W, H):
    s = RayMarchStats(strategy_name=strategy.short_name, scene_name=scene.name)
    s.compute(results, W, H, 1.0)
    return {
        "strategy": s.strategy_name, "scene": s.scene_name, "total_rays": s.total_rays,
        "hit_count": s.hit_count, "miss_count": s.miss_count, "sample_count": s.sample_count,
        "iteration_mean": s.iteration_mean, "iteration_median": s.iteration_median,
        "iteration_std": s.iteration_std, "iteration_min": s.iteration_min,
        "iteration_max": s.iteration_max, "iteration_p95": s.iteration_p95,
        "iteration_p99": s.iteration_p99, "accuracy_mean": s.accuracy_mean,
        "accuracy_max": s.accuracy_max, "accuracy_std": s.accuracy_std, "hit_rate": s.hit_rate,
        "warp_divergence_proxy": s.warp_divergence_proxy,
        "depth_sum": float(s.depth_map.sum()),
    }


def gen_frames(tag, W, H, pairs, mc=None):
    mc = mc or MarchConfig()
    store, stats = {}, {}
    t0 = time.time()
    for sid, kid in pairs:
        scene, strategy, cam, lip, _ = wire(sid, kid, W, H)
        res = march_rows(scene, strategy, cam, mc, 0, H)
        pack(f"s{sid}_k{kid}_", res, cam, mc, lip, 0, H, store)
        stats[f"s{sid}_k{kid}"] = stats_dict(scene, strategy, res, W, H)
        print(f"  [{tag}] {scene.name} / {strategy.short_name}: hits={stats[f's{sid}_k{kid}']['hit_count']} "
              f"sum_iters={stats[f's{sid}_k{kid}']['sample_count']}  ({time.time() - t0:.0f}s)", flush=True)
    np.savez_compressed(os.path.join(OUT, f"frames_{tag}.npz"), **store)
    with open(os.path.join(OUT, f"stats_{tag}.json"), "w", encoding="utf-8") as f:
        json.dump(stats, f, indent=1, ensure_ascii=False)


def gen_rows1080(pairs, row0=536, rows=8):
    """Row-block samples of the 1920x1080 frame: pins full-resolution indexing."""
    mc = MarchConfig()
    store = {}
    for sid, kid in pairs:
        scene, strategy, cam, lip, _ = wire(sid, kid, 1920, 1080)
        res = march_rows(scene, strategy, cam, mc, row0, rows)
        pack(f"s{sid}_k{kid}_", res, cam, mc, lip, row0, rows, store)
        print(f"  [rows1080] {scene.name} / {strategy.short_name}", flush=True)
    np.savez_compressed(os.path.join(OUT, "frames_rows1080.npz"), **store)


def gen_leak():
    """The CLI reuses one RenderConfig across scenes (main.py:167,206): a scene without a
    suggested camera inherits the previous scene's.  Pin Cube rendered right after
    Grazing Plane (non-default, off-axis camera)."""
    mc = MarchConfig()
    store = {}
    rc = RenderConfig(width=64, height=48)
    wire(1, 0, 64, 48, rc)  # Grazing Plane mutates rc
    scene, strategy, cam, lip, _ = wire(2, 0, 64, 48, rc)
    res = march_rows(scene, strategy, cam, mc, 0, 48)
    pack("s2_k0_", res, cam, mc, lip, 0, 48, store)
    np.savez_compressed(os.path.join(OUT, "frames_leak.npz"), **store)


def leak_sequence(W, H, upto):
    """Walk `--scene all` in catalogue order with ONE shared RenderConfig, as cli() does (main.py:167,206):
    returns the RenderConfig as scene `upto` finds it (every earlier scene's suggestion applied)."""
    rc = RenderConfig(width=W, height=H)
    for sid in range(upto):
        wire(sid, 0, W, H, rc)
    return rc


# (scene, strategy) cells whose camera is NOT the scene's own in `--scene all --strategy all` order
# (scenes 2..9 see Grazing Plane's camera, 11 sees Mandelbulb's)
LEAK_CELLS = [(2, 10), (3, 4), (5, 9), (6, 3), (8, 10), (9, 6), (11, 0), (11, 10)]
LEAK_ROWS_1080 = [(2, 0), (9, 0), (11, 10)]


def gen_leakseq():
    """Cells of the CLI's leaked-camera sequence: whole frames at 64x48 and rows 536..543 of 1920x1080."""
    mc = MarchConfig()
    store, stats = {}, {}
    for sid, kid in LEAK_CELLS:
        rc = leak_sequence(64, 48, sid)
        scene, strategy, cam, lip, _ = wire(sid, kid, 64, 48, rc)
        res = march_rows(scene, strategy, cam, mc, 0, 48)
        pack(f"s{sid}_k{kid}_", res, cam, mc, lip, 0, 48, store)
        stats[f"s{sid}_k{kid}"] = stats_dict(scene, strategy, res, 64, 48)
        print(f"  [leakseq] {scene.name} / {strategy.short_name}: camera {rc.camera_position} -> {rc.camera_target}, "
              f"hits {stats[f's{sid}_k{kid}']['hit_count']}", flush=True)
    np.savez_compressed(os.path.join(OUT, "frames_leakseq.npz"), **store)
    with open(os.path.join(OUT, "stats_leakseq.json"), "w", encoding="utf-8") as f:
        json.dump(stats, f, indent=1, ensure_ascii=False)
    store = {}
    for sid, kid in LEAK_ROWS_1080:
        rc = leak_sequence(1920, 1080, sid)
        scene, strategy, cam, lip, _ = wire(sid, kid, 1920, 1080, rc)
        res = march_rows(scene, strategy, cam, mc, 536, 8)
        pack(f"s{sid}_k{kid}_", res, cam, mc, lip, 536, 8, store)
        print(f"  [leakrows1080] {scene.name} / {strategy.short_name}", flush=True)
    np.savez_compressed(os.path.join(OUT, "frames_leakrows1080.npz"), **store)


def gen_schema():
    """Header row and index column of the nine matrix_*.csv files the reference ships under example/
    (data files; BASELINE config 4 asks for CSVs identical in schema to these)."""
    import csv
    ex = os.path.join(REF, "example")
    out = {}
    for fn in sorted(os.listdir(ex)):
        if fn.startswith("matrix_") and fn.endswith(".csv"):
            with open(os.path.join(ex, fn), encoding="utf-8", newline="") as f:
                rows = list(csv.reader(f))
            out[fn] = {"header": rows[0], "index": [r[0] for r in rows[1:]]}
    with open(os.path.join(OUT, "example_matrix_schema.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("example_matrix_schema.json:", len(out), "files")


# ---- non-default strategy parameters ---------------------------------------------------------------
# Parameter names / order of RmStrategyParams (include/rm_hip.h) = rmo_cfg (oracle/rm_oracle.c).
PARAM_ORDER = ["omega", "ar_omega_min", "ar_omega_max", "ar_smoothing", "ar_growth_rate", "ar_decay_rate", "beta",
               "overstep_min_step", "hybrid_stuck_step_ratio", "hybrid_min_step", "margin", "ar_omega_init",
               "overstep_bisection_steps", "hybrid_stuck_threshold", "segment_bisection_steps", "revaa_bisection_steps"]
PARAM_DEFAULTS = dict(omega=1.2, ar_omega_min=1.0, ar_omega_max=2.0, ar_smoothing=0.7, ar_growth_rate=1.05,
                      ar_decay_rate=0.7, beta=0.3, overstep_min_step=0.01, hybrid_stuck_step_ratio=0.001,
                      hybrid_min_step=0.005, margin=0.05, ar_omega_init=1.2, overstep_bisection_steps=16, hybrid_stuck_threshold=5,
                      segment_bisection_steps=8, revaa_bisection_steps=8)


def with_literal(cls, old, new):
    """The reference class with ONE literal of its march() replaced: the method is rebuilt from the reference's
    own code object with that constant swapped (types.CodeType.replace), nothing is re-typed.  Used for the four
    parameters the CPU strategies hold as literals (skipping_spheres.py:30 `margin = 0.05`, auto_relaxed.py:41
    `omega = 1.2`, segment_tracing.py:79 and rev_affine.py:70 `range(8)`); the GLSL seam exposes `margin` as a uniform (gpu/runner.py:115)."""
    import types
    code = cls.march.__code__
    assert sum(1 for c in code.co_consts if type(c) is type(old) and c == old) == 1, (cls.__name__, old, code.co_consts)
    consts = tuple(new if (type(c) is type(old) and c == old) else c for c in code.co_consts)
    fn = types.FunctionType(code.replace(co_consts=consts), cls.march.__globals__, "march", cls.march.__defaults__,
                            cls.march.__closure__)
    return type(cls.__name__ + "Lit", (cls,), {"march": fn})


# (strategy id, constructor kwargs / literal swaps, RmStrategyParams overrides): >= 3 non-default settings per
# tunable strategy; omega / margin values are the reference's own grid (param_grid.py:20-27)
PARAM_CASES = (
    [(1, dict(omega=w), None, dict(omega=w)) for w in (1.4, 1.6, 1.8)] +
    [(2, dict(omega_min=a, omega_max=b, smoothing=c, growth_rate=d, decay_rate=e), None,
      dict(ar_omega_min=a, ar_omega_max=b, ar_smoothing=c, ar_growth_rate=d, ar_decay_rate=e))
     for a, b, c, d, e in ((1.0, 1.6, 0.5, 1.1, 0.5), (1.1, 2.5, 0.9, 1.02, 0.8), (1.0, 3.0, 0.3, 1.2, 0.9))] +
    [(3, dict(beta=b), None, dict(beta=b)) for b in (0.1, 0.5, 0.9)] +
    [(6, dict(min_step_factor=m, bisection_steps=n), None, dict(overstep_min_step=m, overstep_bisection_steps=n))
     for m, n in ((0.02, 8), (0.005, 24), (0.05, 4), (0.01, 0))] +
    [(9, dict(stuck_threshold=k, stuck_step_ratio=r, min_step_factor=m), None,
      dict(hybrid_stuck_threshold=k, hybrid_stuck_step_ratio=r, hybrid_min_step=m))
     for k, r, m in ((3, 0.01, 0.01), (8, 0.0005, 0.002), (2, 0.005, 0.02))] +
    [(7, {}, (0.05, m), dict(margin=m)) for m in (0.02, 0.1, 0.2)] +
    [(2, {}, (1.2, w), dict(ar_omega_init=w)) for w in (1.4, 1.6, 1.8)] +
    [(10, {}, (8, n), dict(segment_bisection_steps=n)) for n in (3, 12, 0)] +
    [(8, {}, (8, n), dict(revaa_bisection_steps=n)) for n in (3, 12, 0)]
)
PARAM_SCENES = (0, 2, 9, 10, 12)


def gen_params(W=48, H=36):
    """Frames marched by the reference's strategy classes constructed with NON-default arguments."""
    mc = MarchConfig()
    store = {}
    n = 0
    for kid, kwargs, literal, overrides in PARAM_CASES:
        cls = STRATEGIES[STRAT_KEYS[kid]]
        if literal is not None:
            cls = with_literal(cls, *literal)
        for sid in PARAM_SCENES:
            scene, _, cam, _, _ = wire(sid, kid, W, H)
            strategy = cls(**kwargs)
            lip = 1.0
            if hasattr(strategy, "lipschitz"):                            # main.py:58-61
                bound = scene.known_lipschitz_bound()
                if bound is not None:
                    strategy.lipschitz = bound
                lip = float(strategy.lipschitz)
            res = march_rows(scene, strategy, cam, mc, 0, H)
            pre = f"c{n}_"
            pack(pre, res, cam, mc, lip, 0, H, store)
            prm = dict(PARAM_DEFAULTS, **overrides)
            store[pre + "ids"] = np.array([sid, kid], dtype=np.int32)
            store[pre + "prm"] = np.array([float(prm[k]) for k in PARAM_ORDER], dtype=np.float64)
            print(f"  [params] case {n}: {scene.name} / {strategy.short_name} {overrides}: hits "
                  f"{sum(1 for r in res if r.hit)} iterations {sum(r.iterations for r in res)}", flush=True)
            n += 1
    store["ncases"] = np.array([n], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, f"frames_params_{W}x{H}.npz"), **store)


def gen_sdf(n=2000):
    """Per-scene SDF values at seeded random points (same generator idea as the
    reference's tests/test_scene_parity.py:88-102)."""
    rng = random.Random(1234)
    pts = np.array([[rng.uniform(-3.5, 3.5) for _ in range(3)] for _ in range(n)], dtype=np.float64)
    store = {"pts": pts}
    for sid, scene in enumerate(SCENES):
        store[f"s{sid}"] = np.array([scene.sdf(Vec3(*p)) for p in pts], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "sdf_points.npz"), **store)


def gen_evals(W=48, H=36, scene_ids=(0, 1, 9, 10, 12, 15)):
    """SDF evaluations per ray: strategy.march is given a counting wrapper of scene.sdf, so the count is what
    the reference's own march() calls (the quantity its GLSL backend exposes as g_evals, scenes.glsl:10-12).
    All 11 strategies on a few scenes; iterations ride along to tie the two together."""
    mc = MarchConfig()
    store = {}
    for sid in scene_ids:
        for kid in range(len(STRAT_KEYS)):
            scene, strategy, cam, lip, _ = wire(sid, kid, W, H)
            calls = [0]

            def counted(p, _sdf=scene.sdf, _c=calls):
                _c[0] += 1
                return _sdf(p)
            evals, iters = [], []
            for py in range(H):
                for px in range(W):
                    calls[0] = 0
                    r = strategy.march(cam.get_ray(px, py), counted, mc)
                    evals.append(calls[0])
                    iters.append(r.iterations)
            pre = f"s{sid}_k{kid}_"
            store[pre + "evals"] = np.array(evals, dtype=np.int16).reshape(H, W)
            store[pre + "iters"] = np.array(iters, dtype=np.int16).reshape(H, W)
            store[pre + "cam"] = cam14(cam)
            store[pre + "meta"] = np.array([W, H, 0, H, mc.max_iterations, mc.hit_threshold, mc.max_distance, lip], dtype=np.float64)
            print(f"  [evals] {scene.name} / {strategy.short_name}: evals {sum(evals)} iterations {sum(iters)}", flush=True)
    np.savez_compressed(os.path.join(OUT, f"evals_{W}x{H}.npz"), **store)


def gen_analytic(W=80, H=60):
    """Closed-form depth / hit / normal of the reference's gpu/analytic.py (:74-208) for its four analytic scenes,
    on the CPU camera's pixel-centre rays (its intersect_* functions take ray arrays; the module is loaded by
    path because the gpu package's __init__ needs moderngl)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_analytic", os.path.join(REF, "raymarching_benchmark", "gpu", "analytic.py"))
    ref = importlib.util.module_from_spec(spec)
    ref.__package__ = "raymarching_benchmark.gpu"
    spec.loader.exec_module(ref)
    store = {}
    for sid in (0, 1, 2, 3):
        scene, _, cam, _, _ = wire(sid, 0, W, H)
        c = cam14(cam)
        u = (2.0 * (np.arange(W) + 0.5) / W - 1.0) * c[12]
        v = (1.0 - 2.0 * (np.arange(H) + 0.5) / H) * c[13]
        d = c[3:6][None, None, :] + c[6:9][None, None, :] * u[None, :, None] + c[9:12][None, None, :] * v[:, None, None]
        d = d / np.sqrt((d * d).sum(2, keepdims=True))
        depth, hit, normal = ref.ANALYTIC_SCENES[scene.name](c[0:3].copy(), d)
        store[f"s{sid}_depth"] = depth.astype("<f8")
        store[f"s{sid}_hitbits"] = np.packbits(hit)
        store[f"s{sid}_normal"] = normal.astype("<f8")
        store[f"s{sid}_cam"] = c
        print(f"  [analytic] {scene.name}: {int(hit.sum())} hits", flush=True)
    np.savez_compressed(os.path.join(OUT, f"analytic_{W}x{H}.npz"), **store)


def gen_viewpoints():
    """The reference's curated viewpoints (viewpoints.py:41-140) for every catalogue scene, as data."""
    from raymarching_benchmark.viewpoints import viewpoints_for
    from raymarching_benchmark.scenes.catalog import get_all_scenes
    out = {}
    for sc in get_all_scenes():
        out[sc.name] = [[v.name, v.category, [float(c) for c in v.position], [float(c) for c in v.target],
                         [float(c) for c in v.up]] for v in viewpoints_for(sc)]
    with open(os.path.join(OUT, "viewpoints.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("viewpoints.json:", sum(len(v) for v in out.values()), "viewpoints of", len(out), "scenes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    all_pairs = [(s, k) for s in range(len(SCENES)) for k in range(len(STRAT_KEYS))]
    graded9 = [0, 1, 2, 3, 4, 5, 6, 9, 10]  # the README's nine strategies (ids in STRATEGIES order)
    if a.only in ("all", "viewpoints"):
        gen_viewpoints()
    if a.only in ("all", "evals"):
        gen_evals()
    if a.only in ("all", "analytic"):
        gen_analytic()
    if a.only in ("all", "sdf"):
        gen_sdf()
    if a.only in ("all", "frames64"):
        gen_frames("64x48", 64, 48, all_pairs)
    if a.only in ("all", "frames160"):
        pairs = [(0, k) for k in graded9] + [(2, k) for k in graded9]
        pairs += [(s, k) for s in (9, 10) for k in (0, 4, 6)] + [(12, 0)]
        gen_frames("160x120", 160, 120, pairs)
    if a.only in ("all", "rows1080"):
        gen_rows1080([(0, 0), (2, 0), (9, 0), (10, 0), (10, 4), (10, 6), (12, 0)])
    if a.only in ("all", "leak"):
        gen_leak()
    if a.only in ("all", "leakseq"):
        gen_leakseq()
    if a.only in ("all", "params"):
        gen_params()
    if a.only in ("all", "schema"):
        gen_schema()
    if a.only in ("all", "small"):
        # max_iterations=100, 16x12: the configuration of the reference's own smoke test
        # (tests/test_smoke.py:31-43), every registry key on the Sphere.
        gen_frames("16x12_it100", 16, 12, [(0, k) for k in range(len(STRAT_KEYS))],
                   MarchConfig(max_iterations=100))


if __name__ == "__main__":
    main()
