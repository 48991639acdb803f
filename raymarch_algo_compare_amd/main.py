"""run_once + CLI with the reference's flags and outputs (reference main.py:29-78, 145-315),
rendering through the gfx950 kernels.

    python -m raymarch_algo_compare_amd --scene Sphere --strategy Standard --width 160 --height 120

Reference quirks kept on purpose: the CLI reuses one RenderConfig, so a scene
without a suggested camera inherits the previous scene's (--fresh-camera turns that off); scene /
strategy lookup rules are the reference's.  One deliberate deviation: an unknown name raises
instead of silently rendering Sphere / Standard (main.py:39-40) unless --compat-fallback is given.
Out of scope (not on the hot path): tables.txt, charts, REPORT.md.
"""
from __future__ import annotations

import argparse
import json
from typing import Optional

from .analyzer import MetricsAnalyzer
from .artifacts import save_outputs
from .camera import Camera
from .collector import HipCollector
from .config import MarchConfig, RenderConfig
from .registry import SCENES, get_all_scenes, get_scene_by_name, get_strategy_by_name, list_strategies
from .stats import RayMarchStats, gpu_warp_divergence_proxy


def run_once(render: Optional[RenderConfig] = None, march: Optional[MarchConfig] = None,
             scene_name: str = "Sphere", strategy_name: str = "Standard",
             hybrid_fallback_iter: int | None = None, *, compat_fallback: bool = False,
             full: bool = True, device_id: int | None = None) -> RayMarchStats:
    """Render one (scene, strategy) frame and return its RayMarchStats (reference main.py:29-78)."""
    render = render or RenderConfig(width=64, height=48)
    march = march or MarchConfig()
    scene = get_scene_by_name(scene_name)
    strategy = get_strategy_by_name(strategy_name)
    if scene is None:
        if not compat_fallback:
            raise KeyError(f"unknown scene {scene_name!r} (the reference would silently render 'Sphere'; "
                           f"pass compat_fallback=True / --compat-fallback for that behaviour)")
        scene = SCENES[0]
    if strategy is None:
        if not compat_fallback:
            raise KeyError(f"unknown strategy {strategy_name!r} (the reference would silently use 'Standard')")
        strategy = get_strategy_by_name("Standard")
    # hybrid_fallback_iter: accepted for signature parity; AdaptiveHybridTracing.march never reads it.
    suggestion = scene.suggested_camera()                                 # main.py:50-55 (mutates `render`)
    if suggestion:
        render.camera_position = suggestion.camera_position
        render.camera_target = suggestion.camera_target
        render.camera_up = suggestion.camera_up
        render.fov_degrees = suggestion.fov_degrees
    if strategy.has_lipschitz:                                            # main.py:58-61
        bound = scene.known_lipschitz_bound()
        if bound is not None:
            strategy.lipschitz = bound
    cam = Camera(render.camera_position, render.camera_target, render.camera_up, render.fov_degrees,
                 render.width, render.height)
    return HipCollector(march, full=full, device_id=device_id).benchmark_strategy(strategy, scene, cam, verbose=False)


def _print_stats(s: RayMarchStats) -> None:
    print(f"Strategy: {s.strategy_name} | Scene: {s.scene_name}")
    print(f"Rays: {s.total_rays}  Hits: {s.hit_count}  Hit rate: {s.hit_rate:.2%}")
    print(f"Iter mean: {s.iteration_mean:.2f}  p95: {s.iteration_p95:.1f}  max: {s.iteration_max}")
    print(f"Time: {s.time_per_ray_us:.4f} us/ray  Total: {s.total_time_seconds:.4f}s"
          + (f"  Kernel: {s.kernel_ms:.3f} ms" if s.kernel_ms is not None else ""))


def attach_gpu_columns(stats: RayMarchStats, scene_name: str, strat_name: str, gpu_rc: RenderConfig,
                       mc: MarchConfig, warmup: int, repeats: int) -> None:
    """The CLI's GPU block (main.py:211-253) on top of run_gpu_benchmark."""
    from .runner import run_gpu_benchmark
    res = run_gpu_benchmark(scene_name, strat_name, gpu_rc, mc, gpu_warmup=warmup, gpu_repeats=repeats)
    if res is None:
        stats.gpu_time_per_ray_us = stats.gpu_time_per_ray_median_us = None
        stats.gpu_time_sample_count = stats.gpu_frame_ms_median = stats.gpu_warp_divergence_proxy = None
        return
    n = gpu_rc.width * gpu_rc.height
    stats.gpu_width, stats.gpu_height = int(gpu_rc.width), int(gpu_rc.height)
    median_s = float(res["render_time_s_median"])
    stats.gpu_time_per_ray_median_us = (median_s / n) * 1e6
    stats.gpu_time_sample_count = int(res["sample_count"])
    stats.gpu_time_per_ray_us = float(res["render_time_s_mean"] / n * 1e6)
    stats.gpu_frame_ms_median = (stats.gpu_time_per_ray_median_us * n) / 1000.0
    stats.gpu_warp_divergence_proxy = gpu_warp_divergence_proxy(res["iterations"])


def cli(argv: Optional[list] = None) -> int:
    p = argparse.ArgumentParser(prog="raymarch-bench-amd", description="Ray marching benchmark runner (MI355X)")
    p.add_argument("--width", type=int, default=64)
    p.add_argument("--height", type=int, default=48)
    p.add_argument("--gpu-width", type=int, default=None)
    p.add_argument("--gpu-height", type=int, default=None)
    p.add_argument("--gpu-1080p", action="store_true")
    p.add_argument("--gpu-warmup", type=int, default=5)
    p.add_argument("--gpu-repeats", type=int, default=10)
    p.add_argument("--hybrid-fallback-iter", type=int, default=None)
    p.add_argument("--scene", type=str, default="Sphere", help="Scene name (comma-separated, or 'all')")
    p.add_argument("--strategy", type=str, default="Standard", help="Strategy name (comma-separated, or 'all')")
    p.add_argument("--output-dir", type=str, default=None)
    p.add_argument("--no-save-images", action="store_true")
    p.add_argument("--json", type=str, default=None)
    p.add_argument("--kappa", type=float, default=None)
    p.add_argument("--min-step-fraction", type=float, default=None)
    # engine-specific
    p.add_argument("--compat-fallback", action="store_true",
                   help="unknown scene/strategy names silently fall back to Sphere/Standard like the reference")
    p.add_argument("--fresh-camera", action="store_true",
                   help="give every scene a fresh RenderConfig (the reference CLI leaks the previous scene's camera)")
    p.add_argument("--no-gpu-columns", action="store_true", help="skip the separate timed GPU pass (gpu_* columns)")
    p.add_argument("--device", type=int, default=None)
    args = p.parse_args(argv)

    rc = RenderConfig(width=args.width, height=args.height)
    if args.gpu_width is not None or args.gpu_height is not None:
        gpu_rc = RenderConfig(width=int(args.gpu_width or args.width), height=int(args.gpu_height or args.height))
    elif args.gpu_1080p:
        gpu_rc = RenderConfig(width=1920, height=1080)
    else:
        gpu_rc = rc
    mc = MarchConfig()
    if args.kappa is not None:
        mc.kappa = float(args.kappa)
    if args.min_step_fraction is not None:
        mc.min_step_fraction = float(args.min_step_fraction)
    results_dir = args.output_dir or "results"

    scene_names = ([s.name for s in get_all_scenes()] if args.scene.lower() == "all"
                   else [s.strip() for s in args.scene.split(",") if s.strip()])
    strategy_names = (list_strategies() if args.strategy.lower() == "all"
                      else [s.strip() for s in args.strategy.split(",") if s.strip()])
    if args.device is not None:
        from . import _native
        _native.init(args.device)

    analyzer = MetricsAnalyzer()
    print(f"Running benchmark: {len(strategy_names)} strategies x {len(scene_names)} scenes at {args.width}x{args.height}")
    for scene_name in scene_names:
        for strat_name in strategy_names:
            print(f"\n>> Scene: {scene_name} | Strategy: {strat_name}")
            this_rc = RenderConfig(width=args.width, height=args.height) if args.fresh_camera else rc
            stats = run_once(render=this_rc, march=mc, scene_name=scene_name, strategy_name=strat_name,
                             hybrid_fallback_iter=args.hybrid_fallback_iter, compat_fallback=args.compat_fallback)
            _print_stats(stats)
            if not args.no_gpu_columns:
                try:
                    attach_gpu_columns(stats, scene_name, strat_name, gpu_rc, mc, int(args.gpu_warmup), int(args.gpu_repeats))
                except Exception as e:                                    # main.py:250-253: tolerate, mark None
                    print(f"  [Warning] GPU timing pass failed: {e}")
                    stats.gpu_time_per_ray_us = None
                    stats.gpu_warp_divergence_proxy = None
            analyzer.add_result(stats)
            if not args.no_save_images:
                save_outputs(stats, results_dir, max_iters=mc.max_iterations)

    if len(analyzer.all_stats) > 1:
        try:
            analyzer.save_csv_matrices(results_dir)
            print(f"  Saved CSV matrices to: {results_dir}")
        except Exception as e:
            print(f"  [Error] Failed to save CSV matrices: {e}")

    if args.json and analyzer.all_stats:
        summary = [{
            "strategy": s.strategy_name, "scene": s.scene_name, "total_rays": int(s.total_rays),
            "hit_count": int(s.hit_count), "hit_rate": float(s.hit_rate),
            "iteration_mean": float(s.iteration_mean), "iteration_p95": float(s.iteration_p95),
            "iteration_max": int(s.iteration_max), "time_per_ray_us": float(s.time_per_ray_us),
            "warp_divergence": float(s.warp_divergence_proxy),
        } for s in analyzer.all_stats]
        with open(args.json, "w", encoding="utf-8") as f:
            json.dump(summary, f, indent=2)
        print(f"\n  Saved summary JSON to: {args.json}")
    return 0


if __name__ == "__main__":
    raise SystemExit(cli())
