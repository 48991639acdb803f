"""run_gpu_benchmark -- same call shape and result keys as the reference's GPU seam
(gpu/runner.py:270-330), backed by the gfx950 kernels instead of a moderngl fragment shader.

Differences from the reference seam, all deliberate: the arithmetic is the CPU
path's (fp64, the reference camera, the catalogue SDFs), every strategy has its own kernel (the
reference maps Slope-AR / Curvature to Standard and Hybrid to Overstep-Bisect), and rows are
top-to-bottom like the CPU maps.
"""
from __future__ import annotations

import statistics

import numpy as np

from . import _native
from .camera import Camera
from .config import MarchConfig, RenderConfig
from .registry import SCENES, get_strategy_by_name


def run_gpu_benchmark(scene_name: str, strategy_name: str, render_cfg: RenderConfig, march_cfg: MarchConfig, *,
                      gpu_warmup: int = 3, gpu_repeats: int = 7, device_id: int | None = None):
    scene = next((s for s in SCENES if s.name == scene_name), None)      # exact name (runner.py:275-280)
    if scene is None:
        return None
    strategy = get_strategy_by_name(strategy_name)
    if strategy is None:
        return None
    if strategy.has_lipschitz and scene.lipschitz is not None:
        strategy.lipschitz = scene.lipschitz
    cam = Camera(render_cfg.camera_position, render_cfg.camera_target, render_cfg.camera_up,
                 render_cfg.fov_degrees, render_cfg.width, render_cfg.height)
    _native.init(device_id)
    desc = _native.make_desc(scene.id, strategy.id, cam.params14(), cam.width, cam.height, 0, None,
                             march_cfg.max_iterations, march_cfg.hit_threshold, march_cfg.max_distance,
                             strategy.lipschitz if strategy.has_lipschitz else 1.0, True)
    repeats = max(1, min(int(gpu_repeats), _native.RM_MAX_TIMED))
    out = _native.render(desc, want_t_raw=True, want_final_sdf=True, warmup=max(0, int(gpu_warmup)), repeats=repeats)
    times = [ms * 1e-3 for ms in out["timing"]["ms_each"]]
    ts = sorted(times)
    median_t = float(statistics.median(ts))
    q1 = float(statistics.median(ts[: len(ts) // 2])) if len(ts) > 1 else median_t
    q3 = float(statistics.median(ts[(len(ts) + 1) // 2:])) if len(ts) > 1 else median_t
    h, w = out["iters"].shape
    pixels = np.empty((h, w, 4), dtype=np.float32)                        # main.glsl:79-84 channel layout
    pixels[..., 0] = out["hit"]
    pixels[..., 1] = out["iters"] / float(march_cfg.max_iterations)
    pixels[..., 2] = out["t_raw"] / float(march_cfg.max_distance)
    pixels[..., 3] = out["final_sdf"]
    return {
        "pixels": pixels,
        "render_times_s": times,
        "render_time_s_median": median_t,
        "render_time_s_iqr": q3 - q1,
        "render_time_s_mean": float(sum(times) / len(times)),
        "sample_count": len(times),
        # exact integer maps for consumers that do not want the float32 round trip
        "iterations": out["iters"], "hit": out["hit"], "depth": out["depth"],
    }
