"""Per-run output artefacts with the reference's file names and stats.json schema
(reference main.py:92-142): iterations.png, hit_map.png, inv_depth.png, depth_map.npy, stats.json
inside <results>/<Scene>__<Strategy>__<timestamp>/."""
from __future__ import annotations

import json
import os
from typing import Optional

import numpy as np

from .stats import RayMarchStats


def safe_name(s: str) -> str:
    return s.replace(" ", "_").replace("/", "_")           # main.py:88-89


def _write_gray(arr01: np.ndarray, path: str) -> None:
    img = (255.0 * np.clip(arr01, 0.0, 1.0)).astype(np.uint8)
    try:
        from PIL import Image
        Image.fromarray(img).save(path)
    except ImportError:
        np.save(path + ".npy", img)


def compact_stats(stats: RayMarchStats) -> dict:
    """The stats.json record (main.py:121-138), key for key."""
    def opt(v, cast):
        return cast(v) if v is not None else None
    return {
        "strategy": stats.strategy_name,
        "scene": stats.scene_name,
        "total_rays": int(stats.total_rays),
        "hit_count": int(stats.hit_count),
        "hit_rate": float(stats.hit_rate),
        "iteration_mean": float(stats.iteration_mean),
        "iteration_p95": float(stats.iteration_p95),
        "iteration_max": int(stats.iteration_max),
        "warp_divergence": float(stats.warp_divergence_proxy),
        "time_us_per_ray": float(stats.time_per_ray_us),
        "gpu_time_us_per_ray": opt(stats.gpu_time_per_ray_us, float),
        "gpu_time_us_per_ray_median": opt(stats.gpu_time_per_ray_median_us, float),
        "gpu_time_sample_count": opt(stats.gpu_time_sample_count, int),
        "gpu_warp_divergence": opt(stats.gpu_warp_divergence_proxy, float),
        "gpu_width": opt(stats.gpu_width, int),
        "gpu_height": opt(stats.gpu_height, int),
    }


def save_outputs(stats: RayMarchStats, results_dir: str, max_iters: Optional[int] = None) -> str:
    timestamp = np.datetime64(np.datetime64("now"), "s").astype(str).replace(":", "-")
    out_dir = os.path.join(results_dir, f"{safe_name(stats.scene_name)}__{safe_name(stats.strategy_name)}__{timestamp}")
    os.makedirs(out_dir, exist_ok=True)
    if stats.iteration_heatmap is not None:
        mx = max(int(max_iters or stats.iteration_max or 1), 1)
        _write_gray(stats.iteration_heatmap.astype(np.float32) / mx, os.path.join(out_dir, "iterations.png"))
    if stats.hit_map is not None:
        _write_gray(stats.hit_map.astype(np.float32), os.path.join(out_dir, "hit_map.png"))
    if stats.depth_map is not None and stats.hit_map is not None:
        inv = np.zeros_like(stats.depth_map, dtype=np.float32)
        mask = stats.hit_map.astype(bool) & (stats.depth_map > 1e-12)
        inv[mask] = 1.0 / stats.depth_map[mask]
        if mask.any():
            inv = inv / max(float(np.percentile(inv[mask], 99.0)), 1e-12)
        _write_gray(inv, os.path.join(out_dir, "inv_depth.png"))
        np.save(os.path.join(out_dir, "depth_map.npy"), stats.depth_map)
    with open(os.path.join(out_dir, "stats.json"), "w", encoding="utf-8") as f:
        json.dump(compact_stats(stats), f, indent=2)
    print(f"  Saved outputs to: {out_dir}")
    return out_dir
