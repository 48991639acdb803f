"""RayMarchStats -- same fields as the reference's core/types.py:24-75, computed from the maps the
GPU returns instead of from a list of per-ray MarchResult objects.

compute_from_maps() reproduces RayMarchStats.compute (core/types.py:77-137) with the same NumPy
calls on the same row-major float64 iteration array (mean / median / std / min / max /
percentile 95, 99), so every scalar is bit-identical to the reference's for identical iteration
maps.  The 8x4-block divergence proxy is evaluated vectorised; each block's population
variance is exact in binary64 (integers <= 2^10 scale), so the vectorised value equals the
reference's per-block np.std, and the mean over blocks runs in the same row-major order.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class RayMarchStats:
    strategy_name: str
    scene_name: str
    total_rays: int = 0
    hit_count: int = 0
    miss_count: int = 0
    sample_count: int = 0
    iteration_counts: List[int] = field(default_factory=list)
    iteration_mean: float = 0.0
    iteration_median: float = 0.0
    iteration_std: float = 0.0
    iteration_min: int = 0
    iteration_max: int = 0
    iteration_p95: float = 0.0
    iteration_p99: float = 0.0
    final_sdf_values: List[float] = field(default_factory=list)
    accuracy_mean: float = 0.0
    accuracy_max: float = 0.0
    accuracy_std: float = 0.0
    hit_rate: float = 0.0
    total_time_seconds: float = 0.0
    time_per_ray_us: float = 0.0
    gpu_time_per_ray_us: Optional[float] = None
    gpu_time_per_ray_median_us: Optional[float] = None
    gpu_time_sample_count: Optional[int] = None
    gpu_frame_ms_median: Optional[float] = None
    gpu_warp_divergence_proxy: Optional[float] = None
    gpu_width: Optional[int] = None
    gpu_height: Optional[int] = None
    warp_divergence_proxy: float = 0.0
    iteration_heatmap: Optional[np.ndarray] = None
    hit_map: Optional[np.ndarray] = None
    depth_map: Optional[np.ndarray] = None
    # extras of this engine (not in the reference)
    kernel_ms: Optional[float] = None          # hipEvent time of the render kernel
    evals_map: Optional[np.ndarray] = None     # SDF evaluations per ray, when requested (not a reference field)

    def compute_from_maps(self, iters: np.ndarray, hit: np.ndarray, depth: np.ndarray,
                          elapsed_seconds: float, final_sdf: Optional[np.ndarray] = None,
                          keep_lists: bool = False):
        height, width = iters.shape
        self.total_rays = int(iters.size)
        self.total_time_seconds = elapsed_seconds
        hit_b = hit.astype(bool)
        self.hit_count = int(hit_b.sum())
        self.miss_count = self.total_rays - self.hit_count
        iter_arr = iters.reshape(-1).astype(np.float64)                  # types.py:103
        if keep_lists:
            self.iteration_counts = iters.reshape(-1).tolist()
        if self.total_rays:
            self.sample_count = int(iter_arr.sum())                       # :106
            self.iteration_mean = float(np.mean(iter_arr))                # :108-114
            self.iteration_median = float(np.median(iter_arr))
            self.iteration_std = float(np.std(iter_arr))
            self.iteration_min = int(np.min(iter_arr))
            self.iteration_max = int(np.max(iter_arr))
            self.iteration_p95 = float(np.percentile(iter_arr, 95))
            self.iteration_p99 = float(np.percentile(iter_arr, 99))
        if final_sdf is not None and self.hit_count:
            sdf_arr = np.abs(final_sdf.reshape(-1)[hit_b.reshape(-1)]).astype(np.float64)   # :116-120
            if keep_lists:
                self.final_sdf_values = sdf_arr.tolist()
            self.accuracy_mean = float(np.mean(sdf_arr))
            self.accuracy_max = float(np.max(sdf_arr))
            self.accuracy_std = float(np.std(sdf_arr))
        self.hit_rate = self.hit_count / max(self.total_rays, 1)         # :122
        self.time_per_ray_us = (elapsed_seconds * 1e6) / max(self.total_rays, 1)
        self.warp_divergence_proxy = warp_divergence_proxy(iters)        # :125-133
        self.iteration_heatmap = iters.astype(np.int32, copy=False)
        self.hit_map = hit_b
        self.depth_map = depth.astype(np.float64)
        return self


def warp_divergence_proxy(iters: np.ndarray) -> float:
    """Mean over full, non-overlapping 4-row x 8-column blocks (origin top-left, partial edge
    blocks dropped) of the population std of the iteration counts (core/types.py:125-133)."""
    height, width = iters.shape
    bh, bw = height // 4, width // 8
    if bh == 0 or bw == 0:
        return 0.0
    blk = iters[: bh * 4, : bw * 8].astype(np.float64).reshape(bh, 4, bw, 8).transpose(0, 2, 1, 3).reshape(bh, bw, 32)
    stds = np.std(blk, axis=2).reshape(-1)
    return float(np.mean(stds.tolist()))


def warp_divergence_from_block_var(block_var: np.ndarray) -> float:
    """Same statistic from the kernel's per-block integers 32*sum(x^2) - sum(x)^2 (= variance * 1024)."""
    if block_var.size == 0:
        return 0.0
    stds = np.sqrt(block_var.reshape(-1).astype(np.float64) / 1024.0)
    return float(np.mean(stds.tolist()))


def gpu_warp_divergence_proxy(iters: np.ndarray) -> float:
    """The CLI's GPU-side proxy (reference main.py:233-243): 8x4 blocks INCLUDING partial edge
    blocks, plain Python mean of the block stds."""
    h, w = iters.shape
    a = iters.astype(np.float64)
    bh, bw = h // 4, w // 8
    stds: list = []
    for yy in range(0, h, 4):
        full_row = yy + 4 <= h
        if full_row and bw:
            blk = a[yy:yy + 4, : bw * 8].reshape(4, bw, 8).transpose(1, 0, 2).reshape(bw, 32)
            stds.extend(np.std(blk, axis=1).tolist())
            xs = range(bw * 8, w, 8)
        else:
            xs = range(0, w, 8)
        for xx in xs:
            block = a[yy: min(yy + 4, h), xx: min(xx + 8, w)].ravel()
            if block.size:
                stds.append(float(block.std()))
    return float(sum(stds) / len(stds)) if stds else 0.0
