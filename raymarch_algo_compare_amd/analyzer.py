"""Cross-run matrix / CSV writer with the reference's schema (metrics/analyzer.py:22-169):
rows = scene names in first-seen order, columns = sorted strategy short names, one
matrix_<metric>.csv per metric, pandas default float formatting, NaN -> empty."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np

from .stats import RayMarchStats

CSV_METRICS = [
    "iteration_mean", "time_per_ray_us", "hit_rate", "warp_divergence_proxy",
    "gpu_time_per_ray_us", "gpu_time_per_ray_median_us", "gpu_frame_ms_median", "gpu_time_sample_count",
    "gpu_warp_divergence_proxy",
]


class MetricsAnalyzer:
    def __init__(self):
        self.all_stats: List[RayMarchStats] = []

    def add_result(self, stats: RayMarchStats):
        self.all_stats.append(stats)

    def add_results(self, stats_list: List[RayMarchStats]):
        self.all_stats.extend(stats_list)

    def get_strategies(self) -> List[str]:
        return sorted(set(s.strategy_name for s in self.all_stats))

    def get_scenes(self) -> List[str]:
        seen: List[str] = []
        for s in self.all_stats:
            if s.scene_name not in seen:
                seen.append(s.scene_name)
        return seen

    def get_stat(self, strategy: str, scene: str) -> Optional[RayMarchStats]:
        for s in self.all_stats:
            if s.strategy_name == strategy and s.scene_name == scene:
                return s
        return None

    def per_scene_matrix(self, metric: str = "iteration_mean") -> Tuple[List[str], List[str], np.ndarray]:
        scenes, strategies = self.get_scenes(), self.get_strategies()
        matrix = np.full((len(scenes), len(strategies)), np.nan)
        for si, scene in enumerate(scenes):
            for sti, strategy in enumerate(strategies):
                stat = self.get_stat(strategy, scene)
                if stat:
                    v = getattr(stat, metric, np.nan)
                    matrix[si, sti] = np.nan if v is None else v
        return scenes, strategies, matrix

    def save_csv_matrices(self, output_dir: str):
        import pandas as pd
        os.makedirs(output_dir, exist_ok=True)
        for metric in CSV_METRICS:
            scenes, strategies, matrix = self.per_scene_matrix(metric)
            pd.DataFrame(matrix, index=scenes, columns=strategies).to_csv(os.path.join(output_dir, f"matrix_{metric}.csv"))
