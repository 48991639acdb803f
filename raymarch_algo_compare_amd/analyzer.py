"""Cross-run result table and the nine `matrix_<metric>.csv` files of a run.

File names, row / column order and number formatting are the reference's CSV schema (its example/ directory;
metrics/analyzer.py:153-169 writes them): one CSV per metric, rows = scenes in the order they were first rendered,
columns = strategy short names in sorted order, floats as pandas prints them, a missing cell left empty.  The table
itself is kept as a mapping (scene, strategy) -> RayMarchStats filled as results arrive.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from .stats import RayMarchStats

CSV_METRICS = [
    "iteration_mean", "time_per_ray_us", "hit_rate", "warp_divergence_proxy",
    "gpu_time_per_ray_us", "gpu_time_per_ray_median_us", "gpu_frame_ms_median", "gpu_time_sample_count",
    "gpu_warp_divergence_proxy",
]


class MetricsAnalyzer:
    """Collects one RayMarchStats per (scene, strategy) cell; the first result of a cell wins, like a linear search
    over the arrival order would."""

    def __init__(self):
        self.all_stats: List[RayMarchStats] = []                       # arrival order (the CLI's --json summary walks it)
        self._cells: Dict[Tuple[str, str], RayMarchStats] = {}
        self._scene_order: Dict[str, int] = {}                          # scene name -> rank of first appearance

    # -- filling ---------------------------------------------------------------------------------------------
    def add_result(self, stats: RayMarchStats) -> None:
        self.all_stats.append(stats)
        self._cells.setdefault((stats.scene_name, stats.strategy_name), stats)
        self._scene_order.setdefault(stats.scene_name, len(self._scene_order))

    def add_results(self, stats_list: Iterable[RayMarchStats]) -> None:
        for st in stats_list:
            self.add_result(st)

    # -- axes --------------------------------------------------------------------------------------------------
    def get_scenes(self) -> List[str]:
        return sorted(self._scene_order, key=self._scene_order.__getitem__)

    def get_strategies(self) -> List[str]:
        return sorted({strategy for _, strategy in self._cells})

    def get_stat(self, strategy: str, scene: str) -> Optional[RayMarchStats]:
        return self._cells.get((scene, strategy))

    # -- tables ------------------------------------------------------------------------------------------------
    def per_scene_matrix(self, metric: str = "iteration_mean") -> Tuple[List[str], List[str], np.ndarray]:
        """(scene names, strategy names, values[scene, strategy]); NaN where a cell is missing or the metric is unset."""
        rows, cols = self.get_scenes(), self.get_strategies()
        row_of = {name: i for i, name in enumerate(rows)}
        col_of = {name: j for j, name in enumerate(cols)}
        table = np.full((len(rows), len(cols)), np.nan)
        for (scene, strategy), st in self._cells.items():
            value = getattr(st, metric, None)
            if value is not None:
                table[row_of[scene], col_of[strategy]] = value
        return rows, cols, table

    def save_csv_matrices(self, output_dir: str) -> None:
        import pandas as pd
        os.makedirs(output_dir, exist_ok=True)
        for metric in CSV_METRICS:
            rows, cols, table = self.per_scene_matrix(metric)
            frame = pd.DataFrame(table, index=rows, columns=cols)
            frame.to_csv(os.path.join(output_dir, f"matrix_{metric}.csv"))
