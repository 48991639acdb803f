"""HipCollector -- the MI355X replacement of MetricsCollector (reference metrics/collector.py:13-66).

benchmark_strategy(strategy, scene, camera) keeps the reference signature and returns a
RayMarchStats with the same fields; the double Python loop over pixels (collector.py:40-44)
becomes ONE launch of the (scene, strategy) gfx950 kernel through rm_render (include/rm_hip.h).
"""
from __future__ import annotations

import time

from . import _native
from .camera import Camera
from .config import MarchConfig
from .registry import SceneInfo, StrategyInfo
from .stats import RayMarchStats


class HipCollector:
    """Collects per-ray results from the GPU and computes aggregate statistics.

    full=True (default) also fetches MarchResult.final_sdf (accuracy_* statistics) and the raw
    fp64 t of every ray, so depth_map is the reference's float64 `t if hit else 0.0`; full=False
    is the 9 B/ray product path (fp32 depth, no accuracy statistics)."""

    def __init__(self, config: MarchConfig, full: bool = True, device_id: int | None = None,
                 tile_rows: int = 0, refill_min: int = 0, grid_waves: int = 0):
        self.config = config
        self.full = full
        self.device_id = device_id
        self.tuning = dict(tile_rows=tile_rows, refill_min=refill_min, grid_waves=grid_waves)

    def _lipschitz(self, strategy: StrategyInfo) -> float:
        return float(strategy.lipschitz) if strategy.has_lipschitz else 1.0

    def render_maps(self, strategy: StrategyInfo, scene: SceneInfo, camera: Camera, row0: int = 0,
                    rows: int | None = None, warmup: int = 0, repeats: int = 0) -> dict:
        _native.init(self.device_id)
        desc = _native.make_desc(
            scene.id, strategy.id, camera.params14(), camera.width, camera.height, row0, rows,
            self.config.max_iterations, self.config.hit_threshold, self.config.max_distance,
            self._lipschitz(strategy), self.full, params=strategy.params, **self.tuning)
        return _native.render(desc, want_t_raw=self.full, want_final_sdf=self.full, warmup=warmup,
                              repeats=repeats)

    def benchmark_batch(self, strategy: StrategyInfo, scene: SceneInfo, cameras, configs=None, want_evals: bool = False,
                        params=None):
        """Render one frame per camera (optionally one MarchConfig per frame) in ONE launch
        (rm_render_batch) and return a RayMarchStats per frame -- the shape of the reference's sweeps
        over curated viewpoints and iteration-budget / epsilon levels (viewpoints.py:41-140,
        sweep.py:96-127).  Cameras must share the resolution.  fp32 depth (the 9 B/ray path).
        `params`: optional list, one dict of RmStrategyParams overrides per frame (the reference's parameter grid,
        param_grid.py:20-27 / sweep.py:181), applied on top of the strategy's own constructor arguments.
        want_evals renders with march.full = 1 so the evals map counts every sdf() call the reference's march()
        makes (its final_sdf evaluations included), like the single-frame goldens."""
        import numpy as np
        cameras = list(cameras)
        if not cameras:
            return []
        w, h = cameras[0].width, cameras[0].height
        if any(c.width != w or c.height != h for c in cameras):
            raise ValueError("all cameras of a batch must share the resolution")
        _native.init(self.device_id)
        if params is not None and len(params) != len(cameras):
            raise ValueError("one parameter dict per frame")
        mcs = list(configs) if configs is not None else [self.config] * len(cameras)
        cfgs = [dict(max_iterations=c.max_iterations, hit_threshold=c.hit_threshold, max_distance=c.max_distance,
                     lipschitz=self._lipschitz(strategy), full=bool(want_evals),
                     params=dict(strategy.params, **(params[i] if params is not None else {})))
                for i, c in enumerate(mcs)]
        shape = _native.make_desc(scene.id, strategy.id, cameras[0].params14(), w, h, **self.tuning)
        start = time.perf_counter()
        out = _native.render_batch(shape, np.stack([c.params14() for c in cameras]), cfgs, want_evals=want_evals)
        elapsed = time.perf_counter() - start
        res = []
        for i in range(len(cameras)):
            st = RayMarchStats(strategy_name=strategy.short_name, scene_name=scene.name)
            st.compute_from_maps(out["iters"][i], out["hit"][i], out["depth"][i], elapsed / len(cameras))
            st.kernel_ms = out["ms_total"] / len(cameras)
            st.evals_map = out["evals"][i] if want_evals else None       # SDF evaluations per ray (RmOutputs.evals)
            res.append(st)
        return res

    def benchmark_strategy(self, strategy: StrategyInfo, scene: SceneInfo, camera: Camera,
                           verbose: bool = True) -> RayMarchStats:
        width, height = camera.width, camera.height
        if verbose:
            print(f"  Benchmarking: {strategy.short_name} on {scene.name} "
                  f"({width}x{height} = {width * height} rays)...")
        start = time.perf_counter()
        out = self.render_maps(strategy, scene, camera, repeats=1)
        elapsed = time.perf_counter() - start        # wall clock incl. launch + copies (collector.py:38,52)
        stats = RayMarchStats(strategy_name=strategy.short_name, scene_name=scene.name)
        if self.full:
            import numpy as np
            depth = np.where(out["hit"] > 0, out["t_raw"], 0.0)          # types.py:93, float64
        else:
            depth = out["depth"]
        stats.compute_from_maps(out["iters"], out["hit"], depth, elapsed, out["final_sdf"])
        stats.kernel_ms = out["timing"]["ms_median"] if out["timing"] else None
        dev = out["stats"]
        if (dev["hit_count"], dev["sum_iters"], dev["total_rays"]) != (stats.hit_count, stats.sample_count, stats.total_rays):
            raise RuntimeError(f"in-kernel frame reduce disagrees with the returned maps: {dev['hit_count']}/"
                               f"{dev['sum_iters']}/{dev['total_rays']} vs {stats.hit_count}/{stats.sample_count}/{stats.total_rays}")
        if verbose:
            print(f"    Done in {elapsed:.2f}s. Hit rate: {stats.hit_rate:.1%}, "
                  f"Mean iters: {stats.iteration_mean:.1f}, Max iters: {stats.iteration_max}")
        return stats
