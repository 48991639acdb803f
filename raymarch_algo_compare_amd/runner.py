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
from .registry import SCENES, get_shader_strategy, get_strategy_by_name


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
                             strategy.lipschitz if strategy.has_lipschitz else 1.0, True, params=strategy.params)
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


# ---- GPURunner: the reference's runner object (gpu/runner.py:17-268) on the gfx950 engine -------------------

# strategy ids of the reference's fragment shader (main.glsl:64-74) -> registry keys of this engine
GLSL_STRATEGY_KEYS = {0: "Standard", 1: "Overstep-Bisect", 2: "Relaxed", 3: "Segment", 4: "Enhanced",
                      5: "Heuristic-Auto-Relaxed", 6: "Skipping-Spheres", 7: "RevAA",
                      8: "Safe-Relaxed", 9: "Dense-March"}      # 8, 9: shader-only (registry.SHADER_ONLY_STRATEGIES)
# Uniforms the shader takes per run (gpu/runner.py:108-124) -> RmStrategyParams fields of the CPU-path strategies
# that read the same constant: `omega` is RelaxedSphereTracing's constructor argument and the start value of
# AutoRelaxedSphereTracing's omega (the shader's two relaxed marchers share the uniform, param_grid.py:21-23),
# `beta` SlopeAutoRelaxed's, `margin` the fattening of Skipping-Spheres.
# `stepScale` scales the step of the shader's standard() and dense_march() (the understep oracle, groundtruth.py:59-61),
# `minStep` is the floor of dense_march()'s stride.
SHADER_UNIFORMS = {"omega": ("omega", "ar_omega_init"), "beta": ("beta",), "margin": ("margin",),
                   "stepScale": ("step_scale",), "minStep": ("dense_min_step",)}
# uniforms that exist only in the GLSL marcher bodies this engine does not build (the shader's own segment tracing):
# only their defaults can be honoured
_GLSL_ONLY_DEFAULTS = {"kappa": 2.0}


class GPURunner:
    """`GPURunner.render` / `GPURunner.capture` with the reference's signatures and result layouts
    (gpu/runner.py:58-166, :168-268), evaluated by the gfx950 kernels with the CPU path's arithmetic.

    Deliberate differences: fp64 and the CPU camera model (the shader's pinhole ignores
    fov), the catalogue SDFs, rows top-to-bottom in both calls, `strategy_id` follows the shader's numbering
    (GLSL_STRATEGY_KEYS; pass `strategy_key=` for this engine's other strategies).  8 = Safe-Relaxed and 9 =
    Dense-March exist only as shader text in the reference: they run the shader's control flow on the CPU path's
    arithmetic and are PARITY UNPINNED (nothing of the reference can check them here).  `params` takes the shader's
    uniform names like the reference seam (gpu/runner.py:120-124): omega / beta / margin / stepScale / minStep reach
    RmStrategyParams (SHADER_UNIFORMS; `minStep` defaults to max(hit_threshold, min_step_fraction * max_distance)
    like the seam, gpu/runner.py:117-118); RmStrategyParams field names are accepted as they are; `kappa` belongs to
    the shader's own segment tracing (non-default values raise NotImplementedError)."""

    def __init__(self, device_id: int | None = None):
        self.device_id = device_id

    @staticmethod
    def _strategy(strategy_id, strategy_key):
        if strategy_key is None:
            if strategy_id not in GLSL_STRATEGY_KEYS:
                raise ValueError(f"strategy id {strategy_id} is not one of the shader's ids (0..9)")
            strategy_key = GLSL_STRATEGY_KEYS[strategy_id]
        st = get_shader_strategy(strategy_key) or get_strategy_by_name(strategy_key)
        if st is None:
            raise KeyError(f"unknown strategy {strategy_key!r}")
        return st

    @staticmethod
    def strategy_params(params) -> dict:
        """Shader uniform overrides (or RmStrategyParams names) -> RmStrategyParams overrides."""
        out = {}
        for k, v in (params or {}).items():
            if k in SHADER_UNIFORMS:
                for f in SHADER_UNIFORMS[k]:
                    out[f] = float(v)
            elif k in _native.DEFAULT_STRATEGY_PARAMS:
                out[k] = v
            elif k in _GLSL_ONLY_DEFAULTS:
                if float(v) != _GLSL_ONLY_DEFAULTS[k]:
                    raise NotImplementedError(f"{k}={v}: a constant of the GLSL marchers only; the CPU-path strategies "
                                              f"have no counterpart ({k}={_GLSL_ONLY_DEFAULTS[k]} is the shader default)")
            else:
                raise KeyError(f"unknown shader parameter {k!r}")
        return out

    def _frame(self, scene_id, strategy, render_cfg, march_cfg, lipschitz, timed, want_evals, params=None):
        if not 0 <= int(scene_id) < len(SCENES):
            raise ValueError(f"scene id {scene_id} out of range")
        cam = Camera(render_cfg.camera_position, render_cfg.camera_target, render_cfg.camera_up,
                     render_cfg.fov_degrees, render_cfg.width, render_cfg.height)
        _native.init(self.device_id)
        lip = float(lipschitz if lipschitz is not None else 1.0) if strategy.has_lipschitz else 1.0
        prm = dict(strategy.params, **self.strategy_params(params))
        if strategy.key == "Dense-March" and "dense_min_step" not in prm:      # the seam's minStep (gpu/runner.py:117-118)
            prm["dense_min_step"] = max(march_cfg.hit_threshold,
                                        getattr(march_cfg, "min_step_fraction", 0.0) * march_cfg.max_distance)
        desc = _native.make_desc(int(scene_id), strategy.id, cam.params14(), cam.width, cam.height, 0, None,
                                 march_cfg.max_iterations, march_cfg.hit_threshold, march_cfg.max_distance, lip, True,
                                 params=prm)
        out = _native.render(desc, want_t_raw=True, want_final_sdf=True, repeats=1 if timed else 0, want_evals=want_evals)
        return cam, out

    @staticmethod
    def _geom(out, march_cfg):
        h, w = out["iters"].shape
        px = np.empty((h, w, 4), dtype=np.float32)                      # main.glsl:79-84
        px[..., 0] = out["hit"]
        px[..., 1] = out["iters"] / float(march_cfg.max_iterations)
        px[..., 2] = out["t_raw"] / float(march_cfg.max_distance)
        px[..., 3] = out["final_sdf"]
        return px

    def render(self, scene_id: int, strategy_id: int, render_cfg: RenderConfig, march_cfg: MarchConfig,
               lipschitz: float = 1.0, params: dict | None = None, *, strategy_key: str | None = None):
        """-> (pixels (H, W, 4) float32 [hit, iterations / max, t / max_distance, final_sdf], seconds)."""
        _, out = self._frame(scene_id, self._strategy(strategy_id, strategy_key), render_cfg, march_cfg, lipschitz, True, False, params)
        return self._geom(out, march_cfg), out["timing"]["ms_median"] * 1e-3

    def capture(self, scene_id: int, strategy_id: int, render_cfg: RenderConfig, march_cfg: MarchConfig,
                lipschitz: float = 1.0, params: dict | None = None, *, strategy_key: str | None = None) -> dict:
        """-> geom (H,W,4), normal (H,W,3), depth (H,W), color (H,W,3), evals (H,W), hit (H,W) bool -- the capture
        targets of main.glsl:79-110: tetrahedron normals from four SDF evaluations (rm_sdf_eval) at the hit point,
        the shader's fixed key light + hemisphere ambient + gamma, its background on misses, and the number of SDF
        evaluations the march itself performed."""
        cam, out = self._frame(scene_id, self._strategy(strategy_id, strategy_key), render_cfg, march_cfg, lipschitz, False, True, params)
        h, w = out["iters"].shape
        hit = out["hit"] > 0
        c = cam.params14()
        u = (2.0 * (np.arange(w) + 0.5) / w - 1.0) * c[12]                   # camera.py:37-38
        v = (1.0 - 2.0 * (np.arange(h) + 0.5) / h) * c[13]
        rd = c[3:6][None, None, :] + c[6:9][None, None, :] * u[None, :, None] + c[9:12][None, None, :] * v[:, None, None]
        rd /= np.sqrt((rd * rd).sum(2, keepdims=True))
        depth = np.where(hit, out["t_raw"], 0.0)
        normal = np.zeros((h, w, 3))
        color = np.empty((h, w, 3))
        tb = 0.5 * (rd[..., 1] + 1.0)                                          # main.glsl background()
        color[:] = (1.0 - tb)[..., None] * np.array([0.06, 0.07, 0.09]) + tb[..., None] * np.array([0.12, 0.14, 0.18])
        if hit.any():
            pos = c[0:3][None, :] + depth[hit][:, None] * rd[hit]
            e = 0.0005                                                         # calcNormal, main.glsl:26-35
            ks = np.array([[1.0, -1.0, -1.0], [-1.0, -1.0, 1.0], [-1.0, 1.0, -1.0], [1.0, 1.0, 1.0]])
            d4 = _native.sdf_eval(int(scene_id), (pos[:, None, :] + e * ks[None, :, :]).reshape(-1, 3)).reshape(-1, 4)
            n = (d4[:, :, None] * ks[None, :, :]).sum(1)
            n /= np.maximum(np.sqrt((n * n).sum(1, keepdims=True)), 1e-300)
            normal[hit] = n
            L = np.array([0.6, 0.7, 0.5]) / np.sqrt(0.6 ** 2 + 0.7 ** 2 + 0.5 ** 2)   # shade(), main.glsl:40-47
            diff = np.maximum(n @ L, 0.0)
            hemi = 0.5 + 0.5 * n[:, 1]
            col = np.array([0.82, 0.80, 0.78])[None, :] * (0.15 * hemi + 0.85 * diff)[:, None]
            color[hit] = np.clip(col, 0.0, 1.0) ** 0.4545
        return {"geom": self._geom(out, march_cfg), "normal": normal.astype(np.float32), "depth": depth.astype(np.float32),
                "color": color.astype(np.float32), "evals": out["evals"].astype(np.float32), "hit": hit}
