"""Scene and strategy registries: the reference's names, order, cameras and lookup rules.

Scene order == get_all_scenes() (reference scenes/catalog.py:640-663) == kernel scene id.
Strategy order == the STRATEGIES dict (strategies/__init__.py:16-28) == kernel strategy id.
The SDFs and marchers themselves are device code (csrc/rm_scenes.h, csrc/rm_strategies.h);
these records only carry what the host needs: names, categories, suggested cameras
(catalog.py suggested_camera overrides) and Lipschitz bounds (known_lipschitz_bound).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

from .config import RenderConfig


@dataclass(frozen=True)
class SceneInfo:
    id: int
    name: str
    category: str
    description: str
    camera_position: Optional[tuple] = None   # suggested_camera(); None = keep the caller's camera
    camera_target: Optional[tuple] = None
    lipschitz: Optional[float] = 1.0           # known_lipschitz_bound()

    def suggested_camera(self) -> Optional[RenderConfig]:
        if self.camera_position is None:
            return None
        return RenderConfig(camera_position=self.camera_position, camera_target=self.camera_target)

    def known_lipschitz_bound(self) -> Optional[float]:
        return self.lipschitz


@dataclass
class StrategyInfo:
    id: int
    key: str          # registry key (STRATEGIES dict)
    short_name: str   # MarchStrategy.short_name: what tables / stats.json / CSV columns show
    name: str         # MarchStrategy.name
    has_lipschitz: bool = False
    lipschitz: float = 1.0
    # RmStrategyParams overrides (include/rm_hip.h) this instance was constructed with; {} = the reference's defaults
    params: Dict[str, float] = field(default_factory=dict)


SCENES: List[SceneInfo] = [
    SceneInfo(0, "Sphere", "smooth", "Unit sphere at origin. Baseline: all methods should handle easily."),
    SceneInfo(1, "Grazing Plane", "stress_grazing",
              "Plane viewed at <5 degree angle. Worst-case for standard sphere tracing (tiny steps).",
              (0.0, 0.6, 8.0), (0.0, -0.4, 0.0)),
    SceneInfo(2, "Cube", "sharp_edges", "Unit cube. Tests sharp edges and corners."),
    SceneInfo(3, "Thin Torus", "thin_features", "Torus with small minor radius. Tests thin feature handling."),
    SceneInfo(4, "Cylinder", "sharp_edges", "Cylinder with sharp circular edges."),
    SceneInfo(5, "Near Miss", "near_miss", "Two spheres nearly touching. Rays through gap test convergence behavior."),
    SceneInfo(6, "Hollow Cube (CSG)", "csg", "Cube with sphere subtracted. Tests CSG boundary handling."),
    SceneInfo(7, "Smooth Blend", "smooth", "Smoothly blended sphere+box. Tests smooth union SDF quality."),
    SceneInfo(8, "Onion Shell", "thin_features", "Nested onion shells of a sphere. Extreme thin feature stress test."),
    SceneInfo(9, "Menger Sponge (iter=3)", "fractal", "Menger sponge fractal. Tests tunneling and high iteration demands."),
    SceneInfo(10, "Mandelbulb", "fractal", "Mandelbulb fractal. Infinite curvature breaks planar assumptions.",
              (0.0, 0.0, 3.0), (0.0, 0.0, 0.0), None),
    SceneInfo(11, "Bad Lipschitz Sphere", "invalid_sdf", "Sphere with SDF scaled by 2x (invalid). Tests overshoot recovery.",
              None, None, 2.0),
    SceneInfo(12, "Pillar Forest", "complex", "Grid of thin cylinders. Many near-miss rays, tests throughput.",
              (1.0, 1.0, 8.0), (0.0, 0.0, 0.0)),
    SceneInfo(13, "Thin Planes Stack", "thin_features", "Multiple thin parallel planes. Tests tunneling through thin geometry.",
              (0.0, 0.25, 5.0), (0.0, 0.25, 0.0)),
    SceneInfo(14, "Sphere Cloud", "expensive_metric", "Union of 24 fixed spheres. Expensive metric SDF: costly eval, sound oracle.",
              (0.0, 0.0, 7.0), (0.0, 0.0, 0.0)),
    SceneInfo(15, "Bumpy Sphere", "expensive_metric", "Sphere + 30 bumps. Expensive metric SDF with grazing crawl + thin features.",
              (0.0, 0.0, 5.0), (0.0, 0.0, 0.0)),
    SceneInfo(16, "Gyroid", "periodic_surface",
              "Gyroid labyrinth clipped to a ball. Smooth periodic curved surface; grazing-rich, metric.",
              (0.0, 0.0, 6.0), (0.0, 0.0, 0.0)),
    SceneInfo(17, "Capped Torus", "thin_features",
              "Open C-shaped torus. Thin feature with a boundary edge → extra silhouette + grazing.",
              (0.0, 0.0, 4.5), (0.0, 0.0, 0.0)),
    SceneInfo(18, "Box Lattice", "near_miss",
              "Finite 5×5×5 box grid (metric). Many near-miss rays through the gaps; throughput + silhouette.",
              (0.0, 0.0, 7.0), (0.0, 0.0, 0.0)),
    SceneInfo(19, "Metaballs", "smooth", "Six spheres fused with polynomial smin. Smooth-union regime; metric (under-estimating).",
              (0.0, 0.0, 5.0), (0.0, 0.0, 0.0)),
]

# key -> (short_name, name, has_lipschitz); order is the reference's dict order
_STRATEGY_ROWS = [
    ("Standard", "Standard", "Standard Sphere Tracing", False),
    ("Relaxed", "Relaxed(ω=1.2)", "Relaxed Sphere Tracing (ω=1.2)", False),
    ("Heuristic-Auto-Relaxed", "AR-ST", "Auto-Relaxed Sphere Tracing", False),
    ("Slope-Auto-Relaxed", "Slope-AR(β=0.3)", "Slope-Based Auto-Relaxed (β=0.3)", False),
    ("Enhanced", "Enhanced", "Enhanced Sphere Tracing", False),
    ("Curvature", "Curvature-Aware Tracing", "Curvature-Aware Tracing", False),
    ("Overstep-Bisect", "Overstep-Bisect", "Overstep-Bisect", False),
    ("Skipping-Spheres", "Skipping-Spheres", "Skipping Spheres (Coarse->Fine)", False),
    ("RevAA", "RevAA", "RevAA (Interval Approx)", False),
    ("Adaptive-Hybrid", "Hybrid", "Adaptive Hybrid", False),
    ("Segment", "Segment", "Segment Tracing", True),
]
STRATEGIES = {row[0]: i for i, row in enumerate(_STRATEGY_ROWS)}   # key -> strategy id
# Strategies that exist only in the reference's fragment shader (gpu/shaders/strategies.glsl:508-541, :559-593; shader
# ids 8 and 9).  Kernel ids 11 and 12: the shader's control flow on the CPU path's arithmetic -- PARITY UNPINNED (no
# Python statement exists, the shader is fp32).  Not registry entries: `list_strategies()`, `--strategy all` and the
# CSV columns stay the reference's eleven; they are reached by key through get_shader_strategy / GPURunner.
_SHADER_ONLY_ROWS = [
    ("Safe-Relaxed", "Safe-Relaxed(ω=1.2)", "Safe Over-Relaxed Sphere Tracing (ω=1.2)", False),
    ("Dense-March", "Dense-March", "Dense March (calibration oracle)", False),
]
SHADER_ONLY_STRATEGIES = {row[0]: len(_STRATEGY_ROWS) + i for i, row in enumerate(_SHADER_ONLY_ROWS)}

# The nine strategies of the reference README (README.md:7-15) and the 14 scenes of its
# published matrix (example/matrix_*.csv) -- the graded 14 x 9 configuration.
GRADED_STRATEGY_KEYS = ["Standard", "Relaxed", "Heuristic-Auto-Relaxed", "Slope-Auto-Relaxed", "Enhanced",
                        "Curvature", "Overstep-Bisect", "Adaptive-Hybrid", "Segment"]
GRADED_SCENE_IDS = list(range(14))


# Constructor keywords of the reference's strategy classes -> RmStrategyParams fields (relaxed_sphere.py:17,
# auto_relaxed.py:21-23, slope_auto_relaxed.py:25, overstep_bisect.py:18, adaptive_hybrid.py:17-19,
# segment_tracing.py:26).  None = accepted and not read by march() (as in the reference).
CTOR_ARGS = {
    "Relaxed": {"omega": "omega"},
    "Heuristic-Auto-Relaxed": {"omega_min": "ar_omega_min", "omega_max": "ar_omega_max", "smoothing": "ar_smoothing",
                               "growth_rate": "ar_growth_rate", "decay_rate": "ar_decay_rate"},
    "Slope-Auto-Relaxed": {"beta": "beta"},
    "Overstep-Bisect": {"min_step_factor": "overstep_min_step", "bisection_steps": "overstep_bisection_steps"},
    "Adaptive-Hybrid": {"stuck_threshold": "hybrid_stuck_threshold", "stuck_step_ratio": "hybrid_stuck_step_ratio",
                        "min_step_factor": "hybrid_min_step", "bisection_steps": None, "fallback_to_segment_after": None},
    "Segment": {"lipschitz": "lipschitz"},
}
# constants the CPU strategies hold as literals, tunable here by their RmStrategyParams name
LITERAL_ARGS = {"Skipping-Spheres": ("margin",), "Heuristic-Auto-Relaxed": ("ar_omega_init",),
                "Segment": ("segment_bisection_steps",), "RevAA": ("revaa_bisection_steps",)}


def _make_strategy(i: int, **ctor) -> StrategyInfo:
    """`STRATEGIES[key](**ctor)` of the reference: the constructor's own keyword names (plus this engine's names for
    the literals).  short_name / name follow the instance like the reference's f-strings do."""
    key, short, name, has_l = _STRATEGY_ROWS[i]
    st = StrategyInfo(i, key, short, name, has_l)
    known = CTOR_ARGS.get(key, {})
    for k, v in ctor.items():
        if k in known:
            if known[k] == "lipschitz":
                st.lipschitz = float(v)
            elif known[k] is not None:
                st.params[known[k]] = v
        elif k in LITERAL_ARGS.get(key, ()):
            st.params[k] = v
        else:
            raise TypeError(f"{key}: unexpected constructor argument {k!r} (accepted: "
                            f"{sorted(known) + list(LITERAL_ARGS.get(key, ()))})")
    if key == "Relaxed" and "omega" in st.params:                  # relaxed_sphere.py:22,26
        w = st.params["omega"]
        st.short_name, st.name = f"Relaxed(ω={w})", f"Relaxed Sphere Tracing (ω={w})"
    if key == "Slope-Auto-Relaxed" and "beta" in st.params:         # slope_auto_relaxed.py:35,39
        b = st.params["beta"]
        st.short_name, st.name = f"Slope-AR(β={b})", f"Slope-Based Auto-Relaxed (β={b})"
    return st


def get_all_scenes() -> List[SceneInfo]:
    return list(SCENES)


def get_scene_by_name(name: str) -> Optional[SceneInfo]:
    """Reference lookup rule (catalog.py:666-681): lower-case, strip SPACES only, exact then
    starts-with; None on a miss ("Pillar_Forest" does not match, "Menger" does)."""
    low = name.lower().replace(" ", "")
    for s in SCENES:
        if s.name.lower().replace(" ", "") == low:
            return s
    for s in SCENES:
        if s.name.lower().replace(" ", "").startswith(low):
            return s
    return None


def get_strategy_by_name(name: str, **ctor) -> Optional[StrategyInfo]:
    """Reference lookup rule (strategies/__init__.py:31-45): case-insensitive exact match on the
    registry KEYS, then substring.  A fresh record per call, like the reference's `strat_class()`.
    Extension: an exact short_name ("AR-ST", "Hybrid", "Slope-AR(β=0.3)") is also accepted,
    tried after the reference's two rules so reference-valid names resolve identically.
    `ctor`: constructor arguments of the reference class (e.g. omega=1.6, min_step_factor=0.02), see CTOR_ARGS."""
    low = name.lower()
    for i, row in enumerate(_STRATEGY_ROWS):
        if row[0].lower() == low:
            return _make_strategy(i, **ctor)
    for i, row in enumerate(_STRATEGY_ROWS):
        if low in row[0].lower():
            return _make_strategy(i, **ctor)
    for i, row in enumerate(_STRATEGY_ROWS):
        if row[1].lower() == low:
            return _make_strategy(i, **ctor)
    return None


def get_shader_strategy(key: str, **params) -> Optional[StrategyInfo]:
    """One of the two shader-only strategies ("Safe-Relaxed": omega; "Dense-March": step_scale, dense_min_step) as a
    StrategyInfo whose id is the kernel id; `params` are RmStrategyParams names.  None for any other key."""
    low = key.lower().replace("_", "-")
    for i, row in enumerate(_SHADER_ONLY_ROWS):
        if row[0].lower() == low:
            st = StrategyInfo(len(_STRATEGY_ROWS) + i, row[0], row[1], row[2], row[3])
            allowed = {"Safe-Relaxed": ("omega",), "Dense-March": ("step_scale", "dense_min_step")}[row[0]]
            for k, v in params.items():
                if k not in allowed:
                    raise TypeError(f"{row[0]}: unexpected parameter {k!r} (accepted: {list(allowed)})")
                st.params[k] = v
            if row[0] == "Safe-Relaxed" and "omega" in st.params:
                w = st.params["omega"]
                st.short_name, st.name = f"Safe-Relaxed(ω={w})", f"Safe Over-Relaxed Sphere Tracing (ω={w})"
            return st
    return None


def list_strategies() -> List[str]:
    return [row[0] for row in _STRATEGY_ROWS]


def get_scenes_by_category(category: str) -> List[SceneInfo]:
    return [s for s in SCENES if s.category == category]
