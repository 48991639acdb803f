"""Render / march configuration -- same fields and defaults as the reference's config.py:8-29."""
from dataclasses import dataclass


@dataclass
class RenderConfig:
    """Rendering resolution and camera settings (reference config.py:8-15)."""
    width: int = 320
    height: int = 240
    fov_degrees: float = 60.0
    camera_position: tuple = (0.0, 0.0, 5.0)
    camera_target: tuple = (0.0, 0.0, 0.0)
    camera_up: tuple = (0.0, 1.0, 0.0)


@dataclass
class MarchConfig:
    """Marching parameters (reference config.py:19-29).  Only max_iterations, hit_threshold and
    max_distance are read by the reference's CPU strategies; the rest are carried for API parity."""
    max_iterations: int = 512
    hit_threshold: float = 1e-4
    max_distance: float = 100.0
    min_step_fraction: float = 0.01
    kappa: float = 2.0
    initial_relaxation: float = 1.6
    bisection_steps: int = 10
    stuck_threshold: int = 5
    stuck_step_ratio: float = 0.001
