"""MI355X-native per-ray SDF sphere-tracing engine.

Drop-in for ONE path of kylegrover/raymarch-algo-compare: the frame render
(camera rays -> scene SDF -> marching strategy -> depth / iterations / hit ->
frame statistics) behind the reference's scene / strategy registry names and CLI.
The compute lives in librm_hip.so (hand-written gfx950 kernels, C ABI in
include/rm_hip.h); this package is the thin ctypes host.  There is no CPU
fallback: importing works anywhere, rendering requires the HIP library and a GPU.
"""
from .config import MarchConfig, RenderConfig
from .registry import (SCENES, STRATEGIES, get_all_scenes, get_scene_by_name, get_strategy_by_name,
                       list_strategies)
from .camera import Camera
from .stats import RayMarchStats
from .collector import HipCollector
from .main import run_once

__all__ = [
    "MarchConfig", "RenderConfig", "SCENES", "STRATEGIES", "get_all_scenes", "get_scene_by_name",
    "get_strategy_by_name", "list_strategies", "Camera", "RayMarchStats", "HipCollector", "run_once",
]
