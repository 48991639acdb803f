"""Curated viewpoints per scene -- the reference's table (viewpoints.py:41-123) as data.

The reference frames every core scene from two or three hand-picked cameras in four categories
(orthogonal / grazing / macro / interior) because grazing behaviour is viewpoint-sensitive; scenes
without an entry fall back to their suggested camera (viewpoints.py:126-140).  The engine renders all
viewpoints of a (scene, strategy) -- times all sweep levels -- in ONE batched launch (sweep.py).

Table format: scene name -> "name category x y z [-> tx ty tz]" entries separated by ';'.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

from .config import RenderConfig

_TABLE: Dict[str, str] = {
    "Sphere": "ortho orthogonal 0 0 3.2; macro macro 0 0 1.7",
    "Grazing Plane": "steep orthogonal 0 4 4 -> 0 -0.5 0; grazing grazing 0 0.6 8 -> 0 -0.4 0; "
                     "extreme-grazing grazing 0 0.28 13 -> 0 -0.46 0",
    "Cube": "face orthogonal 0 0 3.6; corner orthogonal 2.4 2 2.6; grazing-face grazing 3.4 0 0.5",
    "Thin Torus": "ring-face orthogonal 0 0 4; grazing-edge grazing 0 0.35 4; macro macro 0 0 2.2",
    "Mandelbulb": "ortho orthogonal 0 0 3; macro macro 0 0 1.9; angled orthogonal 2 1.4 2",
    "Cylinder": "side orthogonal 0 0 4.2; cap-grazing grazing 3.6 1.55 0.6; macro macro 0 0 2.4",
    "Near Miss": "ortho orthogonal 0 0 5.2; gap-grazing grazing 0 2.6 4.4; macro-gap macro 0 0 3",
    "Hollow Cube (CSG)": "face orthogonal 0 0 4; corner orthogonal 2.4 2 2.6; grazing-face grazing 3.4 0 0.5",
    "Onion Shell": "ortho orthogonal 0 0 5.4; grazing grazing 0 0.55 5.4; macro macro 0 0 3.2",
    "Thin Planes Stack": "ortho orthogonal 0 0.25 5 -> 0 0.25 0; grazing grazing 0 0.12 6.2 -> 0 0.05 0; "
                         "edge orthogonal 3 0.25 4 -> 0 0.25 0",
    "Sphere Cloud": "ortho orthogonal 0 0 7; angled orthogonal 4.2 3.2 5.2; macro macro 0 0 4.6",
    "Bumpy Sphere": "ortho orthogonal 0 0 5; macro macro 0 0 2.6; angled orthogonal 3.2 2.2 3.2",
    "Gyroid": "ortho orthogonal 0 0 6; grazing grazing 5.6 0.5 1.8; macro macro 0 0 3.4",
    "Capped Torus": "face orthogonal 0 0 4.5; grazing-edge grazing 4.3 0.4 1; gap orthogonal 0 3.4 3.2",
    "Box Lattice": "ortho orthogonal 0 0 7; diagonal orthogonal 5.5 5.5 5.5; grazing-row grazing 7 0.35 0.7",
    "Metaballs": "ortho orthogonal 0 0 5; macro macro 0 0 3; angled orthogonal 3 2 3",
    "Menger Sponge (iter=3)": "ortho orthogonal 0 0 4; corner orthogonal 2.4 2 2.6; grazing-face grazing 3.6 0 0.6",
}

CATEGORIES = ("orthogonal", "grazing", "macro", "interior")


@dataclass(frozen=True)
class Viewpoint:
    name: str
    category: str
    position: Tuple[float, float, float]
    target: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    up: Tuple[float, float, float] = (0.0, 1.0, 0.0)

    def render_config(self, width: int, height: int) -> RenderConfig:
        return RenderConfig(width=width, height=height, camera_position=self.position, camera_target=self.target,
                            camera_up=self.up)


def _parse(entry: str) -> Viewpoint:
    head, _, tail = entry.partition("->")
    f = head.split()
    pos = tuple(float(v) for v in f[2:5])
    tgt = tuple(float(v) for v in tail.split()) if tail.strip() else (0.0, 0.0, 0.0)
    if f[1] not in CATEGORIES or len(pos) != 3 or len(tgt) != 3:
        raise ValueError(f"bad viewpoint entry {entry!r}")
    return Viewpoint(f[0], f[1], pos, tgt)


_CURATED: Dict[str, List[Viewpoint]] = {k: [_parse(e) for e in v.split(";")] for k, v in _TABLE.items()}


def viewpoints_for(scene) -> List[Viewpoint]:
    """Curated viewpoints of a scene, else its suggested camera, else (0, 0, 5) looking at the origin."""
    vps = _CURATED.get(scene.name)
    if vps:
        return list(vps)
    sc = scene.suggested_camera()
    if sc is not None:
        return [Viewpoint("default", "orthogonal", tuple(sc.camera_position), tuple(sc.camera_target), tuple(sc.camera_up))]
    return [Viewpoint("default", "orthogonal", (0.0, 0.0, 5.0))]
