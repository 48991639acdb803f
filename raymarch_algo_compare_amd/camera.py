"""Host-side camera: the per-frame constants the kernels need.

Restates Camera.__init__ (reference core/camera.py:11-33) with the same Python float
expressions (`** 0.5` lengths, reciprocal-multiply normalisation, math.tan/radians), so the 14
doubles handed to the kernel are bit-identical to the reference's basis.  Per-pixel ray
generation (Camera.get_ray, camera.py:35-41) happens inside the kernel (csrc/rm_camera.h).
"""
from __future__ import annotations

import math

import numpy as np


def _normalized(v):
    x, y, z = v
    l = (x * x + y * y + z * z) ** 0.5          # vec3.py:46-47
    if l < 1e-12:                               # vec3.py:53-55
        return (0.0, 0.0, 0.0)
    inv = 1.0 / l                               # vec3.py:32-34
    return (x * inv, y * inv, z * inv)


def _cross(a, b):                               # vec3.py:39-44
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


class Camera:
    """Perspective camera; `position/target/up` are 3-tuples (or anything with x, y, z)."""

    def __init__(self, position, target, up, fov_degrees: float, width: int, height: int):
        def tup(v):
            if hasattr(v, "x"):
                return (float(v.x), float(v.y), float(v.z))
            return tuple(float(c) for c in v)

        self.position = tup(position)
        tgt, upv = tup(target), tup(up)
        self.width = int(width)
        self.height = int(height)
        p = self.position
        forward = _normalized((tgt[0] - p[0], tgt[1] - p[1], tgt[2] - p[2]))
        right = _normalized(_cross(forward, upv))
        true_up = _normalized(_cross(right, forward))
        self.forward, self.right, self.up = forward, right, true_up
        aspect = self.width / self.height
        fov_rad = math.radians(fov_degrees)
        self.half_height = math.tan(fov_rad / 2.0)
        self.half_width = aspect * self.half_height

    def params14(self) -> np.ndarray:
        """position, forward, right, up, half_width, half_height (RmFrameDesc.cam)."""
        return np.array([*self.position, *self.forward, *self.right, *self.up,
                         self.half_width, self.half_height], dtype=np.float64)
