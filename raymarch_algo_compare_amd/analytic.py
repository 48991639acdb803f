"""Closed-form ray intersections for the four catalogue scenes that have one -- an independent known answer
for marched depth at any resolution (the role of the reference's gpu/analytic.py:74-208, here with the CPU
path's camera model so the maps line up with HipCollector / GPURunner output pixel for pixel).

    depth, hit, normal = analytic_depth("Thin Torus", camera)

depth is the ray parameter of the first intersection in front of the camera (0 where there is none), normal
the outward unit surface normal there.  Everything is float64 NumPy on the host: this module is a checker and
a ground-truth source for reports, not part of the render path.
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import numpy as np

from .camera import Camera

T_MIN = 1e-6     # roots closer than this are behind / at the eye


def camera_rays(camera: Camera) -> Tuple[np.ndarray, np.ndarray]:
    """(origin (3,), unit directions (H, W, 3)) of the pixel-centre rays (camera.py:35-41)."""
    c = camera.params14()
    u = (2.0 * (np.arange(camera.width) + 0.5) / camera.width - 1.0) * c[12]
    v = (1.0 - 2.0 * (np.arange(camera.height) + 0.5) / camera.height) * c[13]
    d = c[3:6][None, None, :] + c[6:9][None, None, :] * u[None, :, None] + c[9:12][None, None, :] * v[:, None, None]
    return c[0:3].copy(), d / np.sqrt((d * d).sum(2, keepdims=True))


def _finish(o, d, t, ok, grad):
    depth = np.where(ok, t, 0.0)
    n = np.zeros(d.shape)
    if ok.any():
        g = grad(o[None, :] + depth[ok][:, None] * d[ok])
        n[ok] = g / np.maximum(np.sqrt((g * g).sum(1, keepdims=True)), 1e-300)
    return depth, ok, n


def sphere(o, d, radius=1.0, center=(0.0, 0.0, 0.0)):
    oc = o - np.asarray(center, float)
    b = d @ oc
    disc = b * b - (oc @ oc - radius * radius)
    root = np.sqrt(np.maximum(disc, 0.0))
    near, far = -b - root, -b + root
    t = np.where(near > T_MIN, near, far)
    return _finish(o, d, t, (disc >= 0.0) & (t > T_MIN), lambda p: p - np.asarray(center, float))


def plane(o, d, normal=(0.0, 1.0, 0.0), offset=-0.5):
    """The plane dot(p, normal) = offset (Grazing Plane: y = -0.5)."""
    n = np.asarray(normal, float)
    dn = d @ n
    ok = np.abs(dn) > 1e-12
    t = np.where(ok, (offset - o @ n) / np.where(ok, dn, 1.0), 0.0)
    return _finish(o, d, t, ok & (t > T_MIN), lambda p: np.broadcast_to(n, p.shape))


def box(o, d, half=(1.0, 1.0, 1.0)):
    """Axis-aligned box [-half, half]: the slab method; the entered face gives the normal."""
    b = np.asarray(half, float)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        lo, hi = (-b - o) * inv, (b - o) * inv
    par = ~np.isfinite(inv)                                  # parallel to a slab: inside it or never
    t_in = np.where(par, -np.inf, np.minimum(lo, hi))
    t_out = np.where(par, np.inf, np.maximum(lo, hi))
    outside = (par & ((o < -b) | (o > b))).any(2)
    enter, leave = t_in.max(2), t_out.min(2)
    t = np.where(enter > T_MIN, enter, leave)
    ok = (enter <= leave) & ~outside & (t > T_MIN)

    def grad(p):
        q = np.abs(p) / b                                   # the face the point lies on has the largest |p| / half
        g = np.zeros(p.shape)
        ax = q.argmax(1)
        g[np.arange(len(p)), ax] = np.sign(p[np.arange(len(p)), ax])
        return g
    return _finish(o, d, t, ok, grad)


def torus(o, d, major=1.5, minor=0.05):
    """Torus around the y axis: smallest positive real root of the quartic (|P|^2 + R^2 - r^2)^2 = 4 R^2 (Px^2 + Pz^2),
    P = o + t d, for the rays that pass the bounding sphere; all candidates at once through the eigenvalues of
    the stacked companion matrices."""
    H, W, _ = d.shape
    od = d @ o
    cand = (o @ o - od * od) <= ((major + minor) * 1.01) ** 2
    t = np.zeros((H, W))
    ok = np.zeros((H, W), bool)
    if cand.any():
        dd = d[cand]
        s1 = 2.0 * od[cand]
        s0 = o @ o + major * major - minor * minor
        q2 = dd[:, 0] ** 2 + dd[:, 2] ** 2
        q1 = 2.0 * (o[0] * dd[:, 0] + o[2] * dd[:, 2])
        q0 = o[0] ** 2 + o[2] ** 2
        k = 4.0 * major * major
        c3, c2 = 2.0 * s1, s1 * s1 + 2.0 * s0 - k * q2
        c1, c0 = 2.0 * s1 * s0 - k * q1, s0 * s0 - k * q0 + 0.0 * s1
        comp = np.zeros((len(dd), 4, 4))
        comp[:, 0, :] = -np.stack([c3, c2, c1, c0], 1)       # monic quartic t^4 + c3 t^3 + c2 t^2 + c1 t + c0
        comp[:, 1, 0] = comp[:, 2, 1] = comp[:, 3, 2] = 1.0
        roots = np.linalg.eigvals(comp)
        real = np.where((np.abs(roots.imag) < 1e-7) & (roots.real > T_MIN), roots.real, np.inf)
        first = real.min(1)
        good = np.isfinite(first)
        t[cand] = np.where(good, first, 0.0)
        ok[cand] = good

    def grad(p):
        q = np.sqrt(p[:, 0] ** 2 + p[:, 2] ** 2)
        w = (q - major) / np.maximum(q, 1e-300)
        return np.stack([p[:, 0] * w, p[:, 1], p[:, 2] * w], 1)
    return _finish(o, d, t, ok, grad)


ANALYTIC_SCENES: Dict[str, Callable] = {
    "Sphere": lambda o, d: sphere(o, d, 1.0),
    "Grazing Plane": lambda o, d: plane(o, d, (0.0, 1.0, 0.0), -0.5),
    "Cube": lambda o, d: box(o, d, (1.0, 1.0, 1.0)),
    "Thin Torus": lambda o, d: torus(o, d, 1.5, 0.05),
}


def has_analytic(scene_name: str) -> bool:
    return scene_name in ANALYTIC_SCENES


def analytic_depth(scene_name: str, camera: Camera):
    """(depth (H, W) float64, hit (H, W) bool, normal (H, W, 3) float64) of an ANALYTIC_SCENES scene."""
    if scene_name not in ANALYTIC_SCENES:
        raise KeyError(f"no closed form for scene {scene_name!r}")
    o, d = camera_rays(camera)
    return ANALYTIC_SCENES[scene_name](o, d)
