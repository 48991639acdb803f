"""Render / march configuration records.

Field names, order and defaults are those of the reference's configuration dataclasses
(reference config.py:8-29) because they are part of the drop-in interface (`RenderConfig(width=...,
height=...)`, `MarchConfig(max_iterations=...)`, attribute access by name).  Only three MarchConfig
fields are read on the path -- max_iterations, hit_threshold, max_distance -- exactly as in the
reference's CPU strategies; the others are carried so reference call sites keep working.
"""
from dataclasses import field, make_dataclass


def _record(name, doc, fields):
    cls = make_dataclass(name, [(n, t, field(default=d)) for n, t, d in fields])
    cls.__doc__ = doc
    cls.__module__ = __name__
    return cls


RenderConfig = _record(
    "RenderConfig", "Resolution and camera of one frame.",
    [("width", int, 320), ("height", int, 240), ("fov_degrees", float, 60.0),
     ("camera_position", tuple, (0.0, 0.0, 5.0)), ("camera_target", tuple, (0.0, 0.0, 0.0)),
     ("camera_up", tuple, (0.0, 1.0, 0.0))])

MarchConfig = _record(
    "MarchConfig", "Marching parameters shared by every strategy.",
    [("max_iterations", int, 512), ("hit_threshold", float, 1e-4), ("max_distance", float, 100.0),
     # carried for interface parity; not read by the CPU-path strategies or by the kernels
     ("min_step_fraction", float, 0.01), ("kappa", float, 2.0), ("initial_relaxation", float, 1.6),
     ("bisection_steps", int, 10), ("stuck_threshold", int, 5), ("stuck_step_ratio", float, 0.001)])
