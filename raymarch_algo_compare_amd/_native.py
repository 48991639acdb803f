"""ctypes binding of librm_hip.so (C ABI: include/rm_hip.h).

Fails loudly: a missing library or a missing GPU raises RmError -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RM_HIP_LIB: a development build (csrc/Makefile DEV=1 -> librm_hip_dev.so, a few kernels only) for tools/; tests and the
# bench never set it
LIB_PATH = os.environ.get("RM_HIP_LIB") or os.path.join(_HERE, "librm_hip.so")

RM_NUM_SCENES = 20
RM_NUM_STRATEGIES = 11
RM_NUM_STRATEGY_KERNELS = 13      # + the two shader-only strategies (ids 11, 12; parity unpinned)
RM_HIST_BINS = 544
RM_MAX_TIMED = 256

ERROR_NAMES = {0: "RM_OK", -1: "RM_E_BAD_SCENE", -2: "RM_E_BAD_STRATEGY", -3: "RM_E_BAD_DIMS",
               -4: "RM_E_NO_DEVICE", -5: "RM_E_HIP", -6: "RM_E_BAD_ARG", -7: "RM_E_RCCL"}

EXPORTS = [
    "rm_init", "rm_shutdown", "rm_last_error", "rm_device_info", "rm_num_scenes", "rm_num_strategies",
    "rm_default_strategy_params",
    "rm_sdf_eval", "rm_march_rays", "rm_march_rays_team", "rm_render", "rm_render_outputs", "rm_render_device", "rm_stats_device_bytes",
    "rm_read_stats", "rm_bench_device", "rm_alloc_frame", "rm_free_frame", "rm_copy_frame_to_host",
    "rm_bench_store_path", "rm_render_batch", "rm_render_batch_outputs", "rm_set_pass_timing", "rm_get_pass_ms", "rm_last_queue_marks", "rm_long_ray_marks", "rm_set_queue_capacity",
    "rm_comm_unique_id", "rm_comm_init", "rm_comm_destroy", "rm_shard_rows", "rm_gather_frame", "rm_assemble_frame", "rm_gather_frame_root",
    "rm_runtime_info", "rm_stream_create", "rm_stream_synchronize", "rm_stream_destroy", "rm_debug_poison_queues",
    "rm_debug_set_trace", "rm_debug_get_trace",
]


class RmError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


# RmStrategyParams (include/rm_hip.h): the reference strategies' constructor arguments (relaxed_sphere.py:17,
# auto_relaxed.py:21-23, slope_auto_relaxed.py:25, overstep_bisect.py:18, adaptive_hybrid.py:17-19) and four
# literals of their march() bodies; (field, ctype, the reference's default).
STRATEGY_PARAM_FIELDS = [
    ("omega", ctypes.c_double, 1.2), ("ar_omega_min", ctypes.c_double, 1.0), ("ar_omega_max", ctypes.c_double, 2.0),
    ("ar_smoothing", ctypes.c_double, 0.7), ("ar_growth_rate", ctypes.c_double, 1.05),
    ("ar_decay_rate", ctypes.c_double, 0.7), ("beta", ctypes.c_double, 0.3),
    ("overstep_min_step", ctypes.c_double, 0.01), ("hybrid_stuck_step_ratio", ctypes.c_double, 0.001),
    ("hybrid_min_step", ctypes.c_double, 0.005), ("margin", ctypes.c_double, 0.05),
    ("ar_omega_init", ctypes.c_double, 1.2),
    ("overstep_bisection_steps", ctypes.c_int32, 16), ("hybrid_stuck_threshold", ctypes.c_int32, 5),
    ("segment_bisection_steps", ctypes.c_int32, 8), ("revaa_bisection_steps", ctypes.c_int32, 8),
    # uniforms only the reference's fragment shader has (strategies.glsl:24,47,570); the defaults change no bit
    ("step_scale", ctypes.c_double, 1.0), ("dense_min_step", ctypes.c_double, 1e-4),
]
DEFAULT_STRATEGY_PARAMS = {n: d for n, _, d in STRATEGY_PARAM_FIELDS}


class RmStrategyParams(ctypes.Structure):
    _fields_ = [(n, t) for n, t, _ in STRATEGY_PARAM_FIELDS]


class RmMarchConfig(ctypes.Structure):
    _fields_ = [("max_iterations", ctypes.c_int32), ("full", ctypes.c_int32),
                ("hit_threshold", ctypes.c_double), ("max_distance", ctypes.c_double),
                ("lipschitz", ctypes.c_double), ("use_params", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("params", RmStrategyParams)]


def fill_params(cfg: RmMarchConfig, params: dict | None) -> None:
    """Set cfg.params from a dict of RmStrategyParams overrides ({} / None: the reference's defaults)."""
    if not params:
        cfg.use_params = 0
        return
    vals = dict(DEFAULT_STRATEGY_PARAMS)
    for k, v in params.items():
        if k not in vals:
            raise KeyError(f"unknown strategy parameter {k!r} (RmStrategyParams has {sorted(vals)})")
        vals[k] = v
    for n, t, _ in STRATEGY_PARAM_FIELDS:
        setattr(cfg.params, n, int(vals[n]) if t is ctypes.c_int32 else float(vals[n]))
    cfg.use_params = 1


def march_config(max_iterations=512, full=False, hit_threshold=1e-4, max_distance=100.0, lipschitz=1.0,
                 params: dict | None = None) -> RmMarchConfig:
    c = RmMarchConfig()
    c.max_iterations, c.full = int(max_iterations), 1 if full else 0
    c.hit_threshold, c.max_distance, c.lipschitz = float(hit_threshold), float(max_distance), float(lipschitz)
    fill_params(c, params)
    return c


class RmFrameDesc(ctypes.Structure):
    _fields_ = [("scene_id", ctypes.c_int32), ("strategy_id", ctypes.c_int32),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("row0", ctypes.c_int32), ("rows", ctypes.c_int32),
                ("cam", ctypes.c_double * 14), ("march", RmMarchConfig),
                ("tile_rows", ctypes.c_int32), ("refill_min", ctypes.c_int32),
                ("grid_waves", ctypes.c_int32), ("band_rows", ctypes.c_int32),
                ("band_stride", ctypes.c_int32), ("band_offset", ctypes.c_int32),
                ("tile_order_mode", ctypes.c_int32), ("eval_mode", ctypes.c_int32),
                ("suspend_after", ctypes.c_int32 * 2),
                ("resume_grid", ctypes.c_int32), ("resume_mode", ctypes.c_int32),
                ("pipeline", ctypes.c_int32), ("team_grid", ctypes.c_int32), ("queue_first", ctypes.c_int32),
                ("team_steal", ctypes.c_int32), ("queue_refill_min", ctypes.c_int32), ("queue_retry", ctypes.c_int32),
                ("team_retry", ctypes.c_int32), ("age_priority", ctypes.c_int32),
                ("late_teams", ctypes.c_int32), ("exit_backlog", ctypes.c_int32),
                ("keep_busy", ctypes.c_int32), ("early_handover", ctypes.c_int32),
                ("early_trips", ctypes.c_int32), ("reserved1", ctypes.c_int32)]


class RmOutputs(ctypes.Structure):
    _fields_ = [("depth", ctypes.c_void_p), ("iters", ctypes.c_void_p), ("hit", ctypes.c_void_p),
                ("t_raw", ctypes.c_void_p), ("final_sdf", ctypes.c_void_p), ("block_var", ctypes.c_void_p),
                ("evals", ctypes.c_void_p)]


class RmStats(ctypes.Structure):
    _fields_ = [("total_rays", ctypes.c_uint64), ("hit_count", ctypes.c_uint64),
                ("sum_iters", ctypes.c_uint64), ("iter_max", ctypes.c_int32), ("iter_min", ctypes.c_int32),
                ("iter_hist", ctypes.c_uint64 * RM_HIST_BINS), ("sum_evals", ctypes.c_uint64)]


class RmTiming(ctypes.Structure):
    _fields_ = [("warmup", ctypes.c_int32), ("repeats", ctypes.c_int32),
                ("ms_median", ctypes.c_float), ("ms_mean", ctypes.c_float),
                ("ms_min", ctypes.c_float), ("ms_max", ctypes.c_float),
                ("ms_each", ctypes.c_float * RM_MAX_TIMED)]


class RmDeviceInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 128), ("arch", ctypes.c_char * 64),
                ("device_id", ctypes.c_int32), ("compute_units", ctypes.c_int32),
                ("clock_mhz", ctypes.c_int32), ("wavefront_size", ctypes.c_int32),
                ("total_mem_bytes", ctypes.c_uint64)]


class RmRuntimeInfo(ctypes.Structure):
    _fields_ = [("hip_runtime_path", ctypes.c_char * 512), ("hip_runtime_version", ctypes.c_int32),
                ("hip_driver_version", ctypes.c_int32), ("hip_runtimes_loaded", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("other_runtime_path", ctypes.c_char * 512)]


_lib = None
_lock = threading.Lock()
_device = None


def load() -> ctypes.CDLL:
    """dlopen librm_hip.so and declare the prototypes (no GPU call is made)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RmError(-4, f"{LIB_PATH} is missing: build it with "
                              f"`make -C {os.path.join(_HERE, 'csrc')} -j8` (or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        vp, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)
        L.rm_init.argtypes = [ctypes.c_int]
        L.rm_shutdown.restype = None
        L.rm_last_error.restype = ctypes.c_char_p
        L.rm_device_info.argtypes = [ctypes.POINTER(RmDeviceInfo)]
        L.rm_default_strategy_params.argtypes = [ctypes.POINTER(RmStrategyParams)]
        L.rm_default_strategy_params.restype = None
        L.rm_sdf_eval.argtypes = [ctypes.c_int, dp, ctypes.c_size_t, dp]
        L.rm_march_rays.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(RmMarchConfig), dp, dp,
                                    ctypes.c_size_t, vp, dp, vp, dp]
        L.rm_march_rays_team.argtypes = L.rm_march_rays.argtypes
        L.rm_render.argtypes = [ctypes.POINTER(RmFrameDesc), vp, vp, vp, vp, vp, vp,
                                ctypes.POINTER(RmStats), ctypes.POINTER(RmTiming)]
        L.rm_render_outputs.argtypes = [ctypes.POINTER(RmFrameDesc), ctypes.POINTER(RmOutputs), ctypes.POINTER(RmStats),
                                        ctypes.POINTER(RmTiming)]
        L.rm_render_device.argtypes = [ctypes.POINTER(RmFrameDesc), vp, vp, vp, vp, vp]
        L.rm_stats_device_bytes.restype = ctypes.c_size_t
        L.rm_read_stats.argtypes = [vp, vp, ctypes.POINTER(RmStats)]
        L.rm_bench_device.argtypes = [ctypes.POINTER(RmFrameDesc), vp, vp, vp, ctypes.POINTER(RmStats),
                                      ctypes.POINTER(RmTiming)]
        L.rm_render_batch.argtypes = [ctypes.POINTER(RmFrameDesc), ctypes.c_int32, dp, ctypes.POINTER(RmMarchConfig),
                                      vp, vp, vp, ctypes.POINTER(RmStats), ctypes.POINTER(ctypes.c_float)]
        L.rm_render_batch_outputs.argtypes = [ctypes.POINTER(RmFrameDesc), ctypes.c_int32, dp, ctypes.POINTER(RmMarchConfig),
                                              ctypes.POINTER(RmOutputs), ctypes.POINTER(RmStats), ctypes.POINTER(ctypes.c_float)]
        L.rm_alloc_frame.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                     ctypes.POINTER(vp)]
        L.rm_free_frame.argtypes = [vp, vp, vp]
        L.rm_copy_frame_to_host.argtypes = [ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, vp, vp]
        L.rm_bench_store_path.argtypes = [ctypes.c_int32, ctypes.c_int32, vp, vp, vp, ctypes.POINTER(RmTiming)]
        L.rm_set_pass_timing.argtypes = [ctypes.c_int]
        L.rm_set_queue_capacity.argtypes = [ctypes.c_int64]
        L.rm_get_pass_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float)]
        L.rm_last_queue_marks.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        L.rm_long_ray_marks.argtypes = [ctypes.POINTER(ctypes.c_float)]
        L.rm_comm_unique_id.argtypes = [ctypes.c_char_p]
        L.rm_comm_init.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32]
        L.rm_shard_rows.argtypes = [ctypes.c_int32, ctypes.c_int32]
        L.rm_gather_frame.argtypes = [ctypes.POINTER(RmFrameDesc), vp, vp, vp, vp, vp, vp, vp]
        L.rm_assemble_frame.argtypes = [ctypes.c_int32] * 6 + [vp, vp, vp]
        L.rm_gather_frame_root.argtypes = [ctypes.POINTER(RmFrameDesc), vp, vp, vp, vp, vp, vp, ctypes.c_int32, vp]
        L.rm_runtime_info.argtypes = [ctypes.POINTER(RmRuntimeInfo)]
        L.rm_stream_create.argtypes = [ctypes.POINTER(vp)]
        L.rm_stream_synchronize.argtypes = [vp]
        L.rm_stream_destroy.argtypes = [vp]
        L.rm_debug_poison_queues.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        L.rm_debug_set_trace.argtypes = [ctypes.c_int]
        L.rm_debug_get_trace.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), vp, vp, ctypes.c_int64,
                                         ctypes.POINTER(ctypes.c_uint32)]
        for name in EXPORTS:
            if name not in ("rm_shutdown", "rm_last_error", "rm_stats_device_bytes", "rm_default_strategy_params"):
                getattr(L, name).restype = ctypes.c_int
        _lib = L
        return L


def check(rc: int) -> None:
    if rc != 0:
        raise RmError(rc, load().rm_last_error().decode("utf-8", "replace"))


def init(device_id: int | None = None) -> ctypes.CDLL:
    """Bind the library to a GPU (default: $LOCAL_RANK or 0).  Raises RmError without one."""
    global _device
    L = load()
    if device_id is None:
        device_id = _device if _device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    if _device == device_id:
        return L
    if _device is not None:
        L.rm_shutdown()
        _device = None
    check(L.rm_init(int(device_id)))
    _device = device_id
    return L


def runtime_info() -> dict:
    """The HIP runtime librm_hip.so is bound to (no device needed): stream handles must come from this one."""
    L = load()
    info = RmRuntimeInfo()
    check(L.rm_runtime_info(ctypes.byref(info)))
    return {"hip_runtime_path": info.hip_runtime_path.decode(), "hip_runtime_version": int(info.hip_runtime_version),
            "hip_driver_version": int(info.hip_driver_version), "hip_runtimes_loaded": int(info.hip_runtimes_loaded),
            "other_runtime_path": info.other_runtime_path.decode()}


def device_info() -> dict:
    L = init()
    info = RmDeviceInfo()
    check(L.rm_device_info(ctypes.byref(info)))
    return {"name": info.name.decode(), "arch": info.arch.decode(), "device_id": info.device_id,
            "compute_units": info.compute_units, "clock_mhz": info.clock_mhz,
            "wavefront_size": info.wavefront_size, "total_mem_bytes": int(info.total_mem_bytes)}


def make_desc(scene_id, strategy_id, cam14, width, height, row0=0, rows=None, max_iterations=512,
              hit_threshold=1e-4, max_distance=100.0, lipschitz=1.0, full=False, tile_rows=0, refill_min=0,
              grid_waves=0, band_rows=0, band_stride=0, band_offset=0, tile_order_mode=0, eval_mode=0,
              suspend_after=(0, 0), resume_grid=0, resume_mode=0, params: dict | None = None, pipeline=0, team_grid=0,
              queue_first=0, team_steal=0, queue_refill_min=0, queue_retry=0, team_retry=0, age_priority=0, late_teams=0,
              exit_backlog=0, keep_busy=0, early_handover=0, early_trips=0) -> RmFrameDesc:
    d = RmFrameDesc()
    d.scene_id, d.strategy_id = int(scene_id), int(strategy_id)
    d.width, d.height = int(width), int(height)
    d.row0 = int(row0)
    d.rows = int(height - row0 if rows is None else rows)
    cam14 = np.asarray(cam14, dtype=np.float64).ravel()
    if cam14.size != 14:
        raise ValueError("cam14 must hold 14 doubles")
    for i in range(14):
        d.cam[i] = float(cam14[i])
    d.march.max_iterations = int(max_iterations)
    d.march.full = 1 if full else 0
    d.march.hit_threshold = float(hit_threshold)
    d.march.max_distance = float(max_distance)
    d.march.lipschitz = float(lipschitz)
    fill_params(d.march, params)
    d.tile_rows, d.refill_min, d.grid_waves = int(tile_rows), int(refill_min), int(grid_waves)
    d.band_rows, d.band_stride, d.band_offset = int(band_rows), int(band_stride), int(band_offset)
    d.tile_order_mode = int(tile_order_mode)
    d.eval_mode = int(eval_mode)
    d.suspend_after[0], d.suspend_after[1] = int(suspend_after[0]), int(suspend_after[1])
    d.resume_grid = int(resume_grid)
    d.resume_mode = int(resume_mode)
    d.pipeline, d.team_grid, d.queue_first, d.team_steal = int(pipeline), int(team_grid), int(queue_first), int(team_steal)
    d.queue_refill_min, d.queue_retry, d.team_retry = int(queue_refill_min), int(queue_retry), int(team_retry)
    d.age_priority = int(age_priority)
    d.late_teams, d.exit_backlog = int(late_teams), int(exit_backlog)
    d.keep_busy = int(keep_busy)
    d.early_handover = int(early_handover)
    d.early_trips = int(early_trips)
    return d


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def timing_dict(t: RmTiming) -> dict:
    return {"warmup": t.warmup, "repeats": t.repeats, "ms_median": float(t.ms_median),
            "ms_mean": float(t.ms_mean), "ms_min": float(t.ms_min), "ms_max": float(t.ms_max),
            "ms_each": [float(t.ms_each[i]) for i in range(t.repeats)]}


def stats_dict(s: RmStats) -> dict:
    return {"total_rays": int(s.total_rays), "hit_count": int(s.hit_count), "sum_iters": int(s.sum_iters),
            "iter_max": int(s.iter_max), "iter_min": int(s.iter_min), "sum_evals": int(s.sum_evals),
            "iter_hist": np.ctypeslib.as_array(s.iter_hist).astype(np.int64).copy()}


def render(desc: RmFrameDesc, want_t_raw=False, want_final_sdf=False, want_block_var=False, warmup=0,
           repeats=0, want_evals=False) -> dict:
    """rm_render_outputs into fresh NumPy arrays.  Returns depth (f32), iters (i32), hit (u8), optional
    t_raw / final_sdf (f64) / block_var (i64) / evals (i32), stats (dict) and timing (dict or None)."""
    L = init()
    rows, W = desc.rows, desc.width
    out = {"depth": np.empty((rows, W), np.float32), "iters": np.empty((rows, W), np.int32),
           "hit": np.empty((rows, W), np.uint8), "t_raw": None, "final_sdf": None, "block_var": None, "evals": None}
    if want_evals:
        out["evals"] = np.empty((rows, W), np.int32)
    if want_t_raw:
        out["t_raw"] = np.empty((rows, W), np.float64)
    if want_final_sdf:
        out["final_sdf"] = np.empty((rows, W), np.float64)
    if want_block_var:
        out["block_var"] = np.empty((rows // 4, W // 8), np.int64)
    st = RmStats()
    tm = None
    if repeats > 0:
        tm = RmTiming()
        tm.warmup, tm.repeats = int(warmup), int(repeats)
    def addr(a):
        return None if a is None else a.ctypes.data
    o = RmOutputs(addr(out["depth"]), addr(out["iters"]), addr(out["hit"]), addr(out["t_raw"]), addr(out["final_sdf"]),
                  addr(out["block_var"]), addr(out["evals"]))
    check(L.rm_render_outputs(ctypes.byref(desc), ctypes.byref(o), ctypes.byref(st),
                              ctypes.byref(tm) if tm is not None else None))
    out["stats"] = stats_dict(st)
    out["timing"] = timing_dict(tm) if tm is not None else None
    return out


def render_batch(shape: RmFrameDesc, cams, configs=None, want_evals=False) -> dict:
    """rm_render_batch_outputs: `cams` is (n, 14); `configs` an optional list of dicts / RmMarchConfig (one per
    frame).  Returns frame-major depth / iters / hit (/ evals) arrays (n, rows, W), per-frame stats and ms_total."""
    L = init()
    cams = np.ascontiguousarray(cams, dtype=np.float64).reshape(-1, 14)
    n = len(cams)
    rows, W = shape.rows, shape.width
    out = {"depth": np.empty((n, rows, W), np.float32), "iters": np.empty((n, rows, W), np.int32),
           "hit": np.empty((n, rows, W), np.uint8)}
    cfg_arr = None
    if configs is not None:
        if len(configs) != n:
            raise ValueError("one march config per frame")
        cfg_arr = (RmMarchConfig * n)()
        for i, c in enumerate(configs):
            if isinstance(c, RmMarchConfig):
                cfg_arr[i] = c
            else:
                cfg_arr[i] = march_config(c.get("max_iterations", 512), c.get("full"), c.get("hit_threshold", 1e-4),
                                          c.get("max_distance", 100.0), c.get("lipschitz", 1.0), c.get("params"))
    st = (RmStats * n)()
    ms = ctypes.c_float(0.0)
    out["evals"] = np.empty((n, rows, W), np.int32) if want_evals else None
    o = RmOutputs(out["depth"].ctypes.data, out["iters"].ctypes.data, out["hit"].ctypes.data, None, None, None,
                  out["evals"].ctypes.data if want_evals else None)
    check(L.rm_render_batch_outputs(ctypes.byref(shape), n, cams.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cfg_arr,
                                    ctypes.byref(o), st, ctypes.byref(ms)))
    out["stats"] = [stats_dict(st[i]) for i in range(n)]
    out["ms_total"] = float(ms.value)
    return out


def sdf_eval(scene_id: int, pts) -> np.ndarray:
    L = init()
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    out = np.empty(len(pts), np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    check(L.rm_sdf_eval(int(scene_id), pts.ctypes.data_as(dp), len(pts), out.ctypes.data_as(dp)))
    return out


def march_rays(scene_id, strategy_id, origins, dirs, max_iterations=512, hit_threshold=1e-4, max_distance=100.0,
               lipschitz=1.0, team=False, params: dict | None = None):
    L = init()
    origins = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    n = len(origins)
    cfg = march_config(max_iterations, True, hit_threshold, max_distance, lipschitz, params)
    hit, t = np.empty(n, np.uint8), np.empty(n, np.float64)
    iters, fs = np.empty(n, np.int32), np.empty(n, np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    fn = L.rm_march_rays_team if team else L.rm_march_rays
    check(fn(int(scene_id), int(strategy_id), ctypes.byref(cfg), origins.ctypes.data_as(dp),
             dirs.ctypes.data_as(dp), n, _ptr(hit), t.ctypes.data_as(dp), _ptr(iters), fs.ctypes.data_as(dp)))
    return hit, t, iters, fs
