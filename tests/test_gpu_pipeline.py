"""The single-launch pipeline (RmFrameDesc.pipeline = 2, csrc/rm_pipeline.h): producers, queue-0 consumers and
wavefront teams side by side in one kernel.  Who marches a ray, and when, depends on timing -- the results must
not: every knob combination is checked against the reference-generated goldens / the pinned oracle, bit for bit."""
import numpy as np
import pytest

from conftest import golden_frames, golden_param_cases, sha_f64
from test_gpu_parity import _check, _render

pytestmark = pytest.mark.gpu

KNOBS = [dict(), dict(team_grid=1), dict(team_grid=5, queue_first=2), dict(team_grid=300, team_steal=2), dict(queue_first=3, team_grid=7),
         dict(queue_first=1),
         dict(queue_first=1, queue_refill_min=1, queue_retry=1, team_retry=1), dict(queue_refill_min=64, queue_retry=50, team_retry=20),
         dict(grid_waves=4, team_grid=2), dict(grid_waves=1000, team_grid=64, refill_min=1), dict(resume_mode=1),
         dict(late_teams=3, team_grid=1), dict(late_teams=60, exit_backlog=1), dict(late_teams=5, exit_backlog=400, team_grid=2),
         dict(keep_busy=-1), dict(keep_busy=1, team_grid=2), dict(keep_busy=5000, grid_waves=1000, team_grid=3),
         dict(early_handover=-1), dict(early_handover=1, team_grid=4), dict(early_handover=30, keep_busy=-1),
         dict(early_trips=1, early_handover=1), dict(early_trips=6), dict(early_trips=64, team_grid=2)]


def test_mandelbulb_every_strategy_single_launch(hip):
    """Mandelbulb x 11 strategies at 64x48 under tiny budgets (nearly every ray crosses both queues), both
    evaluation modes, every knob set."""
    G = golden_frames("64x48")
    for kid in range(11):
        g = G.get(10, kid)
        for budgets in ((6, 40), (1, 2), (8, 0), (3, 200)):
            for i, knobs in enumerate(KNOBS):
                out = _render(hip, g, 10, kid, True, pipeline=2, suspend_after=budgets, eval_mode=1 + (i + kid) % 2, **knobs)
                assert _check(out, g, 10) == (0, 0), (kid, budgets, knobs)


def test_other_scenes_single_launch(hip):
    """The pipeline is generic: scenes without a team form run producers + queue 0 only; the union scenes have teams."""
    G = golden_frames("160x120")
    cells = [p for p in G.pairs if p[0] in (0, 2, 9, 12)]
    for sid, kid in cells:
        g = G.get(sid, kid)
        for sched in (dict(suspend_after=(4, 0)), dict(suspend_after=(5, 23), team_grid=3), dict(suspend_after=(2, 0), queue_first=2)):
            out = _render(hip, g, sid, kid, True, pipeline=2, **sched)
            assert _check(out, g, sid) == (0, 0), (sid, kid, sched)
    G = golden_frames("64x48")
    for sid in (1, 13, 14, 15, 16):
        for kid in (0, 5, 8, 10):
            g = G.get(sid, kid)
            for sched in (dict(suspend_after=(4, 19)), dict(suspend_after=(3, 0), team_grid=2), dict(suspend_after=(16, 64), team_steal=2)):
                out = _render(hip, g, sid, kid, True, pipeline=2, **sched)
                assert _check(out, g, sid) == (0, 0), (sid, kid, sched)


def test_full_frames_1080p_single_launch_equal_the_oracle(hip):
    """The bench configuration itself: Mandelbulb 1920x1080, three strategies, single launch at the default budgets
    and at two other settings -- every ray against the oracle."""
    import os
    from oracle import oracle
    from raymarch_algo_compare_amd import registry
    from raymarch_algo_compare_amd.camera import Camera
    W, H = 1920, 1080
    sc = registry.SCENES[10]
    cam = Camera(sc.camera_position, sc.camera_target, (0.0, 1.0, 0.0), 60.0, W, H).params14()
    nt = max(1, (os.cpu_count() or 2) - 1)
    for kid, scheds in ((0, (dict(), dict(suspend_after=(32, 64), team_grid=96, tile_order_mode=2), dict(suspend_after=(16, 48), queue_first=2, team_steal=2))),
                        (4, (dict(),)), (10, (dict(team_grid=200),))):
        ref = oracle.render(10, kid, cam, W, H, nthreads=nt)
        for sched in scheds:
            out = hip.render(hip.make_desc(10, kid, cam, W, H, full=True, pipeline=2, **sched), want_t_raw=True, want_final_sdf=True,
                             want_block_var=True)
            assert (out["iters"] == ref.iters).all() and (out["hit"] == ref.hit).all(), (kid, sched)
            assert (out["t_raw"].view(np.uint64) == ref.t.view(np.uint64)).all(), (kid, sched)
            assert (out["final_sdf"].view(np.uint64) == ref.final_sdf.view(np.uint64)).all(), (kid, sched)
            st = out["stats"]
            assert st["total_rays"] == W * H and st["sum_iters"] == int(ref.iters.sum(dtype=np.int64)) and st["hit_count"] == int(ref.hit.sum())
            assert (st["iter_hist"] == np.bincount(ref.iters.ravel(), minlength=len(st["iter_hist"]))).all()
            from raymarch_algo_compare_amd.stats import warp_divergence_from_block_var, warp_divergence_proxy
            assert warp_divergence_from_block_var(out["block_var"]) == warp_divergence_proxy(ref.iters)


def test_batches_shards_parameters_and_full_queues_single_launch(hip):
    """Everything that rides on a parked ray in the single launch: its frame (batches), its row shard, its strategy
    parameters, its evaluation count; and queues too small for the rays that want to park."""
    import math
    from oracle import oracle
    from raymarch_algo_compare_amd import registry
    from raymarch_algo_compare_amd.camera import Camera
    # batches with per-frame parameters
    groups = {}
    for sid, kid, prm, g in golden_param_cases():
        if sid == 10 and kid in (1, 6, 9):
            groups.setdefault(kid, []).append((prm, g))
    for kid, cases in groups.items():
        g0 = cases[0][1]
        cfgs = [dict(max_iterations=512, lipschitz=g["lipschitz"], params=prm) for prm, g in cases]
        shape = hip.make_desc(10, kid, g0["cam"], g0["W"], g0["H"], pipeline=2, suspend_after=(3, 11), team_grid=2)
        out = hip.render_batch(shape, np.stack([g["cam"] for _, g in cases]), cfgs)
        for i, (prm, g) in enumerate(cases):
            assert (out["iters"][i] == g["iters"]).all() and (out["hit"][i] == g["hit"]).all(), (kid, i)
    # ragged frames, row shards, band-cyclic shards
    sc = registry.SCENES[10]
    for w, h in ((100, 37), (67, 50)):
        cam = Camera(sc.camera_position, sc.camera_target, (0.0, 1.0, 0.0), 60.0, w, h).params14()
        ref = oracle.render(10, 0, cam, w, h)
        for kw in (dict(), dict(row0=8, rows=16), dict(row0=0, rows=12, band_rows=4, band_stride=3, band_offset=1)):
            out = hip.render(hip.make_desc(10, 0, cam, w, h, full=True, pipeline=2, suspend_after=(5, 30), **kw), want_t_raw=True,
                             want_evals=True)
            rows = ([kw["row0"] + ((y // 4) * 3 + 1) * 4 + y % 4 for y in range(kw["rows"])] if "band_rows" in kw
                    else list(range(kw.get("row0", 0), kw.get("row0", 0) + kw.get("rows", h))))
            assert (out["iters"] == ref.iters[rows]).all() and (out["t_raw"].view(np.uint64) == ref.t[rows].view(np.uint64)).all(), (w, h, kw)
            one = hip.render(hip.make_desc(10, 0, cam, w, h, full=True, suspend_after=(-1, -1), **kw), want_evals=True)
            assert (out["evals"] == one["evals"]).all()
    # queues of 100 entries
    G = golden_frames("160x120")
    L = hip.load()
    try:
        hip.check(L.rm_set_queue_capacity(100))
        for kid in (0, 6):
            g = G.get(10, kid)
            for sched in (dict(suspend_after=(2, 9)), dict(suspend_after=(3, 0)), dict(suspend_after=(2, 9), team_grid=1, team_steal=2)):
                assert _check(_render(hip, g, 10, kid, True, pipeline=2, **sched), g, 10) == (0, 0), (kid, sched)
    finally:
        hip.check(L.rm_set_queue_capacity(0))
    # temporal tile order across single-launch frames (costs of resumed rays arrive before or after the tile flush)
    g = G.get(10, 0)
    for _ in range(3):
        assert _check(_render(hip, g, 10, 0, False, pipeline=2, suspend_after=(8, 40), tile_order_mode=1), g, 10) == (0, 0)


def test_pass_marks_of_a_single_launch(hip):
    """rm_get_pass_ms splits a single-launch frame at the marks its waves leave: four non-negative spans."""
    import ctypes
    g = golden_frames("160x120").get(10, 0)
    L = hip.load()
    hip.check(L.rm_set_pass_timing(1))
    try:
        _render(hip, g, 10, 0, False, pipeline=2, suspend_after=(8, 40))
        n, ms = ctypes.c_int32(0), (ctypes.c_float * 4)()
        hip.check(L.rm_get_pass_ms(None, ctypes.byref(n), ms))
        assert n.value == 4 and all(ms[i] >= 0.0 for i in range(4)) and sum(ms) > 0.0
    finally:
        hip.check(L.rm_set_pass_timing(0))


def _two_stream_cells():
    G = golden_frames("160x120")
    mb = [k for s, k in G.pairs if s == 10]
    other = [(s, k) for s, k in G.pairs if s != 10][0]
    cells = [(10, mb[0], dict(pipeline=2, suspend_after=(6, 30))), (10, mb[1], dict(pipeline=1, suspend_after=(5, 25))),
             (other[0], other[1], dict(pipeline=2, suspend_after=(4, 12))),
             (10, mb[-1], dict(pipeline=2, suspend_after=(3, 9), tile_order_mode=1)), (10, mb[0], dict(suspend_after=(-1, -1)))]
    return G, cells


def test_frames_in_flight_on_two_streams_do_not_share_state(hip):
    """The parked-ray queues, control block and tile-cost maps are one workspace per device: frames enqueued back to back
    on two HIP streams (rm_render_device is asynchronous, nothing waits in between) must come out as if rendered
    alone -- the library orders them with an event.  Streams come from rm_stream_create, i.e. from the HIP runtime the
    library is bound to."""
    import ctypes
    L = hip.load()
    vp = ctypes.c_void_p
    G, cells = _two_stream_cells()
    streams = [vp(), vp()]
    for st in streams:
        hip.check(L.rm_stream_create(ctypes.byref(st)))
    try:
        for rnd in range(8):
            inflight = []
            for i, st in enumerate(streams):                    # two frames in flight, one per stream
                sid, kid, sched = cells[(2 * rnd + i) % len(cells)]
                g = G.get(sid, kid)
                w, h = g["W"], g["H"]
                desc = hip.make_desc(sid, kid, g["cam"], w, h, 0, h, g["max_iterations"], g["hit_threshold"],
                                     g["max_distance"], g["lipschitz"], False, **sched)
                p = [vp(), vp(), vp()]
                hip.check(L.rm_alloc_frame(w, h, *[ctypes.byref(q) for q in p]))
                hip.check(L.rm_render_device(ctypes.byref(desc), p[0], p[1], p[2], None, st))
                inflight.append((g, p, st, (sid, kid, sched)))
            for g, p, st, what in inflight:
                hip.check(L.rm_stream_synchronize(st))
                w, h = g["W"], g["H"]
                depth, iters, hit = np.empty((h, w), np.float32), np.empty((h, w), np.int32), np.empty((h, w), np.uint8)
                hip.check(L.rm_copy_frame_to_host(w, h, p[0], p[1], p[2], depth.ctypes.data_as(vp), iters.ctypes.data_as(vp),
                                                  hit.ctypes.data_as(vp)))
                hip.check(L.rm_free_frame(*p))
                assert (iters == g["iters"]).all() and (hit.astype(bool) == g["hit"].astype(bool)).all(), (rnd, what)
                assert float(np.abs(depth - g["depth"]).max()) <= 1e-5, (rnd, what)
    finally:
        for st in streams:
            hip.check(L.rm_stream_destroy(st))


def test_two_host_threads_two_streams(hip):
    """The boundary's threading claim (include/rm_hip.h "Threading"): two HOST THREADS, each with its own stream, drive the
    library at the same time -- alloc, asynchronous render, synchronise, copy back, free -- and every frame comes out as
    if rendered alone.  This is round 2's test that aborted on the MI355X with ONE variable changed: the streams are made
    by the runtime the library is bound to (rm_stream_create) instead of `ctypes.CDLL("libamdhip64.so")`, which in a
    process that imported PyTorch after librm_hip.so is a SECOND copy of the HIP runtime (DESIGN.md section 0, row
    "b threading")."""
    import ctypes
    import threading
    L = hip.load()
    vp = ctypes.c_void_p
    G, cells = _two_stream_cells()
    frames = {c[:2]: G.get(c[0], c[1]) for c in cells}          # npz access stays on the main thread
    results, errors = {}, []

    def worker(tid):
        try:
            stream = vp()
            hip.check(L.rm_stream_create(ctypes.byref(stream)))
            for rep in range(6):
                sid, kid, sched = cells[(tid * 2 + rep) % len(cells)]
                g = frames[(sid, kid)]
                w, h = g["W"], g["H"]
                desc = hip.make_desc(sid, kid, g["cam"], w, h, 0, h, g["max_iterations"], g["hit_threshold"],
                                     g["max_distance"], g["lipschitz"], False, **sched)
                p = [vp(), vp(), vp()]
                hip.check(L.rm_alloc_frame(w, h, *[ctypes.byref(q) for q in p]))
                hip.check(L.rm_render_device(ctypes.byref(desc), p[0], p[1], p[2], None, stream))
                hip.check(L.rm_stream_synchronize(stream))
                depth, iters, hit = np.empty((h, w), np.float32), np.empty((h, w), np.int32), np.empty((h, w), np.uint8)
                hip.check(L.rm_copy_frame_to_host(w, h, p[0], p[1], p[2], depth.ctypes.data_as(vp), iters.ctypes.data_as(vp),
                                                  hit.ctypes.data_as(vp)))
                hip.check(L.rm_free_frame(*p))
                results[(tid, rep)] = (sid, kid, bool((iters == g["iters"]).all() and (hit.astype(bool) == g["hit"].astype(bool)).all()
                                                      and float(np.abs(depth - g["depth"]).max()) <= 1e-5))
            hip.check(L.rm_stream_destroy(stream))
        except Exception as e:                                   # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 12 and all(ok for _, _, ok in results.values()), results


def test_a_stream_of_another_runtime_is_refused(hip):
    """librm_hip.so is loaded (the `hip` fixture), THEN PyTorch: its wheel bundles a libamdhip64 of its own, so the process
    holds two HIP runtimes -- round 2's SIGABRT ("b threading").  A handle that rm_stream_create did
    not make is then refused with RM_E_BAD_ARG and never handed to the runtime (which would dereference it: the handle here
    is the address of a host buffer); streams of rm_stream_create keep working."""
    import ctypes
    import torch                                            # noqa: F401 -- maps the second runtime
    L = hip.load()
    info = hip.runtime_info()
    assert "libamdhip64" in info["hip_runtime_path"] and info["hip_runtime_version"] > 0
    if info["hip_runtimes_loaded"] < 2:
        pytest.skip("one HIP runtime in this process (PyTorch was imported before librm_hip.so): nothing foreign to refuse")
    assert info["other_runtime_path"] and info["other_runtime_path"] != info["hip_runtime_path"]
    g = golden_frames("64x48").get(0, 0)
    desc = hip.make_desc(0, 0, g["cam"], g["W"], g["H"])
    vp = ctypes.c_void_p
    p = [vp(), vp(), vp()]
    hip.check(L.rm_alloc_frame(g["W"], g["H"], *[ctypes.byref(q) for q in p]))
    bogus = ctypes.create_string_buffer(4096)
    own = vp()
    hip.check(L.rm_stream_create(ctypes.byref(own)))
    try:
        assert L.rm_render_device(ctypes.byref(desc), p[0], p[1], p[2], None, ctypes.cast(bogus, vp)) == -6
        assert b"copies of the HIP runtime" in L.rm_last_error()
        assert L.rm_stream_synchronize(ctypes.cast(bogus, vp)) == -6
        hip.check(L.rm_render_device(ctypes.byref(desc), p[0], p[1], p[2], None, own))
        hip.check(L.rm_stream_synchronize(own))
        hip.check(L.rm_render_device(ctypes.byref(desc), p[0], p[1], p[2], None, None))     # the library stream too
        hip.check(L.rm_stream_synchronize(None))
    finally:
        hip.check(L.rm_stream_destroy(own))
        hip.check(L.rm_free_frame(*p))


def test_single_launch_after_another_queue_layout(hip):
    """The `ready` word of a queue entry is compared with the launch's generation tag; the queues are reused across frames
    with other entry strides (strategies) and by multi-pass frames, whose entries carry no tag.  Sequence: Segment single
    launch (largest entries), Standard single launch, a multi-pass frame, single launch again -- with both queues filled
    beforehand, and again in the middle, with words EQUAL to the tag of the next single launch (the worst stale content)."""
    import ctypes
    L = hip.load()
    G = golden_frames("64x48")
    nxt = ctypes.c_uint32(0)
    seq = [(10, dict(pipeline=2, suspend_after=(2, 6))), (0, dict(pipeline=2, suspend_after=(2, 6))),
           (4, dict(pipeline=1, suspend_after=(2, 6))), (0, dict(pipeline=2, suspend_after=(3, 9))),
           (10, dict(pipeline=2, suspend_after=(1, 3), queue_first=1)), (0, dict(pipeline=2, suspend_after=(1, 3), queue_first=1))]
    _render(hip, G.get(10, 10), 10, 10, True, pipeline=2, suspend_after=(2, 6))                 # the queues exist
    for rnd in range(2):
        hip.check(L.rm_debug_poison_queues(rnd, ctypes.byref(nxt)))
        assert nxt.value != 0
        for kid, sched in seq:
            g = G.get(10, kid)
            assert _check(_render(hip, g, 10, kid, True, **sched), g, 10) == (0, 0), (rnd, kid, sched)
        hip.check(L.rm_debug_poison_queues(0, None))                                            # tag of the very next launch
        g = G.get(10, 0)
        assert _check(_render(hip, g, 10, 0, True, pipeline=2, suspend_after=(2, 6), team_grid=3), g, 10) == (0, 0), rnd


def test_development_trace_of_a_single_launch(hip):
    """rm_debug_set_trace / rm_debug_get_trace (the aid behind DESIGN.md section 3 "What a trace shows"): every ray a team
    finishes leaves one record whose times are ordered (push <= pop <= end), whose iteration count is the frame's, and whose
    pixel carries a start stamp; the frame itself is unchanged by tracing."""
    import ctypes
    L = hip.load()
    g = golden_frames("160x120").get(10, 0)
    w, h = g["W"], g["H"]
    hip.check(L.rm_debug_set_trace(1))
    try:
        out = _render(hip, g, 10, 0, False, pipeline=2, suspend_after=(4, 12))
        assert _check(out, g, 10) == (0, 0)
        rec = np.empty((1 << 16, 8), np.uint32)
        n = ctypes.c_int64(0)
        start, detach = np.zeros(w * h, np.uint32), np.zeros(w * h, np.uint32)
        t0 = ctypes.c_uint32(0)
        vp = ctypes.c_void_p
        hip.check(L.rm_debug_get_trace(rec.ctypes.data_as(vp), len(rec), ctypes.byref(n), start.ctypes.data_as(vp), detach.ctypes.data_as(vp),
                                       w * h, ctypes.byref(t0)))
    finally:
        hip.check(L.rm_debug_set_trace(0))
    rec = rec[: n.value]
    assert 0 < n.value <= w * h
    gi = rec[:, 0].astype(np.int64)
    assert len(np.unique(gi)) == len(gi) and gi.max() < w * h                  # one record per ray
    assert (rec[:, 1] == g["iters"].reshape(-1)[gi]).all()                    # iterations of that pixel
    assert (rec[:, 2] <= rec[:, 3]).all() and (rec[:, 3] <= rec[:, 4]).all()  # push <= pop <= end (10 ns ticks since launch)
    assert (rec[:, 5] <= rec[:, 6]).all() and (rec[:, 6] >= 12).all()          # evaluations at the pop / at the end; handed over at 12 trips
    assert (start[gi] != 0).all() and (detach[gi] != 0).all()
    again = _render(hip, g, 10, 0, False, pipeline=2, suspend_after=(4, 12))   # tracing off: same frame
    assert (again["iters"] == out["iters"]).all()
