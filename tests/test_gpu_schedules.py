"""Schedules never change results: evaluation mode (whole SDF evaluation vs one trip per turn),
long-ray suspension (rays parked at a trip budget and resumed from their strategy record) and
wavefront teams (three waves sharing the trips of the same 64 rays) are checked against the
reference-generated goldens and against each other, through the C ABI."""
import numpy as np
import pytest

from conftest import golden_frames, sha_f64
from test_gpu_parity import _check, _render

pytestmark = pytest.mark.gpu

# (suspend_after, resume_mode): off / one resume pass / two passes, single waves and teams
SCHEDULES = [dict(suspend_after=(-1, -1)), dict(suspend_after=(8, 0), resume_mode=1), dict(suspend_after=(8, 0), resume_mode=2),
             dict(suspend_after=(6, 40), resume_mode=1), dict(suspend_after=(6, 40), resume_mode=2),
             dict(suspend_after=(6, 40), resume_mode=3), dict()]


def test_mandelbulb_every_strategy_every_schedule(hip):
    """Mandelbulb x 11 strategies (reference goldens, 64x48) under every schedule and both evaluation modes:
    iterations / hits / raw fp64 t / final_sdf / stats identical to the reference each time."""
    G = golden_frames("64x48")
    for kid in range(11):
        g = G.get(10, kid)
        for sched in SCHEDULES:
            for em in (1, 2):
                out = _render(hip, g, 10, kid, True, eval_mode=em, **sched)
                assert _check(out, g, 10) == (0, 0), (kid, sched, em)


def test_suspension_other_scenes(hip):
    """Suspension is generic (the strategy record is the march state): algebraic scenes, strategies with
    multi-evaluation trips (Segment, RevAA, Hybrid, Overstep-Bisect, Slope), goldens at 160x120."""
    G = golden_frames("160x120")
    cells = [p for p in G.pairs if p[0] in (0, 9, 11, 12, 13)]
    assert len(cells) >= 10
    for sid, kid in cells:
        g = G.get(sid, kid)
        for sched in (dict(suspend_after=(4, 0)), dict(suspend_after=(5, 23))):
            out = _render(hip, g, sid, kid, True, **sched)
            assert _check(out, g, sid) == (0, 0), (sid, kid, sched)


def test_block_var_and_tile_cost_with_parked_rays(hip):
    """block_var (8x4 divergence numerators) comes from the finished map when rays were parked; the temporal
    tile order (per-tile cost from the previous frame) keeps working across the resume passes."""
    g = golden_frames("160x120").get(10, 0)
    ref = _render(hip, g, 10, 0, False, suspend_after=(-1, -1))
    it = g["iters"].astype(np.int64)
    H, W = it.shape
    blk = it[: H // 4 * 4, : W // 8 * 8].reshape(H // 4, 4, W // 8, 8)
    want = 32 * (blk * blk).sum(axis=(1, 3)) - blk.sum(axis=(1, 3)) ** 2
    assert (ref["block_var"] == want).all()
    for sched in (dict(suspend_after=(8, 0)), dict(suspend_after=(8, 40)), dict(suspend_after=(8, 40), tile_order_mode=1),
                  dict(suspend_after=(8, 40), tile_order_mode=1)):
        out = _render(hip, g, 10, 0, False, **sched)
        assert (out["block_var"] == want).all(), sched
        assert (out["iters"] == g["iters"]).all() and (out["hit"] == g["hit"]).all()


def test_budget_of_one_trip(hip):
    """A budget of 1 parks every ray that survives its first trip, a budget of 2 parks the survivors again:
    nearly the whole frame goes through both queues -- results unchanged."""
    g = golden_frames("64x48").get(10, 4)
    out = _render(hip, g, 10, 4, True, suspend_after=(1, 2))
    assert _check(out, g, 10) == (0, 0)


def test_march_rays_team_equals_single_wave(hip):
    """rm_march_rays_team == rm_march_rays bit for bit (explicit rays: every strategy, a frame's worth of
    rays including the 512-trip stragglers); scenes without a team form are refused."""
    g = golden_frames("64x48").get(10, 0)
    cam = g["cam"]
    W, H = g["W"], g["H"]
    px, py = np.meshgrid(np.arange(W), np.arange(H))
    u = (2.0 * (px.ravel() + 0.5) / W - 1.0) * cam[12]
    v = (1.0 - 2.0 * (py.ravel() + 0.5) / H) * cam[13]
    dirs = cam[3:6][None, :] + cam[6:9][None, :] * u[:, None] + cam[9:12][None, :] * v[:, None]
    orig = np.repeat(cam[0:3][None, :], len(dirs), 0)
    for kid in range(11):
        a = hip.march_rays(10, kid, orig, dirs)
        b = hip.march_rays(10, kid, orig, dirs, team=True)
        for x, y in zip(a, b):
            assert x.tobytes() == y.tobytes(), kid
    hit, t, iters, fs = hip.march_rays(10, 0, orig, dirs, team=True)
    assert (iters.reshape(H, W) == g["iters"]).all() and sha_f64(t.reshape(H, W)) == g["sha_t"]
    with pytest.raises(hip.RmError):
        hip.march_rays(0, 0, orig[:4], dirs[:4], team=True)


def test_bad_schedule_arguments(hip):
    g = golden_frames("64x48").get(0, 0)
    for bad in (dict(eval_mode=3), dict(resume_mode=4), dict(resume_grid=-1)):
        with pytest.raises(hip.RmError):
            _render(hip, g, 0, 0, False, **bad)


def test_ragged_frames_and_row_shards_with_parked_rays(hip):
    """Widths that are no multiple of 64 / 8, heights no multiple of 4, row shards and band-cyclic shards:
    parked rays carry their output index, so every layout must reproduce the oracle frame."""
    from oracle import oracle
    from raymarch_algo_compare_amd import registry
    from raymarch_algo_compare_amd.camera import Camera
    sc = registry.SCENES[10]
    for w, h in ((100, 37), (67, 50)):
        cam = Camera(sc.camera_position, sc.camera_target, (0.0, 1.0, 0.0), 60.0, w, h).params14()
        ref = oracle.render(10, 0, cam, w, h)
        for kw in (dict(), dict(row0=8, rows=16), dict(row0=0, rows=12, band_rows=4, band_stride=3, band_offset=1)):
            for sched in (dict(suspend_after=(5, 30), resume_mode=2), dict(suspend_after=(5, 30), resume_mode=1), dict(suspend_after=(7, 0))):
                out = hip.render(hip.make_desc(10, 0, cam, w, h, full=True, **kw, **sched), want_t_raw=True, want_final_sdf=True)
                if "band_rows" in kw:
                    rows = [kw["row0"] + ((y // 4) * 3 + 1) * 4 + y % 4 for y in range(kw["rows"])]
                else:
                    rows = list(range(kw.get("row0", 0), kw.get("row0", 0) + kw.get("rows", h)))
                assert (out["iters"] == ref.iters[rows]).all() and (out["hit"] == ref.hit[rows]).all(), (w, h, kw, sched)
                assert (out["t_raw"].view(np.uint64) == ref.t[rows].view(np.uint64)).all(), (w, h, kw, sched)
                assert (out["final_sdf"].view(np.uint64) == ref.final_sdf[rows].view(np.uint64)).all(), (w, h, kw, sched)
                assert out["stats"]["sum_iters"] == int(ref.iters[rows].sum())


def test_batch_with_parked_rays(hip):
    """rm_render_batch with suspension: parked rays of different frames share the queues (the entry's output
    index names the frame; camera and march configuration are looked up per ray on resume)."""
    import math
    from raymarch_algo_compare_amd.camera import Camera
    w, h = 160, 90
    cams, cfgs = [], []
    for i in range(6):
        ang = 2.0 * math.pi * i / 6.0
        cams.append(Camera((3.0 * math.sin(ang), 0.5 * math.cos(2 * ang), 3.0 * math.cos(ang)), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0),
                           60.0, w, h).params14())
        cfgs.append(dict(max_iterations=[512, 200, 77][i % 3], hit_threshold=[1e-4, 1e-3][i % 2], max_distance=100.0, lipschitz=1.0))
    for kid in (0, 4):
        for sched in (dict(suspend_after=(6, 40), resume_mode=2), dict(suspend_after=(6, 0), resume_mode=1), dict()):
            out = hip.render_batch(hip.make_desc(10, kid, cams[0], w, h, **sched), np.array(cams), cfgs)
            for i in range(6):
                one = hip.render(hip.make_desc(10, kid, cams[i], w, h, suspend_after=(-1, -1), **cfgs[i]))
                assert (out["iters"][i] == one["iters"]).all() and (out["hit"][i] == one["hit"]).all(), (kid, sched, i)
                assert (out["depth"][i].view(np.uint32) == one["depth"].view(np.uint32)).all(), (kid, sched, i)
                assert out["stats"][i]["sum_iters"] == one["stats"]["sum_iters"]


def test_full_queue_leaves_rays_in_place(hip):
    """rm_set_queue_capacity bounds the parked-ray queues; rays that find them full march on in their lanes.
    100 entries against ~10 000 rays that want to park (twice): results unchanged."""
    G = golden_frames("160x120")
    L = hip.load()
    try:
        hip.check(L.rm_set_queue_capacity(100))
        for kid in (0, 4, 6):
            g = G.get(10, kid)
            for sched in (dict(suspend_after=(2, 9), resume_mode=2), dict(suspend_after=(3, 0), resume_mode=1), dict(suspend_after=(2, 9), resume_mode=3)):
                out = _render(hip, g, 10, kid, True, **sched)
                assert _check(out, g, 10) == (0, 0), (kid, sched)
        g = G.get(12, 0)
        assert _check(_render(hip, g, 12, 0, True, suspend_after=(3, 11)), g, 12) == (0, 0)
    finally:
        hip.check(L.rm_set_queue_capacity(0))
    with pytest.raises(hip.RmError):
        hip.check(L.rm_set_queue_capacity(-1))


def test_union_scenes_team_form(hip):
    """Sphere Cloud / Bumpy Sphere: a team evaluates the union three ways (each wave a third of the sphere
    list, minimum of the three).  Goldens at 64x48 for every strategy under single-wave and team schedules,
    and rm_march_rays_team against rm_march_rays."""
    G = golden_frames("64x48")
    for sid in (14, 15):
        for kid in range(11):
            g = G.get(sid, kid)
            for sched in (dict(suspend_after=(-1, -1)), dict(suspend_after=(4, 0), resume_mode=2), dict(suspend_after=(4, 19), resume_mode=2),
                          dict(suspend_after=(4, 19), resume_mode=3), dict(suspend_after=(5, 0), resume_mode=1), dict(eval_mode=2), dict()):
                out = _render(hip, g, sid, kid, True, **sched)
                assert _check(out, g, sid) == (0, 0), (sid, kid, sched)
        g = G.get(sid, 0)
        cam, W, H = g["cam"], g["W"], g["H"]
        px, py = np.meshgrid(np.arange(W), np.arange(H))
        u = (2.0 * (px.ravel() + 0.5) / W - 1.0) * cam[12]
        v = (1.0 - 2.0 * (py.ravel() + 0.5) / H) * cam[13]
        dirs = cam[3:6][None, :] + cam[6:9][None, :] * u[:, None] + cam[9:12][None, :] * v[:, None]
        orig = np.repeat(cam[0:3][None, :], len(dirs), 0)
        a, b = hip.march_rays(sid, 0, orig, dirs), hip.march_rays(sid, 0, orig, dirs, team=True)
        assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
        assert (b[2].reshape(H, W) == g["iters"]).all()


def test_randomised_parity_campaign(hip):
    """150 random (scene, strategy, frame, camera, march configuration, schedule) cases of tests/fuzz_parity.py
    against the pinned oracle: every combination of evaluation mode, suspension budgets, resume mode, refill
    threshold, grid sizes, tile order and row shard must give the reference's bits."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(2024)
    for _ in range(150):
        ok, info = fz.one_case(rng)
        assert ok, info
