"""Full-size (1920x1080 and larger) checks on the GPU through the C ABI.  The reference cannot run
at this size in test time, so these use size-independent properties plus oracle spot checks:
row shards and band-cyclic shards reassemble to the unsharded frame bit-for-bit, runs are
deterministic, the in-kernel frame reduce equals the maps, schedule parameters never change
results, sampled rows equal the CPU oracle, and the Sphere agrees with the closed-form depth."""
import numpy as np
import pytest

from raymarch_algo_compare_amd import registry, sharding
from raymarch_algo_compare_amd.camera import Camera
from raymarch_algo_compare_amd.stats import warp_divergence_from_block_var, warp_divergence_proxy

pytestmark = pytest.mark.gpu

W, H = 1920, 1080
GRADED = [registry.STRATEGIES[k] for k in registry.GRADED_STRATEGY_KEYS]


def _cam(scene, w=W, h=H):
    return Camera(scene.camera_position or (0.0, 0.0, 5.0), scene.camera_target or (0.0, 0.0, 0.0),
                  (0.0, 1.0, 0.0), 60.0, w, h)


def _lip(sid, kid):
    sc = registry.SCENES[sid]
    return sc.lipschitz if (kid == registry.STRATEGIES["Segment"] and sc.lipschitz) else 1.0


def _render(hip, sid, kid, w=W, h=H, **kw):
    want_bv = kw.pop("want_block_var", False)
    desc = hip.make_desc(sid, kid, _cam(registry.SCENES[sid], w, h).params14(), w, h, lipschitz=_lip(sid, kid), **kw)
    return hip.render(desc, want_t_raw=True, want_block_var=want_bv)


def _same(a, b):
    return ((a["iters"] == b["iters"]).all() and (a["hit"] == b["hit"]).all()
            and (a["depth"].view(np.uint32) == b["depth"].view(np.uint32)).all()
            and (a["t_raw"].view(np.uint64) == b["t_raw"].view(np.uint64)).all())


@pytest.mark.parametrize("sid,kid", [(0, 0), (2, 10), (10, 0), (10, 6), (9, 4), (12, 0)])
def test_row_shards_reassemble(hip, sid, kid):
    full = _render(hip, sid, kid)
    for N in (2, 8):
        parts_c, parts_b, plans_c, plans_b = [], [], [], []
        for r in range(N):
            per = ((H // 4 + N - 1) // N) * 4                     # contiguous 4-aligned blocks
            r0, r1 = min(r * per, H), min((r + 1) * per, H)
            parts_c.append(_render(hip, sid, kid, row0=r0, rows=r1 - r0))
            plans_c.append((r0, r1))
        rebuilt = {k: np.concatenate([p[k] for p in parts_c], axis=0) for k in ("iters", "hit", "depth", "t_raw")}
        assert _same(rebuilt, full), ("contiguous", N)
        if H % (4 * N) == 0:
            for r in range(N):
                plan = sharding.plan_rows(H, N, r)
                assert plan.cyclic
                parts_b.append(_render(hip, sid, kid, **plan.desc_kwargs()))
                plans_b.append(plan)
            rebuilt = {k: sharding.assemble([p[k] for p in parts_b], plans_b) for k in ("iters", "hit", "depth", "t_raw")}
            assert _same(rebuilt, full), ("band-cyclic", N)
            assert sum(p["stats"]["sum_iters"] for p in parts_b) == full["stats"]["sum_iters"]


def test_deterministic_and_schedule_invariant(hip):
    for sid, kid in ((10, 0), (13, 9), (5, 5)):
        ref = _render(hip, sid, kid)
        for kw in ({}, dict(refill_min=1), dict(refill_min=64), dict(tile_rows=4), dict(grid_waves=64), dict(grid_waves=100000),
                   dict(tile_order_mode=1), dict(tile_order_mode=1)):   # first call records costs, second uses them
            assert _same(_render(hip, sid, kid, **kw), ref), (sid, kid, kw)


def test_in_kernel_reduce_matches_maps_all_graded_cells(hip):
    """All 14 x 9 graded cells at 1080p: stats block, histogram and block variances vs the maps."""
    for sid in registry.GRADED_SCENE_IDS:
        for kid in GRADED:
            out = _render(hip, sid, kid, want_block_var=True)
            st, it = out["stats"], out["iters"]
            assert st["total_rays"] == W * H
            assert st["hit_count"] == int(out["hit"].sum()) and st["sum_iters"] == int(it.sum(dtype=np.int64))
            assert st["iter_max"] == int(it.max()) and st["iter_min"] == int(it.min())
            assert (st["iter_hist"] == np.bincount(it.reshape(-1), minlength=len(st["iter_hist"]))).all()
            assert warp_divergence_from_block_var(out["block_var"]) == warp_divergence_proxy(it)
            assert (out["depth"] == np.where(out["hit"] > 0, out["t_raw"], 0.0).astype(np.float32)).all()
            assert it.max() <= 512 + 9


def test_sampled_rows_match_oracle_at_1080p(hip):
    """Eight scattered rows of every graded scene (Standard + one other strategy) against the CPU
    oracle at full resolution: bit-exact iterations / hits / t."""
    from oracle import oracle
    rows = [0, 137, 401, 539, 540, 777, 1001, 1079]
    for sid in registry.GRADED_SCENE_IDS:
        for kid in (0, GRADED[(sid % 8) + 1]):
            cam = _cam(registry.SCENES[sid]).params14()
            for r in rows:
                out = hip.render(hip.make_desc(sid, kid, cam, W, H, row0=r, rows=1, lipschitz=_lip(sid, kid)), want_t_raw=True)
                ref = oracle.render(sid, kid, cam, W, H, row0=r, rows=1, lipschitz=_lip(sid, kid))
                assert (out["iters"] == ref.iters).all() and (out["hit"] == ref.hit).all(), (sid, kid, r)
                assert (out["t_raw"].view(np.uint64) == ref.t.view(np.uint64)).all(), (sid, kid, r)


def test_sphere_depth_matches_closed_form(hip):
    """Independent known answer: ray / unit-sphere intersection in closed form (float64).  A hit is
    declared within hit_threshold of the surface, so the marched depth may stop short of the root by
    at most ~1e-4 / cos(incidence); every strategy must agree with the analytic silhouette."""
    sc = registry.SCENES[0]
    c = _cam(sc).params14()
    pos, fwd, right, up, hw, hh = c[0:3], c[3:6], c[6:9], c[9:12], c[12], c[13]
    u = (2.0 * (np.arange(W) + 0.5) / W - 1.0) * hw
    v = (1.0 - 2.0 * (np.arange(H) + 0.5) / H) * hh
    d = fwd[None, None, :] + right[None, None, :] * u[None, :, None] + up[None, None, :] * v[:, None, None]
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    b = (d * pos).sum(2)
    disc = b * b - ((pos * pos).sum() - 1.0)
    t_exact = -b - np.sqrt(np.maximum(disc, 0.0))
    inside = disc > 1e-3          # clearly inside the silhouette
    outside = disc < -1e-3        # clearly outside
    for kid in GRADED:
        out = _render(hip, 0, kid)
        assert (out["hit"][inside] == 1).all() and (out["hit"][outside] == 0).all(), kid
        err = np.abs(out["t_raw"][inside] - t_exact[inside])
        assert err.max() < 5e-3, (kid, err.max())
        assert np.median(err) < 1.5e-4, (kid, np.median(err))


def _rays(sid):
    c = _cam(registry.SCENES[sid]).params14()
    pos, fwd, right, up, hw, hh = c[0:3], c[3:6], c[6:9], c[9:12], c[12], c[13]
    u = (2.0 * (np.arange(W) + 0.5) / W - 1.0) * hw
    v = (1.0 - 2.0 * (np.arange(H) + 0.5) / H) * hh
    d = fwd[None, None, :] + right[None, None, :] * u[None, :, None] + up[None, None, :] * v[:, None, None]
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    return pos, d


def test_plane_and_cube_match_closed_form(hip):
    """Independent known answers (cf. the reference's gpu/analytic.py idea, CPU camera model):
    Grazing Plane = the plane y = -0.5, Cube = the slab intersection with [-1, 1]^3."""
    # plane: t = (-0.5 - o.y) / d.y for d.y < 0
    pos, d = _rays(1)
    t_plane = np.where(d[..., 1] < 0, (-0.5 - pos[1]) / np.minimum(d[..., 1], -1e-300), np.inf)
    sure_hit = t_plane < 90.0          # well inside max_distance = 100
    sure_miss = ~(d[..., 1] < 0)
    for kid in GRADED:
        out = _render(hip, 1, kid)
        hits = out["hit"] > 0
        assert not hits[sure_miss].any(), kid
        ok = hits & sure_hit
        assert ok.sum() > 0.3 * sure_hit.sum(), (kid, ok.sum(), sure_hit.sum())      # grazing rays may exhaust 512 steps
        err = np.abs(out["t_raw"][ok] - t_plane[ok]) * np.abs(d[..., 1][ok])          # distance to the plane at the stop
        assert err.max() < 1.2e-4, (kid, err.max())
    # cube: slab method
    pos, d = _rays(2)
    inv = 1.0 / np.where(np.abs(d) < 1e-300, 1e-300, d)
    t1, t2 = (-1.0 - pos) * inv, (1.0 - pos) * inv
    tn = np.minimum(t1, t2).max(axis=2)
    tf = np.maximum(t1, t2).min(axis=2)
    margin = tf - tn
    inside, outside = margin > 2e-2, margin < -2e-2
    for kid in GRADED:
        out = _render(hip, 2, kid)
        assert (out["hit"][outside] == 0).all(), kid
        # the two over-relaxing strategies miss part of the cube in the reference itself (SURVEY Appendix A:
        # Relaxed and AR-ST hit 324 of 400 pixels at 64x48); every other strategy must hit all thick chords
        thick = margin > 0.5
        if kid in (registry.STRATEGIES["Relaxed"], registry.STRATEGIES["Heuristic-Auto-Relaxed"]):
            continue                    # they also converge on other faces after overshooting; nothing more to pin
        assert (out["hit"][thick] == 1).all(), kid
        both = thick & (out["hit"] > 0)
        err = np.abs(out["t_raw"][both] - tn[both])
        # a hit is any point with |sdf| < 1e-4: outside the face, or up to 1e-4 deep behind it when a
        # min-step / bisection strategy overshoots (Overstep-Bisect min_step 0.01, Hybrid 0.005)
        tol = 1.1e-2 if kid in (registry.STRATEGIES["Overstep-Bisect"], registry.STRATEGIES["Adaptive-Hybrid"]) else 2e-3
        assert err.max() < tol, (kid, float(err.max()))
        assert np.median(err) < 2e-4, (kid, float(np.median(err)))


def test_thin_torus_matches_the_quartic(hip):
    """Thin Torus (R = 1.5, r = 0.05, axis y): the smallest positive root of the ray/torus quartic
    F(t) = (|P|^2 + R^2 - r^2)^2 - 4 R^2 (Px^2 + Pz^2), P = o + t d (the closed form of the reference's
    gpu/analytic.py:131-192, CPU camera model), on every 5th pixel near the ring."""
    R, r = 1.5, 0.05
    pos, d = _rays(3)
    od = (d * pos).sum(2)
    dist2 = (pos * pos).sum() - od * od                      # squared distance of the ray line from the centre
    ys, xs = np.nonzero(dist2 <= (R + r) ** 2 * 1.02)
    ys, xs = ys[::5], xs[::5]
    dd = d[ys, xs]
    s1 = 2.0 * od[ys, xs]
    s0 = (pos * pos).sum() + R * R - r * r
    q2 = dd[:, 0] ** 2 + dd[:, 2] ** 2
    q1 = 2.0 * (pos[0] * dd[:, 0] + pos[2] * dd[:, 2])
    q0 = pos[0] ** 2 + pos[2] ** 2
    coef = np.stack([np.ones_like(s1), 2.0 * s1, s1 * s1 + 2.0 * s0 - 4 * R * R * q2, 2.0 * s1 * s0 - 4 * R * R * q1,
                     s0 * s0 - 4 * R * R * q0 + 0.0 * s1], 1)
    t_exact = np.full(len(ys), np.inf)
    gap = np.full(len(ys), np.inf)          # how clearly the ray hits / misses: min |Im| of the nearest root pair
    for i, c in enumerate(coef):
        rt = np.roots(c)
        real = rt[np.abs(rt.imag) < 1e-9].real
        real = real[real > 1e-6]
        if real.size:
            t_exact[i] = real.min()
        gap[i] = np.abs(rt.imag).min()
    hit_exact = np.isfinite(t_exact)
    assert hit_exact.sum() > 1000 and (~hit_exact).sum() > 1000
    for kid in GRADED:
        out = _render(hip, 3, kid)
        hit = out["hit"][ys, xs] > 0
        t = out["t_raw"][ys, xs]
        # a marched hit is any point within hit_threshold of the surface: rays passing closer than that are hits too
        clear_miss = (~hit_exact) & (gap > 2e-2)
        assert not hit[clear_miss].any(), kid
        # chords well inside the tube (the two front roots far apart) must be found by the sound strategies
        ok = hit & hit_exact
        assert ok.sum() > 0.5 * hit_exact.sum(), (kid, int(ok.sum()), int(hit_exact.sum()))
        P = pos[None, :] + t[ok][:, None] * dd[ok]
        sdf = np.sqrt((np.sqrt(P[:, 0] ** 2 + P[:, 2] ** 2) - R) ** 2 + P[:, 1] ** 2) - r     # closed-form distance at the stop
        assert np.abs(sdf).max() < 1.1e-4, (kid, float(np.abs(sdf).max()))
        front = ok & (t < t_exact + 0.02)              # stopped on the first sheet of the tube
        err = np.abs(t[front] - t_exact[front])
        assert np.median(err) < 3e-4, (kid, float(np.median(err)))


def test_8k_frame_smoke(hip):
    """7680x4320 (BASELINE config 5 shape): one band-cyclic rank-0-of-8 shard vs the matching rows of
    a contiguous render of the same image rows."""
    w, h = 7680, 4320
    plan = sharding.plan_rows(h, 8, 3)
    a = _render(hip, 12, 0, w, h, **plan.desc_kwargs())
    rows = plan.image_rows()
    for band in (0, 57, 134):
        r0 = int(rows[band * 4])
        b = _render(hip, 12, 0, w, h, row0=r0, rows=4)
        assert (a["iters"][band * 4: band * 4 + 4] == b["iters"]).all()
        assert (a["t_raw"][band * 4: band * 4 + 4].view(np.uint64) == b["t_raw"].view(np.uint64)).all()


def test_batch_equals_single_frames(hip):
    """rm_render_batch: 12 viewpoints x different iteration budgets / thresholds in one call, each frame
    bit-identical to its own rm_render (cf. the reference's viewpoint and budget sweeps)."""
    import math
    w, h = 384, 216
    for sid, kid in ((10, 0), (9, 10), (0, 6)):
        cams, cfgs = [], []
        for i in range(12):
            ang = 2.0 * math.pi * i / 12.0
            rad = 3.0 if sid == 10 else 5.0
            cams.append(Camera((rad * math.sin(ang), 0.4 * math.cos(3 * ang), rad * math.cos(ang)), (0.0, 0.0, 0.0),
                               (0.0, 1.0, 0.0), 60.0, w, h).params14())
            cfgs.append(dict(max_iterations=[64, 128, 512][i % 3], hit_threshold=[1e-3, 1e-4][i % 2], max_distance=100.0,
                             lipschitz=_lip(sid, kid)))
        shape = hip.make_desc(sid, kid, cams[0], w, h)
        out = hip.render_batch(shape, np.array(cams), cfgs)
        assert out["ms_total"] > 0
        for i in range(12):
            one = hip.render(hip.make_desc(sid, kid, cams[i], w, h, **cfgs[i]))
            assert (out["iters"][i] == one["iters"]).all() and (out["hit"][i] == one["hit"]).all(), (sid, kid, i)
            assert (out["depth"][i].view(np.uint32) == one["depth"].view(np.uint32)).all(), (sid, kid, i)
            assert out["stats"][i]["sum_iters"] == one["stats"]["sum_iters"]
            assert out["stats"][i]["hit_count"] == one["stats"]["hit_count"]


def test_throughput_regime_frame_at_7680x4320(hip):
    """The 7680x4320 frame runs through the plain render kernel (throughput regime) and counts every ray.  The register
    cliff this test once guarded by wall time (two waves per SIMD need <= 256 registers) is asserted where it can be read:
    in the code objects' metadata, tests/test_code_objects.py (CPU)."""
    sc = registry.SCENES[10]
    cam = Camera(sc.camera_position, sc.camera_target, (0.0, 1.0, 0.0), 60.0, 7680, 4320).params14()
    out = hip.render(hip.make_desc(10, 0, cam, 7680, 4320), warmup=0, repeats=1)
    assert out["stats"]["total_rays"] == 7680 * 4320
