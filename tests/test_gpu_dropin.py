"""Drop-in surface on the GPU: run_once / CLI artefacts / run_gpu_benchmark seam / non-default
march configurations, checked against the reference-generated goldens and the CPU oracle."""
import json
import os

import numpy as np
import pytest

from conftest import golden_frames
from raymarch_algo_compare_amd import MarchConfig, RenderConfig, registry, run_once
from raymarch_algo_compare_amd.camera import Camera

pytestmark = pytest.mark.gpu


def test_run_once_reproduces_reference_stats(hip):
    """run_once(...) -> RayMarchStats: every scalar the reference computes, bit-for-bit (64x48 goldens
    come from the reference's own run_once wiring: suggested cameras, Lipschitz bound)."""
    G = golden_frames("64x48")
    for sid, kid in [(0, 0), (2, 1), (9, 10), (10, 0), (10, 9), (11, 10), (12, 3), (13, 5), (16, 4), (19, 6)]:
        ref = G.stats[f"s{sid}_k{kid}"]
        st = run_once(RenderConfig(width=64, height=48), MarchConfig(), registry.SCENES[sid].name,
                      registry.list_strategies()[kid])
        assert (st.strategy_name, st.scene_name) == (ref["strategy"], ref["scene"])
        for key in ("total_rays", "hit_count", "miss_count", "sample_count", "iteration_mean", "iteration_median",
                    "iteration_std", "iteration_min", "iteration_max", "iteration_p95", "iteration_p99", "hit_rate",
                    "warp_divergence_proxy", "accuracy_mean", "accuracy_max", "accuracy_std"):
            assert getattr(st, key) == ref[key], (sid, kid, key, getattr(st, key), ref[key])
        assert float(st.depth_map.sum()) == ref["depth_sum"]          # float64 depth map, exact
        assert st.depth_map.dtype == np.float64 and st.iteration_heatmap.dtype == np.int32 and st.hit_map.dtype == bool


def test_unknown_names_raise_unless_compat(hip):
    with pytest.raises(KeyError):
        run_once(RenderConfig(width=16, height=12), None, "Grazing_Plane", "Standard")
    with pytest.raises(KeyError):
        run_once(RenderConfig(width=16, height=12), None, "Sphere", "Slope-AR")
    # the reference silently renders Sphere / Standard for those names (tests/test_smoke.py:56-72 rely on it)
    st = run_once(RenderConfig(width=16, height=12), MarchConfig(max_iterations=100), "Grazing_Plane", "Nope",
                  compat_fallback=True)
    g = golden_frames("16x12_it100").get(0, 0)
    assert (st.scene_name, st.strategy_name) == ("Sphere", "Standard")
    assert (st.iteration_heatmap == g["iters"]).all()


def test_cli_writes_reference_artefacts(hip, tmp_path):
    from raymarch_algo_compare_amd.main import cli
    out = tmp_path / "res"
    rc = cli(["--scene", "Grazing Plane,Cube", "--strategy", "Standard,Segment", "--width", "64", "--height", "48",
              "--gpu-width", "96", "--gpu-height", "64", "--gpu-warmup", "1", "--gpu-repeats", "2",
              "--output-dir", str(out), "--json", str(out / "summary.json")])
    assert rc == 0
    runs = sorted(d for d in os.listdir(out) if "__" in d)
    assert [r.split("__")[:2] for r in runs] == [["Cube", "Segment"], ["Cube", "Standard"],
                                                 ["Grazing_Plane", "Segment"], ["Grazing_Plane", "Standard"]]
    for r in runs:
        assert sorted(os.listdir(out / r)) == ["depth_map.npy", "hit_map.png", "inv_depth.png", "iterations.png", "stats.json"]
        rec = json.load(open(out / r / "stats.json", encoding="utf-8"))
        assert rec["gpu_width"] == 96 and rec["gpu_height"] == 64 and rec["gpu_time_sample_count"] == 2
        assert rec["gpu_time_us_per_ray_median"] > 0
    # the reference CLI reuses one RenderConfig: Cube inherits Grazing Plane's camera
    leak = golden_frames("leak").get(2, 0)
    rec = json.load(open(out / [r for r in runs if r.startswith("Cube__Standard")][0] / "stats.json", encoding="utf-8"))
    assert rec["hit_count"] == int(leak["hit"].sum())
    assert np.load(out / [r for r in runs if r.startswith("Cube__Standard")][0] / "depth_map.npy").sum() == leak["depth"].sum()
    lines = open(out / "matrix_iteration_mean.csv", encoding="utf-8").read().splitlines()
    assert lines[0] == ",Segment,Standard" and [l.split(",")[0] for l in lines[1:]] == ["Grazing Plane", "Cube"]
    summary = json.load(open(out / "summary.json", encoding="utf-8"))
    assert len(summary) == 4 and set(summary[0]) == {"strategy", "scene", "total_rays", "hit_count", "hit_rate",
                                                     "iteration_mean", "iteration_p95", "iteration_max",
                                                     "time_per_ray_us", "warp_divergence"}


def test_run_gpu_benchmark_seam(hip):
    """Same keys / shapes as the reference seam (gpu/runner.py:323-330) and the channel layout of
    main.glsl:79-84; unknown scene -> None."""
    from raymarch_algo_compare_amd.runner import run_gpu_benchmark
    rc, mc = RenderConfig(width=96, height=64), MarchConfig()
    assert run_gpu_benchmark("No Such Scene", "Standard", rc, mc) is None
    res = run_gpu_benchmark("Sphere", "Standard", rc, mc, gpu_warmup=1, gpu_repeats=3)
    assert set(res) >= {"pixels", "render_times_s", "render_time_s_median", "render_time_s_iqr", "render_time_s_mean",
                        "sample_count"}
    assert res["pixels"].shape == (64, 96, 4) and res["pixels"].dtype == np.float32
    assert res["sample_count"] == 3 and len(res["render_times_s"]) == 3
    from oracle import oracle
    ref = oracle.render(0, 0, Camera(rc.camera_position, rc.camera_target, rc.camera_up, 60.0, 96, 64).params14(), 96, 64)
    assert (np.rint(res["pixels"][..., 1] * mc.max_iterations).astype(np.int32) == ref.iters).all()
    assert (res["pixels"][..., 0] == ref.hit).all()


def test_non_default_march_configs_match_oracle(hip):
    """Budget / epsilon / far-plane sweeps (sweep.py:96-127): GPU vs the pinned oracle, bit-exact."""
    from oracle import oracle
    w, h = 96, 64
    rng = np.random.default_rng(5)
    for sid, kid in [(0, 0), (2, 6), (3, 10), (8, 2), (9, 9), (10, 0), (10, 10), (12, 4), (13, 8), (16, 7), (18, 5)]:
        sc = registry.SCENES[sid]
        cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, w, h).params14()
        for mi, thr, far in ((17, 1e-3, 100.0), (64, 1e-5, 20.0), (300, 3e-4, 7.5), (int(rng.integers(20, 500)), 1e-4, 55.0)):
            lip = sc.lipschitz if (kid == 10 and sc.lipschitz) else 1.0
            out = hip.render(hip.make_desc(sid, kid, cam, w, h, max_iterations=mi, hit_threshold=thr, max_distance=far,
                                           lipschitz=lip, full=True), want_t_raw=True, want_final_sdf=True)
            ref = oracle.render(sid, kid, cam, w, h, max_iterations=mi, hit_threshold=thr, max_distance=far, lipschitz=lip)
            assert (out["iters"] == ref.iters).all() and (out["hit"] == ref.hit).all(), (sid, kid, mi, thr, far)
            assert (out["t_raw"].view(np.uint64) == ref.t.view(np.uint64)).all(), (sid, kid, mi, thr, far)
            assert (out["final_sdf"].view(np.uint64) == ref.final_sdf.view(np.uint64)).all(), (sid, kid, mi, thr, far)


def test_sweep_cells_equal_single_renders(hip, tmp_path):
    """sweep.run_sweep: every (viewpoint, level) row of a batched (scene, strategy) cell carries the statistics of
    that frame rendered on its own; both axes; Segment gets the scene's Lipschitz bound like run_once."""
    from raymarch_algo_compare_amd import sweep, viewpoints
    w, h = 64, 48
    rows = sweep.run_sweep(["Sphere", "Mandelbulb", "Bad Lipschitz"], ["Standard", "Segment"], "budget", w, h, budgets=[16, 64, 512],
                           out_path=str(tmp_path / "sweep.csv"))
    n_vp = {"Sphere": 2, "Mandelbulb": 3, "Bad Lipschitz Sphere": 1}
    assert len(rows) == sum(n_vp.values()) * 2 * 3
    assert len(open(tmp_path / "sweep.csv", encoding="utf-8").read().splitlines()) == len(rows) + 1
    for r in rows:
        sc = registry.get_scene_by_name(r["scene"])
        st = registry.get_strategy_by_name("Segment" if r["strategy"] == "Segment" else "Standard")
        vp = [v for v in viewpoints.viewpoints_for(sc) if v.name == r["viewpoint"]][0]
        cam = Camera(vp.position, vp.target, vp.up, 60.0, w, h).params14()
        lip = (sc.lipschitz or 1.0) if st.key == "Segment" else 1.0
        one = hip.render(hip.make_desc(sc.id, st.id, cam, w, h, max_iterations=r["max_iterations"],
                                       hit_threshold=r["hit_threshold"], lipschitz=lip, full=True), want_evals=True)
        assert r["iters_mean"] == float(one["iters"].mean()) and r["iters_max"] == float(one["iters"].max()), r
        assert r["evals_mean"] == float(one["evals"].mean()) and r["evals_max"] == float(one["evals"].max()), r
        assert r["evals_mean"] >= r["iters_mean"] or r["strategy"] != "Standard"
        assert r["hit_rate"] == float((one["hit"] > 0).mean()) and r["divergence_proxy"] == sweep.divergence_proxy(one["iters"])
        if r["max_iterations"] == 512:
            assert r["depth_mae_vs_finest"] == 0.0 and r["hit_flips_vs_finest"] == 0
    # a tighter budget can only lose hits relative to the finest level on the convex Sphere
    sph = [r for r in rows if r["scene"] == "Sphere" and r["strategy"] == "Standard" and r["viewpoint"] == "ortho"]
    assert [r["hit_rate"] for r in sph] == sorted(r["hit_rate"] for r in sph)
    res = sweep.run_sweep(["Cube"], ["Enhanced"], "residual", w, h, epsilons=[1e-2, 1e-4], cap=200)
    assert [r["hit_threshold"] for r in res] == [1e-2, 1e-4] * 3 and all(r["max_iterations"] == 200 for r in res)
    assert all(r["depth_mae_vs_finest"] == 0.0 for r in res if r["hit_threshold"] == 1e-4)


def test_random_cameras_match_oracle(hip):
    """Off-axis positions, tilted up vectors, fields of view from 25 to 110 degrees, non-4:3 frames: the camera
    basis comes from the host (Camera.__init__ expressions), the per-pixel ray from the kernel -- against the
    pinned oracle, bit for bit (iterations, hits, raw fp64 t)."""
    from oracle import oracle
    rng = np.random.default_rng(11)
    for sid, kid in [(0, 0), (2, 4), (9, 3), (10, 0), (12, 10), (16, 1)]:
        for _ in range(3):
            w, h = int(rng.integers(40, 131)), int(rng.integers(24, 91))
            pos = tuple(float(v) for v in rng.normal(size=3) * 2.0 + np.array([0.0, 0.0, 4.0]))
            tgt = tuple(float(v) for v in rng.normal(size=3) * 0.3)
            up = tuple(float(v) for v in (rng.normal(size=3) * 0.3 + np.array([0.0, 1.0, 0.0])))
            fov = float(rng.uniform(25.0, 110.0))
            cam = Camera(pos, tgt, up, fov, w, h).params14()
            assert (cam == oracle.camera14(pos, tgt, up, fov, w, h)).all()
            lip = registry.SCENES[sid].lipschitz if (kid == 10 and registry.SCENES[sid].lipschitz) else 1.0
            out = hip.render(hip.make_desc(sid, kid, cam, w, h, lipschitz=lip), want_t_raw=True)
            ref = oracle.render(sid, kid, cam, w, h, lipschitz=lip)
            assert (out["iters"] == ref.iters).all() and (out["hit"] == ref.hit).all(), (sid, kid, pos, fov)
            assert (out["t_raw"].view(np.uint64) == ref.t.view(np.uint64)).all(), (sid, kid, pos, fov)


def test_gpurunner_render_and_capture(hip):
    """GPURunner (reference gpu/runner.py:58-268): render -> (pixels, seconds), capture -> geom / normal / depth /
    color / evals / hit, with the shader's strategy numbering, on the CPU path's arithmetic."""
    import os
    from conftest import GOLDEN
    from raymarch_algo_compare_amd.runner import GLSL_STRATEGY_KEYS, GPURunner
    rc, mc = RenderConfig(width=48, height=36), MarchConfig()
    r = GPURunner()
    z = np.load(os.path.join(GOLDEN, "evals_48x36.npz"))
    for gid, key in GLSL_STRATEGY_KEYS.items():
        if key in registry.SHADER_ONLY_STRATEGIES:                 # ids 8, 9: parity unpinned, tests/test_gpu_shader_only.py
            continue
        kid = registry.STRATEGIES[key]
        px, secs = r.render(0, gid, rc, mc, lipschitz=1.0, params={"omega": 1.2, "stepScale": 1.0, "minStep": 0.5})
        assert px.shape == (36, 48, 4) and px.dtype == np.float32 and secs > 0
        assert (np.rint(px[..., 1] * mc.max_iterations).astype(np.int32) == z[f"s0_k{kid}_iters"]).all(), key
        cap = r.capture(0, gid, rc, mc)
        assert (cap["evals"] == z[f"s0_k{kid}_evals"]).all(), key                # the march's own SDF evaluations
        assert (cap["geom"] == px).all() and (cap["hit"] == (px[..., 0] > 0.5)).all()
    cap = r.capture(0, 0, rc, mc)
    hit = cap["hit"]
    assert hit.sum() > 50 and (cap["depth"][~hit] == 0).all() and (cap["normal"][~hit] == 0).all()
    # unit sphere at the origin: the tetrahedron normal is the radial direction up to the technique's O(e) = 5e-4 error
    cam = Camera(rc.camera_position, rc.camera_target, rc.camera_up, rc.fov_degrees, 48, 36).params14()
    u = (2.0 * (np.arange(48) + 0.5) / 48 - 1.0) * cam[12]
    v = (1.0 - 2.0 * (np.arange(36) + 0.5) / 36) * cam[13]
    rd = cam[3:6][None, None, :] + cam[6:9][None, None, :] * u[None, :, None] + cam[9:12][None, None, :] * v[:, None, None]
    rd /= np.linalg.norm(rd, axis=2, keepdims=True)
    pos = cam[0:3][None, :] + cap["depth"][hit][:, None].astype(np.float64) * rd[hit]
    assert np.abs(cap["normal"][hit] - pos / np.linalg.norm(pos, axis=1, keepdims=True)).max() < 6e-4
    n = cap["normal"][hit].astype(np.float64)
    L = np.array([0.6, 0.7, 0.5]) / np.linalg.norm([0.6, 0.7, 0.5])
    want = np.clip(np.array([0.82, 0.80, 0.78])[None, :] * (0.15 * (0.5 + 0.5 * n[:, 1]) + 0.85 * np.maximum(n @ L, 0.0))[:, None], 0, 1) ** 0.4545
    assert np.abs(cap["color"][hit] - want).max() < 1e-5
    tb = 0.5 * (rd[..., 1][~hit] + 1.0)
    assert np.abs(cap["color"][~hit] - ((1 - tb)[:, None] * np.array([0.06, 0.07, 0.09]) + tb[:, None] * np.array([0.12, 0.14, 0.18]))).max() < 1e-6
    # this engine's other strategies by key; unknown shader ids, the shader's own segment-tracing knob and bad names are refused
    px, _ = r.render(10, 0, RenderConfig(width=48, height=36, camera_position=(0.0, 0.0, 3.0)), mc, strategy_key="Curvature")
    assert (np.rint(px[..., 1] * 512).astype(np.int32) == z["s10_k5_iters"]).all()
    for bad, exc in ((dict(strategy_id=10), ValueError), (dict(strategy_id=3, params={"kappa": 3.0}), NotImplementedError),
                     (dict(strategy_id=0, params={"gain": 1.0}), KeyError)):
        with pytest.raises(exc):
            r.render(0, bad.pop("strategy_id"), rc, mc, **bad)
    with pytest.raises(ValueError):
        r.render(99, 0, rc, mc)


def test_marched_frames_against_analytic_ground_truth(hip):
    """analytic.analytic_depth as ground truth for the four analytic scenes at 320x200: the marched depth of the
    sound strategies stops within the hit threshold (over the incidence cosine) of the closed-form root, and the
    tetrahedron normals of GPURunner.capture agree with the closed-form normals."""
    from raymarch_algo_compare_amd import analytic
    from raymarch_algo_compare_amd.runner import GPURunner
    w, h = 320, 200
    r = GPURunner()
    for name in ("Sphere", "Grazing Plane", "Cube", "Thin Torus"):
        sc = registry.get_scene_by_name(name)
        rc = RenderConfig(width=w, height=h, camera_position=sc.camera_position or (0.0, 0.0, 5.0),
                          camera_target=sc.camera_target or (0.0, 0.0, 0.0))
        cam = Camera(rc.camera_position, rc.camera_target, rc.camera_up, rc.fov_degrees, w, h)
        depth, hit, normal = analytic.analytic_depth(name, cam)
        _, rd = analytic.camera_rays(cam)
        for gid in (0, 4):                                        # Standard, Enhanced
            cap = r.capture(sc.id, gid, rc, MarchConfig())
            both = cap["hit"] & hit
            assert both.sum() > 0.6 * hit.sum(), (name, gid, int(both.sum()), int(hit.sum()))
            cosi = np.abs((normal[both] * rd[both]).sum(1))      # a stop within eps of the surface is eps / cos(incidence) short
            short = (depth[both] - cap["depth"][both].astype(np.float64)) * cosi
            assert (short > -2e-4).all() and np.median(np.abs(short)) < 1.2e-4, (name, gid, float(short.min()), float(np.median(np.abs(short))))
            facing = cosi > 0.3                                    # away from silhouettes / edges the finite-difference normal is clean
            assert np.abs(cap["normal"][both][facing] - normal[both][facing]).max() < 5e-3, (name, gid)
