"""Strategy parameters across the C ABI (RmStrategyParams, include/rm_hip.h) on the GPU.

tests/golden/frames_params_48x36.npz holds 140 frames the REFERENCE marched with its strategy classes constructed
with non-default arguments (relaxed_sphere.py:17, auto_relaxed.py:21-23, slope_auto_relaxed.py:25,
overstep_bisect.py:18, adaptive_hybrid.py:17-19) or with one literal of march() swapped (margin, AR-ST's start
omega, the two bisection counts): >= 3 settings per tunable strategy on 5 scenes.  The kernels must reproduce them
bit for bit through every path that carries parameters: one-frame launches (parameters in scalar registers),
batched launches (one parameter set per frame, per-lane), parked / resumed / team-finished rays, explicit rays."""
import numpy as np
import pytest

from conftest import golden_param_cases, sha_f64
from raymarch_algo_compare_amd import MarchConfig, RenderConfig, registry

pytestmark = pytest.mark.gpu


def _desc(hip, sid, kid, g, prm, full=True, **kw):
    return hip.make_desc(sid, kid, g["cam"], g["W"], g["H"], g["row0"], g["rows"], g["max_iterations"], g["hit_threshold"],
                         g["max_distance"], g["lipschitz"], full, params=prm, **kw)


def _overrides(hip, prm):
    return {k: v for k, v in prm.items() if v != hip.DEFAULT_STRATEGY_PARAMS[k]}


def test_single_frames_match_the_reference_for_all_parameter_cases(hip):
    n = 0
    seen = set()
    for sid, kid, prm, g in golden_param_cases():
        out = hip.render(_desc(hip, sid, kid, g, prm), want_t_raw=True, want_final_sdf=True)
        assert (out["iters"] == g["iters"]).all() and (out["hit"] == g["hit"]).all(), (sid, kid, _overrides(hip, prm))
        assert sha_f64(out["t_raw"]) == g["sha_t"] and sha_f64(out["final_sdf"]) == g["sha_fs"], (sid, kid, _overrides(hip, prm))
        assert np.abs(out["depth"].astype(np.float64) - g["depth"]).max() <= 1e-5
        # the lean path (full = 0) takes the same decisions
        lean = hip.render(_desc(hip, sid, kid, g, prm, full=False))
        assert (lean["iters"] == g["iters"]).all() and (lean["hit"] == g["hit"]).all()
        seen.add((kid, tuple(sorted(_overrides(hip, prm).items()))))
        n += 1
    assert n == 140
    per_strategy = {}
    for kid, ov in seen:
        per_strategy.setdefault(kid, set()).add(ov)
    assert all(len(v) >= 3 for v in per_strategy.values()) and set(per_strategy) == {1, 2, 3, 6, 7, 8, 9, 10}


def test_batched_frames_carry_their_own_parameters(hip):
    """rm_render_batch: all parameter settings of one (scene, strategy) in ONE launch, one RmStrategyParams per frame
    (the reference's parameter grid, sweep.py:181,222-223) -- each frame bit-identical to the reference's."""
    groups = {}
    for sid, kid, prm, g in golden_param_cases():
        groups.setdefault((sid, kid), []).append((prm, g))
    assert len(groups) == 40
    for (sid, kid), cases in groups.items():
        g0 = cases[0][1]
        # add a default-parameter frame and vary the budget too, so frames of one launch differ in everything
        cfgs = [dict(max_iterations=g["max_iterations"], hit_threshold=g["hit_threshold"], max_distance=g["max_distance"],
                     lipschitz=g["lipschitz"], params=prm) for prm, g in cases]
        cfgs.append(dict(max_iterations=77, hit_threshold=1e-3, max_distance=50.0, lipschitz=g0["lipschitz"]))
        cams = np.stack([g["cam"] for _, g in cases] + [g0["cam"]])
        shape = hip.make_desc(sid, kid, g0["cam"], g0["W"], g0["H"])
        out = hip.render_batch(shape, cams, cfgs)
        for i, (prm, g) in enumerate(cases):
            assert (out["iters"][i] == g["iters"]).all() and (out["hit"][i] == g["hit"]).all(), (sid, kid, _overrides(hip, prm))
            assert np.abs(out["depth"][i].astype(np.float64) - g["depth"]).max() <= 1e-5
        one = hip.render(hip.make_desc(sid, kid, g0["cam"], g0["W"], g0["H"], **cfgs[-1]))
        assert (out["iters"][-1] == one["iters"]).all() and (out["depth"][-1].view(np.uint32) == one["depth"].view(np.uint32)).all()


def test_parked_and_team_finished_rays_keep_their_parameters(hip):
    """Long-ray suspension with non-default parameters: a parked ray is resumed with the configuration of its frame
    (single waves and wavefront teams), one frame and batched."""
    mandel = [(kid, prm, g) for sid, kid, prm, g in golden_param_cases() if sid == 10]
    assert len(mandel) == 28
    for kid, prm, g in mandel:
        for sched in (dict(suspend_after=(3, 11), resume_mode=2), dict(suspend_after=(4, 0), resume_mode=1),
                      dict(suspend_after=(2, 9), resume_mode=3)):
            out = hip.render(_desc(hip, 10, kid, g, prm, **sched), want_t_raw=True, want_final_sdf=True)
            assert (out["iters"] == g["iters"]).all() and (out["hit"] == g["hit"]).all(), (kid, _overrides(hip, prm), sched)
            assert sha_f64(out["t_raw"]) == g["sha_t"] and sha_f64(out["final_sdf"]) == g["sha_fs"], (kid, sched)
    by_kid = {}
    for kid, prm, g in mandel:
        by_kid.setdefault(kid, []).append((prm, g))
    for kid, cases in by_kid.items():
        g0 = cases[0][1]
        cfgs = [dict(max_iterations=512, lipschitz=g["lipschitz"], params=prm) for prm, g in cases]
        shape = hip.make_desc(10, kid, g0["cam"], g0["W"], g0["H"], suspend_after=(3, 11), resume_mode=2)
        out = hip.render_batch(shape, np.stack([g["cam"] for _, g in cases]), cfgs)
        for i, (prm, g) in enumerate(cases):
            assert (out["iters"][i] == g["iters"]).all() and (out["hit"][i] == g["hit"]).all(), (kid, _overrides(hip, prm))


def test_restated_defaults_equal_no_parameters(hip):
    """use_params = 1 with every field at its default == use_params = 0, bit for bit (all 11 strategies)."""
    from conftest import golden_frames
    G = golden_frames("64x48")
    for kid in range(11):
        g = G.get(9, kid)
        a = hip.render(hip.make_desc(9, kid, g["cam"], 64, 48, lipschitz=g["lipschitz"], full=True), want_t_raw=True)
        b = hip.render(hip.make_desc(9, kid, g["cam"], 64, 48, lipschitz=g["lipschitz"], full=True,
                                     params=dict(hip.DEFAULT_STRATEGY_PARAMS)), want_t_raw=True)
        assert (a["iters"] == g["iters"]).all() and (b["iters"] == g["iters"]).all()
        assert (a["t_raw"].view(np.uint64) == b["t_raw"].view(np.uint64)).all()


def test_explicit_rays_with_parameters_match_the_oracle(hip):
    """rm_march_rays / rm_march_rays_team (MarchStrategy.march) with non-default parameters against the pinned oracle."""
    from oracle import oracle
    rng = np.random.default_rng(3)
    n = 500
    o = np.tile(np.array([0.0, 0.0, 3.0]), (n, 1)) + rng.normal(size=(n, 3)) * 0.05
    d = np.array([0.0, 0.0, -1.0]) + rng.normal(size=(n, 3)) * 0.25
    for kid, prm in ((1, {"omega": 1.7}), (6, {"overstep_min_step": 0.03, "overstep_bisection_steps": 5}),
                     (9, {"hybrid_stuck_threshold": 2, "hybrid_min_step": 0.02}), (2, {"ar_omega_init": 1.9, "ar_decay_rate": 0.5}),
                     (7, {"margin": 0.15}), (10, {"segment_bisection_steps": 2}), (8, {"revaa_bisection_steps": 11}),
                     (3, {"beta": 0.7})):
        for sid, team in ((10, False), (10, True), (12, False)):
            hit, t, it, fs = hip.march_rays(sid, kid, o, d, team=team, params=prm)
            rh, rt, ri, rf = oracle.march_rays(sid, kid, o, d, params=prm)
            assert (hit == rh).all() and (it == ri).all(), (sid, kid, prm, team)
            assert (t.view(np.uint64) == rt.view(np.uint64)).all() and (fs.view(np.uint64) == rf.view(np.uint64)).all(), (sid, kid, prm, team)


def test_reference_seams_accept_parameters(hip, tmp_path):
    """The three host seams that carry parameters in the reference: constructing a strategy with arguments
    (STRATEGIES[key](omega=...)), GPURunner.render(params={uniform: value}) (gpu/runner.py:120-124) and the sweep's
    parameter grid (sweep.py:181,222-223) -- each against the reference-marched goldens."""
    from raymarch_algo_compare_amd import HipCollector, sweep
    from raymarch_algo_compare_amd.camera import Camera
    from raymarch_algo_compare_amd.runner import GPURunner
    cases = {(sid, kid, tuple(sorted(_overrides(hip, prm).items()))): g for sid, kid, prm, g in golden_param_cases()}
    W, H = 48, 36
    # constructor arguments through the collector (the reference: MetricsCollector.benchmark_strategy(strategy, ...))
    for key, ctor, ov in (("Relaxed", dict(omega=1.6), (("omega", 1.6),)),
                          ("Overstep-Bisect", dict(min_step_factor=0.02, bisection_steps=8),
                           (("overstep_bisection_steps", 8), ("overstep_min_step", 0.02))),
                          ("Adaptive-Hybrid", dict(stuck_threshold=3, stuck_step_ratio=0.01, min_step_factor=0.01),
                           (("hybrid_min_step", 0.01), ("hybrid_stuck_step_ratio", 0.01), ("hybrid_stuck_threshold", 3)))):
        st = registry.get_strategy_by_name(key, **ctor)
        for sid in (0, 10):
            g = cases[(sid, st.id, ov)]
            sc = registry.SCENES[sid]
            cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H)
            stats = HipCollector(MarchConfig()).benchmark_strategy(st, sc, cam, verbose=False)
            assert (stats.iteration_heatmap == g["iters"]).all() and (stats.hit_map == (g["hit"] > 0)).all(), (key, sid)
            assert (stats.depth_map == g["depth"]).all()
            assert stats.strategy_name == st.short_name
    # shader uniforms through GPURunner: omega drives Relaxed (id 2) and the start omega of AR-ST (id 5), margin id 6
    r = GPURunner()
    for gid, kid, uniform, ov in ((2, 1, {"omega": 1.8}, (("omega", 1.8),)), (5, 2, {"omega": 1.4}, (("ar_omega_init", 1.4),)),
                                  (6, 7, {"margin": 0.2}, (("margin", 0.2),))):
        for sid in (2, 9):
            g = cases[(sid, kid, ov)]
            sc = registry.SCENES[sid]
            px, secs = r.render(sid, gid, RenderConfig(width=W, height=H), MarchConfig(), params=dict(uniform, minStep=1.0, stepScale=1.0))
            assert (np.rint(px[..., 1] * 512).astype(np.int32) == g["iters"]).all() and (px[..., 0] == g["hit"]).all(), (gid, sid)
            cap = r.capture(sid, gid, RenderConfig(width=W, height=H), MarchConfig(), params=uniform)
            assert (cap["hit"] == (g["hit"] > 0)).all()
    # the sweep's parameter grid: every (viewpoint, combo, level) row equals that frame rendered on its own
    rows = sweep.run_sweep(["Cube"], ["Relaxed", "Skipping-Spheres", "Standard"], "budget", 64, 48, budgets=[32, 512], grid=True,
                           out_path=str(tmp_path / "grid.csv"))
    from raymarch_algo_compare_amd import viewpoints
    nvp = len(viewpoints.viewpoints_for(registry.SCENES[2]))
    assert len(rows) == nvp * 2 * (4 + 4 + 1)
    assert sorted({r["params"] for r in rows if r["strategy"].startswith("Relaxed")}) == ["omega=1.2", "omega=1.4", "omega=1.6", "omega=1.8"]
    assert {r["params"] for r in rows if r["strategy"] == "Standard"} == {"default"}
    for row in rows:
        st = registry.get_strategy_by_name("Relaxed" if row["strategy"].startswith("Relaxed") else row["strategy"])
        vp = [v for v in viewpoints.viewpoints_for(registry.SCENES[2]) if v.name == row["viewpoint"]][0]
        cam = Camera(vp.position, vp.target, vp.up, 60.0, 64, 48).params14()
        prm = {} if row["params"] == "default" else GPURunner.strategy_params({row["params"].split("=")[0]: float(row["params"].split("=")[1])})
        one = hip.render(hip.make_desc(2, st.id, cam, 64, 48, max_iterations=row["max_iterations"], full=True, params=prm), want_evals=True)
        assert row["iters_mean"] == float(one["iters"].mean()) and row["hit_rate"] == float((one["hit"] > 0).mean()), row
        assert row["evals_mean"] == float(one["evals"].mean()), row
    d12 = [r for r in rows if r["params"] == "omega=1.2" and r["max_iterations"] == 512]
    d18 = [r for r in rows if r["params"] == "omega=1.8" and r["max_iterations"] == 512]
    assert [r["iters_mean"] for r in d12] != [r["iters_mean"] for r in d18]                 # the parameter does something
