"""Host side of the sweep row: the curated viewpoint table against the
reference-generated fixture, sweep levels, the adjacent-pixel divergence proxy.  No GPU needed."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from raymarch_algo_compare_amd import registry, sweep, viewpoints


def test_viewpoints_equal_the_reference_table():
    """tests/golden/viewpoints.json is written by oracle/gen_golden.py from the reference's viewpoints_for
    (viewpoints.py:126-140) for all 20 catalogue scenes: names, categories, positions, targets, ups."""
    ref = json.load(open(os.path.join(GOLDEN, "viewpoints.json"), encoding="utf-8"))
    assert len(ref) == len(registry.SCENES) == 20
    total = 0
    for sc in registry.SCENES:
        got = [[v.name, v.category, list(map(float, v.position)), list(map(float, v.target)), list(map(float, v.up))]
               for v in viewpoints.viewpoints_for(sc)]
        assert got == ref[sc.name], sc.name
        total += len(got)
    assert total == 53
    # scenes without a curated entry fall back to their suggested camera, then to (0, 0, 5)
    assert [v.name for v in viewpoints.viewpoints_for(registry.SCENES[12])] == ["default"]
    assert viewpoints.viewpoints_for(registry.SCENES[12])[0].position == (1.0, 1.0, 8.0)
    assert viewpoints.viewpoints_for(registry.SCENES[7])[0].position == (0.0, 0.0, 5.0)
    rc = viewpoints.viewpoints_for(registry.SCENES[1])[2].render_config(320, 200)
    assert (rc.width, rc.height, rc.camera_position, rc.camera_target) == (320, 200, (0.0, 0.28, 13.0), (0.0, -0.46, 0.0))


def test_levels_follow_the_reference_axes():
    """budget: max_iterations sweeps at a fixed epsilon; residual: epsilon sweeps at a fixed cap
    (reference sweep.py:96-127, defaults :48-54)."""
    lv = sweep.build_levels("budget")
    assert [int(v) for v, _, _ in lv] == [32, 64, 128, 256, 512]
    assert all(mc.hit_threshold == 1e-4 and mc.max_iterations == int(v) and ex["sweep_axis"] == "budget" for v, mc, ex in lv)
    lv = sweep.build_levels("residual", cap=300)
    assert [v for v, _, _ in lv] == [1e-2, 3e-3, 1e-3, 3e-4, 1e-4, 3e-5, 1e-5]
    assert all(mc.max_iterations == 300 and mc.hit_threshold == v and ex["hit_threshold"] == v for v, mc, ex in lv)
    assert sweep.finest_index("budget", sweep.build_levels("budget")) == 4
    assert sweep.finest_index("residual", sweep.build_levels("residual")) == 6
    with pytest.raises(ValueError):
        sweep.build_levels("nope")


def test_divergence_proxy():
    a = np.array([[1, 3, 3], [2, 3, 7]])
    # |dx|: 2 0 / 1 4 -> 7 ; |dy|: 1 0 4 -> 5 ; 7 edges
    assert sweep.divergence_proxy(a) == pytest.approx(12.0 / 7.0)
    assert sweep.divergence_proxy(np.zeros((1, 1))) == 0.0


def test_rows_roundtrip_csv_and_json(tmp_path):
    row = {k: 0 for k in sweep.ROW_FIELDS}
    row.update(scene="Sphere", strategy="Relaxed(ω=1.2)", viewpoint="ortho", category="orthogonal", sweep_axis="budget")
    sweep.write_rows([row, row], str(tmp_path / "s.csv"))
    lines = open(tmp_path / "s.csv", encoding="utf-8").read().splitlines()
    assert lines[0].split(",") == sweep.ROW_FIELDS and len(lines) == 3
    sweep.write_rows([row], str(tmp_path / "s.json"))
    assert json.load(open(tmp_path / "s.json", encoding="utf-8"))[0]["strategy"] == "Relaxed(ω=1.2)"


def test_unknown_names_raise():
    with pytest.raises(KeyError):
        sweep.run_sweep(["No Such Scene"], ["Standard"])


def test_analytic_module_equals_the_reference_closed_forms():
    """raymarch_algo_compare_amd.analytic against tests/golden/analytic_80x60.npz, written by oracle/gen_golden.py
    from the reference's gpu/analytic.py intersect_* functions on the same rays: hit masks identical, depth and
    normals to rounding."""
    from raymarch_algo_compare_amd import analytic
    from raymarch_algo_compare_amd.camera import Camera
    z = np.load(os.path.join(GOLDEN, "analytic_80x60.npz"))
    for sid in (0, 1, 2, 3):
        sc = registry.SCENES[sid]
        assert analytic.has_analytic(sc.name)
        cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 80, 60)
        assert (cam.params14() == z[f"s{sid}_cam"]).all()
        depth, hit, normal = analytic.analytic_depth(sc.name, cam)
        ref_hit = np.unpackbits(z[f"s{sid}_hitbits"])[:4800].reshape(60, 80).astype(bool)
        assert (hit == ref_hit).all() and hit.sum() > 30, sc.name
        assert np.abs(depth - z[f"s{sid}_depth"]).max() < 1e-9, sc.name
        assert np.abs(normal - z[f"s{sid}_normal"]).max() < 1e-9, sc.name
        assert np.abs(np.linalg.norm(normal[hit], axis=1) - 1.0).max() < 1e-12
    assert not analytic.has_analytic("Mandelbulb")
    with pytest.raises(KeyError):
        analytic.analytic_depth("Mandelbulb", cam)


def test_param_grid_follows_the_reference():
    """param_grid.py:20-27: omega for the two relaxed marchers, margin for Skipping-Spheres; first value = default;
    uniform names reach RmStrategyParams through runner.SHADER_UNIFORMS."""
    from raymarch_algo_compare_amd.runner import GPURunner
    assert sweep.param_combos("Standard") == [{}]
    assert sweep.param_combos("Relaxed") == [{"omega": w} for w in (1.2, 1.4, 1.6, 1.8)]
    assert sweep.param_combos("Skipping-Spheres") == [{"margin": m} for m in (0.02, 0.05, 0.1, 0.2)]
    assert sweep.param_label({}) == "default" and sweep.param_label({"omega": 1.6}) == "omega=1.6"
    assert GPURunner.strategy_params({"omega": 1.6, "minStep": 1.0, "kappa": 2.0, "stepScale": 0.6}) == {
        "omega": 1.6, "ar_omega_init": 1.6, "dense_min_step": 1.0, "step_scale": 0.6}
    assert GPURunner.strategy_params({"margin": 0.1, "beta": 0.5, "hybrid_stuck_threshold": 3}) == {"margin": 0.1, "beta": 0.5, "hybrid_stuck_threshold": 3}
    with pytest.raises(NotImplementedError):
        GPURunner.strategy_params({"kappa": 3.0})
    with pytest.raises(KeyError):
        GPURunner.strategy_params({"gain": 1.0})
    assert "params" in sweep.ROW_FIELDS
