"""The CPU oracle (oracle/rm_oracle.c) against the reference's own outputs (tests/golden, written
by oracle/gen_golden.py importing the reference): bit-for-bit on iterations, hit masks, raw fp64
t and final_sdf (sha256 of the little-endian doubles).  This is what pins the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_frames, golden_param_cases, sha_f64
from oracle import oracle


def _render(g, sid, kid):
    return oracle.render(sid, kid, g["cam"], g["W"], g["H"], g["row0"], g["rows"], g["max_iterations"],
                         g["hit_threshold"], g["max_distance"], g["lipschitz"])


@pytest.mark.parametrize("tag", ["64x48", "160x120", "rows1080", "leak", "leakseq", "leakrows1080", "16x12_it100"])
def test_oracle_matches_reference(tag):
    G = golden_frames(tag)
    assert G.pairs
    for sid, kid in G.pairs:
        g = G.get(sid, kid)
        fr = _render(g, sid, kid)
        assert (fr.iters == g["iters"]).all(), (tag, sid, kid)
        assert (fr.hit == g["hit"]).all(), (tag, sid, kid)
        assert sha_f64(fr.t) == g["sha_t"], (tag, sid, kid)
        assert sha_f64(fr.final_sdf) == g["sha_fs"], (tag, sid, kid)
        assert (np.where(fr.hit > 0, fr.t, 0.0) == g["depth"]).all()


def test_oracle_matches_reference_with_non_default_strategy_parameters():
    """Strategy constructor arguments (relaxed_sphere.py:17, auto_relaxed.py:21-23, slope_auto_relaxed.py:25,
    overstep_bisect.py:18, adaptive_hybrid.py:17-19) and the three march() literals (margin, two bisection counts):
    140 frames the reference marched with non-default values."""
    n = 0
    for sid, kid, prm, g in golden_param_cases():
        fr = oracle.render(sid, kid, g["cam"], g["W"], g["H"], g["row0"], g["rows"], g["max_iterations"],
                           g["hit_threshold"], g["max_distance"], g["lipschitz"], params=prm)
        assert (fr.iters == g["iters"]).all() and (fr.hit == g["hit"]).all(), (sid, kid, prm)
        assert sha_f64(fr.t) == g["sha_t"] and sha_f64(fr.final_sdf) == g["sha_fs"], (sid, kid, prm)
        n += 1
    assert n == 140


def test_oracle_sdf_points():
    z = np.load(os.path.join(GOLDEN, "sdf_points.npz"))
    for sid in range(20):
        got = oracle.sdf_eval(sid, z["pts"])
        assert (got.view(np.uint64) == z[f"s{sid}"].view(np.uint64)).all(), sid


def test_known_answers_survey_appendix_a():
    """SURVEY.md Appendix A rows captured from the reference: hits, sum of iterations, max."""
    G = golden_frames("64x48")
    table = {(0, 0): (216, 30992, 83), (0, 10): (216, 19136, 43), (2, 0): (400, 30712, 29),
             (2, 6): (400, 32552, 29), (10, 0): (748, 63947, 160), (10, 6): (738, 51686, 66),
             (9, 0): (384, 34600, 113), (9, 10): (384, 21056, 57)}
    for (sid, kid), (hits, total, mx) in table.items():
        g = G.get(sid, kid)
        assert (int(g["hit"].sum()), int(g["iters"].sum()), int(g["iters"].max())) == (hits, total, mx)


def test_multithreaded_render_is_identical():
    g = golden_frames("64x48").get(9, 0)
    a = _render(g, 9, 0)
    b = oracle.render(9, 0, g["cam"], g["W"], g["H"], nthreads=4)
    assert (a.iters == b.iters).all() and (a.t.view(np.uint64) == b.t.view(np.uint64)).all()


def test_edge_cases():
    cam = oracle.camera14((0, 0, 5), (0, 0, 0), (0, 1, 0), 60.0, 8, 4)
    # zero rows, zero iterations budget, budget below the bisection reserve (overstep_bisect.py:40-41)
    assert oracle.render(0, 0, cam, 8, 4, row0=2, rows=0).iters.shape == (0, 8)
    fr = oracle.render(0, 0, cam, 8, 4, max_iterations=0)
    assert (fr.iters == 0).all() and (fr.hit == 0).all()
    fr = oracle.render(0, 6, cam, 8, 4, max_iterations=10)
    assert (fr.iters == 0).all() and (fr.hit == 0).all()
