"""The product's kernel headers (csrc/rm_scenes.h, rm_strategies.h, rm_camera.h, rm_math*.h)
compiled for the HOST by g++ (tests/native/host_check.cpp) and diffed against the reference
goldens.  This checks, without a GPU, the very source the gfx950 kernels are built from; on the
host the not-yet-exact transcendentals resolve to libm, so every scene is expected bit-exact."""
import ctypes

import numpy as np
import pytest

from conftest import PARAM_ORDER, build_native, golden_frames, golden_param_cases, sha_f64


@pytest.fixture(scope="module")
def hostlib():
    L = ctypes.CDLL(build_native("host_check"))
    dp = ctypes.POINTER(ctypes.c_double)
    L.rmh_render.argtypes = ([ctypes.c_int] * 3 + [ctypes.c_double] * 3 + [ctypes.c_int, dp] + [ctypes.c_int] * 4
                             + [ctypes.c_void_p, dp, ctypes.c_void_p, dp, dp])
    L.rmh_sdf_eval.argtypes = [ctypes.c_int, dp, ctypes.c_size_t, dp]
    return L


def _render(L, g, sid, kid, full, prm=None):
    n = g["rows"] * g["W"]
    hit, t = np.empty(n, np.uint8), np.empty(n, np.float64)
    it, fs = np.empty(n, np.int32), np.empty(n, np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    cam = np.ascontiguousarray(g["cam"])
    rc = L.rmh_render(sid, kid, g["max_iterations"], g["hit_threshold"], g["max_distance"], g["lipschitz"], full,
                      cam.ctypes.data_as(dp), g["W"], g["H"], g["row0"], g["rows"], hit.ctypes.data, t.ctypes.data_as(dp),
                      it.ctypes.data, fs.ctypes.data_as(dp),
                      None if prm is None else np.array([float(prm[k]) for k in PARAM_ORDER]).ctypes.data_as(dp))
    assert rc == 0
    return hit, t, it, fs


@pytest.mark.parametrize("full", [1, 0])
def test_state_machines_match_reference_64x48(hostlib, full):
    G = golden_frames("64x48")
    for sid, kid in G.pairs:
        g = G.get(sid, kid)
        hit, t, it, fs = _render(hostlib, g, sid, kid, full)
        assert (it == g["iters"].reshape(-1)).all(), (sid, kid)
        assert (hit == g["hit"].reshape(-1)).all(), (sid, kid)
        assert sha_f64(t) == g["sha_t"], (sid, kid)
        if full:
            assert sha_f64(fs) == g["sha_fs"], (sid, kid)


@pytest.mark.parametrize("tag", ["16x12_it100", "leak", "rows1080"])
def test_state_machines_other_shapes(hostlib, tag):
    G = golden_frames(tag)
    for sid, kid in G.pairs:
        g = G.get(sid, kid)
        hit, t, it, fs = _render(hostlib, g, sid, kid, 1)
        assert (it == g["iters"].reshape(-1)).all() and (hit == g["hit"].reshape(-1)).all(), (tag, sid, kid)
        assert sha_f64(t) == g["sha_t"] and sha_f64(fs) == g["sha_fs"], (tag, sid, kid)


def test_tiny_budgets_match_oracle(hostlib):
    """max_iterations 0, 1, 2, 15, 16, 17: empty loops and Overstep-Bisect's phase-1 reserve."""
    from oracle import oracle
    g = golden_frames("16x12_it100").get(0, 0)
    for mi in (0, 1, 2, 15, 16, 17):
        for kid in range(11):
            gg = dict(g, max_iterations=mi)
            hit, t, it, fs = _render(hostlib, gg, 0, kid, 1)
            fr = oracle.render(0, kid, g["cam"], g["W"], g["H"], max_iterations=mi)
            assert (it == fr.iters.reshape(-1)).all() and (hit == fr.hit.reshape(-1)).all(), (mi, kid)
            assert (t.view(np.uint64) == fr.t.reshape(-1).view(np.uint64)).all(), (mi, kid)
            assert (fs.view(np.uint64) == fr.final_sdf.reshape(-1).view(np.uint64)).all(), (mi, kid)


def test_state_machines_with_non_default_strategy_parameters(hostlib):
    """StratParams (csrc/rm_core.h) through the resumable state machines, against frames the reference's classes
    marched with non-default constructor arguments (tests/golden/frames_params_48x36.npz)."""
    n = 0
    for sid, kid, prm, g in golden_param_cases():
        hit, t, it, fs = _render(hostlib, g, sid, kid, 1, prm)
        assert (it == g["iters"].reshape(-1)).all() and (hit == g["hit"].reshape(-1)).all(), (sid, kid, prm)
        assert sha_f64(t) == g["sha_t"] and sha_f64(fs) == g["sha_fs"], (sid, kid, prm)
        n += 1
    assert n == 140
