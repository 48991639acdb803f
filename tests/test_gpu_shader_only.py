"""The strategies and uniforms that exist only in the reference's fragment shader -- Safe-Relaxed (kernel id 11,
gpu/shaders/strategies.glsl:508-541), Dense-March (12, :559-593) and `stepScale` of standard() (:47) -- on the CPU
path's arithmetic.  PARITY UNPINNED: no Python statement of them exists in the reference and its shader computes in
fp32 (moderngl is absent here), so no fixture covers them.  What these tests hold: the HIP state machines equal the
oracle's restatement of the same shader text bit for bit, through every launch structure, and GPURunner reaches
them under the shader's own ids and uniform names."""
import numpy as np
import pytest

from conftest import golden_frames
from raymarch_algo_compare_amd import registry
from raymarch_algo_compare_amd.camera import Camera

pytestmark = pytest.mark.gpu

CASES = [(11, dict()), (11, dict(omega=1.6)), (11, dict(omega=1.0)), (12, dict(step_scale=0.5, dense_min_step=0.002)),
         (12, dict()), (12, dict(step_scale=0.6, dense_min_step=0.01)), (0, dict(step_scale=0.6))]


def _same(out, ref):
    return ((out["iters"] == ref.iters).all() and (out["hit"] == ref.hit).all()
            and (out["t_raw"].view(np.uint64) == ref.t.view(np.uint64)).all()
            and (out["final_sdf"].view(np.uint64) == ref.final_sdf.view(np.uint64)).all())


def test_frames_equal_the_oracle_text(hip):
    from oracle import oracle
    W, H = 160, 120
    for sid in (0, 2, 3, 8, 9, 10, 12, 13, 14, 16):
        sc = registry.SCENES[sid]
        cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
        for kid, prm in CASES:
            for mi in (96, 900):
                ref = oracle.render(sid, kid, cam, W, H, max_iterations=mi, params=prm, nthreads=8)
                for sched in (dict(), dict(suspend_after=(5, 21), pipeline=2), dict(suspend_after=(7, 0), pipeline=1), dict(tile_order_mode=3)):
                    desc = hip.make_desc(sid, kid, cam, W, H, 0, H, mi, 1e-4, 100.0, 1.0, True, params=prm, **sched)
                    out = hip.render(desc, want_t_raw=True, want_final_sdf=True)
                    assert _same(out, ref), (sid, kid, prm, mi, sched)


def test_mandelbulb_1080p_dense_march_and_understep_full_frame(hip):
    """The calibration configuration of the reference (groundtruth.py:73-87: stepScale 0.5, minStep 0.002) and the understep
    oracle (stepScale 0.6) at the bench's frame size, single launch, every ray against the oracle."""
    from oracle import oracle
    sc = registry.SCENES[10]
    cam = Camera(sc.camera_position, sc.camera_target, (0.0, 1.0, 0.0), 60.0, 1920, 1080).params14()
    for kid, prm in ((12, dict(step_scale=0.5, dense_min_step=0.002)), (0, dict(step_scale=0.6)), (11, dict(omega=1.4))):
        ref = oracle.render(10, kid, cam, 1920, 1080, params=prm, nthreads=64)
        out = hip.render(hip.make_desc(10, kid, cam, 1920, 1080, full=True, params=prm), want_t_raw=True, want_final_sdf=True)
        assert _same(out, ref), (kid, prm)


def test_explicit_rays_and_batches(hip):
    from oracle import oracle
    rng = np.random.default_rng(3)
    o = rng.normal(size=(3000, 3)) * 0.3 + np.array([0.0, 0.0, 4.0])
    d = -o + rng.normal(size=(3000, 3)) * 0.6
    for sid in (0, 9, 10):
        for kid, prm in CASES:
            hit, t, it, fs = hip.march_rays(sid, kid, o, d, max_iterations=300, params=prm)
            rh, rt, ri, rf = oracle.march_rays(sid, kid, o, d, max_iterations=300, params=prm)
            assert (it == ri).all() and (hit == rh).all(), (sid, kid, prm)
            assert (t.view(np.uint64) == rt.view(np.uint64)).all() and (fs.view(np.uint64) == rf.view(np.uint64)).all(), (sid, kid, prm)
    # one batch launch, per-frame parameters: Dense-March at three (stepScale, minStep) pairs
    sc = registry.SCENES[10]
    W, H = 96, 72
    cam = Camera(sc.camera_position, sc.camera_target, (0.0, 1.0, 0.0), 60.0, W, H).params14()
    prms = [dict(step_scale=0.5, dense_min_step=0.002), dict(step_scale=1.0, dense_min_step=1e-4), dict(step_scale=0.7, dense_min_step=0.02)]
    shape = hip.make_desc(10, 12, cam, W, H)
    res = hip.render_batch(shape, np.tile(cam, (3, 1)), [dict(max_iterations=400, params=p) for p in prms])
    for f, p in enumerate(prms):
        ref = oracle.render(10, 12, cam, W, H, max_iterations=400, params=p)
        assert (res["iters"][f] == ref.iters).all() and (res["hit"][f] == ref.hit).all(), p


def test_gpurunner_reaches_them_by_shader_id_and_uniform_name(hip):
    from oracle import oracle
    from raymarch_algo_compare_amd.config import MarchConfig, RenderConfig
    from raymarch_algo_compare_amd.runner import GPURunner
    rc = RenderConfig(width=80, height=60)
    mc = MarchConfig(max_iterations=600, hit_threshold=1e-5)
    cam = Camera(rc.camera_position, rc.camera_target, rc.camera_up, rc.fov_degrees, 80, 60).params14()
    r = GPURunner()
    px, _ = r.render(2, 9, rc, mc, params={"stepScale": 0.5, "minStep": 0.002})           # Cube, dense_march
    ref = oracle.render(2, 12, cam, 80, 60, max_iterations=600, hit_threshold=1e-5, params=dict(step_scale=0.5, dense_min_step=0.002))
    assert (px[..., 0] == ref.hit).all() and (np.round(px[..., 1] * 600) == ref.iters).all()
    px, _ = r.render(0, 8, rc, mc, params={"omega": 1.5})                                  # Sphere, safe_relaxed
    ref = oracle.render(0, 11, cam, 80, 60, max_iterations=600, hit_threshold=1e-5, params=dict(omega=1.5))
    assert (px[..., 0] == ref.hit).all() and (np.round(px[..., 1] * 600) == ref.iters).all()
    px, _ = r.render(0, 0, rc, mc, params={"stepScale": 0.6})                              # the understep oracle
    ref = oracle.render(0, 0, cam, 80, 60, max_iterations=600, hit_threshold=1e-5, params=dict(step_scale=0.6))
    assert (px[..., 0] == ref.hit).all() and (np.round(px[..., 1] * 600) == ref.iters).all()
    # the seam's default minStep for dense_march: max(hit_threshold, min_step_fraction * max_distance)
    px, _ = r.render(0, 9, rc, mc)
    ref = oracle.render(0, 12, cam, 80, 60, max_iterations=600, hit_threshold=1e-5,
                        params=dict(dense_min_step=max(1e-5, mc.min_step_fraction * mc.max_distance)))
    assert (px[..., 0] == ref.hit).all() and (np.round(px[..., 1] * 600) == ref.iters).all()
