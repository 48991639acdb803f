"""The frame's only exchange, from C: rm_comm_* / rm_gather_frame / rm_assemble_frame (RCCL all-gather of the three maps
+ device-side placement of the rows; include/rm_hip.h, BASELINE config 5).  On the one-GPU box: the assembly of 8
band-cyclic and 3 uneven contiguous shards rendered one after the other, and the whole entry-point sequence on a
communicator of one rank.  With two or more GPUs: one process per GPU under torch.distributed.run."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from raymarch_algo_compare_amd import registry, sharding
from raymarch_algo_compare_amd.camera import Camera

pytestmark = pytest.mark.gpu
vp = ctypes.c_void_p


def _cam(sid, w, h):
    sc = registry.SCENES[sid]
    return Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, w, h).params14()


def _alloc(hip, L, w, rows):
    p = [vp(), vp(), vp()]
    hip.check(L.rm_alloc_frame(w, rows, *[ctypes.byref(q) for q in p]))
    return p


def _to_host(hip, L, w, h, p):
    depth, iters, hit = np.empty((h, w), np.float32), np.empty((h, w), np.int32), np.empty((h, w), np.uint8)
    hip.check(L.rm_copy_frame_to_host(w, h, p[0], p[1], p[2], depth.ctypes.data_as(vp), iters.ctypes.data_as(vp), hit.ctypes.data_as(vp)))
    return depth, iters, hit


@pytest.mark.parametrize("sid,w,h,N", [(12, 7680, 4320, 8), (10, 1920, 1080, 3), (0, 200, 96, 4), (9, 333, 50, 3), (2, 64, 16, 4)])
def test_device_side_assembly_of_sequentially_rendered_shards(hip, sid, w, h, N):
    """Every rank's shard rendered on this GPU straight into its slot of a rank-major buffer (what ncclAllGather
    leaves), then rm_assemble_frame x 3: the image equals the unsharded frame bit for bit (band-cyclic at 7680x4320 / 8
    and 200x96 / 4 and 64x16 / 4, uneven contiguous blocks otherwise; widths that are no multiple of 16 bytes too)."""
    L = hip.load()
    cam = _cam(sid, w, h)
    plans = [sharding.plan_rows(h, N, r) for r in range(N)]
    per = plans[0].rows if plans[0].cyclic else L.rm_shard_rows(h, N)
    gathered = _alloc(hip, L, w, per * N)
    full = _alloc(hip, L, w, h)
    eb = (4, 4, 1)
    try:
        for r, plan in enumerate(plans):
            if plan.rows == 0:
                continue
            desc = hip.make_desc(sid, 0, cam, w, h, **plan.desc_kwargs())
            slot = [vp(gathered[k].value + r * per * w * eb[k]) for k in range(3)]
            hip.check(L.rm_render_device(ctypes.byref(desc), slot[0], slot[1], slot[2], None, None))
        for k in range(3):
            hip.check(L.rm_assemble_frame(N, h, w, per, 1 if plans[0].cyclic else 0, eb[k], gathered[k], full[k], None))
        depth, iters, hit = _to_host(hip, L, w, h, full)
        ref = hip.render(hip.make_desc(sid, 0, cam, w, h))
        assert (iters == ref["iters"]).all() and (hit == ref["hit"]).all()
        assert (depth.view(np.uint32) == ref["depth"].view(np.uint32)).all()
    finally:
        hip.check(L.rm_free_frame(*gathered))
        hip.check(L.rm_free_frame(*full))


def test_gather_entry_points_on_a_communicator_of_one(hip):
    """rm_comm_unique_id -> rm_comm_init(world 1) -> rm_render_device -> rm_gather_frame -> rm_comm_destroy: RCCL is
    loaded, the collective runs (one rank: a copy) and the frame lands in image order; wrong plans are refused."""
    L = hip.load()
    ident = ctypes.create_string_buffer(128)
    hip.check(L.rm_comm_unique_id(ident))
    assert any(ident.raw)
    hip.check(L.rm_comm_init(ident.raw, 1, 0))
    try:
        assert L.rm_comm_init(ident.raw, 1, 0) == -6                       # one communicator at a time
        w, h = 320, 180
        cam = _cam(10, w, h)
        desc = hip.make_desc(10, 0, cam, w, h)
        shard, full = _alloc(hip, L, w, h), _alloc(hip, L, w, h)
        hip.check(L.rm_render_device(ctypes.byref(desc), shard[0], shard[1], shard[2], None, None))
        hip.check(L.rm_gather_frame(ctypes.byref(desc), shard[0], shard[1], shard[2], full[0], full[1], full[2], None))
        depth, iters, hit = _to_host(hip, L, w, h, full)
        ref = hip.render(desc)
        assert (iters == ref["iters"]).all() and (hit == ref["hit"]).all() and (depth.view(np.uint32) == ref["depth"].view(np.uint32)).all()
        # the gather-to-root form on the same communicator (one rank: the root's own shard is placed, nothing is sent)
        hip.check(L.rm_free_frame(*full))
        full = _alloc(hip, L, w, h)
        hip.check(L.rm_gather_frame_root(ctypes.byref(desc), shard[0], shard[1], shard[2], full[0], full[1], full[2], 0, None))
        d2, i2, h2 = _to_host(hip, L, w, h, full)
        assert (i2 == ref["iters"]).all() and (h2 == ref["hit"]).all() and (d2.view(np.uint32) == ref["depth"].view(np.uint32)).all()
        assert L.rm_gather_frame_root(ctypes.byref(desc), shard[0], shard[1], shard[2], full[0], full[1], full[2], 1, None) == -6   # no such root
        assert L.rm_gather_frame_root(ctypes.byref(desc), shard[0], shard[1], shard[2], None, None, None, 0, None) == -6         # the root needs buffers
        bad = hip.make_desc(10, 0, cam, w, h, row0=0, rows=90)               # not this rank's shard of a 1-rank plan
        assert L.rm_gather_frame(ctypes.byref(bad), shard[0], shard[1], shard[2], full[0], full[1], full[2], None) == -3
        bad = hip.make_desc(10, 0, cam, w, h, row0=0, rows=60, band_rows=4, band_stride=3, band_offset=1)
        assert L.rm_gather_frame(ctypes.byref(bad), shard[0], shard[1], shard[2], full[0], full[1], full[2], None) == -3
        hip.check(L.rm_free_frame(*shard))
        hip.check(L.rm_free_frame(*full))
    finally:
        hip.check(L.rm_comm_destroy())
    desc = hip.make_desc(0, 0, _cam(0, 64, 48), 64, 48)
    p = _alloc(hip, L, 64, 48)
    assert L.rm_gather_frame(ctypes.byref(desc), p[0], p[1], p[2], p[0], p[1], p[2], None) == -7    # RM_E_RCCL: no communicator
    hip.check(L.rm_free_frame(*p))


def test_two_rank_gather_over_rccl():
    """Two processes, one GPU each (torch.distributed.run only launches and carries the communicator id): band-cyclic
    and contiguous plans.  Skipped on a one-GPU box -- the multi-GPU leg has not run on hardware in this round."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the test box has one)")
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for args in (("12", "0", "1920", "1080"), ("10", "0", "640", "360"), ("0", "0", "200", "50")):
        with socket.socket() as sk:                      # a port that is free now (no fixed port: other jobs share the host)
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                              "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "mp_gather_check.py"), *args],
                             capture_output=True, text=True, timeout=180, env=env, cwd=root)
        assert "GATHER_OK" in out.stdout, (args, out.stdout[-2000:], out.stderr[-2000:])
