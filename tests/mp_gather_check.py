#!/usr/bin/env python3
"""One rank of the multi-GPU gather check (launched by tests/test_gpu_gather.py through torch.distributed.run, one
process per GPU): render this rank's row shard of a frame, gather it with rm_gather_frame (RCCL from C -- torch is
only the launcher and carries the 128-byte communicator id), compare the assembled frame with the unsharded render.
usage (under torchrun): mp_gather_check.py <scene> <strategy> <width> <height>"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    from raymarch_algo_compare_amd import _native, registry, sharding
    from raymarch_algo_compare_amd.camera import Camera
    sid, kid, W, H = (int(v) for v in sys.argv[1:5])
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dist.init_process_group("gloo")                     # the id exchange only; the data path is RCCL inside librm_hip.so
    L = _native.init(local)
    ident = ctypes.create_string_buffer(128)
    if rank == 0:
        _native.check(L.rm_comm_unique_id(ident))
    box = [ident.raw]
    dist.broadcast_object_list(box, src=0)
    _native.check(L.rm_comm_init(box[0], world, rank))
    sc = registry.SCENES[sid]
    cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
    plan = sharding.plan_rows(H, world, rank)
    desc = _native.make_desc(sid, kid, cam, W, H, **plan.desc_kwargs())
    vp = ctypes.c_void_p
    shard = [vp(), vp(), vp()]
    full = [vp(), vp(), vp()]
    _native.check(L.rm_alloc_frame(W, max(plan.rows, 1), *[ctypes.byref(p) for p in shard]))
    _native.check(L.rm_alloc_frame(W, H, *[ctypes.byref(p) for p in full]))
    ok = True
    for _ in range(3):                                   # repeated frames reuse the landing buffers
        _native.check(L.rm_render_device(ctypes.byref(desc), shard[0], shard[1], shard[2], None, None))
        _native.check(L.rm_gather_frame(ctypes.byref(desc), shard[0], shard[1], shard[2], full[0], full[1], full[2], None))
        depth, iters, hit = np.empty((H, W), np.float32), np.empty((H, W), np.int32), np.empty((H, W), np.uint8)
        _native.check(L.rm_copy_frame_to_host(W, H, full[0], full[1], full[2], depth.ctypes.data_as(vp), iters.ctypes.data_as(vp),
                                              hit.ctypes.data_as(vp)))
        ref = _native.render(_native.make_desc(sid, kid, cam, W, H))
        ok = ok and (iters == ref["iters"]).all() and (hit == ref["hit"]).all() and (depth.view(np.uint32) == ref["depth"].view(np.uint32)).all()
        # the gather-to-root form: only the last rank ends up with the image
        root = world - 1
        _native.check(L.rm_gather_frame_root(ctypes.byref(desc), shard[0], shard[1], shard[2], full[0] if rank == root else None,
                                             full[1] if rank == root else None, full[2] if rank == root else None, root, None))
        if rank == root:
            _native.check(L.rm_copy_frame_to_host(W, H, full[0], full[1], full[2], depth.ctypes.data_as(vp), iters.ctypes.data_as(vp),
                                                  hit.ctypes.data_as(vp)))
            ok = ok and (iters == ref["iters"]).all() and (hit == ref["hit"]).all() and (depth.view(np.uint32) == ref["depth"].view(np.uint32)).all()
        else:
            _native.check(L.rm_stream_synchronize(None))
    _native.check(L.rm_comm_destroy())
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    dist.destroy_process_group()
    if rank == 0:
        print("GATHER_OK" if all(flags) else f"GATHER_MISMATCH {flags}", flush=True)
    return 0 if all(flags) else 1


if __name__ == "__main__":
    raise SystemExit(main())
