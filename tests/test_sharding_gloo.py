"""Row sharding across ranks: plan arithmetic and the world-size-2 gather on the `gloo` backend
(CPU stand-in for the RCCL all-gather; the shards come from the golden maps, no GPU)."""
import os
import socket

import numpy as np
import pytest

from conftest import golden_frames
from raymarch_algo_compare_amd import sharding


@pytest.mark.parametrize("H,N", [(4320, 8), (1080, 2), (1080, 8), (48, 2), (50, 3), (7, 2), (4320, 1)])
def test_plans_partition_the_rows(H, N):
    plans = [sharding.plan_rows(H, N, r) for r in range(N)]
    rows = np.concatenate([p.image_rows() for p in plans])
    assert sorted(rows.tolist()) == list(range(H))
    for p in plans:
        ir = p.image_rows()
        if p.rows:
            assert ir[0] % 4 == 0                      # 8x4 divergence blocks never straddle ranks
        if p.cyclic:
            assert p.rows == H // N and (ir.reshape(-1, 4)[:, 0] % (4 * N) == 4 * p.rank).all()
    if H % (4 * N) == 0 and N > 1:
        assert all(p.cyclic for p in plans)


def test_assemble_host_side():
    g = golden_frames("64x48").get(10, 0)
    plans = [sharding.plan_rows(48, 4, r) for r in range(4)]
    parts = [g["iters"][p.image_rows()] for p in plans]
    assert (sharding.assemble(parts, plans) == g["iters"]).all()


def _worker(rank, world, port, tag, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for H_rows, sid in ((48, 10), (46, 0)):       # cyclic plan and contiguous fallback (46 % 8 != 0)
            g = golden_frames(tag).get(sid, 0)
            full = g["iters"][:H_rows]
            plan = sharding.plan_rows(H_rows, world, rank)
            local = torch.from_numpy(np.ascontiguousarray(full[plan.image_rows()]))
            out = sharding.all_gather_frame(local, plan)
            ok = ok and bool((out.numpy() == full).all())
            hits = torch.tensor([int(g["hit"][:H_rows][plan.image_rows()].sum())])
            dist.all_reduce(hits)
            ok = ok and int(hits) == int(g["hit"][:H_rows].sum())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_all_gather_world_size_2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, "64x48", q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
