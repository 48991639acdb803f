#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sphere-tracing path on MI355X.

Metric (BASELINE.json): Mrays/s + mean iterations/ray, 1920x1080 Mandelbulb / Standard, fp64
parity arithmetic, 1/2/4/8 GPUs.  A "step" is one frame render into HBM-resident depth / iterations /
hit buffers: ONE launch of the (Mandelbulb, Standard) pipeline kernel -- producer workgroups render the tiles and
hand the rays still marching after 48 trips to wavefront teams that run beside them --
plus the small stats / block-variance kernels, all on one stream.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workloads (--workload):
  mandelbulb-1080p  (default) every rank renders a full 1920x1080 frame per step: the rays of an
                    N-frame batch are sharded one frame per GPU, no data-path collective (weak scaling).
  rowshard-8k       BASELINE config 5: ONE 7680x4320 frame per step, rows dealt band-cyclically to
                    the ranks, the three maps all-gathered with RCCL over xGMI by the library itself
                    (rm_gather_frame: ncclAllGather x 3 + device-side row placement; torch.distributed only
                    launches the ranks and carries the 128-byte communicator id) -- strong scaling.

The timed region holds exactly K steps between barrier + torch.cuda.synchronize(); rank 0 prints
ONE JSON line.  `roofline` prices the render kernel against HBM (9 algorithmic bytes per ray; the
kernel is fp64-VALU/latency bound, so the fraction is tiny by construction -- DESIGN.md) and
`cpu_baseline` times the CPU oracle (oracle/, test infrastructure) on the host cores.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X vector fp64 spec peak (half the 157.3 TF fp32 vector rate)
BYTES_PER_RAY = 9               # fp32 depth + int32 iterations + uint8 hit

WORKLOADS = {
    "mandelbulb-1080p": dict(scene=10, strategy="Standard", width=1920, height=1080, sharded=False),
    "rowshard-8k": dict(scene=10, strategy="Standard", width=7680, height=4320, sharded=True),
    "pillars-8k": dict(scene=12, strategy="Standard", width=7680, height=4320, sharded=True),
}


def cpu_baseline(scene_id, strategy_id, cam14, width, height, lipschitz, gpu_maps=None, rows_cap=1080):
    """Oracle (CPU restatement of the reference, oracle/rm_oracle.c) on a bounded sample of the
    same workload (about 10-30 s of CPU work): every host thread on the whole frame (capped at
    1080 rows) + one thread on 24 centre rows."""
    from oracle import oracle
    threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64))
    rows_mt = min(height, rows_cap)
    r0 = (height - rows_mt) // 2
    t0 = time.perf_counter()
    fr = oracle.render(scene_id, strategy_id, cam14, width, height, row0=r0, rows=rows_mt, lipschitz=lipschitz,
                       nthreads=threads)
    dt = time.perf_counter() - t0
    rows_1t = min(height, 24)
    r1 = (height - rows_1t) // 2
    t1 = time.perf_counter()
    oracle.render(scene_id, strategy_id, cam14, width, height, row0=r1, rows=rows_1t, lipschitz=lipschitz, nthreads=1)
    dt1 = time.perf_counter() - t1
    parity = None
    if gpu_maps is not None:
        # the oracle frame is already here: use it as the checker for the GPU maps of the timed run
        import numpy as np
        g_depth, g_iters, g_hit = gpu_maps
        sl = slice(r0, r0 + rows_mt)
        both = (g_hit[sl] > 0) & (fr.hit > 0)
        parity = {"iter_mismatch_count": int((g_iters[sl] != fr.iters).sum()),
                  "hit_mismatch_count": int((g_hit[sl] != fr.hit).sum()),
                  "max_abs_depth_err": float(np.abs(g_depth[sl].astype(np.float64) - fr.t)[both].max()) if both.any() else 0.0,
                  "rays_compared": int(fr.iters.size)}
    return {
        "parity_vs_oracle": parity,
        "value": rows_mt * width / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"rows {r0}..{r0 + rows_mt - 1} of the {width}x{height} frame ({rows_mt * width} rays, "
                  f"{dt:.2f} s, OpenMP over rows); mean iters/ray of the sample {float(fr.iters.mean()):.2f}",
        "single_thread_us_per_ray": dt1 / (rows_1t * width) * 1e6,
        "single_thread_sample": f"rows {r1}..{r1 + rows_1t - 1}, {dt1:.2f} s",
    }


def csrc_fingerprint():
    """sha256 over the kernel sources: a profile's counters describe ONE build of the kernels."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raymarch_algo_compare_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def idle_team_pace(L, _native, scene_id, strat_id, cam14, W, H, iters_map, lip):
    """The frame's own longest rays (>= 500 iterations) marched again by wavefront teams with nothing else to do
    (rm_march_rays_team: three waves per 64 rays; filler workgroups keep the rest of the chip busy, as a frame's finished
    producers do -- KEEP BUSY, csrc/rm_kernels.h): wall time of the call / iterations of the longest ray = microseconds per
    evaluation of a dependent chain at its best -- the floor a frame cannot go below."""
    import numpy as np
    ys, xs = np.nonzero(iters_map >= 500)
    if len(ys) == 0:
        return None
    ys, xs = ys[:192], xs[:192]
    pos, fwd, right, up = (np.asarray(cam14[i:i + 3], dtype=np.float64) for i in (0, 3, 6, 9))
    hw, hh = float(cam14[12]), float(cam14[13])
    u = (2.0 * (xs + 0.5) / W - 1.0) * hw                      # camera.py:37-40, same IEEE operations as the kernel
    v = (1.0 - 2.0 * (ys + 0.5) / H) * hh
    dirs = (fwd[None, :] + right[None, :] * u[:, None]) + up[None, :] * v[:, None]
    origins = np.repeat(pos[None, :], len(xs), axis=0)
    best, it = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        hit, t, it, fs = _native.march_rays(scene_id, strat_id, origins, dirs, lipschitz=lip, team=True)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    if not (it == iters_map[ys, xs]).all():
        return {"error": "rm_march_rays_team disagrees with the frame on its longest rays"}
    return {"rays": int(len(xs)), "iter_max": int(it.max()), "call_ms": best * 1e3, "us_per_evaluation": best * 1e6 / float(it.max()),
            "note": "wall time of one rm_march_rays_team call (copies included) over the frame's own >= 500-iteration pixels / iter_max"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="mandelbulb-1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--refill-min", type=int, default=0)
    ap.add_argument("--grid-waves", type=int, default=0)
    ap.add_argument("--temporal-grid-waves", type=int, default=0,
                    help="persistent wavefronts for the temporal_order figure (0 = library default)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from raymarch_algo_compare_amd import _native, registry, sharding
    from raymarch_algo_compare_amd.camera import Camera

    L = _native.init(local_rank)
    wl = WORKLOADS[args.workload]
    scene = registry.SCENES[wl["scene"]]
    strat_id = registry.STRATEGIES[wl["strategy"]]
    W, H = wl["width"], wl["height"]
    cam = Camera(scene.camera_position or (0.0, 0.0, 5.0), scene.camera_target or (0.0, 0.0, 0.0),
                 (0.0, 1.0, 0.0), 60.0, W, H).params14()
    lip = scene.lipschitz if (wl["strategy"] == "Segment" and scene.lipschitz) else 1.0
    tuning = dict(tile_rows=args.tile_rows, refill_min=args.refill_min, grid_waves=args.grid_waves)
    if wl["sharded"]:
        plan = sharding.plan_rows(H, world, rank)
        desc = _native.make_desc(scene.id, strat_id, cam, W, H, lipschitz=lip, **tuning, **plan.desc_kwargs())
        rows_local = plan.rows
    else:
        plan = None
        desc = _native.make_desc(scene.id, strat_id, cam, W, H, lipschitz=lip, **tuning)
        rows_local = H

    dev = torch.device("cuda", local_rank)
    d_depth = torch.empty((rows_local, W), dtype=torch.float32, device=dev)
    d_iters = torch.empty((rows_local, W), dtype=torch.int32, device=dev)
    d_hit = torch.empty((rows_local, W), dtype=torch.uint8, device=dev)
    d_stats = torch.zeros(L.rm_stats_device_bytes() // 8, dtype=torch.int64, device=dev)
    # a dedicated (non-default) torch stream: its handle is non-NULL, so the kernel, the bracketing
    # events and the RCCL gather all sit on this one stream (NULL would mean "library stream")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = ctypes.c_void_p(stream.cuda_stream)
    assert sptr.value, "expected a non-default stream handle"

    full = None
    if plan is not None and world > 1:
        # the library's own RCCL communicator (include/rm_hip.h): rank 0 makes the id, torch.distributed ships it
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            _native.check(L.rm_comm_unique_id(ident))
        box = [ident.raw]
        dist.broadcast_object_list(box, src=0)
        _native.check(L.rm_comm_init(box[0], world, rank))
        full = (torch.empty((H, W), dtype=torch.float32, device=dev), torch.empty((H, W), dtype=torch.int32, device=dev),
                torch.empty((H, W), dtype=torch.uint8, device=dev))

    def step(ev0=None, ev1=None):
        if ev0 is not None:
            ev0.record(stream)
        _native.check(L.rm_render_device(ctypes.byref(desc), ctypes.c_void_p(d_depth.data_ptr()),
                                         ctypes.c_void_p(d_iters.data_ptr()), ctypes.c_void_p(d_hit.data_ptr()),
                                         ctypes.c_void_p(d_stats.data_ptr()), sptr))
        if ev1 is not None:
            ev1.record(stream)
        if full is not None:
            # the frame's only exchange: RCCL all-gather of the three maps (xGMI) + row placement, on the same stream
            _native.check(L.rm_gather_frame(ctypes.byref(desc), ctypes.c_void_p(d_depth.data_ptr()), ctypes.c_void_p(d_iters.data_ptr()),
                                            ctypes.c_void_p(d_hit.data_ptr()), ctypes.c_void_p(full[0].data_ptr()),
                                            ctypes.c_void_p(full[1].data_ptr()), ctypes.c_void_p(full[2].data_ptr()), sptr))
        return full

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(*evs[i])
    barrier()
    elapsed = time.perf_counter() - t0

    st = _native.RmStats()
    _native.check(L.rm_read_stats(ctypes.c_void_p(d_stats.data_ptr()), sptr, ctypes.byref(st)))

    # Secondary figure (never `value`): the same frames scheduled longest-tile-first from the per-tile
    # cost the previous frame left behind (tile_order_mode 1) on a smaller persistent grid.  Every ray
    # is recomputed; only the order in which tiles are handed to the waves changes.
    temporal = None
    if not wl["sharded"]:
        desc_t = _native.make_desc(scene.id, strat_id, cam, W, H, lipschitz=lip, tile_rows=args.tile_rows,
                                   refill_min=args.refill_min, grid_waves=args.temporal_grid_waves, tile_order_mode=1)

        def step_t():
            _native.check(L.rm_render_device(ctypes.byref(desc_t), ctypes.c_void_p(d_depth.data_ptr()),
                                             ctypes.c_void_p(d_iters.data_ptr()), ctypes.c_void_p(d_hit.data_ptr()),
                                             ctypes.c_void_p(d_stats.data_ptr()), sptr))
        for _ in range(max(2, args.warmup)):
            step_t()
        barrier()
        tt0 = time.perf_counter()
        for _ in range(args.steps):
            step_t()
        barrier()
        temporal = time.perf_counter() - tt0

    # Isolated store path (the flush code of the render kernel, no marching): what "fraction of the HBM
    # roofline on the write path" can mean for a kernel that writes 9 B per ~20 000 fp64 operations.
    tm = _native.RmTiming()
    tm.warmup, tm.repeats = 3, 20
    store_gbps = None
    p_depth, p_iters, p_hit = torch.empty_like(d_depth), torch.empty_like(d_iters), torch.empty_like(d_hit)   # probe scratch
    if L.rm_bench_store_path(W, rows_local, ctypes.c_void_p(p_depth.data_ptr()), ctypes.c_void_p(p_iters.data_ptr()),
                             ctypes.c_void_p(p_hit.data_ptr()), ctypes.byref(tm)) == 0 and tm.ms_median > 0:
        store_gbps = BYTES_PER_RAY * rows_local * W / (tm.ms_median * 1e-3) / 1e9
    # the same probe at 7680x4320 (BASELINE config 5's frame): a 1080p launch writes 18.7 MB in ~8 us, too short to
    # fill the memory pipeline; the north star's ">= 40 % of the HBM roofline on the write path" is read at this size
    store8k_gbps = None
    tp8k = None
    if rank == 0:
        W8, H8 = 7680, 4320
        q_depth = torch.empty((H8, W8), dtype=torch.float32, device=dev)
        q_iters = torch.empty((H8, W8), dtype=torch.int32, device=dev)
        q_hit = torch.empty((H8, W8), dtype=torch.uint8, device=dev)
        tm8 = _native.RmTiming()
        tm8.warmup, tm8.repeats = 3, 20
        if L.rm_bench_store_path(W8, H8, ctypes.c_void_p(q_depth.data_ptr()), ctypes.c_void_p(q_iters.data_ptr()),
                                 ctypes.c_void_p(q_hit.data_ptr()), ctypes.byref(tm8)) == 0 and tm8.ms_median > 0:
            store8k_gbps = BYTES_PER_RAY * W8 * H8 / (tm8.ms_median * 1e-3) / 1e9
        # the throughput regime: the same scene / strategy at 7680x4320 (BASELINE config 5's frame, unsharded), events
        # inside the library around each of 5 frames after 2 warm-ups -- reported as `throughput_8k`, never `value`
        if not wl["sharded"]:
            sc8 = Camera(scene.camera_position or (0.0, 0.0, 5.0), scene.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W8, H8).params14()
            d8 = _native.make_desc(scene.id, strat_id, sc8, W8, H8, lipschitz=lip)
            t8, s8 = _native.RmTiming(), _native.RmStats()
            t8.warmup, t8.repeats = 2, 5
            if L.rm_bench_device(ctypes.byref(d8), ctypes.c_void_p(q_depth.data_ptr()), ctypes.c_void_p(q_iters.data_ptr()),
                                 ctypes.c_void_p(q_hit.data_ptr()), ctypes.byref(s8), ctypes.byref(t8)) == 0 and t8.ms_median > 0:
                tp8k = {"value": W8 * H8 / (t8.ms_median * 1e-3) / 1e6, "unit": "Mrays/s", "ms_per_frame": t8.ms_median,
                        "workload": f"{scene.name}/{wl['strategy']} {W8}x{H8}, one GPU, default schedule",
                        "mean_iters_per_ray": s8.sum_iters / max(s8.total_rays, 1)}
        del q_depth, q_iters, q_hit
    # per-pass device time of a frame (events inside the library, between the passes on `stream`): a short
    # untimed loop after the measurement, so the numbers can be held against the rocprofv3 kernel stats
    passes = None
    long_marks = None
    if L.rm_set_pass_timing(1) == 0:
        acc_ms, nfr = [0.0] * 4, 5
        for _ in range(nfr):
            step()
            npass = ctypes.c_int32(0)
            pms = (ctypes.c_float * 4)()
            _native.check(L.rm_get_pass_ms(sptr, ctypes.byref(npass), pms))
            for i in range(npass.value):
                acc_ms[i] += pms[i] / nfr
        # one launch per pass: the kernels one by one; single launch: spans between the marks its waves leave
        names = (["launch -> tile counter exhausted", "-> last producer wave out of fresh pixels", "-> last producer wave exited",
                  "-> end of the kernel (teams' tail)"] if npass.value == 4
                 else ["render_kernel (first pass)", "resume pass 1", "resume pass 2"])
        passes = {names[i]: acc_ms[i] for i in range(npass.value)}
        lm = (ctypes.c_float * 4)()
        if npass.value == 4 and L.rm_long_ray_marks(lm) == 0:
            long_marks = [round(float(lm[i]), 3) for i in range(4)]
        L.rm_set_pass_timing(0)
    # Secondary figure (never `value`): FOUR such frames in ONE launch (rm_render_batch, the sweep path of the reference's
    # real workload): the tails of the frames overlap, so the device time per frame is the throughput regime's
    batched = None
    if rank == 0 and not wl["sharded"]:
        try:
            import numpy as np
            shape = _native.make_desc(scene.id, strat_id, cam, W, H, lipschitz=lip)
            _native.render_batch(shape, np.tile(np.asarray(cam, dtype=np.float64), (4, 1)))      # warm-up
            ob = _native.render_batch(shape, np.tile(np.asarray(cam, dtype=np.float64), (4, 1)))
            same = bool((ob["iters"][0] == ob["iters"][3]).all())
            batched = {"frames_per_launch": 4, "ms_per_frame": ob["ms_total"] / 4.0, "value": 4.0 * W * H / (ob["ms_total"] * 1e-3) / 1e6,
                       "unit": "Mrays/s", "frames_identical": same,
                       "note": "rm_render_batch: 4 frames of this workload in one launch, device time of the launch / 4 (hipEvents)"}
        except Exception as e:
            batched = {"error": str(e)}
    pace = None
    if rank == 0 and scene.id == 10 and not wl["sharded"]:
        try:
            pace = idle_team_pace(L, _native, scene.id, strat_id, cam, W, H, d_iters.cpu().numpy(), lip)
        except Exception as e:      # a side measurement must not hide the headline
            pace = {"error": str(e)}
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    local = torch.tensor([elapsed, float(st.total_rays), float(st.sum_iters), sum(kernel_ms) / len(kernel_ms)],
                         dtype=torch.float64, device=dev)
    if world > 1:
        mx = local.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = local.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed_max, rays_step, iters_step, kms = float(mx[0]), float(sm[1]), float(sm[2]), float(mx[3])
    else:
        elapsed_max, rays_step, iters_step, kms = elapsed, float(st.total_rays), float(st.sum_iters), local[3].item()

    if rank == 0:
        total_rays = rays_step * args.steps
        value = total_rays / elapsed_max / 1e6
        rays_per_launch = float(st.total_rays)
        achieved = BYTES_PER_RAY * rays_per_launch / (kms * 1e-3) / 1e9
        # HBM bytes per launch from the committed counter profile -- only while it describes THIS build of the kernels
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_r03.json")
        if os.path.exists(pmc_path) and args.workload == "mandelbulb-1080p":
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("csrc_sha16") == csrc_fingerprint():
                    traffic, traffic_src = pmc.get("hbm_bytes_per_launch"), pmc.get("source")
                else:
                    traffic_src = f"profiles/pmc_r03.json was collected on kernel sources {pmc.get('csrc_sha16')}, this build is {csrc_fingerprint()}: stale, not reported"
            except Exception:
                traffic = None
        # fp64 work estimate: evaluations x ~1.76 fractal iterations x ~1100 flop
        line = {
            "metric": "Mrays/sec (mean iters/ray alongside), 1920x1080 Mandelbulb/Standard" if not wl["sharded"]
                      else f"Mrays/sec, {W}x{H} {scene.name}/Standard row-sharded",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if wl["sharded"] else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "mean_iters_per_ray": iters_step / max(rays_step, 1.0),
            "mean_sdf_evals_per_ray": float(st.sum_evals) / max(float(st.total_rays), 1.0),
            "config": {"workload": f"{scene.name}/{wl['strategy']} {W}x{H}, MarchConfig(512, 1e-4, 100.0), "
                                   f"camera {scene.camera_position or (0.0, 0.0, 5.0)}",
                       "frames_per_step": 1 if wl["sharded"] else world,
                       "rays_per_step": int(rays_step),
                       "parallelism": (f"rowshard{world}-bandcyclic4+allgather" if wl["sharded"] else f"frame-per-gpu x{world}"),
                       "schedule": ("library defaults (RmFrameDesc knobs all 0): one launch per frame, tile order centre-out, rays struck "
                                    "from their tile at 16 trips and handed to wavefront teams at 48 (near-surface rays at once), keep_busy on (finished producer "
                                    "workgroups execute fp32 filler until the teams are through: include/rm_hip.h)")
                                   if scene.id == 10 and not wl["sharded"] else "library defaults (RmFrameDesc knobs all 0)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms_avg": kms, "passes_ms": passes, "bytes_per_ray": BYTES_PER_RAY,
                         "store_path_GBps": store_gbps,
                         "store_path_frac": (store_gbps / HBM_PEAK_GBPS) if store_gbps else None,
                         "store_path_8k_GBps": store8k_gbps,
                         "store_path_8k_frac": (store8k_gbps / HBM_PEAK_GBPS) if store8k_gbps else None,
                         # what actually binds the frame (the contract's `bound` is hbm | mfma; neither binds this path)
                         "binding": "fp64 dependent-chain latency of the frame's longest ray (see chain_latency, valu_fp64)",
                         "note": "write-only path, 9 B/ray; kernel_ms_avg = device time of one frame (events on the launch "
                                 "stream), passes_ms = its spans / kernels; the frame is fp64-VALU / ray-latency bound "
                                 "(DESIGN.md); store_path_* = the flush code alone at this frame size, store_path_8k_* = the "
                                 "same at 7680x4320"},
        }
        # the longest ray is one dependent chain: iter_max evaluations, each as fast as a wavefront team runs it when it is
        # all there is to run -- measured here, on the frame's own longest rays (idle_team_pace)
        if pace is not None and "us_per_evaluation" in pace:
            chain_floor_ms = float(st.iter_max) * pace["us_per_evaluation"] * 1e-3
            line["chain_latency"] = {"iter_max": int(st.iter_max), "idle_team": pace, "floor_ms": chain_floor_ms,
                                     "frac": chain_floor_ms / kms if kms > 0 else None, "long_ray_marks_ms": long_marks,
                                     "note": "frame time cannot go below the longest ray's chain; frac = floor / measured frame; "
                                             "long_ray_marks_ms = [earliest, latest hand-over to a team, shortest, longest stay with a team] "
                                             "of the rays that end at >= 500 iterations, device clock inside the timed kernel's twin"}
        elif pace is not None:
            line["chain_latency"] = pace
        # what actually bounds the kernel: fp64 vector work.  530 fp64 flop per fractal iteration
        # (SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 of profiles/, one lane) x 1.76 fractal iterations per SDF
        # evaluation on this view -- an estimate, reported next to the vector-fp64 peak.
        if scene.id == 10:
            flop = iters_step * 1.76 * 530.0
            tf = flop / (elapsed_max / args.steps) / 1e12
            line["valu_fp64"] = {"achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "flop_per_ray": flop / max(rays_step, 1.0),
                                 "note": "estimate from counter-measured flop per fractal iteration; the frame is bound by "
                                         "the latency of its longest rays, not by vector throughput (DESIGN.md)"}
        if temporal is not None:
            line["temporal_order"] = {
                "value": rays_step * args.steps / temporal / 1e6 if world == 1 else None, "unit": "Mrays/s",
                "ms_per_step": temporal / args.steps * 1e3, "grid_waves": args.temporal_grid_waves,
                "note": "tile_order_mode=1: tiles handed out longest-first using the previous frame's per-tile "
                        "max-iteration map; identical outputs, every ray recomputed; rank-0 local figure"}
        if tp8k is not None:
            line["throughput_8k"] = tp8k
        if batched is not None:
            line["batched_frames"] = batched
        if not args.no_cpu_baseline:
            try:
                # rank 0 only, every N; with N > 1 a shorter sample (the other ranks wait at the end of the run)
                maps = (d_depth.cpu().numpy(), d_iters.cpu().numpy(), d_hit.cpu().numpy()) if plan is None else None
                line["cpu_baseline"] = cpu_baseline(scene.id, strat_id, cam, W, H, lip, maps, rows_cap=1080 if world == 1 else 360)
            except Exception as e:  # the oracle is a checker; a failure here must not hide the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port",
                                        "sample": f"unavailable: {e}"}
        print(json.dumps(line), flush=True)
    if full is not None:
        _native.check(L.rm_comm_destroy())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
