/*
 * rm_hip.h -- C ABI of librm_hip.so, the MI355X (gfx950) sphere-tracing engine.
 *
 * Drop-in boundary for ONE path of kylegrover/raymarch-algo-compare: the per-ray
 * SDF sphere-tracing frame render.  Each entry point names the reference
 * interface it replaces (paths relative to raymarching_benchmark/).  Plain
 * pointers and sizes only; no C++ or torch types cross this boundary.  Nothing
 * here falls back to a CPU implementation: every call fails with RM_E_NO_DEVICE /
 * RM_E_HIP when no gfx950 device is usable.
 *
 * Ids: scene_id = index in get_all_scenes() (scenes/catalog.py:640-663), 0..19;
 *      strategy_id = index in the STRATEGIES dict (strategies/__init__.py:16-28), 0..10.
 * Threading: every entry point may be called from any host thread; calls are serialised by one lock inside the
 * library (one in-flight CALL per device), calls are synchronous unless stated, and frames enqueued on different streams
 * are ordered on the device by an event (they share one workspace).  A `stream` argument must be a hipStream_t of the
 * HIP runtime this library is bound to (rm_runtime_info).  A process can hold two copies of libamdhip64 -- this library
 * loaded before PyTorch, whose wheel bundles its own -- and HIP dereferences whatever handle it is given, so a stream of
 * the other copy corrupts the runtime (round 2's SIGABRT).  While two copies are mapped, only streams made by
 * rm_stream_create are accepted (anything else: RM_E_BAD_ARG, the handle is not touched); with one copy mapped every
 * hipStream_t of the process is that runtime's and is accepted (e.g. a torch stream when torch was imported first).
 * Hosts without a HIP binding of their own take streams from rm_stream_create.
 */
#ifndef RM_HIP_H
#define RM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RM_OK 0
#define RM_E_BAD_SCENE (-1)
#define RM_E_BAD_STRATEGY (-2)
#define RM_E_BAD_DIMS (-3)
#define RM_E_NO_DEVICE (-4)
#define RM_E_HIP (-5)
#define RM_E_BAD_ARG (-6)
#define RM_E_RCCL (-7)     /* librccl could not be loaded, or one of its calls failed (text in rm_last_error) */

#define RM_NUM_SCENES 20
#define RM_NUM_STRATEGIES 11
/* Strategy ids [0, RM_NUM_STRATEGIES) are the reference's CPU registry (strategies/__init__.py:16-28).  Two more exist
 * only in its fragment shader (gpu/shaders/strategies.glsl:508-541 safe_relaxed = 11, :559-593 dense_march = 12; shader
 * ids 8 and 9): here they run the shader's control flow on the CPU path's arithmetic (binary64, this camera).  PARITY
 * UNPINNED: no Python statement of them exists and the shader computes in fp32, so nothing of the reference can check
 * them; they are not part of the registry (rm_num_strategies() stays 11, names / CLI "all" unchanged). */
#define RM_NUM_STRATEGY_KERNELS 13
#define RM_HIST_BINS 544 /* iterations histogram bins; counts >= RM_HIST_BINS-1 share the last bin */
#define RM_MAX_TIMED 256

/* The constructor arguments of the reference's strategy classes -- what `STRATEGIES[key](**kwargs)` takes and
 * what its accelerator seam threads through as per-run uniform overrides (gpu/runner.py:108-124, swept by
 * param_grid.py:20-27 / sweep.py:181,222-223) -- plus four constants the CPU march() bodies hold as literals.
 * Defaults (rm_default_strategy_params) are the reference's; with them every result is bit-identical to the
 * registry's no-argument instances.  A strategy reads only its own fields. */
typedef struct RmStrategyParams {
    double omega;                     /* 1.2   RelaxedSphereTracing(omega)                 strategies/relaxed_sphere.py:17 */
    double ar_omega_min;              /* 1.0   AutoRelaxedSphereTracing(omega_min,          strategies/auto_relaxed.py:21-23 */
    double ar_omega_max;              /* 2.0     omega_max, */
    double ar_smoothing;              /* 0.7     smoothing, */
    double ar_growth_rate;            /* 1.05    growth_rate, */
    double ar_decay_rate;             /* 0.7     decay_rate) */
    double beta;                      /* 0.3   SlopeAutoRelaxed(beta)                      strategies/slope_auto_relaxed.py:25 */
    double overstep_min_step;         /* 0.01  OverstepBisectTracing(min_step_factor,       strategies/overstep_bisect.py:18 */
    double hybrid_stuck_step_ratio;   /* 0.001 AdaptiveHybridTracing(stuck_step_ratio,      strategies/adaptive_hybrid.py:17-19 */
    double hybrid_min_step;           /* 0.005   min_step_factor, */
    double margin;                    /* 0.05  SkippingSpheresTracing: the literal `margin` of march() (skipping_spheres.py:30),
                                               uniform `margin` of the GLSL seam (gpu/runner.py:115) */
    double ar_omega_init;             /* 1.2   AutoRelaxedSphereTracing: the literal start value `omega = 1.2` of march() (auto_relaxed.py:41);
                                               the GLSL seam's `omega` uniform drives its auto-relaxed marcher too (param_grid.py:23) */
    int32_t overstep_bisection_steps; /* 16      bisection_steps)                           strategies/overstep_bisect.py:18 */
    int32_t hybrid_stuck_threshold;   /* 5       stuck_threshold)                           strategies/adaptive_hybrid.py:17 */
    int32_t segment_bisection_steps;  /* 8     SegmentTracing: the literal `range(8)` of march()      segment_tracing.py:79 */
    int32_t revaa_bisection_steps;    /* 8     RevAAApproxTracing: the literal `range(8)` of march()  rev_affine.py:70 */
    /* shader-only uniforms (the CPU strategies have no such constant; defaults change no bit of the registry's results) */
    double step_scale;                /* 1.0   `stepScale`: standard() multiplies every step by it (strategies.glsl:24,47 -- the
                                               understep oracle of gpu/groundtruth.py:59-61 uses 0.6), dense_march() too (:570) */
    double dense_min_step;            /* 1e-4  `minStep` as dense_march reads it (strategies.glsl:570; the seam's default is
                                               max(hit_threshold, min_step_fraction * max_distance), gpu/runner.py:117-118) */
} RmStrategyParams;

/* MarchConfig (config.py:19-29) -- the three fields the CPU strategies read -- plus
 * SegmentTracing.lipschitz as wired by run_once (main.py:58-61) and the strategy's constructor arguments.
 * full != 0 also produces MarchResult.final_sdf (costs the reference's tail evaluations).
 * use_params == 0: `params` is ignored and every strategy constant has the reference's default (a zeroed
 * record is therefore a valid default configuration); != 0: `params` is read (fill it with
 * rm_default_strategy_params first and override what the run sweeps). */
typedef struct RmMarchConfig {
    int32_t max_iterations; /* 512   */
    int32_t full;
    double hit_threshold;   /* 1e-4  */
    double max_distance;    /* 100.0 */
    double lipschitz;       /* 1.0   */
    int32_t use_params;
    int32_t reserved;       /* 0 */
    RmStrategyParams params;
} RmMarchConfig;

/* One frame (or a row shard of it).  cam[14] = position, forward, right, up,
 * half_width, half_height exactly as Camera.__init__ computes them (core/camera.py:11-33);
 * rows [row0, row0+rows) of a width x height image are rendered, row 0 = top
 * (core/types.py:90).  Output arrays hold rows*width elements, row-major from row0. */
typedef struct RmFrameDesc {
    int32_t scene_id;
    int32_t strategy_id;
    int32_t width, height;
    int32_t row0, rows;
    double cam[14];
    RmMarchConfig march;
    int32_t tile_rows;   /* 0 = default; rows per 64-pixel-wide wave tile: 4, or 1 for the single launch (pipeline = 2 with
                          * suspension) of scenes with a team form -- default there when tile_order_mode = 1 */
    int32_t refill_min;  /* 0 = default; idle lanes required before a wave refills */
    int32_t grid_waves;  /* 0 = default (fill the device); persistent wavefront count */
    /* Band-cyclic row sharding (multi-GPU load balance): local row y of this slice is image row
     *   row0 + ((y / band_rows) * band_stride + band_offset) * band_rows + y % band_rows.
     * band_rows = 0 (or band_stride <= 1) means contiguous rows.  band_rows must be a multiple of
     * the tile height; `rows` counts LOCAL rows and the mapped rows must stay below `height`. */
    int32_t band_rows;
    int32_t band_stride;
    int32_t band_offset;
    /* Tile scheduling order (results are identical in every mode; only the schedule changes -- a frame ends with its
     * longest ray, and that ray starts when the order reaches its tile).
     *   0  the library's choice: centre-out for one-frame launches of every scene but the three whose geometry runs to
     *      the horizon (Grazing Plane, Thin Planes Stack: natural; Pillar Forest: 4), natural for batches of cheap scenes
     *   1  longest-first using the per-tile cost (max iterations) the previous render of the SAME frame shape left in the
     *      library workspace -- rays that ran long last frame start first; the order of mode 0 when no matching previous frame exists
     *   2  centre-out: a static permutation of the frame shape, cached until the shape changes (the registry's cameras
     *      look at their object, so the object's grazing / fractal rays start first)
     *   3  natural (row-major tiles)
     *   4  middle rows first (centre-out with the horizontal distance weighted 1/16): for geometry that runs to the
     *      horizon, whose long rays lie along the horizon line (the library's choice for Pillar Forest) */
    int32_t tile_order_mode;
    /* Scenes whose SDF is a data-dependent loop (Mandelbulb, catalog.py:266-293).  0 = default
     * (library's choice, currently 2); 1 = a whole SDF evaluation per wave turn; 2 = one trip of the
     * SDF loop per wave turn, lanes advance to their next evaluation independently.  Results are
     * identical in either mode; ignored for every other scene. */
    int32_t eval_mode;
    /* Long-ray suspension.  A frame cannot finish before its longest ray, and that ray only starts
     * when the tile order reaches its pixel.  Pass 1 parks every ray still marching after
     * suspend_after[0] trips of its strategy loop and goes on with fresh pixels; pass 2 restarts all
     * parked rays at once (dense wavefronts of long rays) and parks those beyond suspend_after[1]
     * trips for a last sparse pass.  A parked ray continues from its saved strategy state, so results
     * are identical with or without suspension.  Per entry: 0 = library default, < 0 = off. */
    int32_t suspend_after[2];
    int32_t resume_grid;   /* 0 = default (one per compute unit); workgroups of the resume passes */
    /* How parked rays are finished.  0 = default (2 where the scene has a team form), 1 = one wavefront per
     * 64 rays, 2 = wavefront TEAMS for the last pass: three waves carry the same 64 rays and each evaluates
     * one of the three independent transcendental chains of a Mandelbulb trip (rm_march_rays_team), 3 = teams
     * for both resume passes (measured slower than 2; kept for experiments). */
    int32_t resume_mode;
    /* How the passes of a frame with long-ray suspension are launched.  0 = default (library's choice), 1 = one
     * launch per pass (first pass, resume, last resume), 2 = ONE launch: producer workgroups render fresh tiles,
     * take parked rays out of queue 0 as lanes fall idle and park them again at suspend_after[1] trips in queue 1,
     * which `team_grid` workgroups of wavefront teams consume while the producers are still rendering -- a long ray
     * moves to the next form the moment it crosses a threshold instead of waiting for a kernel boundary.  Identical
     * results in every mode.  The remaining fields tune the single launch (0 = library default):
     *   team_grid        workgroups that run as teams (scenes with a team form; the rest are producers)
     *   queue_first      1 = idle producer lanes take parked rays before fresh pixels, 2 = fresh pixels first,
     *                    3 = queue 0 is not used: at suspend_after[0] trips a ray is struck from its tile (the tile is
     *                    flushed without it) and marches on in its lane until suspend_after[1] (default with teams)
     *   team_steal       1 = teams take queue 0 entries while queue 1 is empty, 2 = never
     *   queue_refill_min idle lanes a producer wave needs before it looks at queue 0
     *   queue_retry      turns between two looks of a producer wave whose lanes stay idle
     *   team_retry       evaluations between two looks of a team that still carries rays
     *   age_priority     accepted and ignored by this build (an experiment: waves raising their issue priority with the
     *                    trip count of their oldest ray measured no gain on any scene and cost the cheap scenes 6-8 %;
     *                    rm::kAgePriority in csrc/rm_kernels.h builds it) */
    int32_t pipeline;
    int32_t team_grid;
    int32_t queue_first;
    int32_t team_steal;
    int32_t queue_refill_min;
    int32_t queue_retry;
    int32_t team_retry;
    int32_t age_priority;
    /* Single launch, roles that change while the frame runs (scenes with a team form, queue_first = 0 / 3).  A frame's
     * first milliseconds want every workgroup as a producer -- the last of its long rays is found when the tile order
     * reaches it -- and its tail wants teams.  `late_teams` further team workgroups are put BEHIND the grid that is
     * resident at once: the dispatcher starts one whenever a producer workgroup has left, and up to `late_teams` producer
     * workgroups leave early (stop taking tiles, hand their rays to queue 1) when queue 1 holds `exit_backlog` (default
     * 64) rays more than the teams have taken.  0 = none (measured: no gain at the default sizes, csrc/rm_capi.hip).
     * Results are identical for every setting. */
    int32_t late_teams;
    int32_t exit_backlog;
    /* KEEP BUSY.  The same march loop runs at two speeds on this chip: the full one while most compute units execute
     * vector instructions, and 15-60 % slower when few wavefronts are live (tools/ubench/
     * sparse_share.hip) -- which is exactly the state of a frame whose last long rays are finished by a few team
     * wavefronts.  With keep_busy the workgroups that have run out of work do not leave: they execute `keep_busy`
     * fp32 multiply-adds per lane between two looks at a counter until the teams are through (bounded: 30 ms), and
     * the chains of the long rays run at the full speed (Mandelbulb 1920x1080: 9.4 -> 8.1 ms).  0 = library default
     * (256 in launches that end with teams, off elsewhere: where other workgroups still render tiles the filler only
     * takes their issue slots), < 0 = off, > 0 = explicit burst length.  Results are identical for every setting. */
    int32_t keep_busy;
    /* EARLY HAND-OVER (single launch, scenes with a team form and a per-evaluation cost measure: the Mandelbulb).  A ray that
     * has been struck from its tile and whose last evaluation ran every iteration of the fractal loop -- a near-surface ray,
     * eight producer turns per evaluation -- is handed to the teams at once instead of at suspend_after[1] trips.
     * 0 = library default (on from the strike budget; off for Overstep-Bisect, Skipping-Spheres and Adaptive-Hybrid, which
     * measured 1-2 % slower with it), < 0 = off, > 0 = the earliest trip.  `early_trips`: the iterations an evaluation must
     * have run to count as near-surface (0 = library default: 8 of 8; 1 ... 8 explicit -- 6 with suspend_after = {16, 64} is
     * 1-8 % faster from two of the Mandelbulb's three curated cameras and writes half as much again to HBM).  Results are identical. */
    int32_t early_handover;
    int32_t early_trips;
    int32_t reserved1;
} RmFrameDesc;

/* Frame reduce computed in-kernel (the integer part of RayMarchStats.compute, core/types.py:77-137). */
typedef struct RmStats {
    uint64_t total_rays;
    uint64_t hit_count;
    uint64_t sum_iters; /* sample_count */
    int32_t iter_max;
    int32_t iter_min;
    uint64_t iter_hist[RM_HIST_BINS];
    uint64_t sum_evals; /* SDF evaluations of all rays (see RmOutputs.evals; in batches only when the evals map is requested) */
} RmStats;

/* in: warmup, repeats (<= RM_MAX_TIMED).  out: per-launch kernel milliseconds measured with
 * hipEvents on the launch stream (cf. gpu/runner.py:139-165, main.py:152-153). */
typedef struct RmTiming {
    int32_t warmup;
    int32_t repeats;
    float ms_median, ms_mean, ms_min, ms_max;
    float ms_each[RM_MAX_TIMED];
} RmTiming;

typedef struct RmDeviceInfo {
    char name[128];
    char arch[64];
    int32_t device_id;
    int32_t compute_units;
    int32_t clock_mhz;
    int32_t wavefront_size;
    uint64_t total_mem_bytes;
} RmDeviceInfo;

/* Select the device and create the library's stream + workspace.  Must be called
 * before anything else; idempotent for the same device. */
int rm_init(int device_id);
void rm_shutdown(void);
/* Thread-local text of the last error on this thread (never NULL). */
const char* rm_last_error(void);
int rm_device_info(RmDeviceInfo* out);
int rm_num_scenes(void);
int rm_num_strategies(void);
/* The reference's defaults (the values its registry's no-argument constructors use); never fails. */
void rm_default_strategy_params(RmStrategyParams* out);

/* SDFScene.sdf (scenes/base.py:29-32) over n points; host pointers, xyz is n x 3. */
int rm_sdf_eval(int scene_id, const double* xyz, size_t n, double* out);

/* MarchStrategy.march (strategies/base.py:25-38) over n explicit rays; host pointers.
 * Directions are normalised as Ray.__init__ does (core/ray.py:11-13). */
int rm_march_rays(int scene_id, int strategy_id, const RmMarchConfig* cfg, const double* origins,
                  const double* dirs, size_t n, uint8_t* hit, double* t, int32_t* iters, double* final_sdf);

/* Same contract, evaluated by wavefront TEAMS (scenes whose SDF is a loop of independent
 * transcendental chains -- Mandelbulb: three waves carry the same 64 rays and each evaluates one
 * chain per trip; see rm_kernels.h).  Identical results; RM_E_BAD_SCENE for scenes without a team form. */
int rm_march_rays_team(int scene_id, int strategy_id, const RmMarchConfig* cfg, const double* origins,
                       const double* dirs, size_t n, uint8_t* hit, double* t, int32_t* iters, double* final_sdf);

/* MetricsCollector.benchmark_strategy (metrics/collector.py:19-66): render the frame and
 * copy the maps back.  Host pointers; depth/iters/hit are required (depth is fp32: t if hit
 * else 0, core/types.py:93), the rest optional (NULL):
 *   t_raw      fp64 termination parameter of every ray (MarchResult.t)
 *   final_sdf  fp64 MarchResult.final_sdf (requires desc->march.full)
 *   block_var  (rows/4) x (width/8) int64: 32*sum(x^2) - sum(x)^2 of the iteration counts of each
 *              full 8x4 block (the population variance x 1024 behind warp_divergence_proxy,
 *              core/types.py:125-133); requires row0 % 4 == 0
 * With timing != NULL the kernel is launched warmup + repeats times and timed per launch. */
int rm_render(const RmFrameDesc* desc, float* depth, int32_t* iters, uint8_t* hit, double* t_raw,
              double* final_sdf, int64_t* block_var, RmStats* stats, RmTiming* timing);

/* rm_render with the output pointers in a record, plus one more optional map:
 *   evals   int32 per ray: the number of SDF evaluations its march performed -- what the reference's GLSL
 *           backend counts in g_evals (gpu/shaders/scenes.glsl:10-12).  With march.full = 1 this is the number
 *           of sdf() calls the reference's CPU march() makes for that ray (it also evaluates once more for
 *           final_sdf on a miss and at bisection exits); with full = 0 those final_sdf-only evaluations are skipped.
 * Iterations are not evaluations: Segment, RevAA and Hybrid evaluate more than once per counted iteration. */
typedef struct RmOutputs {
    float* depth;        /* required */
    int32_t* iters;      /* required */
    uint8_t* hit;        /* required */
    double* t_raw;       /* optional, as rm_render */
    double* final_sdf;
    int64_t* block_var;
    int32_t* evals;
} RmOutputs;
int rm_render_outputs(const RmFrameDesc* desc, const RmOutputs* out, RmStats* stats, RmTiming* timing);

/* Same render, outputs left in device memory (caller-owned device pointers, e.g. torch
 * tensors handed to RCCL afterwards).  Asynchronous on `stream` (a hipStream_t, NULL = the
 * library stream).  d_stats: device buffer of rm_stats_device_bytes() bytes or NULL to use
 * the library workspace; decode it with rm_read_stats after synchronising. */
int rm_render_device(const RmFrameDesc* desc, void* d_depth, void* d_iters, void* d_hit, void* d_stats,
                     void* stream);
size_t rm_stats_device_bytes(void);
int rm_read_stats(const void* d_stats, void* stream, RmStats* out);

/* hipEvent-timed loop of rm_render_device on the library stream (device outputs stay resident). */
int rm_bench_device(const RmFrameDesc* desc, void* d_depth, void* d_iters, void* d_hit, RmStats* stats,
                    RmTiming* timing);

/* N frames of one (scene, strategy, frame shape) rendered in ONE launch: the reference's real
 * workload is sweeps of many small frames -- curated viewpoints (viewpoints.py:41-140) and
 * iteration-budget / epsilon levels (sweep.py:96-127) -- and a small frame is bound by the latency
 * of its longest ray, not by throughput.  The tiles of all frames feed one persistent grid (every
 * ray carries its own frame's camera origin and march configuration), so the long-ray tails of the
 * frames overlap and the whole device stays busy.
 *   shape     scene / strategy / width / height / row0 / rows (cam and march of `shape` are ignored)
 *   cams      nframes x 14 doubles (one camera per frame, as RmFrameDesc.cam)
 *   configs   nframes march configs, or NULL to use shape->march for every frame
 *   depth / iters / hit   host arrays of nframes x rows x width elements (frame-major)
 *   stats     nframes RmStats or NULL;  ms_total  device time of the batch launch (hipEvents) or NULL
 *   all configs must share `full`; tile_order_mode must be 0
 * Every frame is bit-identical to what rm_render returns for it alone. */
int rm_render_batch(const RmFrameDesc* shape, int32_t nframes, const double* cams, const RmMarchConfig* configs,
                    float* depth, int32_t* iters, uint8_t* hit, RmStats* stats, float* ms_total);

/* Entries each of the two parked-ray queues may hold (device memory: entries x 64..136 bytes, allocated on
 * first use; default 4 Mi, 0 restores it).  A ray that finds the queue full simply marches on where it is,
 * so the capacity bounds memory, never results. */
int rm_set_queue_capacity(int64_t entries);

/* Per-pass device time of the LAST frame launched (rm_render / rm_render_device / ...): a frame is one
 * render pass plus, with long-ray suspension, up to two resume passes (RmFrameDesc.suspend_after).  For a
 * single-launch frame (RmFrameDesc.pipeline = 2) four spans inside the one kernel are returned instead, read
 * from the device clock by the waves themselves: launch -> the tile counter ran out, -> the last producer wave
 * finished its fresh pixels, -> the last producer wave exited, -> the end of the kernel (the teams' tail).
 * rm_set_pass_timing(1) makes every launch record hipEvents between its passes on the launch stream;
 * rm_get_pass_ms synchronises that stream and returns the count and the milliseconds of each pass
 * (ms must hold RM_MAX_PASSES floats). */
#define RM_MAX_PASSES 4
int rm_set_pass_timing(int enable);
int rm_get_pass_ms(void* stream, int32_t* npasses, float* ms);
/* Single-launch frames: milliseconds after launch of the last push into and the last pop out of queue 1, as decoded
 * by the latest rm_get_pass_ms (0 when there was none).  A tuning aid: pop long after push = the teams fell behind. */
int rm_last_queue_marks(float* last_push_ms, float* last_pop_ms);
/* The same for the rays that ended with >= 500 iterations (the frame's longest chains): ms[0], ms[1] = earliest and
 * latest time after launch at which one of them was handed to the teams, ms[2], ms[3] = shortest and longest time one
 * of them then spent with a team. */
int rm_long_ray_marks(float ms[4]);

/* rm_render_batch with an outputs record: depth / iters / hit as above plus the optional frame-major evals map
 * (RmOutputs.evals; t_raw, final_sdf and block_var must be NULL).  With evals, stats[f].sum_evals is filled. */
int rm_render_batch_outputs(const RmFrameDesc* shape, int32_t nframes, const double* cams, const RmMarchConfig* configs,
                            const RmOutputs* out, RmStats* stats, float* ms_total);

/* Library-owned device frame buffers for callers without their own allocator. */
int rm_alloc_frame(int32_t width, int32_t rows, void** d_depth, void** d_iters, void** d_hit);
int rm_free_frame(void* d_depth, void* d_iters, void* d_hit);
int rm_copy_frame_to_host(int32_t width, int32_t rows, const void* d_depth, const void* d_iters,
                          const void* d_hit, float* depth, int32_t* iters, uint8_t* hit);

/* ---- BASELINE config 5: one frame row-sharded over the GPUs of a node, one process per GPU ------------------
 * The frame shards with no exchange while it is rendered (rays are independent); its only communication is the
 * final gather of the three maps.  These entry points do that gather from C with RCCL over xGMI -- no PyTorch on the
 * path.  RCCL is loaded on first use (dlopen "librccl.so.1"); the library has no link-time dependency on it.
 *   rank 0:      rm_comm_unique_id(id);  ship the 128 bytes to the other ranks (any channel: a file, MPI, a socket)
 *   every rank:  rm_init(local device);  rm_comm_init(id, world_size, rank);
 *   per frame:   rm_render_device(shard desc, ...);  rm_gather_frame(...)  -> the whole frame on every rank
 *   at exit:     rm_comm_destroy();
 * Row plan (the same on every rank; raymarch_algo_compare_amd/sharding.py states it in Python): height a multiple of
 * 4 * world_size -> band-cyclic, rank r renders the 4-row bands r, r + N, r + 2N ... (RmFrameDesc.band_rows = 4,
 * band_stride = N, band_offset = r, rows = height / N), so every rank holds the same mix of sky and object rows and
 * no 8x4 divergence block straddles two ranks; otherwise contiguous 4-aligned blocks of rm_shard_rows() rows (the
 * last rank takes the remainder). */
#define RM_COMM_ID_BYTES 128
int rm_comm_unique_id(uint8_t id[RM_COMM_ID_BYTES]);
int rm_comm_init(const uint8_t id[RM_COMM_ID_BYTES], int32_t world_size, int32_t rank);
int rm_comm_destroy(void);
/* Rows per rank of the contiguous plan: ceil(ceil(height / 4) / world_size) * 4. */
int32_t rm_shard_rows(int32_t height, int32_t world_size);
/* All-gather the local shard (RmFrameDesc.rows x width elements per map, device pointers, as rm_render_device left
 * them) from every rank -- three ncclAllGather on `stream` (NULL = the library stream) -- and put the rows in image
 * order: d_full_* hold height x width elements on every rank.  `shard` is the descriptor the shard was rendered
 * with.  Asynchronous on the stream. */
int rm_gather_frame(const RmFrameDesc* shard, const void* d_depth, const void* d_iters, const void* d_hit,
                    void* d_full_depth, void* d_full_iters, void* d_full_hit, void* stream);
/* The same exchange when only ONE rank needs the image (BASELINE config 5 asks for the gathered frame, not for eight
 * copies of it): every other rank sends its shard to `root` (grouped ncclSend / ncclRecv over the direct xGMI links: the
 * root receives 7/8 of the frame, nobody else receives anything).  Contiguous plan: the shards land straight in the image
 * (no second pass over the frame); band-cyclic plan: in a rank-major buffer that one placement pass turns into the image.
 * d_full_* are read on the root only (NULL elsewhere).  Asynchronous on the stream.  UNVERIFIED ON MORE THAN ONE RANK in
 * this build environment (one GPU): covered with a communicator of one and by construction (tests/test_gpu_gather.py). */
int rm_gather_frame_root(const RmFrameDesc* shard, const void* d_depth, const void* d_iters, const void* d_hit,
                         void* d_full_depth, void* d_full_iters, void* d_full_hit, int32_t root, void* stream);
/* The second half of rm_gather_frame on its own: `d_gathered` holds the shards of all ranks one after the other
 * (rank-major, rows_per_rank x width elements of elem_bytes each, as ncclAllGather leaves them); writes the
 * height x width image.  cyclic != 0: band-cyclic plan with 4-row bands; else contiguous blocks.  Asynchronous. */
int rm_assemble_frame(int32_t world_size, int32_t height, int32_t width, int32_t rows_per_rank, int32_t cyclic,
                      int32_t elem_bytes, const void* d_gathered, void* d_full, void* stream);

/* Which HIP runtime the library's calls resolve to (path of the loaded libamdhip64 and its version numbers).  Needs no
 * device.  Stream handles passed to this library must come from THIS runtime. */
typedef struct RmRuntimeInfo {
    char hip_runtime_path[512];
    int32_t hip_runtime_version;
    int32_t hip_driver_version;
    int32_t hip_runtimes_loaded;     /* copies of libamdhip64 mapped in this process right now (1 is the healthy case) */
    int32_t reserved;
    char other_runtime_path[512];    /* one of the copies that is NOT ours ("" when there is none) */
} RmRuntimeInfo;
int rm_runtime_info(RmRuntimeInfo* out);
/* Streams of the library's own runtime (non-blocking streams on the bound device) for hosts that have no HIP binding:
 * what rm_render_device / rm_gather_frame / rm_read_stats accept as `stream`.  rm_stream_synchronize(NULL) waits for the
 * library stream; it does not hold the library lock while it waits. */
int rm_stream_create(void** stream);
int rm_stream_synchronize(void* stream);
int rm_stream_destroy(void* stream);

/* Test aid: fills both parked-ray queues with the 32-bit word (tag of the next single-launch frame + word_offset) -- the
 * worst stale content a queue can hold -- and marks them as written by an unknown layout, which is what obliges the next
 * single-launch frame to clear them (see State::qkey in csrc/rm_capi.hip).  next_generation (optional) receives that tag. */
int rm_debug_poison_queues(uint32_t word_offset, uint32_t* next_generation);

/* Development trace of single-launch frames (off by default; costs a store per ray and an atomic per ray a team
 * finishes).  After rm_debug_set_trace(1) every single-launch frame records, for each ray a wavefront team finished, eight
 * 32-bit words { output index, iterations, push / pop / end time in 10 ns ticks since launch, SDF evaluations at the pop
 * and at the end, team workgroup | rays alive in the team << 16 } and, per pixel, the device-clock ticks (low 32 bits)
 * at which its ray started and was struck from its tile (0 = never); launch_tick = the same clock at launch.
 * tools/trace_pipeline.py turns this into the time line DESIGN.md section 3 quotes. */
int rm_debug_set_trace(int enable);
int rm_debug_get_trace(uint32_t* records, int64_t max_records, int64_t* nrecords, uint32_t* start_ticks,
                       uint32_t* detach_ticks, int64_t npix, uint32_t* launch_tick);

/* Store-path probe: writes the 9 B/ray outputs with the render kernel's flush code and no
 * marching, to measure the isolated HBM write bandwidth of the path. */
int rm_bench_store_path(int32_t width, int32_t rows, void* d_depth, void* d_iters, void* d_hit,
                        RmTiming* timing);

#ifdef __cplusplus
}
#endif
#endif /* RM_HIP_H */
