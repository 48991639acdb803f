// rm_kernels.h -- gfx950 kernels of the sphere-tracing frame render.
//
// One kernel instantiation per (scene, strategy): camera ray generation
// (camera.py:35-41), the scene SDF (catalog.py) and the strategy state machine
// (strategies/*.py) are all inlined; this replaces the reference's frame loop
// MetricsCollector.benchmark_strategy (metrics/collector.py:40-44).
//
// Execution shape (CDNA4, 64-lane wavefronts; VALU/fp64 bound, not HBM bound):
//  * persistent 256-thread workgroups (4 waves); each WAVE independently pulls
//    64 x TILE_H pixel tiles from a global atomic counter until none are left (every
//    wave reaches the exit test, the grid always drains).  The four waves share one
//    LDS copy of the glibc lookup tables their scene needs (<= 47 KB for Mandelbulb):
//    the table gathers of the exact libm restatements are dependent loads on the
//    critical path of every ray, and LDS latency is several times lower than L2's;
//  * one ray per lane, all per-ray state in VGPRs.  A lane that finishes its ray
//    takes the next unassigned pixel ("lane refill"), so a 512-step straggler ray
//    does not idle the other 63 lanes.  A wave keeps up to kSlots tiles in flight:
//    when the current tile is fully handed out it opens the next one while the
//    earlier tiles drain, so lanes only idle when a very long ray pins every slot.
//    Refills are batched (refill_min idle lanes) so ray set-up runs with a usefully
//    full EXEC mask;
//  * results are staged per tile in LDS (6 B/pixel) and a finished tile is flushed
//    as full-row 256-byte (depth, iterations) / 64-byte (hit) contiguous stores: the
//    only HBM traffic of the path, 9 bytes per ray (4 fp32 depth + 4 int32
//    iterations + 1 uint8 hit);
//  * frame statistics ride along: per-wave register accumulators (hit count,
//    iteration sum/max/min), a per-workgroup LDS iteration histogram flushed once at
//    kernel exit into one of 64 partial stats blocks, and the reference's 8x4-block
//    "warp divergence" variance numerators (core/types.py:125-133) reduced in-wave;
//  * scenes whose SDF is a data-dependent loop (Mandelbulb) run ONE trip of it per
//    turn (INTERLEAVE), so lanes with short evaluations do not wait for long ones;
//  * long rays are parked (strategy record -> queue) and finished by resume_kernel /
//    resume_team_kernel: dense wavefronts of long rays, and for the longest ones
//    TEAMS of three wavefronts that split the trip's three transcendental chains.
#pragma once

#define RM_TABLES_IN_LDS 1   // device math reads the LDS mirror filled by rm_load_tables()

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rm_camera.h"
#include "rm_scenes.h"
#include "rm_strategies.h"

namespace rm {

constexpr int kTileW = 64;          // one tile row == one wavefront-wide store
constexpr int kHistBins = 544;      // iterations <= max_iterations + 9 (Segment 521, RevAA 520) for 512
constexpr int kStatsHead = 24;
constexpr int kStatsWords = kStatsHead + kHistBins;  // u64 words, layout below
constexpr int kQueues = 2;          // suspended-ray queues (ping-pong between resume levels)
// The frame totals and the histogram are accumulated in kStatsParts partial blocks (workgroup % kStatsParts)
// behind the canonical block and summed into it by stats_reduce_kernel at the end of the frame: thousands of
// waves adding to the same five words and the same few histogram bins when they all finish queue up at the
// L2 atomic units (measured: 0.18 ms of a 0.33 ms Cube frame).
constexpr int kStatsParts = 64;
constexpr int kStatsBlocks = 1 + kStatsParts;

// device-side stats block (u64 words):
//  [0] tile counter   [1] hit_count   [2] sum_iters   [3] iter_max   [4] 0x7fffffff - iter_min
//  [5] rays written   [6..7] entries pushed to suspended-ray queue 0 / 1   [8..9] entries handed out
//  of queue 0 / 1   [10] SDF evaluations of the finished rays   [11] fused-reduce tickets
//  [12], [13] team passes (resume_team_kernel, level 0 / 1): team workgroups that have left (KernelArgs.keep_busy)
//  [13..15], [17..23] single-launch pipeline: time marks (rm_pipeline.h; written only when KernelArgs.marks)
//  [16] pipeline protocol error (0 = none)
//  [kStatsHead .. kStatsHead + kHistBins) = [24 .. 568) histogram of iterations
// Block 0 holds the counters ([0], [6..9]) and, after stats_reduce_kernel, the totals; blocks 1..kStatsParts
// hold the partial sums of words [1..5], [10] and of the histogram.
// One frame of a launch: its camera and march configuration.  A launch renders `nframes` frames of the
// same shape (1 for rm_render; rm_render_batch renders a whole viewpoint / budget sweep in one launch);
// tile ids run frame-major, the output arrays are frame-major too.
struct FrameParams {
    CameraParams cam;
    MarchCfg cfg;
};

// One-frame launches: the march configuration is wave-uniform (kernel arguments).  Left to itself the compiler
// re-reads the fields a strategy consults from the kernarg segment INSIDE the march loop -- scalar registers are
// short in these kernels -- i.e. s_load + s_waitcnt on every turn: measured +6 % (Cube), +10 % (Sphere), +4.7 %
// (Mandelbulb 7680x4320) against per-lane copies.  Passing a field through an empty asm with a VGPR constraint
// makes it an ordinary per-lane value that stays in vector registers for the life of the wave; fields the
// strategy never reads disappear together with their asm.
#if defined(__HIP_DEVICE_COMPILE__)
#define RM_PIN_LANE(x) asm("" : "+v"(x))
#else
#define RM_PIN_LANE(x) (void)(x)      // host pass of the translation unit
#endif
__device__ __forceinline__ void pin_lane(double& x) { RM_PIN_LANE(x); }
__device__ __forceinline__ void pin_lane(int32_t& x) { RM_PIN_LANE(x); }
template <class T> __device__ __forceinline__ void pin_lane(T*& x) { RM_PIN_LANE(x); }
__device__ __forceinline__ void pin_cfg(MarchCfg& c)
{
    pin_lane(c.hit_threshold); pin_lane(c.max_distance); pin_lane(c.lipschitz); pin_lane(c.max_iterations);
    StratParams& p = c.prm;
    pin_lane(p.omega); pin_lane(p.ar_omega_min); pin_lane(p.ar_omega_max); pin_lane(p.ar_smoothing); pin_lane(p.ar_growth_rate);
    pin_lane(p.ar_decay_rate); pin_lane(p.beta); pin_lane(p.overstep_min_step); pin_lane(p.hybrid_stuck_step_ratio);
    pin_lane(p.hybrid_min_step); pin_lane(p.margin); pin_lane(p.ar_omega_init); pin_lane(p.overstep_bisection_steps);
    pin_lane(p.hybrid_stuck_threshold); pin_lane(p.segment_bisection_steps); pin_lane(p.revaa_bisection_steps);
    pin_lane(p.step_scale); pin_lane(p.dense_min_step);
}

// Issue priority by ray age (RmFrameDesc.age_priority): built only when this is true.  Measured on the MI355X: no gain on
// any scene (Sphere 0.47 ms, Cube 0.23, Mandelbulb 10.0-10.6 ms with or without), and its per-turn bookkeeping cost the
// cheap scenes' render kernel 15 % more scalar instructions and the scalar registers that kept its output pointers
// resident (Cube +6 %, Pillar Forest +8 % kernel time) -- so the field is accepted and ignored.
constexpr bool kAgePriority = false;

struct KernelArgs {
    FrameParams single;          // the frame of a one-frame launch (travels in the kernel arguments)
    const FrameParams* frames;   // device array [nframes] for batches, nullptr for one frame
    int32_t nframes;
    int32_t tiles_per_frame;     // tiles_x * tiles_y
    int32_t full;                // all frames of a launch share cfg.full
    int32_t width, height, row0, rows;
    int32_t tiles_x, tiles_y;
    int32_t tile_h;              // rows of a tile: 4 (64x4 pixels), or 1 in the single-launch pipeline of scenes with teams
    int32_t refill_min;
    int32_t interleave;          // scenes with a resumable SDF: one trip of the SDF loop per turn (see render_kernel)
    int32_t hist_bins;
    int32_t band_rows, band_stride, band_offset;   // band-cyclic row map (band_rows == 0: identity)        // clamp for the LDS histogram (<= kHistBins)
    float* depth;             // rows*width, t if hit else 0 (types.py:93), fp32
    int32_t* iters;           // rows*width
    uint8_t* hit;             // rows*width
    double* t_raw;            // optional: raw fp64 t of every ray (parity tests)
    int32_t* evals;           // optional: SDF evaluations the march of every ray performed (needs cfg.full for the
                              // reference's count: its march() also evaluates for final_sdf)
    double* final_sdf;        // optional: MarchResult.final_sdf (needs cfg.full)
    int32_t raw_outputs;      // any of t_raw / evals / final_sdf is set (the kernels test this before touching the pointers)
    int32_t fused_reduce;     // render_kernel: the last workgroup to finish folds the partial stats blocks (no reduce launch)
    long long* block_var;     // optional: (rows/4) x (width/8) variance numerators 32*sum(x^2)-sum(x)^2
    unsigned long long* stats;
    // Long-ray suspension (see resume_kernel): a ray still marching when its loop index reaches
    // `suspend_after` is parked in queue `suspend_queue` (state = the strategy record) and its lane
    // takes a fresh pixel; 0 = never.  Queues hold `queue_cap` entries of `queue_stride` bytes.
    int32_t suspend_after, suspend_queue, queue_cap, queue_stride;
    unsigned char* queue[kQueues];
    const int32_t* tile_order;  // optional: permutation of the tile ids (longest-first schedule)
    int32_t* tile_cost;         // optional: per tile, the largest iteration count of its rays
    // Single-launch pipeline (rm_pipeline.h): workgroups [0, team_wgs) are wavefront teams, the rest producers
    // (`producer_waves` waves in all); rays resumed from queue 0 are parked again at suspend_after2 trips in
    // queue 1, which the teams consume.  `generation` tags the queue entries of this launch.
    unsigned long long* ctl;    // the launch's hot counters, one per 128-byte line (rm_pipeline.h), zeroed before the launch
    int32_t team_wgs, producer_waves;
    uint32_t generation;
    int32_t suspend_after2;
    int32_t q0_first;           // idle lanes take parked rays before fresh pixels (else only once the wave has no fresh pixels left)
    int32_t q0_detach;          // queue 0 unused: a ray at suspend_after trips is struck from its tile and marches on in its lane
    int32_t q0_refill_min;      // idle lanes required before a wave looks at queue 0
    int32_t q0_retry;           // turns between two looks at queue 0 while lanes idle
    int32_t team_retry;         // evaluations between two looks of a team with idle lanes
    int32_t team_steal;         // teams take queue 0 entries while queue 1 is empty
    int32_t max_spins;          // polls after which a producer wave with nothing to do stops waiting for the others
    int32_t team_prio;          // s_setprio level of the team waves (0..3)
    int32_t age_prio;           // > 0: a producer wave's issue priority = (trips of its oldest ray) / age_prio, capped at 2
    int32_t marks;              // single launch: waves leave device-clock marks in stats block 0 (rm_set_pass_timing); off in production
    int32_t early_handover;     // single launch: > 0 = a struck ray at this trip or later whose evaluation ran >= early_trips iterations
    int32_t early_trips;        // (Scene::eval_trips) goes to the teams at once
    int32_t keep_busy;          // > 0: workgroups without work stay and execute this many fp32 fmas per lane between two looks at
                                // the count of finished team workgroups (keep_busy_until below; RmFrameDesc.keep_busy)
    // Roles that change during the launch (rm_pipeline.h "late teams"): workgroups [late_team_first, gridDim.x) are teams
    // the dispatcher starts when producer workgroups have left; producer workgroups [team_wgs, team_wgs + early_exit_wgs)
    // leave (stop taking tiles, hand their detached rays to queue 1) once queue 1 holds exit_backlog rays per pending
    // conversion more than the teams have taken.
    int32_t late_team_first, early_exit_wgs, exit_backlog;
    // Development trace of a single-launch frame (rm_debug_set_trace; nullptr in production): trace[0] counts the records,
    // 8-word records from trace + 8 (one per ray a team finished: output index, iterations, push / pop / end ticks since
    // launch, evaluations at the pop and at the end, team workgroup | live lanes << 16); per pixel the low 32 bits of
    // the device clock when its ray started and when it was struck from its tile.
    uint32_t* trace;
    uint32_t trace_cap;
    uint32_t* trace_start;
    uint32_t* trace_detach;
};

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int rank_in_mask(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// t_raw / final_sdf are parity-test outputs (fp64, every ray): written straight to global
// memory when requested, never staged (they are not part of the 9 B/ray product path).
// Callers test a per-lane copy of KernelArgs.raw_outputs first: a finished ray of the product path must not pull the
// block of output pointers back into scalar registers (the compiler re-reads it from the kernel arguments at every
// site that looks at one of these pointers).
__device__ __forceinline__ void store_raw(const KernelArgs& a, uint32_t gi, const Result& r, int nev)
{
    if (a.t_raw) a.t_raw[gi] = r.t;
#ifdef RM_DEV_STAMP      // development (tools/ray_times.py): `evals` = finish time, `final_sdf` = start time, 100 MHz device clock
    if (a.evals) a.evals[gi] = (int)(__builtin_amdgcn_s_memrealtime() & 0x3fffffffull);
#else
    if (a.final_sdf) a.final_sdf[gi] = r.final_sdf;
    if (a.evals) a.evals[gi] = nev;
#endif
}

constexpr int kWavesPerWG = 4;         // 256-thread workgroups: four waves share one LDS copy of the libm tables

// KEEP BUSY (RmFrameDesc.keep_busy, include/rm_hip.h).  This chip runs the same instruction stream at two speeds: the full
// one while most compute units execute vector instructions, a 15-60 % slower one when few wavefronts are live -- the state
// of a launch whose last long rays are being finished by a few team wavefronts (tools/ubench/sparse_share.hip: Sphere's
// march loop 0.34 against 0.52-0.60 us per iteration, the Mandelbulb's 10.0 against 11.5-12.4; what matters is vector work on
// many lanes -- fp64, fp32 or integer alike -- not waves that merely stay resident: s_sleep in the same place gains nothing).
// A wave that has nothing left to do therefore stays and executes `burst` fp32 fmas on all 64 lanes between two looks at
// *done, until `target` team workgroups have reported or 30 ms of the 100 MHz clock have passed (no wait is unbounded).
__device__ __forceinline__ void keep_busy_until(const unsigned long long* done, unsigned long long target, int burst)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float f0 = (float)lane_id(), f1 = f0 + 1.0f, f2 = f0 + 2.0f, f3 = f0 + 3.0f;
    for (;;) {
        unsigned long long n = 0;
        if (lane_id() == 0) n = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)n);
        if (n >= target || __builtin_amdgcn_s_memrealtime() - t0 > 3000000ull) break;
        for (int i = 0; i < burst; ++i) {
            f0 = __builtin_fmaf(f0, 1.0000001f, 1e-9f); f1 = __builtin_fmaf(f1, 1.0000001f, 1e-9f);
            f2 = __builtin_fmaf(f2, 1.0000001f, 1e-9f); f3 = __builtin_fmaf(f3, 1.0000001f, 1e-9f);
        }
    }
    asm volatile("" :: "v"(f0), "v"(f1), "v"(f2), "v"(f3));      // the values are never used; the loop must stay
}

// Copy the gathered libm tables this scene needs from constant memory into LDS (all threads of the
// workgroup cooperate; ends with a barrier).  With RM_TABLES_IN_LDS the math headers read rm_s_*.
template <class T>
__device__ __forceinline__ void copy_table(T* dst, const T* src, int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

template <class Scene>
__device__ __forceinline__ void rm_load_tables()
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(RM_TABLES_IN_LDS)
    constexpr unsigned tb = SceneTables<Scene>::value;
    // mirrors (row strides: rm_tables.h)
    if constexpr (tb & TB_POW) {
        for (int i = threadIdx.x; i < 128 * kPowLogStride; i += blockDim.x) rm_s_pow_log_tab[i] = rm_g_pow_log_tab[i];
        for (int i = threadIdx.x; i < 128 * kExpStride; i += blockDim.x) rm_s_exp_tab[i] = rm_g_exp_tab[i];
    }
    if constexpr (tb & TB_SINCOS)
        for (int i = threadIdx.x; i < 110 * kSinCosStride; i += blockDim.x) {
            const int row = i / kSinCosStride, f = i - row * kSinCosStride;
            rm_s_sincostab[i] = f < 4 ? rm_g_sincostab[4 * row + f] : 0.0;
        }
    if constexpr (tb & TB_ACOS) {
        copy_table(rm_s_asncs, rm_g_asncs, 2808);
        copy_table(rm_s_inroot, rm_g_inroot, 128);
    }
    if constexpr (tb & TB_ATAN) copy_table(rm_s_cij, rm_g_cij, 1687);
    if constexpr (tb & TB_LOG)
        for (int i = threadIdx.x; i < 128 * kLogStride; i += blockDim.x) {
            const int row = i / kLogStride, f = i - row * kLogStride;
            rm_s_log_tab[i] = f < 2 ? rm_g_log_tab[2 * row + f] : 0.0;
        }
    __syncthreads();
#endif
}

// Orders this wave's earlier LDS accesses before its later ones (cross-lane hand-off through a
// wave-private LDS region; DS operations of one wave execute in order, this stops the compiler
// from reordering them and is free at run time).
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr int kSlots = 3;   // tiles a wave may have in flight: one being handed out + two draining
constexpr uint32_t kSuspended = 0xffffffffu;   // staging mark of a pixel whose ray was parked (resume_kernel writes it)

// A parked ray: where its result goes (element index in the frame-major output arrays) and the
// strategy record, which is the whole march state between two SDF evaluations.
template <class Strat>
struct QEntry {
    uint32_t gi;
    uint32_t nev;   // SDF evaluations performed so far (KernelArgs.evals)
    uint32_t ready; // single-launch pipeline: == KernelArgs.generation once the entry is published (rm_pipeline.h)
    uint32_t pad;
    Strat s;
};

// Wave-uniform call: lanes with `want` append their ray to queue q (one atomic per wave).  Returns
// per lane whether the ray was parked; a full queue leaves the ray where it is.
template <class Strat>
__device__ __forceinline__ bool push_suspended(const KernelArgs& a, int q, bool want, uint32_t gi, const Strat& s, int nev)
{
    const unsigned long long m = __ballot(want);
    if (m == 0) return false;
    // 64-bit throughout: rays refused by a full queue bump the counter again on later turns, and a counter
    // truncated to 32 bits could wrap back below queue_cap and overwrite live entries
    unsigned long long base = 0;
    if (lane_id() == 0) base = atomicAdd(&a.stats[6 + q], (unsigned long long)__popcll(m));
    base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    const unsigned long long idx = base + (unsigned long long)rank_in_mask(m);
    const bool ok = want && idx < (unsigned long long)a.queue_cap;
    if (ok) {
        QEntry<Strat>* e = (QEntry<Strat>*)a.queue[q] + idx;
        e->gi = gi;
        e->nev = (uint32_t)nev;
        e->s = s;
    }
    return ok;
}

// the partial stats block this workgroup adds to
__device__ __forceinline__ unsigned long long* stats_part(unsigned long long* stats)
{
    return stats + (size_t)(1 + blockIdx.x % kStatsParts) * kStatsWords;
}

// One-pass frames (no parked rays): the frame reduce without a second launch.  Every workgroup takes a ticket once its
// own partial sums have been performed; the one that draws the last ticket folds the kStatsParts partial blocks into
// block 0 -- what stats_reduce_kernel does for the multi-pass frames -- and leaves the buffer as a zeroed one looks to
// the next frame: partial blocks cleared, tile counter and ticket back at 0 (block 0's totals are overwritten, never
// added to).  The host then skips the reduce launch, and for its own statistics buffer the memset of the next frame:
// two launches and their gaps, 11-13 us of a 0.19 ms Cube frame.
constexpr int kWDone = 11;        // stats block 0: tickets of the workgroups that have finished (fused reduce)
__device__ __forceinline__ void frame_reduce_by_last_workgroup(unsigned long long* stats)
{
    __shared__ unsigned int s_last;
    // This thread's partial sums are device-scope atomics: performed at L2 once they have drained.  No release fence --
    // on gfx950 it would write the L2's dirty lines back, i.e. the frame's own output tiles, once per workgroup (measured:
    // Cube 0.19 -> 0.24 ms with __threadfence here).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = atomicAdd(&stats[kWDone], 1ull);
        s_last = (t == (unsigned long long)gridDim.x - 1ull) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;          // workgroup-uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // The histogram is empty above the frame's largest iteration count: fold iter_max first (64 loads), then only the
    // words up to that bin -- 64 x 64 loads for a Cube frame (36 iterations) instead of 64 x 568.
    __shared__ unsigned int s_top;
    if (threadIdx.x < 64) {
        unsigned long long m = 0;
        for (int p = 1 + (int)threadIdx.x; p <= kStatsParts; p += 64) {
            const unsigned long long v = stats[(size_t)p * kStatsWords + 3];
            m = v > m ? v : m;
        }
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(m, off); m = o > m ? o : m; }
        if (threadIdx.x == 0) s_top = (unsigned int)(m < (unsigned long long)(kHistBins - 1) ? m : (unsigned long long)(kHistBins - 1));
    }
    __syncthreads();
    const int nwords = kStatsHead + (int)s_top + 1;
    for (int w = threadIdx.x; w < kStatsWords; w += blockDim.x) {
        if (w == 0 || (w >= 6 && w < 10) || (w > 10 && w < kStatsHead)) continue;     // counters live in block 0 only
        if (w >= nwords) { stats[w] = 0ull; continue; }                                 // empty bins: block 0 may hold an older frame's
        const bool is_max = (w == 3 || w == 4);
        unsigned long long acc = 0;
        // plain loads (they pipeline; atomic loads came back one at a time: +30 us): the partial blocks were only ever
        // touched by atomics, which do not allocate in this CU's vector cache, and the acquire fence above emptied it
#pragma unroll 16
        for (int p = 1; p <= kStatsParts; ++p) {
            unsigned long long* const q = stats + (size_t)p * kStatsWords + w;
            const unsigned long long v = *q;
            acc = is_max ? (v > acc ? v : acc) : acc + v;
            *q = 0ull;
        }
        stats[w] = acc;
    }
    if (threadIdx.x == 0) { stats[0] = 0ull; stats[kWDone] = 0ull; }
}

// Per-wave frame totals kept in registers; one atomic each at kernel exit.
struct WaveAcc {
    unsigned long long hits = 0, iters = 0, rays = 0, evals = 0;
    int mx = 0, mn = 0x7fffffff;
    __device__ __forceinline__ void add(int it, int h)
    {
        hits += (unsigned)h; iters += (unsigned)it; rays += 1;
        mx = max(mx, it); mn = min(mn, it);
    }
    __device__ __forceinline__ void flush(unsigned long long* stats)
    {
        for (int off = 32; off > 0; off >>= 1) {
            hits += __shfl_xor(hits, off);
            iters += __shfl_xor(iters, off);
            rays += __shfl_xor(rays, off);
            evals += __shfl_xor(evals, off);
            mx = max(mx, __shfl_xor(mx, off));
            mn = min(mn, __shfl_xor(mn, off));
        }
        if (lane_id() == 0 && rays) {
            atomicAdd(&stats[1], hits);
            atomicAdd(&stats[2], iters);
            atomicMax(&stats[3], (unsigned long long)mx);
            atomicMax(&stats[4], (unsigned long long)(0x7fffffff - mn));   // zero-initialised => store the complement
            atomicAdd(&stats[5], rays);
        }
        if (lane_id() == 0 && evals) atomicAdd(&stats[10], evals);
    }
};

// Tile geometry (wave-uniform): frame, origin inside the frame slice, valid extent, image row of its
// first row, element offset of the frame in the frame-major output arrays.
struct TileGeom {
    int frame, x0, y0, tw, th, gy0;
    size_t out0;
};
template <int TILE_H>
__device__ __forceinline__ TileGeom tile_geom(const KernelArgs& a, int tile)
{
    TileGeom g;
    g.frame = tile / a.tiles_per_frame;
    tile -= g.frame * a.tiles_per_frame;
    g.out0 = (size_t)g.frame * (size_t)a.rows * (size_t)a.width;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    g.x0 = tx * kTileW;
    g.y0 = ty * TILE_H;                                   // relative to row0
    g.tw = min(kTileW, a.width - g.x0);                   // pixels of this tile that exist in the slice
    g.th = min(TILE_H, a.rows - g.y0);
    g.gy0 = a.band_rows > 0                               // tiles never straddle a band
        ? a.row0 + ((g.y0 / a.band_rows) * a.band_stride + a.band_offset) * a.band_rows + (g.y0 % a.band_rows)
        : a.row0 + g.y0;
    return g;
}

// Per-lane evaluation state of the trip-interleaved mode (empty for scenes without a resumable SDF).
struct NoEval {};
template <class Scene, bool ITER> struct EvalOf { using type = NoEval; };
template <class Scene> struct EvalOf<Scene, true> { using type = typename Scene::Eval; };

// INTERLEAVE (scenes with a resumable SDF only): a turn of the wave loop runs ONE trip of the SDF's
// inner loop for every lane instead of a whole evaluation; a lane whose value is ready consumes it
// (strategy step) and begins its next evaluation while its neighbours are still iterating.  The
// lanes of a wavefront rarely need the same trip count (Mandelbulb: 1..8, mean 1.8), so a whole
// evaluation per turn leaves more than half of the lanes idle inside the SDF loop.
//
// BATCH: the launch renders a frame table (rm_render_batch) and every ray carries the march configuration of
// its own frame in vector registers (only the fields its strategy reads survive).  A one-frame launch reads
// the configuration -- thresholds and the strategy's constructor arguments -- from the kernel arguments, i.e.
// from scalar registers.
template <class Scene, class Strat, int TILE_H, bool INTERLEAVE, bool BATCH>
// (two workgroups per CU at least: without the bound the register allocator may take more than 256 registers for the
// Mandelbulb instantiations -- one wave per SIMD, the 7680x4320 frame 52 -> 69 ms when two more values became live)
__global__ __launch_bounds__(64 * kWavesPerWG, 2) void render_kernel(const KernelArgs a)
{
    static_assert(!INTERLEAVE || SceneIterative<Scene>::value, "INTERLEAVE needs Scene::Eval");
    constexpr int TILE_PIX = kTileW * TILE_H;
    // per wave and tile slot: fp32 depth + (iterations | hit << 31) of every pixel of the tile
    __shared__ float s_depth_all[kWavesPerWG][kSlots][TILE_PIX];
    __shared__ uint32_t s_ih_all[kWavesPerWG][kSlots][TILE_PIX];
    __shared__ unsigned int s_hist[kHistBins];

    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    float (*const s_depth)[TILE_PIX] = s_depth_all[wave];
    uint32_t (*const s_ih)[TILE_PIX] = s_ih_all[wave];
    const int ntiles = a.tiles_per_frame * a.nframes;

    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) s_hist[b] = 0u;
    rm_load_tables<Scene>();
    __syncthreads();

    WaveAcc acc;

    // wave-uniform scheduler state: the tiles in flight.  Whether a slot still has rays marching is asked of the lanes when
    // the scheduler runs (one ballot per slot) -- not counted on every finish: the scheduler only has something to do when
    // enough lanes are idle for a refill, and a finished tile can wait for that moment to be flushed (round 3: Sphere -2 %;
    // generating rays 64 at a time into an LDS pool so that idle lanes only load their direction was measured as well:
    // -1.5 % more on Sphere, nothing on Cube -- a refill already serves 40-64 lanes there -- and not kept).
    int slot_tile[kSlots];                        // tile id (-1 = free)
#pragma unroll
    for (int k = 0; k < kSlots; ++k) slot_tile[k] = -1;
    int cur = 0;                                  // slot currently handing out pixels
    int pool_next = TILE_PIX;                     // next unassigned pixel id of slot `cur`
    bool more_tiles = true;                       // the global tile queue may still hold work
    bool first_tile = true;                       // wave-uniform: the next tile is this wave's static one
    TileGeom cg = { 0, 0, 0, 0, 0, 0, 0 };        // geometry of slot `cur`

    // per-lane state: the ray this lane carries
    bool active = false;
    int my_slot = 0, my_pix = 0;                  // where its result goes: slot, tile-linear index y*64+x
    uint32_t my_gi = 0;                           // ... and its element index in the output arrays
    int nev = 0;                                  // SDF evaluations this ray's march has performed
    vec3 origin = v3(0.0, 0.0, 0.0), dir = v3(0.0, 0.0, 0.0);
    MarchCfg lane_cfg = a.single.cfg;             // BATCH: of the frame this lane's ray belongs to
    int raw_out = a.raw_outputs;                  // parity outputs requested (kept per lane, see store_raw)
    pin_lane(raw_out);
    if constexpr (!BATCH) pin_cfg(lane_cfg);      // one frame: per-lane copies of the kernel arguments (see pin_cfg)
    const MarchCfg& cfg = lane_cfg;
    Strat s;
    typename EvalOf<Scene, INTERLEAVE>::type ev;  // INTERLEAVE: the SDF evaluation in flight
    bool ready = false;                           // INTERLEAVE: its value can be consumed
    // wave-uniform: the scheduler (flush + refill) has something to look at -- a ray finished, a
    // refill handed out pixels or opened a tile.  Otherwise a turn goes straight to the SDF.
    bool dirty = true;
    int prio_level = 0;                           // current s_setprio level of this wave (age_prio)
    int since_sched = 0;                          // turns since the scheduler last ran

    for (;;) {
        if (dirty) {
        dirty = false;
        // ---- 1. flush every tile whose rays have all finished (and whose pool is handed out) -----
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            const bool pool_done = (k != cur) || pool_next >= TILE_PIX;
            if (slot_tile[k] >= 0 && pool_done && __ballot(active && my_slot == k) == 0) {   // wave-uniform
                wave_lds_fence();     // staged results of all 64 lanes are visible
                const TileGeom g = tile_geom<TILE_H>(a, slot_tile[k]);
                const int gx = g.x0 + lane;
                const bool col_ok = lane < g.tw;
                long long bs = 0, bq = 0;   // per-column sums for the 8x4 block statistic
                const bool want_bv = a.block_var != nullptr;      // (kernel-uniform: no sums, no shuffles when nobody asked)
                // the three output bases are read from the kernel arguments once per tile, not once per row
                float* o_depth = a.depth; int32_t* o_iters = a.iters; uint8_t* o_hit = a.hit;
                pin_lane(o_depth); pin_lane(o_iters); pin_lane(o_hit);
#pragma unroll
                for (int r = 0; r < TILE_H; ++r) {
                    if (r < g.th && col_ok) {
                        // lane == column: one 64-pixel row per store instruction
                        const int li = r * kTileW + lane;
                        const size_t gi = g.out0 + (size_t)(g.y0 + r) * (size_t)a.width + (size_t)gx;
                        const uint32_t ih = s_ih[k][li];
                        if (ih != kSuspended) {     // a parked ray's pixel is written by resume_kernel
                            const int it = (int)(ih & 0x7fffffffu);
                            const int h = (int)(ih >> 31);
                            o_depth[gi] = s_depth[k][li];
                            o_iters[gi] = it;
                            o_hit[gi] = (uint8_t)h;
                            acc.add(it, h);
                            atomicAdd(&s_hist[min(it, a.hist_bins - 1)], 1u);
                            if (want_bv) { bs += it; bq += (long long)it * it; }
                        }
                    }
                    if ((r & 3) == 3 && want_bv) {
                        // reduce the 8 columns of each 8x4 block (lanes 8k..8k+7)
                        long long S = bs, Q = bq;
                        S += __shfl_xor(S, 1); Q += __shfl_xor(Q, 1);
                        S += __shfl_xor(S, 2); Q += __shfl_xor(Q, 2);
                        S += __shfl_xor(S, 4); Q += __shfl_xor(Q, 4);
                        const int brow = g.y0 + (r - 3);        // first row of this block, relative to row0
                        if (a.block_var && (lane & 7) == 0 && gx + 8 <= a.width && brow + 4 <= a.rows) {
                            // full blocks only (types.py:128-131)
                            a.block_var[(size_t)g.frame * (size_t)(a.rows >> 2) * (size_t)(a.width >> 3) +
                                        (size_t)(brow >> 2) * (size_t)(a.width >> 3) + (size_t)(gx >> 3)] = 32 * Q - S * S;
                        }
                        bs = 0; bq = 0;
                    }
                }
                if (a.tile_cost) {
                    int tmax = 0;
#pragma unroll
                    for (int r = 0; r < TILE_H; ++r)
                        if (r < g.th && col_ok) {
                            const uint32_t ih = s_ih[k][r * kTileW + lane];
                            tmax = max(tmax, ih == kSuspended ? a.suspend_after : (int)(ih & 0x7fffffffu));
                        }
                    for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));
                    if (lane == 0) a.tile_cost[slot_tile[k]] = tmax;
                }
                wave_lds_fence();     // the slot may be reused
                slot_tile[k] = -1;
            }
        }

        // ---- 2. lane refill: idle lanes take the next unassigned pixels (batched by refill_min) ----
        const unsigned long long idle = __ballot(!active);
        const int nidle = __popcll(idle);
        if (nidle >= a.refill_min || nidle == 64) {
            if (pool_next >= TILE_PIX && more_tiles) {
                // current tile fully handed out: open the next tile in a free slot, if any
                int f = -1;
#pragma unroll
                for (int k = kSlots - 1; k >= 0; --k) f = (slot_tile[k] < 0) ? k : f;
                if (f >= 0) {
                    // the first tile of a wave is its own index (no atomic: thousands of waves asking one
                    // counter at once queue up behind each other); later tiles come from the shared counter,
                    // which therefore starts behind the statically assigned ones
                    int tile = 0;
                    if (first_tile) {
                        tile = (int)(blockIdx.x * kWavesPerWG) + wave;
                        first_tile = false;
                    } else {
                        if (lane == 0) tile = (int)atomicAdd(&a.stats[0], 1ull) + (int)(gridDim.x * kWavesPerWG);
                        tile = __builtin_amdgcn_readfirstlane(tile);
                    }
                    if (tile < ntiles) {
                        if (a.tile_order) tile = __builtin_amdgcn_readfirstlane(a.tile_order[tile]);
                        cur = f;
                        pool_next = 0;
                        dirty = true;
                        cg = tile_geom<TILE_H>(a, tile);
#pragma unroll
                        for (int k = 0; k < kSlots; ++k) slot_tile[k] = (k == f) ? tile : slot_tile[k];
                    } else {
                        more_tiles = false;   // every wave gets here: the grid always drains
                    }
                }
            }
            if (pool_next < TILE_PIX) {
                if (!active) {
                    // block-major pixel order: 32 consecutive ids form one 8x4 block, so a fresh
                    // wave starts on a compact 16x4 patch (coherent rays, similar trip counts)
                    const int id = pool_next + rank_in_mask(idle);
                    const int blk = id >> 5, within = id & 31;
                    const int px = (blk & 7) * 8 + (within & 7);
                    const int py = (blk >> 3) * 4 + (within >> 3);
                    if (id < TILE_PIX && px < cg.tw && py < cg.th) {
                        my_slot = cur;
                        my_pix = py * kTileW + px;
                        my_gi = (uint32_t)(cg.out0 + (size_t)(cg.y0 + py) * (size_t)a.width + (size_t)(cg.x0 + px));
                        if constexpr (BATCH) {                            // cg.frame is wave-uniform: scalar loads
                            const FrameParams& fp = a.frames[cg.frame];
                            camera_ray(fp.cam, a.width, a.height, cg.x0 + px, cg.gy0 + py, origin, dir);
                            lane_cfg = fp.cfg;
                            lane_cfg.full = a.full;
                        } else {
                            camera_ray(a.single.cam, a.width, a.height, cg.x0 + px, cg.gy0 + py, origin, dir);
                        }
                        nev = 0;
#ifdef RM_DEV_STAMP
                        if (a.final_sdf) a.final_sdf[my_gi] = (double)(__builtin_amdgcn_s_memrealtime() & 0x3fffffffull);
#endif
                        if (s.start(cfg)) {
                            s_depth[cur][my_pix] = s.res.hit ? (float)s.res.t : 0.0f;
                            s_ih[cur][my_pix] = (uint32_t)s.res.iters | ((uint32_t)s.res.hit << 31);
                            if (raw_out) store_raw(a, my_gi, s.res, nev);
                            acc.evals += (unsigned)nev;
                        } else {
                            active = true;
                            if constexpr (INTERLEAVE) {
                                ready = Scene::begin(ev, origin + dir * s.te);   // ray.py:15-17
                            }
                        }
                    }
                }
                pool_next += nidle;
                dirty = true;
            }
        }
        // issue priority by age: a frame ends with its longest ray, and that ray's wave shares its SIMD with waves full
        // of short rays for most of the frame.  A wave raises its priority with the trip count of its oldest ray
        // (throughput-neutral: the other waves get the slots a dependent chain leaves free anyway).
        if (kAgePriority && a.age_prio > 0) {
            int age = active ? s.i : 0;
            for (int off = 32; off > 0; off >>= 1) age = max(age, __shfl_xor(age, off));
            const int lvl = age / a.age_prio;
            if (lvl != prio_level) {
                prio_level = lvl;
                if (lvl <= 0) __builtin_amdgcn_s_setprio(0);
                else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
                else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
            }
        }
        }   // dirty

        // ---- 3. exit / idle turn ---------------------------------------------------------------------
        if (!__any(active)) {
            bool in_flight = false;
#pragma unroll
            for (int k = 0; k < kSlots; ++k) in_flight = in_flight || (slot_tile[k] >= 0);
            if (!more_tiles && !in_flight) break;     // wave-uniform; nothing left anywhere
            dirty = true;
            continue;                                 // a flush or a refill makes progress next turn
        }

        // ---- 4. one SDF evaluation (INTERLEAVE: one trip of it) for every live ray -----------------
        bool fin = false, park = false;
        bool consume = active;
        if constexpr (INTERLEAVE) consume = active && ready;
        if (!INTERLEAVE || __any(consume)) {
        if (consume) {
            double d;
            if constexpr (INTERLEAVE) d = Scene::value(ev);
            else d = Scene::sdf(origin + dir * s.te);   // ray.py:15-17
            ++nev;
            if (s.step(d, cfg)) {
                active = false;
                fin = true;
                s_depth[my_slot][my_pix] = s.res.hit ? (float)s.res.t : 0.0f;   // types.py:93
                s_ih[my_slot][my_pix] = (uint32_t)s.res.iters | ((uint32_t)s.res.hit << 31);
                if (raw_out) store_raw(a, my_gi, s.res, nev);
                acc.evals += (unsigned)nev;
            } else if (a.suspend_after > 0 && s.i >= a.suspend_after) {
                park = true;
            } else if constexpr (INTERLEAVE) {
                ready = Scene::begin(ev, origin + dir * s.te);   // ray.py:15-17
            }
        }
        if (a.suspend_after > 0 && __any(park)) {
            // long rays leave the wave: their lanes take fresh pixels, resume_kernel finishes them
            const bool parked = push_suspended(a, a.suspend_queue, park, my_gi, s, nev);
            if (parked) {
                s_ih[my_slot][my_pix] = kSuspended;
                active = false;
                fin = true;
            } else if (park) {
                if constexpr (INTERLEAVE) ready = Scene::begin(ev, origin + dir * s.te);   // queue full: march on
            }
        }
        if (__any(fin)) {
            // the scheduler has work only when a refill is due (idle lanes >= refill_min; finished tiles are flushed then)
            const int nidle_now = 64 - __popcll(__ballot(active));
            dirty = dirty || nidle_now >= a.refill_min;
        }
        }
        // with age priority on, the scheduler also runs every 16 turns so an ageing ray is noticed without a finish
        if (kAgePriority && a.age_prio > 0 && ++since_sched >= 16) { since_sched = 0; dirty = true; }
        if constexpr (INTERLEAVE) {
            if (active && !ready) ready = Scene::trip(ev);
        }
    }

    // ---- per-wave totals -> one atomic each; histogram flush -----------------------
    unsigned long long* const part = stats_part(a.stats);
    acc.flush(part);
    __syncthreads();   // every wave of the workgroup has left its loop
    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) {
        const unsigned int c = s_hist[b];
        if (c) atomicAdd(&part[kStatsHead + b], (unsigned long long)c);
    }
    if (a.fused_reduce) frame_reduce_by_last_workgroup(a.stats);      // kernel-uniform
}

// ---- long-ray resume ------------------------------------------------------------------------
//
// A frame's time is bounded below by its longest ray, and a long ray only starts when the tile
// order reaches its pixel.  render_kernel therefore parks every ray that is still marching after
// `suspend_after` loop trips (a few per cent of the frame) and moves on, so the first pass is over
// quickly; this kernel then restarts all parked rays AT ONCE -- dense wavefronts of long rays, every
// one of them running from the first microsecond of the pass -- and may park the longest of them
// again for a last, sparse pass.  The march state between two SDF evaluations is the strategy
// record, so a resumed ray continues bit-for-bit where it stopped.  Results go straight to the
// output arrays (scattered 4 + 4 + 1 byte stores of a few per cent of the pixels).
template <class Scene, class Strat, bool INTERLEAVE, bool BATCH>
__global__ __launch_bounds__(64 * kWavesPerWG, 2) void resume_kernel(const KernelArgs a, const int level)
{
    static_assert(!INTERLEAVE || SceneIterative<Scene>::value, "INTERLEAVE needs Scene::Eval");
    using Entry = QEntry<Strat>;
    __shared__ unsigned int s_hist[kHistBins];
    const int lane = lane_id();
    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) s_hist[b] = 0u;
    rm_load_tables<Scene>();
    __syncthreads();

    const unsigned long long pushed = a.stats[6 + level];          // final: the producer kernel has completed
    const unsigned int count = (unsigned int)min(pushed, (unsigned long long)a.queue_cap);
    const Entry* const queue = (const Entry*)a.queue[level];
    const uint32_t frame_elems = (uint32_t)a.rows * (uint32_t)a.width;

    WaveAcc acc;
    // wave-uniform: the queue may still hold entries.  Waves take 64 rays at a time: measured, a pass of
    // few very long rays runs fastest packed into few wavefronts (dealing them out thinly over every
    // SIMD was 1.5x slower -- the whole device busy with one chain per SIMD clocks lower).
    bool more = count > 0;
    bool active = false;
    uint32_t my_gi = 0;
    int nev = 0;
    vec3 origin = v3(0.0, 0.0, 0.0), dir = v3(0.0, 0.0, 0.0);
    MarchCfg lane_cfg = a.single.cfg;             // BATCH: of the frame this lane's ray belongs to
    int raw_out = a.raw_outputs;                  // parity outputs requested (kept per lane, see store_raw)
    pin_lane(raw_out);
    if constexpr (!BATCH) pin_cfg(lane_cfg);      // one frame: per-lane copies of the kernel arguments (see pin_cfg)
    const MarchCfg& cfg = lane_cfg;
    Strat s;
    typename EvalOf<Scene, INTERLEAVE>::type ev;
    bool ready = false;

    for (;;) {
        // ---- refill idle lanes from the queue ----------------------------------------------------
        const unsigned long long idle = __ballot(!active);
        const int nidle = __popcll(idle);
        if (more && (nidle >= a.refill_min || nidle == 64)) {
            const int take = nidle;
            unsigned int base = 0;
            if (lane == 0) base = (unsigned int)atomicAdd(&a.stats[8 + level], (unsigned long long)take);
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            if (base >= count) {
                more = false;                     // every wave gets here: the grid always drains
            } else if (!active) {
                const int rank = rank_in_mask(idle);
                const unsigned int idx = base + (unsigned int)rank;
                if (rank < take && idx < count) {
                    const Entry e = queue[idx];
                    my_gi = e.gi;
                    s = e.s;
                    nev = (int)e.nev;
                    const uint32_t frame = my_gi / frame_elems;
                    const uint32_t pix = my_gi - frame * frame_elems;
                    const int y = (int)(pix / (uint32_t)a.width);
                    const int x = (int)(pix - (uint32_t)y * (uint32_t)a.width);
                    const int gy = a.band_rows > 0
                        ? a.row0 + ((y / a.band_rows) * a.band_stride + a.band_offset) * a.band_rows + (y % a.band_rows)
                        : a.row0 + y;
                    // the camera ray is recomputed: same bits as in the first pass
                    if constexpr (BATCH) {
                        const FrameParams& fp = a.frames[frame];
                        camera_ray(fp.cam, a.width, a.height, x, gy, origin, dir);
                        lane_cfg = fp.cfg;
                        lane_cfg.full = a.full;
                    } else {
                        camera_ray(a.single.cam, a.width, a.height, x, gy, origin, dir);
                    }
                    active = true;
                    if constexpr (INTERLEAVE) ready = Scene::begin(ev, origin + dir * s.te);
                }
            }
        }
        if (!__any(active)) {
            if (!more) break;                     // wave-uniform
            continue;                             // the next turn refills (nidle == 64) or ends the queue
        }

        // ---- one SDF evaluation (INTERLEAVE: one trip of it) for every live ray ---------------------
        bool park = false;
        bool consume = active;
        if constexpr (INTERLEAVE) consume = active && ready;
        if (!INTERLEAVE || __any(consume)) {
            if (consume) {
                double d;
                if constexpr (INTERLEAVE) d = Scene::value(ev);
                else d = Scene::sdf(origin + dir * s.te);   // ray.py:15-17
                ++nev;
            if (s.step(d, cfg)) {
                    active = false;
                    const int it = s.res.iters, h = s.res.hit;
                    a.depth[my_gi] = h ? (float)s.res.t : 0.0f;   // types.py:93
                    a.iters[my_gi] = it;
                    a.hit[my_gi] = (uint8_t)h;
                    if (raw_out) store_raw(a, my_gi, s.res, nev);
                    acc.evals += (unsigned)nev;
                    acc.add(it, h);
                    atomicAdd(&s_hist[min(it, a.hist_bins - 1)], 1u);
                    if (a.tile_cost) {
                        const uint32_t frame = my_gi / frame_elems;
                        const uint32_t pix = my_gi - frame * frame_elems;
                        const uint32_t y = pix / (uint32_t)a.width, x = pix - y * (uint32_t)a.width;
                        atomicMax(&a.tile_cost[frame * (uint32_t)a.tiles_per_frame + (y / 4u) * (uint32_t)a.tiles_x + (x >> 6)], it);
                    }
                } else if (a.suspend_after > 0 && s.i >= a.suspend_after) {
                    park = true;
                } else if constexpr (INTERLEAVE) {
                    ready = Scene::begin(ev, origin + dir * s.te);
                }
            }
            if (a.suspend_after > 0 && __any(park)) {
                const bool parked = push_suspended(a, a.suspend_queue, park, my_gi, s, nev);
                if (parked) {
                    active = false;
                } else if (park) {
                    if constexpr (INTERLEAVE) ready = Scene::begin(ev, origin + dir * s.te);
                }
            }
        }
        if constexpr (INTERLEAVE) {
            if (active && !ready) ready = Scene::trip(ev);
        }
    }

    unsigned long long* const part = stats_part(a.stats);
    acc.flush(part);
    __syncthreads();
    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) {
        const unsigned int c = s_hist[b];
        if (c) atomicAdd(&part[kStatsHead + b], (unsigned long long)c);
    }
}

// ---- test entry points: explicit points / rays, one per thread ----------------------

template <class Scene>
__global__ void sdf_eval_kernel(const double* __restrict__ xyz, size_t n, double* __restrict__ out)
{
    rm_load_tables<Scene>();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = Scene::sdf(v3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
}

template <class Scene, class Strat>
__global__ void march_rays_kernel(MarchCfg cfg, const double* __restrict__ origins, const double* __restrict__ dirs,
                                  size_t n, uint8_t* hit, double* t, int32_t* iters, double* final_sdf)
{
    rm_load_tables<Scene>();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vec3 o = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
    vec3 d = normalized(v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));   // Ray.__init__ normalises
    Result r = march_one<Scene, Strat>(o, d, cfg);
    hit[i] = (uint8_t)r.hit; t[i] = r.t; iters[i] = r.iters; final_sdf[i] = r.final_sdf;
}

// ---- wavefront teams ------------------------------------------------------------------------
//
// The time of a frame is bounded below by its longest ray, and a ray is one dependent chain of
// instructions: more lanes do not shorten it.  More WAVEFRONTS do: the three transcendental chains
// of a Mandelbulb trip (Scene::trip_part 0..2) are independent, so a team of kTeam wavefronts -- one
// per SIMD of a compute unit -- carries the same 64 rays, every wave evaluates ONE of the chains for
// all 64 rays and the results are exchanged through LDS (one workgroup barrier per trip).  All waves
// hold the same ray state and run the same control flow (same data, same decisions), so ballots,
// loop counts and barriers line up by construction.  The instruction stream of a trip drops from
// ~1060 to ~360 per wave; the total work stays about the same.
constexpr int kTeam = 3;

// Exchange buffer of a team: [trip parity][value 0..5][lane]; double-buffered so one barrier per trip
// suffices (a wave can only overwrite a buffer after every wave has passed the barrier that follows
// the reads of its previous use).
struct TeamXch {
    double v[2][2 * kTeam][64];
};

// One trip for a team; `turn` counts the team's trips (buffer parity).  Wave-uniform call.
template <class Scene>
__device__ __forceinline__ bool team_trip(typename Scene::Eval& ev, bool go, int part, int lane, TeamXch& x, int& turn)
{
    double o0 = 0.0, o1 = 0.0;
    if (go) Scene::trip_part(ev, part, o0, o1);
    double (*buf)[64] = x.v[turn & 1];
    ++turn;
    buf[2 * part][lane] = o0;
    buf[2 * part + 1][lane] = o1;
    __syncthreads();
    bool done = true;
    if (go) done = Scene::trip_join(ev, buf[0][lane], buf[1][lane], buf[2][lane], buf[3][lane], buf[4][lane], buf[5][lane]);
    return done;
}

// rm_march_rays for a team: 64 rays per workgroup of kTeam waves (no lane refill)
template <class Scene, class Strat>
__global__ __launch_bounds__(64 * kTeam) void march_rays_team_kernel(MarchCfg cfg, const double* __restrict__ origins,
                                                                     const double* __restrict__ dirs, size_t n, uint8_t* hit,
                                                                     double* t, int32_t* iters, double* final_sdf,
                                                                     unsigned long long* busy)
{
    __shared__ TeamXch xch;
    // workgroups behind the teams only keep the chip busy until the teams are through (KEEP BUSY above; busy = nullptr: none)
    const unsigned int nteams = (unsigned int)((n + 63) / 64);
    if (blockIdx.x >= nteams) {
        if (busy) keep_busy_until(busy, (unsigned long long)nteams, 256);
        return;
    }
    rm_load_tables<Scene>();
    const int lane = lane_id();
    const int part = (int)(threadIdx.x >> 6);
    const size_t i = (size_t)blockIdx.x * 64 + (size_t)lane;
    const bool have = i < n;
    vec3 o = v3(0.0, 0.0, 0.0), d = v3(0.0, 0.0, 1.0);
    if (have) {
        o = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        d = normalized(v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));   // Ray.__init__ normalises
    }
    Strat s;
    bool done = true;
    if (have) done = s.start(cfg);
    typename Scene::Eval ev;
    int turn = 0;
    while (__any(!done)) {
        bool ready = true;
        if (!done) ready = Scene::begin(ev, o + d * s.te);
        while (__any(!ready)) {
            const bool fin = team_trip<Scene>(ev, !ready, part, lane, xch, turn);
            if (!ready) ready = fin;
        }
        if (!done) done = s.step(Scene::value(ev), cfg);
    }
    if (have && part == 0) {
        hit[i] = (uint8_t)s.res.hit; t[i] = s.res.t; iters[i] = s.res.iters; final_sdf[i] = s.res.final_sdf;
    }
    if (busy && threadIdx.x == 0) __hip_atomic_fetch_add(busy, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// resume_kernel for teams: one team per workgroup carries 64 parked rays at a time; every wave holds
// the same state and takes the same decisions, wave 0 owns the side effects (queue pops, stores, stats).
// Whole evaluations per turn: the rays that reach this pass are long, near-surface rays whose
// evaluations all take most of the 8 trips.
template <class Scene, class Strat, bool BATCH>
__global__ __launch_bounds__(64 * kTeam) void resume_team_kernel(const KernelArgs a, const int level)
{
    using Entry = QEntry<Strat>;
    __shared__ TeamXch xch;
    __shared__ unsigned int s_hist[kHistBins];
    __shared__ unsigned int s_base;
    const int lane = lane_id();
    const int part = (int)(threadIdx.x >> 6);
    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) s_hist[b] = 0u;
    rm_load_tables<Scene>();
    __syncthreads();

    const unsigned long long pushed = a.stats[6 + level];
    const unsigned int count = (unsigned int)min(pushed, (unsigned long long)a.queue_cap);
    const Entry* const queue = (const Entry*)a.queue[level];
    const uint32_t frame_elems = (uint32_t)a.rows * (uint32_t)a.width;
    // KEEP BUSY (above): the workgroups behind the a.team_wgs teams of this pass wait until the queue is handed out -- while
    // the teams are full, filler would only take their issue slots -- and then keep their compute units busy until the
    // teams have reported in stats word 12 + level.
    if (a.keep_busy > 0 && (int)blockIdx.x >= a.team_wgs) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(&a.stats[8 + level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)count &&
               __builtin_amdgcn_s_memrealtime() - t0 < 3000000ull)
            __builtin_amdgcn_s_sleep(127);
        keep_busy_until(&a.stats[12 + level], (unsigned long long)a.team_wgs, a.keep_busy);
        return;
    }

    WaveAcc acc;
    bool more = count > 0;
    bool active = false;
    uint32_t my_gi = 0;
    int nev = 0;
    vec3 origin = v3(0.0, 0.0, 0.0), dir = v3(0.0, 0.0, 0.0);
    MarchCfg lane_cfg = a.single.cfg;             // BATCH: of the frame this lane's ray belongs to
    int raw_out = a.raw_outputs;                  // parity outputs requested (kept per lane, see store_raw)
    pin_lane(raw_out);
    if constexpr (!BATCH) pin_cfg(lane_cfg);      // one frame: per-lane copies of the kernel arguments (see pin_cfg)
    const MarchCfg& cfg = lane_cfg;
    Strat s;
    typename Scene::Eval ev;
    int turn = 0;

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const int nidle = __popcll(idle);
        if (more && (nidle >= a.refill_min || nidle == 64)) {      // same decision in every wave of the team
            if (part == 0 && lane == 0) s_base = (unsigned int)atomicAdd(&a.stats[8 + level], (unsigned long long)nidle);
            __syncthreads();
            const unsigned int base = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_base);
            __syncthreads();                      // s_base may be rewritten on the next turn
            if (base >= count) {
                more = false;
            } else if (!active) {
                const unsigned int idx = base + (unsigned int)rank_in_mask(idle);
                if (idx < count) {
                    const Entry e = queue[idx];
                    my_gi = e.gi;
                    s = e.s;
                    nev = (int)e.nev;
                    const uint32_t frame = my_gi / frame_elems;
                    const uint32_t pix = my_gi - frame * frame_elems;
                    const int y = (int)(pix / (uint32_t)a.width);
                    const int x = (int)(pix - (uint32_t)y * (uint32_t)a.width);
                    const int gy = a.band_rows > 0
                        ? a.row0 + ((y / a.band_rows) * a.band_stride + a.band_offset) * a.band_rows + (y % a.band_rows)
                        : a.row0 + y;
                    if constexpr (BATCH) {
                        const FrameParams& fp = a.frames[frame];
                        camera_ray(fp.cam, a.width, a.height, x, gy, origin, dir);
                        lane_cfg = fp.cfg;
                        lane_cfg.full = a.full;
                    } else {
                        camera_ray(a.single.cam, a.width, a.height, x, gy, origin, dir);
                    }
                    active = true;
                }
            }
        }
        if (!__any(active)) {
            if (!more) break;
            continue;
        }

        // ---- one whole SDF evaluation for every live ray, trips shared by the team -------------------
        bool ready = true;
        if (active) ready = Scene::begin(ev, origin + dir * s.te);   // ray.py:15-17
        while (__any(!ready)) {
            const bool fin = team_trip<Scene>(ev, !ready, part, lane, xch, turn);
            if (!ready) ready = fin;
        }
        bool park = false;
        if (active) {
            ++nev;
            if (!s.step(Scene::value(ev), cfg)) {
                park = a.suspend_after > 0 && s.i >= a.suspend_after;
            } else {
            active = false;
            if (part == 0) {
                const int it = s.res.iters, h = s.res.hit;
                a.depth[my_gi] = h ? (float)s.res.t : 0.0f;   // types.py:93
                a.iters[my_gi] = it;
                a.hit[my_gi] = (uint8_t)h;
                if (raw_out) store_raw(a, my_gi, s.res, nev);
                acc.evals += (unsigned)nev;
                acc.add(it, h);
                atomicAdd(&s_hist[min(it, a.hist_bins - 1)], 1u);
                if (a.tile_cost) {
                    const uint32_t frame = my_gi / frame_elems;
                    const uint32_t pix = my_gi - frame * frame_elems;
                    const uint32_t y = pix / (uint32_t)a.width, x = pix - y * (uint32_t)a.width;
                    atomicMax(&a.tile_cost[frame * (uint32_t)a.tiles_per_frame + (y / 4u) * (uint32_t)a.tiles_x + (x >> 6)], it);
                }
            }
            }
        }
        if (a.suspend_after > 0 && __any(park)) {
            // park the longest rays once more: wave 0 reserves the queue slots, every wave learns the outcome
            const unsigned long long m = __ballot(park);
            if (part == 0 && lane == 0) {
                const unsigned long long b64 = atomicAdd(&a.stats[6 + a.suspend_queue], (unsigned long long)__popcll(m));
                s_base = b64 > 0xffffff00ull ? 0xffffff00u : (unsigned int)b64;      // saturate: never wraps below queue_cap
            }
            __syncthreads();
            const unsigned int base = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_base);
            __syncthreads();
            const unsigned long long idx = (unsigned long long)base + (unsigned long long)rank_in_mask(m);
            if (park && idx < (unsigned long long)a.queue_cap) {
                if (part == 0) {
                    Entry* e = (Entry*)a.queue[a.suspend_queue] + idx;
                    e->gi = my_gi;
                    e->nev = (uint32_t)nev;
                    e->s = s;
                }
                active = false;
            }
        }
    }

    unsigned long long* const spart = stats_part(a.stats);
    if (part == 0) acc.flush(spart);
    __syncthreads();
    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) {
        const unsigned int c = s_hist[b];
        if (c) atomicAdd(&spart[kStatsHead + b], (unsigned long long)c);
    }
    if (a.keep_busy > 0 && threadIdx.x == 0) __hip_atomic_fetch_add(&a.stats[12 + level], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Per-scene launch table, filled by rm_scene_tu.hip (one translation unit per scene).
struct SceneLaunchers {
    hipError_t (*render)(int strategy, int tile_h, const KernelArgs& a, int grid, hipStream_t s);
    hipError_t (*resume)(int strategy, int level, const KernelArgs& a, int grid, hipStream_t s);
    hipError_t (*resume_team)(int strategy, int level, const KernelArgs& a, int grid, hipStream_t s);   // nullptr: no team form
    hipError_t (*pipeline)(int strategy, const KernelArgs& a, int grid, hipStream_t s);                  // rm_pipeline.h (a.tile_h: 4, or 1 where has_teams)
    hipError_t (*occupancy_pipeline)(int strategy, int interleave, int batch, int* blocks_per_cu);
    bool has_teams;
    int (*entry_bytes)(int strategy);   // sizeof(QEntry<Strat>)
    hipError_t (*occupancy)(int strategy, int tile_h, int interleave, int batch, int* blocks_per_cu);
    hipError_t (*sdf_eval)(const double* xyz, size_t n, double* out, hipStream_t s);
    hipError_t (*march_rays)(int strategy, const MarchCfg& cfg, const double* o, const double* d, size_t n,
                             uint8_t* hit, double* t, int32_t* iters, double* fs, hipStream_t s);
    // nullptr: no team form.  busy: a zeroed device word -> `fillers` more workgroups keep the chip busy until the teams are through
    hipError_t (*march_rays_team)(int strategy, const MarchCfg& cfg, const double* o, const double* d, size_t n,
                                  uint8_t* hit, double* t, int32_t* iters, double* fs, unsigned long long* busy, int fillers, hipStream_t s);
};

}  // namespace rm
