// rm_capi.hip -- host side of librm_hip.so: the C ABI declared in include/rm_hip.h.
//
// Owns the device selection, one HIP stream, a grow-only device workspace and the
// (scene, strategy) -> kernel dispatch.  No CPU implementation of the path exists in
// this library: without a usable gfx950 device every entry point returns an error.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is loaded with dlopen on first use

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/rm_hip.h"
#include "rm_kernels.h"
#include "rm_pipeline.h"

static_assert(RM_HIST_BINS == rm::kHistBins, "histogram size mismatch between ABI and kernels");
static_assert(RM_NUM_SCENES == 20 && RM_NUM_STRATEGIES == 11 && RM_NUM_STRATEGY_KERNELS == 13, "registry size");

namespace rm {
#if defined(RM_DEV_STRATEGIES)
// development library (make DEV=1): only some scenes' translation units are linked; the others resolve to null
#define RM_X(id, S) const SceneLaunchers* scene_launchers_##id() __attribute__((weak));
#else
#define RM_X(id, S) const SceneLaunchers* scene_launchers_##id();
#endif
RM_SCENE_LIST(RM_X)
#undef RM_X
static const SceneLaunchers* scene(int id)
{
    switch (id) {
#if defined(RM_DEV_STRATEGIES)
#define RM_X(id, S) case id: return scene_launchers_##id ? scene_launchers_##id() : nullptr;
#else
#define RM_X(id, S) case id: return scene_launchers_##id();
#endif
        RM_SCENE_LIST(RM_X)
#undef RM_X
    }
    return nullptr;
}
}  // namespace rm

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(RM_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return RM_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(RM_E_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return RM_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
};

struct State {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    bool stats_ready = false;   // the library's statistics buffer was left clean by a fused-reduce frame (launch_frame)
    Buf stats, depth, iters, hit, traw, fs, bvar, evals, in0, in1, out0, out1, out2, out3, tcost, torder, queue[rm::kQueues];
    // shape of the frame whose per-tile costs sit in `tcost` (tile_order_mode 1 needs a match)
    long long cost_key[10] = { -1 };
    bool cost_valid = false;
    hipEvent_t ev[2 * RM_MAX_TIMED];
    bool events = false;
    Buf bstats;   // rm_render_batch: the device frame table
    Buf ctl;      // single-launch pipeline: its hot counters, one per 128-byte line
    Buf busy;     // rm_march_rays_team: the counter its filler workgroups watch (its own word: a frame in flight owns `ctl`)
    Buf trace, trace_start, trace_detach;   // development trace of single-launch frames (rm_debug_set_trace)
    bool tracing = false;
    size_t trace_pix = 0;
    unsigned long long trace_t0 = 0;        // low bits irrelevant: the start mark of the traced frame is read back from the stats block
    Buf ccost, corder;   // single-launch pipeline: the centre-out tile order of the frame shape `corder_key`
    long long corder_key[12] = { -1 };
    bool corder_valid = false;
    // optional per-pass timing of the last frame (rm_set_pass_timing): events around the passes
    bool pass_timing = false;
    hipEvent_t pev[RM_MAX_PASSES + 1];
    bool pev_ready = false;
    int pass_count = 0;
    bool last_was_pipeline = false;                  // rm_get_pass_ms decodes the in-kernel marks of `last_stats`
    const unsigned long long* last_stats = nullptr;
    float long_marks[4] = { 0.f, 0.f, 0.f, 0.f };    // longest rays: earliest / latest push, shortest / longest stay with a team
    float last_push_ms = 0.f, last_pop_ms = 0.f;     // queue-1 marks of the last single-launch frame decoded by rm_get_pass_ms
    uint32_t generation = 0;                         // tag of the queue entries of the latest single-launch frame
    // Who wrote each parked-ray queue last.  A single-launch consumer takes an entry for published when the word at the
    // entry's `ready` offset equals the launch's generation tag; a frame with another entry stride (another strategy), or
    // one that follows a multi-pass frame (entries without tags), would put those offsets on stale payload words -- old
    // output indices, counts, halves of doubles -- that can equal a small tag.  The span a different writer may have
    // touched is therefore cleared before a single launch uses the queue (launch_frame).
    struct QueueKey { int stride = 0; int writer = 0; size_t used = 0; } qkey[rm::kQueues];   // writer: 1 = pass per launch, 2 = single launch, 3 = unknown
    hipEvent_t frame_ev = nullptr;                   // end of the latest frame, on `frame_stream` (frames share one workspace)
    hipStream_t frame_stream = nullptr;
    bool frame_ev_valid = false;
} g;

std::mutex g_mu;

constexpr size_t kStatsBlockBytes = sizeof(unsigned long long) * rm::kStatsWords;   // the canonical block (what the host reads)
constexpr size_t kStatsBytes = kStatsBlockBytes * rm::kStatsBlocks;                 // + the partial blocks (device buffer size)

int check_ready()
{
    if (!g.ready) return fail(RM_E_NO_DEVICE, "rm_init() has not succeeded: no gfx950 device bound");
    return RM_OK;
}

// A caller's hipStream_t must belong to the HIP runtime this library is bound to.  A process can hold two copies of
// libamdhip64 (librm_hip.so loaded before PyTorch, whose wheel bundles its own runtime under the file name
// libamdhip64.so: the loader does not match it with the already loaded SONAME libamdhip64.so.7); a stream made by the
// other copy is a pointer into a foreign runtime's heap, and HIP 7.2 dereferences whatever it is handed (measured on the
// MI355X box: hipStreamGetFlags on a pointer that is no stream faults instead of returning an error), so a handle cannot be
// validated by asking the runtime.  What CAN be known without touching the handle: streams this library made itself
// (rm_stream_create) are ours; and while only ONE libamdhip64 is mapped in the process, every hipStream_t there is must
// be that runtime's.  With two copies mapped a handle we did not make is refused.
struct HipCopies { int count; char other[256]; const void* mine; };
int count_hip_runtimes_cb(struct dl_phdr_info* info, size_t, void* data)
{
    HipCopies* c = (HipCopies*)data;
    const char* name = info->dlpi_name ? info->dlpi_name : "";
    const char* base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (strncmp(base, "libamdhip64.so", 14) != 0) return 0;
    ++c->count;
    // is this the copy our own calls resolve to?  (its load segments contain the function's address)
    bool ours = false;
    for (int i = 0; i < info->dlpi_phnum; ++i) {
        const ElfW(Phdr)& ph = info->dlpi_phdr[i];
        if (ph.p_type != PT_LOAD) continue;
        const char* lo = (const char*)info->dlpi_addr + ph.p_vaddr;
        if ((const char*)c->mine >= lo && (const char*)c->mine < lo + ph.p_memsz) ours = true;
    }
    if (!ours) snprintf(c->other, sizeof c->other, "%s", name);
    return 0;
}
HipCopies hip_runtimes_loaded()
{
    HipCopies c;
    memset(&c, 0, sizeof c);
    c.mine = reinterpret_cast<const void*>(&hipStreamCreateWithFlags);
    dl_iterate_phdr(count_hip_runtimes_cb, &c);
    return c;
}
std::vector<void*> g_own_streams, g_seen_streams;      // guarded by g_mu
int check_stream(void* stream)
{
    if (!stream) return RM_OK;
    for (void* p : g_own_streams) if (p == stream) return RM_OK;
    for (void* p : g_seen_streams) if (p == stream) return RM_OK;
    const HipCopies c = hip_runtimes_loaded();
    if (c.count > 1)
        return fail(RM_E_BAD_ARG, "stream %p was not made by rm_stream_create, and this process holds %d copies of the HIP runtime "
                                  "(also %s): a stream of another copy cannot be used here.  Create the stream with rm_stream_create, or "
                                  "load the other runtime's owner (e.g. import torch) BEFORE this library so both share one runtime "
                                  "(rm_runtime_info)", stream, c.count, c.other);
    if (g_seen_streams.size() < 4096) g_seen_streams.push_back(stream);
    return RM_OK;
}

int check_desc(const RmFrameDesc* d)
{
    if (!d) return fail(RM_E_BAD_ARG, "desc is NULL");
    if (d->scene_id < 0 || d->scene_id >= RM_NUM_SCENES) return fail(RM_E_BAD_SCENE, "scene_id %d out of range", d->scene_id);
    if (!rm::scene(d->scene_id)) return fail(RM_E_BAD_SCENE, "scene %d is not built into this (development) library", d->scene_id);
    if (d->strategy_id < 0 || d->strategy_id >= RM_NUM_STRATEGY_KERNELS)
        return fail(RM_E_BAD_STRATEGY, "strategy_id %d out of range", d->strategy_id);
    if (d->width <= 0 || d->height <= 0 || d->row0 < 0 || d->rows < 0 ||
        (!(d->band_rows > 0 && d->band_stride > 1) && d->row0 + d->rows > d->height))
        return fail(RM_E_BAD_DIMS, "bad frame slice: %dx%d rows [%d,%d)", d->width, d->height, d->row0, d->row0 + d->rows);
    if ((long long)d->width * d->height > (1ll << 31) - 1) return fail(RM_E_BAD_DIMS, "frame too large");
    if (d->tile_rows != 0 && d->tile_rows != 4 && d->tile_rows != 1) return fail(RM_E_BAD_ARG, "tile_rows must be 0, 4 or 1");
    if (d->tile_order_mode < 0 || d->tile_order_mode > 4) return fail(RM_E_BAD_ARG, "tile_order_mode must be 0 .. 4");
    if (d->eval_mode < 0 || d->eval_mode > 2) return fail(RM_E_BAD_ARG, "eval_mode must be 0, 1 or 2");
    if (d->resume_mode < 0 || d->resume_mode > 3) return fail(RM_E_BAD_ARG, "resume_mode must be 0..3");
    if (d->resume_grid < 0) return fail(RM_E_BAD_ARG, "negative resume_grid");
    if (d->pipeline < 0 || d->pipeline > 2) return fail(RM_E_BAD_ARG, "pipeline must be 0, 1 or 2");
    if (d->team_grid < 0 || d->queue_first < 0 || d->queue_first > 3 || d->team_steal < 0 || d->team_steal > 2 ||
        d->queue_refill_min < 0 || d->queue_refill_min > 64 || d->queue_retry < 0 || d->team_retry < 0 || d->age_priority < 0)
        return fail(RM_E_BAD_ARG, "bad single-launch tuning field");
    if (d->exit_backlog < 0 || d->late_teams > 65536) return fail(RM_E_BAD_ARG, "bad late-team field");
    if (d->keep_busy > (1 << 20)) return fail(RM_E_BAD_ARG, "keep_busy out of range");
    if (d->early_handover > (1 << 20)) return fail(RM_E_BAD_ARG, "early_handover out of range");
    if (d->early_trips < 0 || d->early_trips > 64) return fail(RM_E_BAD_ARG, "early_trips out of range");
    if (d->band_rows < 0 || d->band_stride < 0 || d->band_offset < 0) return fail(RM_E_BAD_ARG, "negative band parameter");
    if (d->band_rows > 0 && d->band_stride > 1) {
        const int th = d->tile_rows ? d->tile_rows : 4;
        if (d->band_rows % th) return fail(RM_E_BAD_ARG, "band_rows must be a multiple of the tile height %d", th);
        if (d->band_offset >= d->band_stride) return fail(RM_E_BAD_ARG, "band_offset must be < band_stride");
        if (d->rows > 0) {
            const long long y = d->rows - 1;
            const long long last = d->row0 + ((y / d->band_rows) * d->band_stride + d->band_offset) * d->band_rows + y % d->band_rows;
            if (last >= d->height) return fail(RM_E_BAD_DIMS, "band-cyclic slice maps row %lld beyond height %d", last, d->height);
        }
    }
    return RM_OK;
}

rm::MarchCfg to_cfg(const RmMarchConfig& m)
{
    rm::MarchCfg c;
    c.hit_threshold = m.hit_threshold;
    c.max_distance = m.max_distance;
    c.lipschitz = m.lipschitz;
    c.max_iterations = m.max_iterations;
    c.full = m.full ? 1 : 0;
    c.prm = rm::default_strat_params();
    if (m.use_params) {
        static_assert(sizeof(RmStrategyParams) == sizeof(rm::StratParams), "RmStrategyParams and rm::StratParams must match");
        memcpy(&c.prm, &m.params, sizeof c.prm);
    }
    return c;
}

// Fill kernel arguments + choose the persistent grid.
int make_args(const RmFrameDesc* d, float* depth, int32_t* iters, uint8_t* hit, double* traw, double* fs,
              long long* bvar, unsigned long long* stats, rm::KernelArgs* a, int* tile_h, int* grid)
{
    const int th = d->tile_rows ? d->tile_rows : 4;
    memset(a, 0, sizeof *a);
    for (int i = 0; i < 14; ++i) a->single.cam.v[i] = d->cam[i];
    a->single.cfg = to_cfg(d->march);
    a->frames = nullptr;
    a->nframes = 1;
    a->full = d->march.full ? 1 : 0;
    a->width = d->width; a->height = d->height; a->row0 = d->row0; a->rows = d->rows;
    a->tiles_x = (d->width + rm::kTileW - 1) / rm::kTileW;
    a->tiles_y = (d->rows + th - 1) / th;
    a->tiles_per_frame = a->tiles_x * a->tiles_y;
    a->tile_h = th;
    // refill batching: ray set-up (~250 instructions) is amortised over the idle lanes it serves.  8 idle lanes is the
    // default (Pillar Forest 1.93 -> 1.69 ms against the 24 used earlier); the scenes whose rays are short -- set-up is a
    // larger share of a ray -- measured better at 16 under the centre-out order (Cube 0.182 -> 0.174 ms, Cylinder
    // 0.366 -> 0.346, Hollow Cube 0.250 -> 0.244, Box Lattice 0.377 -> 0.370, Sphere 0.353 -> 0.345, Metaballs
    // 1.52 -> 1.49, Thin Torus 0.666 -> 0.655), the others not (Menger 0.675 -> 0.702, Pillar Forest 1.70 -> 1.74).
    int refill_default = 8;
    switch (d->scene_id) { case 0: case 2: case 3: case 4: case 6: case 18: case 19: refill_default = 16; break; default: break; }
    a->refill_min = (d->refill_min > 0 && d->refill_min <= 64) ? d->refill_min : refill_default;
    a->hist_bins = rm::kHistBins;
    // one trip per turn pays where the trip count varies (Mandelbulb); the one-trip union scenes run whole evaluations
    a->interleave = d->eval_mode == 2 || (d->eval_mode == 0 && d->scene_id == 10);
    a->age_prio = d->age_priority > 0 ? d->age_priority : 0;
    if (d->band_rows > 0 && d->band_stride > 1) {
        a->band_rows = d->band_rows; a->band_stride = d->band_stride; a->band_offset = d->band_offset;
    }
    a->depth = depth; a->iters = iters; a->hit = hit; a->t_raw = traw; a->final_sdf = fs;
    a->block_var = bvar; a->stats = stats;
    *tile_h = th;
    // persistent grid of 4-wave workgroups; every wave pulls tiles on its own
    const long long ntiles = (long long)a->tiles_x * a->tiles_y;
    const long long max_wgs = (ntiles + rm::kWavesPerWG - 1) / rm::kWavesPerWG;
    long long wgs;
    if (d->grid_waves > 0) {
        wgs = (d->grid_waves + rm::kWavesPerWG - 1) / rm::kWavesPerWG;
    } else {
        int per_cu = 0;
        hipError_t e = rm::scene(d->scene_id)->occupancy(d->strategy_id, th, a->interleave, 0, &per_cu);
        if (e != hipSuccess || per_cu <= 0) per_cu = 2;
        // three workgroups per CU at most: the cheap scenes fit four, and measured 10-16 % slower with
        // four (Sphere 0.50 -> 0.45 ms, Cube 0.40 -> 0.34) while the long-ray scenes are indifferent
        per_cu = std::min(per_cu, 3);
        wgs = (long long)g.prop.multiProcessorCount * per_cu;
    }
    *grid = (int)std::max<long long>(1, std::min<long long>(wgs, max_wgs));
    return RM_OK;
}

// Longest-first tile order from last frame's per-tile cost: a counting sort by descending cost
// (one workgroup; ties keep no particular order -- the order only affects the schedule).
__global__ __launch_bounds__(1024) void order_tiles_kernel(const int32_t* __restrict__ cost, int32_t* __restrict__ order, int n)
{
    constexpr int BINS = 1024;
    __shared__ int hist[BINS];
    for (int b = threadIdx.x; b < BINS; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&hist[min(max(cost[i], 0), BINS - 1)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int b = BINS - 1; b >= 0; --b) { const int c = hist[b]; hist[b] = run; run += c; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int pos = atomicAdd(&hist[min(max(cost[i], 0), BINS - 1)], 1);
        order[pos] = i;
    }
}

// Static priorities: tiles nearer the image centre (where the camera looks) get a higher cost.  xweight = 1: centre-out
// (a disc grows from the middle); xweight < 1 flattens the disc into an ellipse -- at 1/16 the middle ROWS go first, centre
// columns leading: where the geometry runs to the horizon (planes, pillar grids) the long rays lie along the horizon line.
__global__ void center_cost_kernel(int32_t* __restrict__ cost, int tiles_x, int tiles_y, int nframes, int tile_h, int width, int height,
                                   int row0, int band_rows, int band_stride, int band_offset, float xweight)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles_x * tiles_y * nframes) return;
    const int tf = t % (tiles_x * tiles_y);               // tile ids run frame-major: every frame of a batch centre-out
    const int tx = tf % tiles_x, ty = tf / tiles_x;
    const int y0 = ty * tile_h;
    const int gy = band_rows > 0 ? row0 + ((y0 / band_rows) * band_stride + band_offset) * band_rows + (y0 % band_rows) : row0 + y0;
    const float cx = (tx * 64 + 32 - 0.5f * width) / (0.5f * height);      // both axes in units of half the image height
    const float cy = (gy + 0.5f * tile_h - 0.5f * height) / (0.5f * height);
    const float r = sqrtf(xweight * cx * cx + cy * cy);
    cost[t] = max(0, 1023 - (int)(r * 256.0f));
}

// 8x4-block variance numerators 32*sum(x^2) - sum(x)^2 of the finished iteration map (core/types.py:125-133),
// one thread per full block: used instead of the in-flush reduction when rays were parked (their
// pixels are not in the tile when it is flushed).
__global__ void block_var_kernel(const int32_t* __restrict__ iters, int width, int rows, int nframes,
                                 long long* __restrict__ out)
{
    const int bw = width >> 3, bh = rows >> 2;
    const long long n = (long long)bw * bh * nframes;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const int f = (int)(b / ((long long)bw * bh));
    const int r = (int)(b - (long long)f * bw * bh);
    const int by = r / bw, bx = r - by * bw;
    const int32_t* p = iters + (size_t)f * (size_t)rows * (size_t)width + (size_t)(by * 4) * (size_t)width + (size_t)(bx * 8);
    long long S = 0, Q = 0;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const long long v = p[(size_t)y * (size_t)width + x];
            S += v; Q += v * v;
        }
    out[b] = 32 * Q - S * S;
}

// Sum the partial stats blocks into the canonical block 0 (totals: add; iter_max and the complemented iter_min: max).
__global__ void stats_reduce_kernel(unsigned long long* __restrict__ stats)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;      // word index within a block
    if (w >= rm::kStatsWords || w == 0 || (w >= 6 && w < 10) || (w > 10 && w < rm::kStatsHead)) return;   // counters live in block 0 only
    unsigned long long acc = 0;
    const bool is_max = (w == 3 || w == 4);
    for (int p = 1; p <= rm::kStatsParts; ++p) {
        const unsigned long long v = stats[(size_t)p * rm::kStatsWords + w];
        acc = is_max ? (v > acc ? v : acc) : acc + v;
    }
    stats[w] = acc;
}

constexpr long long kQueueCapMax = 1ll << 22;   // default entries per suspended-ray queue (a full queue leaves rays in place)
long long g_queue_cap = kQueueCapMax;           // rm_set_queue_capacity

// Trip budgets of pass 1 / pass 2 (0 = that pass does not park).  desc->suspend_after: 0 = library
// default, < 0 = off, > 0 = explicit.
// Which launch structure a frame with long-ray suspension uses (RmFrameDesc.pipeline; 0 leaves it to the library).
// Measured on Mandelbulb, every strategy, 640x360 ... 5120x2880: the single launch is
// 4-27 % faster than one launch per pass (1080p Standard 11.0-12.4 -> 9.9 ms, Enhanced 9.7 -> 7.9, Hybrid 6.2 ->
// 5.0, 3840x2160 21.4 -> 15.7 ms); at 7680x4320 the frame is throughput-bound and wants every workgroup as a
// producer (52.5 ms without suspension, 59 with the pipeline).  Other scenes keep their measured pass schedules.
int pipeline_mode(const RmFrameDesc* d, long long rays)
{
    if (d->pipeline != 0) return d->pipeline;
    return (d->scene_id == 10 && rays <= 24000000ll && d->march.max_iterations > 128) ? 2 : 1;
}

void suspend_levels(const RmFrameDesc* d, long long rays, int mode, int* park)
{
    // Default: on for Mandelbulb launches of up to ~16 M rays -- those are bound by the latency of a few
    // hundred 512-trip rays (1080p: 15.2 -> 11.1 ms at 32 / 128 trips; 3840x2160: 22.8 -> 20.2 and
    // 5120x2880: 30.3 -> 28.4 ms at 48 / 192); larger launches are throughput-bound (7680x4320: 50 ms
    // without, 53-58 with), every other scene's SDF is too cheap for the extra passes to pay, and the
    // strategies whose rays end early or whose loop index restarts (Overstep-Bisect, Skipping-Spheres)
    // measured no faster or slower with it.
    const bool strat_ok = d->strategy_id != 6 && d->strategy_id != 7;
    const bool dflt = d->scene_id == 10 && strat_ok && rays <= 16000000ll && d->march.max_iterations > 128;
    // Segment and RevAA evaluate the SDF twice per loop trip: half the trip budgets (Segment 19.7 -> 16.4 ms,
    // RevAA 22.2 -> 19.5 ms at 16 / 64)
    const int two = (d->strategy_id == 10 || d->strategy_id == 8) ? 2 : 1;
    const int d0 = (rays <= 3000000ll ? 32 : 48) / two, d1 = (rays <= 3000000ll ? 128 : 192) / two;
    park[0] = d->suspend_after[0] > 0 ? d->suspend_after[0] : (d->suspend_after[0] == 0 && dflt ? d0 : 0);
    park[1] = d->suspend_after[1] > 0 ? d->suspend_after[1] : (d->suspend_after[1] == 0 && dflt && d->suspend_after[0] == 0 ? d1 : 0);
    // Grazing Plane and Thin Planes Stack: a large share of the frame runs hundreds of trips (whole pixel rows
    // skim the planes; mean 40-68 trips).  Parking at 128 trips turns those rays into dense wavefronts of
    // their own instead of dragging them along with short rays -- lane compaction: 20-45 % faster for every strategy but
    // Skipping-Spheres (Grazing Plane / Segment 2.16 -> 1.22 ms, Thin Planes / Hybrid 2.23 -> 1.49).  No other
    // scene gains (17 scenes x 4 strategies measured).
    if ((d->scene_id == 1 || d->scene_id == 13) && d->strategy_id != 7 && d->suspend_after[0] == 0 && rays <= 16000000ll &&
        d->march.max_iterations > 128)
        park[0] = 128;
    // Sphere Cloud and Bumpy Sphere (unions of 24 / 31 spheres: a pow per sphere and evaluation).  The long rays are
    // parked at 16 trips and finished by TEAMS, each wave taking a third of the sphere list: 3.8 -> 3.0 ms and
    // 6.4 -> 5.0-5.2 ms (Curvature 6.7 -> 5.6, Segment 7.3 -> 5.7).  Not for Adaptive-Hybrid, whose rays end early
    // (2.33 vs 1.88 ms and 3.23 vs 3.10 without).
    if ((d->scene_id == 14 || d->scene_id == 15) && strat_ok && d->strategy_id != 9 && d->suspend_after[0] == 0 &&
        rays <= 16000000ll && d->march.max_iterations > 128)
        park[0] = 16 / two;
    // Gyroid (three sincos per evaluation, long skimming rays inside the ball) parked at 24 trips in round 1 (1.88 ->
    // 1.63 ms in natural tile order).  With the centre-out order its long rays start early anyway: parking measured
    // 1.63 vs 1.60 ms without (Adaptive-Hybrid 1.17 vs 1.05), so it no longer parks.
    // Single launch (Mandelbulb): rays are struck from their tile at 16 trips (the tile slot is free again) and handed
    // to the teams at 48; larger frames, and Segment whose trips evaluate twice, at 32 / 64.  Every strategy gains,
    // Overstep-Bisect and Skipping-Spheres included (3.56 -> 3.08 ms, 11.3 -> 10.2 ms).
    if (mode == 2 && d->scene_id == 10 && d->march.max_iterations > 128) {
        // (with the previous frame's tile costs the long rays start first: 24 / 56 with 7/16 of the grid as teams measured
        // 7.3-7.5 ms, 32 / 64 7.5-7.6, 16 / 48 8.5)
        const bool small = rays <= 3000000ll && d->strategy_id != 10 && d->tile_order_mode != 1;
        const bool ordered = rays <= 3000000ll && d->strategy_id != 10 && d->tile_order_mode == 1;
        // (a strike at 24 measured the same as 16 over the Mandelbulb's three curated viewpoints x Standard / Enhanced / Adaptive-Hybrid
        // -- sums 28.98 / 21.30 / 17.00 ms against 28.96 / 21.32 / 16.93, profiles/r03/viewpoint_budgets.jsonl)
        // (with the early hand-over of near-surface rays the strike is back at 16: those rays leave the producer as soon as they
        // are struck)
        if (d->suspend_after[0] == 0) park[0] = small ? 16 : (ordered ? 24 : 32);
        if (d->suspend_after[1] == 0 && d->suspend_after[0] == 0) park[1] = small ? 48 : (ordered ? 56 : 64);
    }
    if (park[0] == 0) park[1] = 0;
    if (park[1] > 0 && park[1] <= park[0]) park[1] = 0;
}

void frame_key(const RmFrameDesc* d, int tile_h, long long* k)
{
    k[0] = d->scene_id; k[1] = d->strategy_id; k[2] = d->width; k[3] = d->height; k[4] = d->row0; k[5] = d->rows;
    k[6] = d->band_rows; k[7] = d->band_stride; k[8] = d->band_offset; k[9] = tile_h;
}

// The library's tile order (RmFrameDesc.tile_order_mode = 0).  A frame ends with its longest ray, and that ray starts
// when the order reaches its tile; the registry's cameras look at their object, so handing tiles out from the image
// centre outwards starts the object -- and its grazing / fractal rays -- first.  Measured at 1920x1080, Standard
// (natural -> centre-out): Sphere 0.46 -> 0.40 ms, Cube 0.239 -> 0.219, Menger 0.87 -> 0.74, Near Miss 0.66 -> 0.58,
// Cylinder 0.51 -> 0.39, Hollow Cube 0.35 -> 0.28, Box Lattice 0.52 -> 0.38, Metaballs 1.78 -> 1.47, Mandelbulb single
// launch 10.5 -> 9.7 (16 of 20 scenes gain, 3-27 %); worse where the long rays are NOT in the middle -- planes and
// pillars to the horizon: Grazing Plane 0.58 -> 0.70, Thin Planes Stack 0.92 -> 1.05, Pillar Forest 1.95 -> 2.10 --
// which keep the natural order (Bad Lipschitz Sphere: no difference) -- except that Pillar Forest, whose long rays lie along
// the horizon line in the middle rows, takes the middle-rows-first order (4): Standard 1.76 -> 1.72, Segment 2.24 -> 2.00
// (Grazing Plane 0.58 -> 0.63 and Thin Planes Stack 0.90 -> 0.91 do not gain: their horizon is not the middle row / the
// natural order already reaches it in time).  The permutation is cached
// per frame shape.  Batches keep the natural order (their tiles run frame-major).
int default_tile_order(const RmFrameDesc* d, int nframes)
{
    if (nframes > 1) return d->scene_id == 10 ? 2 : 3;     // Mandelbulb sweeps: centre-out within every frame
    // Round 3 held the choice against EVERY curated viewpoint of the reference (viewpoints.py:41-123; 53 cameras at
    // 1920x1080, Standard, profiles/r03/viewpoint_orders.jsonl): centre-out is within 5 % of the best static order for all
    // viewpoints of 15 scenes; the plane scenes want the natural order from every camera (centre-out +15...26 %); Pillar
    // Forest, Thin Torus (ring seen edge-on: -13 %, -8 %, -4 %) and Near Miss (the gap between the spheres: -13 %, -10 %,
    // -4 %) the middle rows first.  Intermediate ellipses (horizontal weight 1/4, 1/2) were measured too: never the best.
    // No single static order is within 5 % everywhere (centre-out: 17 of 53 cameras behind, natural 35, middle rows 29).
    switch (d->scene_id) {
        case 1: case 13: return 3;
        case 3: case 5: case 12: return 4;
        default: return 2;
    }
}

int launch_frame(const RmFrameDesc* d, rm::KernelArgs a, int tile_h, int grid, hipStream_t s);

// One frame: (optional) longest-first tile order from the previous frame's costs, stats reset, render.
int launch(const RmFrameDesc* d, rm::KernelArgs a, int tile_h, int grid, hipStream_t s)
{
    // The parked-ray queues, tile costs / orders, the control block and the pass events are ONE workspace: frames
    // are serialised on the device.  A frame enqueued on another stream than the previous one first waits for it
    // (callers may still overlap their own copies and other kernels with a frame).
    if (g.frame_ev_valid && g.frame_stream != s) HIP_TRY(hipStreamWaitEvent(s, g.frame_ev, 0));
    const int rc_frame = launch_frame(d, a, tile_h, grid, s);
    if (rc_frame) return rc_frame;
    if (!g.frame_ev_valid) {
        HIP_TRY(hipEventCreateWithFlags(&g.frame_ev, hipEventDisableTiming));
        g.frame_ev_valid = true;
    }
    HIP_TRY(hipEventRecord(g.frame_ev, s));
    g.frame_stream = s;
    return RM_OK;
}

int launch_frame(const RmFrameDesc* d, rm::KernelArgs a, int tile_h, int grid, hipStream_t s)
{
    a.raw_outputs = (a.t_raw || a.final_sdf || a.evals) ? 1 : 0;
    // launch structure and trip budgets first: the single launch of a scene with teams uses one-row tiles
    const long long rays_total = (long long)a.rows * a.width * a.nframes;
    int park[2];
    const int mode = pipeline_mode(d, rays_total);
    suspend_levels(d, rays_total, mode, park);
    // One-pass frames fold their statistics in the render kernel itself (frame_reduce_by_last_workgroup) and leave the
    // buffer zeroed where the next frame needs it: no reduce launch, and -- for the library's own buffer, which nobody
    // else writes -- no memset either.  A caller's buffer is always cleared (its contents are not ours to trust).
    const bool own_stats = a.stats == (unsigned long long*)g.stats.p;
    a.fused_reduce = (park[0] == 0 && a.nframes == 1 && d->rows > 0) ? 1 : 0;
    if (!(own_stats && g.stats_ready)) HIP_TRY(hipMemsetAsync(a.stats, 0, kStatsBytes, s));
    if (own_stats) g.stats_ready = a.fused_reduce != 0;
    if (d->rows == 0) return RM_OK;
    const rm::SceneLaunchers* const sc = rm::scene(d->scene_id);
    if (d->tile_rows == 1 && !(park[0] > 0 && mode == 2 && sc->has_teams))
        return fail(RM_E_BAD_ARG, "tile_rows = 1 exists for the single launch (pipeline = 2 with suspension) of scenes with a team form");
    if (park[0] > 0 && mode == 2 && sc->has_teams && (d->tile_rows == 1 || (d->tile_rows == 0 && d->tile_order_mode == 1))) {
        // 64x1 tiles: all 64 pixels of a tile start when the tile is opened.  With 64x4 tiles the last pixels of a
        // tile wait in its pixel pool for lanes that rays of 16-48 trips hold (~1 ms each).  Default only with the
        // previous frame's tile costs (tile_order_mode 1: the long rays' tiles are opened first, so their pixels
        // should not queue inside them -- 1080p 9.6 -> 8.5 ms); with a static order 64x4 tiles measured better
        // (9.9 against 10.9 ms: the tile order does not know where the long rays are).
        tile_h = 1;
        a.tile_h = 1;
        a.tiles_y = a.rows;
        a.tiles_per_frame = a.tiles_x * a.tiles_y;
    }
    const int ntiles = a.tiles_per_frame * a.nframes;
    // Tile order: 1 = longest-first from the previous frame's costs, 2 = centre-out, 3 = natural, 4 = middle rows first,
    // 0 = the library's choice
    int order = d->tile_order_mode;
    if (order == 0) order = default_tile_order(d, a.nframes);
    if (order == 1) {
        int rc;
        if ((rc = g.tcost.ensure((size_t)ntiles * 4)) || (rc = g.torder.ensure((size_t)ntiles * 4))) return rc;
        long long key[10];
        frame_key(d, tile_h, key);
        if (g.cost_valid && memcmp(key, g.cost_key, sizeof key) == 0) {
            hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, s, (const int32_t*)g.tcost.p,
                               (int32_t*)g.torder.p, ntiles);
            HIP_TRY(hipGetLastError());
            a.tile_order = (const int32_t*)g.torder.p;
        }
        a.tile_cost = (int32_t*)g.tcost.p;      // this frame's costs feed the next frame's order
        memcpy(g.cost_key, key, sizeof key);
        g.cost_valid = true;
        if (!a.tile_order) order = default_tile_order(d, a.nframes);    // no costs yet: the first frame takes the static order
    }
    if (order == 2 || order == 4) {
        // A static permutation of the frame shape, computed once and kept until the shape changes: no extra launch per frame.
        long long key[12];
        frame_key(d, tile_h, key);
        key[0] = key[1] = 0;                    // pure geometry: the same permutation for every scene and strategy
        key[10] = a.nframes; key[11] = order;
        int rc2;
        if ((rc2 = g.corder.ensure((size_t)ntiles * 4))) return rc2;
        if (!g.corder_valid || memcmp(key, g.corder_key, sizeof key) != 0) {
            if ((rc2 = g.ccost.ensure((size_t)ntiles * 4))) return rc2;
            hipLaunchKernelGGL(center_cost_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, s, (int32_t*)g.ccost.p, a.tiles_x,
                               a.tiles_y, a.nframes, tile_h, a.width, a.height, a.row0, a.band_rows, a.band_stride, a.band_offset,
                               order == 4 ? 1.0f / 16.0f : 1.0f);
            hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, s, (const int32_t*)g.ccost.p, (int32_t*)g.corder.p, ntiles);
            HIP_TRY(hipGetLastError());
            memcpy(g.corder_key, key, sizeof key);
            g.corder_valid = true;
        }
        a.tile_order = (const int32_t*)g.corder.p;
    }
    // long-ray suspension: pass 1 parks rays beyond suspend_after[0] trips, pass 2 restarts them all at
    // once and parks those beyond suspend_after[1], pass 3 finishes the few that remain
    long long* const block_var = a.block_var;
    if (park[0] > 0) {
        const long long total = (long long)a.rows * a.width * a.nframes;
        const long long cap = std::min<long long>(total, g_queue_cap);
        const int stride = rm::scene(d->scene_id)->entry_bytes(d->strategy_id);
        int rc;
        for (int q = 0; q < (park[1] > 0 ? 2 : 1); ++q) {
            const size_t need = (size_t)cap * (size_t)stride;
            State::QueueKey& key = g.qkey[q];
            if (need > g.queue[q].cap) {
                if ((rc = g.queue[q].ensure(need))) return rc;
                // fresh memory: no word of it may look like a published entry of a later launch (QEntry.ready)
                HIP_TRY(hipMemsetAsync(g.queue[q].p, 0, need, s));
                key.used = 0;
            }
            const int writer = mode == 2 ? 2 : 1;
            if (writer == 2 && key.used > 0 && (key.writer != 2 || key.stride != stride)) {
                // another layout wrote here: stale payload words now sit at this launch's `ready` offsets
                HIP_TRY(hipMemsetAsync(g.queue[q].p, 0, std::min(key.used, g.queue[q].cap), s));
                key.used = 0;
            }
            key.writer = writer;
            key.stride = stride;
            key.used = std::max(key.used, need);
            a.queue[q] = (unsigned char*)g.queue[q].p;
        }
        a.queue_cap = (int32_t)cap;
        a.queue_stride = stride;
        a.suspend_after = park[0];
        a.suspend_queue = 0;
        a.block_var = nullptr;      // parked pixels are missing at flush time: reduced from the finished map below
    }
    const bool pt = g.pass_timing && g.pev_ready;
    g.pass_count = 0;
    g.last_was_pipeline = false;
    if (pt) HIP_TRY(hipEventRecord(g.pev[0], s));
    if (park[0] > 0 && mode == 2) {
        // ---- the whole frame in ONE launch (rm_pipeline.h): producers + queue-0 consumers + teams side by side
        const bool teams = sc->has_teams && park[1] > 0 && d->resume_mode != 1;
        int per_cu = 0;
        if (sc->occupancy_pipeline(d->strategy_id, a.interleave, a.frames != nullptr, &per_cu) != hipSuccess || per_cu <= 0) per_cu = 2;
        per_cu = std::min(per_cu, 3);
        const long long resident = (long long)g.prop.multiProcessorCount * per_cu;
        // producers and teams wait for one another (bounded), so the grid never exceeds what is resident at once.
        // A team workgroup also carries kSharedProducers producer waves when rm::kTeamShare (rm_pipeline.h).
        const long long team_pw = rm::kTeamShare ? rm::kSharedProducers : 0;      // producer waves of a team workgroup
        long long team_wgs = 0;
        if (teams) {
            // Share of the resident workgroups that run as teams (512 resident at 2 per CU).  Measured after the guarded
            // square root made a team's trip shorter (Mandelbulb / Standard, ms per frame by team workgroups):
            //   960x540     128: 8.9   192: 8.4   224: 8.1            1280x720   128: 8.4   192: 8.0   224: 8.1
            //   1920x1080   128: 9.8   160: 9.4   192: 9.4   224: 10.1  (previous frame's costs: 128: 8.5  192: 7.7  240: 7.3-7.5)
            //   2560x1440    64: 11.7   96: 10.8  128: 11.3  192: 11.3   3840x2160  64: 19.2   96: 15.1  128: 15.2  192: 17.9
            //   5120x2880    64: 27.6   96: 24.4  128: 26.2  192: 31.0
            // Small frames are all tail (the chains of the long rays): more teams; large frames are fresh-pixel
            // throughput with a short tail: more producers.
            long long share16 = rays_total <= 1000000ll ? 7 : (rays_total <= 3000000ll ? 6 : 3);      // sixteenths of the grid
            // (with keep_busy, over the three curated viewpoints: 1280x720 wants 128-160 teams, not 224 -- Enhanced 17.6 -> 15.6 ms in
            // the sum at 128, Relaxed / Auto-Relaxed / Slope / Curvature 2-5 % at 160, Standard flat; 960x540 and 1920x1080 stay)
            if (rays_total > 600000ll && rays_total <= 1000000ll) share16 = d->strategy_id == 4 ? 4 : 5;
            if (d->tile_order_mode == 1 && rays_total <= 3000000ll) share16 = 7;                      // (15/32 measured 2 % better still)
            if (a.nframes > 1 && rays_total > 3000000ll) share16 = 4;        // sweeps: 64 x 384^2 viewpoints 29.0 ms (96: 31, 192: 34.6)
            // Strategies whose rays end early hand few rays to the teams: Overstep-Bisect 2.95 / 3.03 / 3.32 ms and
            // Adaptive-Hybrid 4.67 / 4.69 / 4.71 at 96 / 128 / 192 teams (with the previous frame's costs 2.81 vs 3.46
            // and 3.79 vs 4.81 at 128 vs 224; Skipping-Spheres 6.6 vs 6.8).  The other eight gain from the larger share
            // like Standard (Enhanced 7.26 -> 7.1, RevAA 15.1 -> 14.2; ordered: Enhanced 6.3 -> 5.6, RevAA 12.1 -> 10.4).
            if (d->strategy_id == 6 || d->strategy_id == 9) share16 = d->tile_order_mode == 1 ? 4 : 3;
            if (d->strategy_id == 7 && d->tile_order_mode == 1) share16 = 4;
            if (rm::kTeamShare) share16 = 8;
            team_wgs = d->team_grid > 0 ? d->team_grid : std::max<long long>(1, resident * share16 / 16);
            team_wgs = std::min<long long>(team_wgs, std::max<long long>(1, rm::kTeamShare ? resident : resident / 2));
        }
        // Late teams (rm_pipeline.h): by default a third of the team workgroups are resident from the start and the
        // others are put behind the resident grid -- their places are held by producers until queue 1 fills.
        long long late = 0;
        const bool detach_mode = d->queue_first == 3 || (d->queue_first == 0 && teams);
        if (teams && detach_mode && !rm::kTeamShare && d->grid_waves == 0) {
            // measured at 1080p (Mandelbulb / Standard, trace of round 3): with 64 resident + 128 late teams the tile counter
            // runs out at 3.8 instead of 4.6 ms and the last long ray enters queue 1 a millisecond earlier (2.5 vs 3.5 ms),
            // but a producer workgroup only leaves when it would open its next tile (every 1-2 ms per wave in the object's
            // tiles), the late teams arrive at 1.8-2.5 ms and the rays pushed meanwhile wait ~1.3 ms: 9.6 vs 9.8 ms.  Off by
            // default; the demand for teams jumps from 0 to ~140 workgroups within 0.5 ms.
            if (d->late_teams > 0) late = d->late_teams;
        }
        const long long want_pw = d->grid_waves > 0 ? d->grid_waves : (long long)resident * rm::kPipeWaves;   // producer waves asked for
        long long pure = (std::min<long long>(want_pw, ntiles) - team_wgs * team_pw + rm::kPipeWaves - 1) / rm::kPipeWaves;
        pure = std::max<long long>(team_pw > 0 && team_wgs > 0 ? 0 : 1, std::min<long long>(pure, resident - team_wgs));
        late = std::min<long long>(late, std::max<long long>(0, pure - 1));      // one producer workgroup at least stays to the end
        const long long pwgs = pure + late;       // grid = static teams + producers + late teams
        a.team_wgs = (int32_t)team_wgs;
        a.producer_waves = (int32_t)(team_wgs * team_pw + pure * rm::kPipeWaves);
        a.late_team_first = (int32_t)(team_wgs + pure);
        a.early_exit_wgs = (int32_t)late;
        a.exit_backlog = d->exit_backlog > 0 ? d->exit_backlog : 64;
        // KEEP BUSY (rm_kernels.h): finished producer workgroups stay until the teams are through -- 1080p Mandelbulb / Standard
        // 9.4 -> 8.1 ms, Enhanced 7.0 -> 6.2 (burst 16 ... 2048 alike; fp64, fp32 and integer filler alike; s_sleep in the same
        // place: nothing).  Not with late teams, which need the producers' places.
        a.keep_busy = (teams && late == 0) ? (d->keep_busy > 0 ? d->keep_busy : (d->keep_busy == 0 ? 256 : 0)) : 0;
        // EARLY HAND-OVER (rm_pipeline.h): struck near-surface rays go to the teams at once.  Over the Mandelbulb's three curated
        // viewpoints at 1080p with the strike at 16: Standard 8.19 / 13.19 / 7.84 -> 7.81 / 13.02 / 7.81 ms, Enhanced 6.33 / 9.18 / 6.30
        // -> 6.34 / 9.06 / 6.38 (sums -2.0 % / -0.1 %; strikes of 8 ... 24 alike, a regular hand-over later than 48 worse)
        // By strategy (default camera, on / off): Relaxed 7.90 / 8.36, Auto-Relaxed 7.92 / 8.19, Slope 6.67 / 7.10, Curvature 7.65 / 8.12,
        // Segment 10.6 / 11.9, Safe-Relaxed 7.75 / 8.08; RevAA and Dense-March alike; the three whose rays end early or whose loop
        // index restarts lose 1-2 % (Overstep-Bisect 3.09 / 3.03, Skipping-Spheres 8.77 / 8.66, Adaptive-Hybrid 4.74 / 4.69): off there.
        // early_trips: how many of the eight fractal iterations make an evaluation "near-surface".  Six, together with a regular
        // hand-over at 64 instead of 48 trips, measured Standard 7.75 / 11.81 / 7.96 ms against 7.85 / 12.88 / 7.77 from the three
        // curated cameras (Auto-Relaxed 7.83 / 11.99 / 7.54 against 8.01 / 12.61 / 7.71; the bench line 266-270 Mrays/s instead of
        // 261-264) -- but twice as many rays go through the queue: 88.9 MB of HBM traffic per frame instead of 58.2, for 1-2 % from
        // the default camera and a loss from the angled one.  The default stays 8 of 8; the knob is there.
        a.early_trips = d->early_trips > 0 ? d->early_trips : 8;
        const bool eh_default = d->strategy_id != 6 && d->strategy_id != 7 && d->strategy_id != 9;
        a.early_handover = (teams && detach_mode)
            ? (d->early_handover > 0 ? d->early_handover : (d->early_handover == 0 && eh_default ? std::max(1, park[0]) : 0)) : 0;
        a.suspend_after2 = teams ? park[1] : 0;
        {
            int rc2;
            if ((rc2 = g.ctl.ensure(sizeof(unsigned long long) * rm::kCtlWords))) return rc2;
            HIP_TRY(hipMemsetAsync(g.ctl.p, 0, sizeof(unsigned long long) * rm::kCtlWords, s));
            a.ctl = (unsigned long long*)g.ctl.p;
        }
        if (++g.generation == 0) g.generation = 1;
        a.generation = g.generation;
        // default: with teams, rays below suspend_after[1] never leave their lane (no queue-0 traffic); without
        // teams queue 0 is the lane-compaction queue, parked rays first
        a.q0_detach = d->queue_first == 3 || (d->queue_first == 0 && teams);
        a.q0_first = d->queue_first == 0 ? 1 : (d->queue_first == 1 ? 1 : 0);
        a.q0_refill_min = d->queue_refill_min > 0 ? d->queue_refill_min : 16;
        a.q0_retry = d->queue_retry > 0 ? d->queue_retry : 16;
        a.team_retry = d->team_retry > 0 ? d->team_retry : 8;      // (with keep_busy: 2: 8.29, 4: 8.19, 8: 7.97, 16: 8.12, 32: 8.33 ms; other strategies flat)
        a.team_steal = d->team_steal == 0 ? 1 : (d->team_steal == 1 ? 1 : 0);
        a.max_spins = 50000;
        a.marks = (g.pass_timing || g.tracing) ? 1 : 0;      // device-clock marks only when somebody will read them (rm_get_pass_ms)
        if (g.tracing) {
            constexpr size_t kTraceRecords = 1u << 20;
            const size_t npix = (size_t)a.rows * a.width * a.nframes;
            int rc3;
            if ((rc3 = g.trace.ensure((8 + 8 * kTraceRecords) * 4)) || (rc3 = g.trace_start.ensure(npix * 4)) ||
                (rc3 = g.trace_detach.ensure(npix * 4)))
                return rc3;
            HIP_TRY(hipMemsetAsync(g.trace.p, 0, 32, s));
            HIP_TRY(hipMemsetAsync(g.trace_start.p, 0, npix * 4, s));
            HIP_TRY(hipMemsetAsync(g.trace_detach.p, 0, npix * 4, s));
            a.trace = (uint32_t*)g.trace.p;
            a.trace_cap = (uint32_t)kTraceRecords;
            a.trace_start = (uint32_t*)g.trace_start.p;
            a.trace_detach = (uint32_t*)g.trace_detach.p;
            g.trace_pix = npix;
        }
        a.team_prio = 3;      // 0 / 1 / 3 measured alike (10.0-10.4 ms): what slows a ray next to producers is not the issue slot      // ~50 ms of polling: only reached when part of the grid is not resident
        if (a.tile_cost) HIP_TRY(hipMemsetAsync(a.tile_cost, 0, (size_t)ntiles * 4, s));   // resumed rays may report before the tile flush
        HIP_TRY(sc->pipeline(d->strategy_id, a, (int)(pwgs + team_wgs), s));
        if (pt) HIP_TRY(hipEventRecord(g.pev[++g.pass_count], s));
        g.last_was_pipeline = true;
        g.last_stats = a.stats;
    } else {
    HIP_TRY(sc->render(d->strategy_id, tile_h, a, grid, s));
    if (pt) HIP_TRY(hipEventRecord(g.pev[++g.pass_count], s));
    }
    if (park[0] > 0 && mode != 2) {
        rm::KernelArgs b = a;
        b.suspend_after = park[1];
        b.suspend_queue = 1;
        b.refill_min = 16;
        // one workgroup per compute unit: measured best for both the dense second pass and the sparse last one
        const int rgrid = d->resume_grid > 0 ? d->resume_grid : std::min(grid, g.prop.multiProcessorCount);
        const bool team = rm::scene(d->scene_id)->resume_team != nullptr && d->resume_mode != 1;
        // KEEP BUSY (rm_kernels.h): a team pass is followed by as many filler workgroups, which start when its queue is handed out
        b.team_wgs = rgrid;
        b.keep_busy = team ? (d->keep_busy > 0 ? d->keep_busy : (d->keep_busy == 0 ? 256 : 0)) : 0;
        const int tgrid = b.keep_busy > 0 ? 2 * rgrid : rgrid;
        if (team && (park[1] == 0 || d->resume_mode == 3))
            HIP_TRY(rm::scene(d->scene_id)->resume_team(d->strategy_id, 0, b, tgrid, s));
        else
            HIP_TRY(rm::scene(d->scene_id)->resume(d->strategy_id, 0, b, rgrid, s));
        if (pt) HIP_TRY(hipEventRecord(g.pev[++g.pass_count], s));
        if (park[1] > 0) {
            b.suspend_after = 0;
            b.interleave = 0;       // a sparse pass of very long rays is latency-bound: whole evaluations per turn
            if (team)
                HIP_TRY(rm::scene(d->scene_id)->resume_team(d->strategy_id, 1, b, tgrid, s));
            else
                HIP_TRY(rm::scene(d->scene_id)->resume(d->strategy_id, 1, b, rgrid, s));
            if (pt) HIP_TRY(hipEventRecord(g.pev[++g.pass_count], s));
        }
    }
    if (park[0] > 0 && block_var) {
        const long long nb = (long long)(a.width >> 3) * (a.rows >> 2) * a.nframes;
        if (nb > 0) {
            hipLaunchKernelGGL(block_var_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, a.iters, a.width,
                               a.rows, a.nframes, block_var);
            HIP_TRY(hipGetLastError());
        }
    }
    if (!a.fused_reduce) {
        hipLaunchKernelGGL(stats_reduce_kernel, dim3((rm::kStatsWords + 255) / 256), dim3(256), 0, s, a.stats);
        HIP_TRY(hipGetLastError());
    }
    return RM_OK;
}

// a single-launch frame whose queue protocol ran into one of its wait bounds (rm_pipeline.h) reports it here
int check_pipeline_error(const unsigned long long* w)
{
    if (w[rm::kWError] != 0)
        return fail(RM_E_HIP, "single-launch pipeline: a wait of the queue protocol hit its bound (code %llu); results are incomplete", w[rm::kWError]);
    return RM_OK;
}

void decode_stats(const unsigned long long* w, RmStats* out)
{
    out->hit_count = w[1];
    out->sum_iters = w[2];
    out->iter_max = (int32_t)w[3];
    out->total_rays = w[5];
    out->sum_evals = w[10];
    out->iter_min = w[5] ? (int32_t)(0x7fffffffull - w[4]) : 0;
    for (int b = 0; b < RM_HIST_BINS; ++b) out->iter_hist[b] = w[rm::kStatsHead + b];
}

void summarise(RmTiming* t)
{
    const int n = t->repeats;
    if (n <= 0) { t->ms_median = t->ms_mean = t->ms_min = t->ms_max = 0.f; return; }
    std::vector<float> v(t->ms_each, t->ms_each + n);
    std::sort(v.begin(), v.end());
    t->ms_min = v.front(); t->ms_max = v.back();
    t->ms_median = (n & 1) ? v[n / 2] : 0.5f * (v[n / 2 - 1] + v[n / 2]);
    double sum = 0; for (float x : v) sum += x;
    t->ms_mean = (float)(sum / n);
}

int ensure_events()
{
    if (g.events) return RM_OK;
    for (auto& e : g.ev) HIP_TRY(hipEventCreate(&e));
    g.events = true;
    return RM_OK;
}

// Launch warmup + repeats times; each timed launch bracketed by events on the launch stream.
int timed_launches(const RmFrameDesc* d, const rm::KernelArgs& a, int tile_h, int grid, RmTiming* t)
{
    if (t->repeats < 1 || t->repeats > RM_MAX_TIMED || t->warmup < 0)
        return fail(RM_E_BAD_ARG, "timing: repeats must be 1..%d, warmup >= 0", RM_MAX_TIMED);
    int rc = ensure_events();
    if (rc) return rc;
    for (int i = 0; i < t->warmup; ++i)
        if ((rc = launch(d, a, tile_h, grid, g.stream))) return rc;
    for (int i = 0; i < t->repeats; ++i) {
        // events bracket one whole frame: stats reset, optional tile ordering, render kernel
        HIP_TRY(hipEventRecord(g.ev[2 * i], g.stream));
        if ((rc = launch(d, a, tile_h, grid, g.stream))) return rc;
        HIP_TRY(hipEventRecord(g.ev[2 * i + 1], g.stream));
    }
    HIP_TRY(hipStreamSynchronize(g.stream));
    for (int i = 0; i < t->repeats; ++i) HIP_TRY(hipEventElapsedTime(&t->ms_each[i], g.ev[2 * i], g.ev[2 * i + 1]));
    summarise(t);
    return RM_OK;
}

// ---- RCCL, loaded on first use ------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    ncclComm_t comm = nullptr;
    int world = 0, rank = -1;
    Buf gather[3], pad[3];      // all-gather landing buffers (rank-major) and padded send buffers of a short last shard
} R;

int rccl_load()
{
    if (R.handle) return RM_OK;
    void* h = nullptr;
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return fail(RM_E_RCCL, "librccl.so.1 could not be loaded: %s", dlerror());
#define RM_SYM(field, sym)                                                                     \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(h, #sym));                              \
    if (!R.field) { dlclose(h); return fail(RM_E_RCCL, "librccl lacks %s", #sym); }
    RM_SYM(GetUniqueId, ncclGetUniqueId)
    RM_SYM(CommInitRank, ncclCommInitRank)
    RM_SYM(CommDestroy, ncclCommDestroy)
    RM_SYM(AllGather, ncclAllGather)
    RM_SYM(Send, ncclSend)
    RM_SYM(Recv, ncclRecv)
    RM_SYM(GroupStart, ncclGroupStart)
    RM_SYM(GroupEnd, ncclGroupEnd)
    RM_SYM(GetErrorString, ncclGetErrorString)
#undef RM_SYM
    R.handle = h;
    return RM_OK;
}

#define RCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess)                                                                 \
            return fail(RM_E_RCCL, "%s failed: %s (%s:%d)", #expr, R.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// Rows of the gathered shards (rank-major) -> image order.  One thread per 16-byte (or 1-byte) piece of a row.
template <class T>
__global__ void assemble_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int world, int height, long long row_elems,
                                     int rows_per_rank, int cyclic)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_elems) return;
    const int y = blockIdx.y;                          // image row
    int r, local;
    if (cyclic) {
        const int band = y >> 2;                       // 4-row bands dealt round-robin
        r = band % world;
        local = (band / world) * 4 + (y & 3);
    } else {
        r = y / rows_per_rank;
        local = y - r * rows_per_rank;
    }
    dst[(long long)y * row_elems + i] = src[((long long)r * rows_per_rank + local) * row_elems + i];
}

int assemble(int world, int height, int width, int rows_per_rank, int cyclic, int elem_bytes, const void* src, void* dst, hipStream_t s)
{
    const long long row_bytes = (long long)width * elem_bytes;
    if (height <= 0 || row_bytes <= 0) return RM_OK;
    const bool vec = (row_bytes % 16 == 0) && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    if (vec) {
        const long long n = row_bytes / 16;
        hipLaunchKernelGGL((assemble_rows_kernel<uint4>), dim3((unsigned)((n + 255) / 256), (unsigned)height), dim3(256), 0, s,
                           (const uint4*)src, (uint4*)dst, world, height, n, rows_per_rank, cyclic);
    } else {
        hipLaunchKernelGGL((assemble_rows_kernel<unsigned char>), dim3((unsigned)((row_bytes + 255) / 256), (unsigned)height), dim3(256), 0, s,
                           (const unsigned char*)src, (unsigned char*)dst, world, height, row_bytes, rows_per_rank, cyclic);
    }
    HIP_TRY(hipGetLastError());
    return RM_OK;
}

// ---- store-path probe: the flush of render_kernel without the march --------------------
__global__ __launch_bounds__(64) void store_path_kernel(float* depth, int32_t* iters, uint8_t* hit, int width,
                                                        int rows, int tiles_x, int ntiles)
{
    constexpr int TILE_H = 4;
    __shared__ float s_depth[64 * TILE_H];
    __shared__ int32_t s_iters[64 * TILE_H];
    __shared__ uint8_t s_hit[64 * TILE_H];
    const int lane = threadIdx.x;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int x0 = (tile % tiles_x) * 64, y0 = (tile / tiles_x) * TILE_H;
        for (int r = 0; r < TILE_H; ++r) {
            s_depth[r * 64 + lane] = (float)(tile + r);
            s_iters[r * 64 + lane] = tile ^ lane;
            s_hit[r * 64 + lane] = (uint8_t)((tile + lane) & 1);
        }
        __syncthreads();
        const int gx = x0 + lane;
        for (int r = 0; r < TILE_H; ++r)
            if (gx < width && y0 + r < rows) {
                const size_t gi = (size_t)(y0 + r) * width + gx;
                depth[gi] = s_depth[r * 64 + lane];
                iters[gi] = s_iters[r * 64 + lane];
                hit[gi] = s_hit[r * 64 + lane];
            }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

const char* rm_last_error(void) { return g_err; }
int rm_num_scenes(void) { return RM_NUM_SCENES; }
int rm_num_strategies(void) { return RM_NUM_STRATEGIES; }

void rm_default_strategy_params(RmStrategyParams* out)
{
    if (!out) return;
    const rm::StratParams p = rm::default_strat_params();
    memcpy(out, &p, sizeof *out);
}
size_t rm_stats_device_bytes(void) { return kStatsBytes; }

int rm_init(int device_id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.ready && g.device == device_id) return RM_OK;
    if (g.ready) return fail(RM_E_BAD_ARG, "already initialised on device %d; call rm_shutdown() first", g.device);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RM_E_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(RM_E_NO_DEVICE, "device %d out of range (have %d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipGetDeviceProperties(&g.prop, device_id));
    if (strncmp(g.prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RM_E_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device_id, g.prop.gcnArchName);
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    int rc = g.stats.ensure(kStatsBytes);
    if (rc) return rc;
    g.device = device_id;
    g.ready = true;
    return RM_OK;
}

void rm_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipStreamSynchronize(g.stream);
    g.corder_valid = false;
    g.cost_valid = false;
    g.stats_ready = false;
    for (auto& k : g.qkey) k = State::QueueKey();
    g.tracing = false;
    g_seen_streams.clear();
    for (Buf* b : { &g.trace, &g.trace_start, &g.trace_detach, &g.ccost, &g.corder, &g.ctl, &g.busy, &g.bstats, &g.stats, &g.depth, &g.iters, &g.hit, &g.traw, &g.fs, &g.bvar, &g.evals, &g.in0, &g.in1, &g.out0, &g.out1,
                    &g.out2, &g.out3, &g.tcost, &g.torder, &g.queue[0], &g.queue[1] })
        b->release();
    if (g.frame_ev_valid) (void)hipEventDestroy(g.frame_ev);
    g.frame_ev_valid = false;
    if (g.events) for (auto& e : g.ev) (void)hipEventDestroy(e);
    g.events = false;
    if (g.pev_ready) for (auto& e : g.pev) (void)hipEventDestroy(e);
    g.pev_ready = false;
    g.pass_timing = false;

    (void)hipStreamDestroy(g.stream);
    g.stream = nullptr;
    g.ready = false;
    g.device = -1;
}

int rm_device_info(RmDeviceInfo* out)
{
    int rc = check_ready();
    if (rc) return rc;
    if (!out) return fail(RM_E_BAD_ARG, "out is NULL");
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", g.prop.name);
    snprintf(out->arch, sizeof out->arch, "%s", g.prop.gcnArchName);
    out->device_id = g.device;
    out->compute_units = g.prop.multiProcessorCount;
    out->clock_mhz = g.prop.clockRate / 1000;
    out->wavefront_size = g.prop.warpSize;
    out->total_mem_bytes = g.prop.totalGlobalMem;
    return RM_OK;
}

int rm_sdf_eval(int scene_id, const double* xyz, size_t n, double* out)
{
    int rc = check_ready();
    if (rc) return rc;
    if (scene_id < 0 || scene_id >= RM_NUM_SCENES || !rm::scene(scene_id)) return fail(RM_E_BAD_SCENE, "scene_id %d out of range", scene_id);
    if (n == 0) return RM_OK;
    if (!xyz || !out) return fail(RM_E_BAD_ARG, "NULL buffer");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = g.in0.ensure(n * 24)) || (rc = g.out0.ensure(n * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(g.in0.p, xyz, n * 24, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(rm::scene(scene_id)->sdf_eval((const double*)g.in0.p, n, (double*)g.out0.p, g.stream));
    HIP_TRY(hipMemcpyAsync(out, g.out0.p, n * 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return RM_OK;
}

static int march_rays_impl(bool team, int scene_id, int strategy_id, const RmMarchConfig* cfg, const double* origins,
                           const double* dirs, size_t n, uint8_t* hit, double* t, int32_t* iters, double* final_sdf)
{
    int rc = check_ready();
    if (rc) return rc;
    if (scene_id < 0 || scene_id >= RM_NUM_SCENES || !rm::scene(scene_id)) return fail(RM_E_BAD_SCENE, "scene_id %d out of range", scene_id);
    if (strategy_id < 0 || strategy_id >= RM_NUM_STRATEGY_KERNELS)
        return fail(RM_E_BAD_STRATEGY, "strategy_id %d out of range", strategy_id);
    if (!cfg) return fail(RM_E_BAD_ARG, "cfg is NULL");
    if (team && !rm::scene(scene_id)->march_rays_team) return fail(RM_E_BAD_SCENE, "scene %d has no wavefront-team form", scene_id);
    if (n == 0) return RM_OK;
    if (!origins || !dirs || !hit || !t || !iters || !final_sdf) return fail(RM_E_BAD_ARG, "NULL buffer");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = g.in0.ensure(n * 24)) || (rc = g.in1.ensure(n * 24)) || (rc = g.out0.ensure(n)) ||
        (rc = g.out1.ensure(n * 8)) || (rc = g.out2.ensure(n * 4)) || (rc = g.out3.ensure(n * 8)))
        return rc;
    HIP_TRY(hipMemcpyAsync(g.in0.p, origins, n * 24, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(g.in1.p, dirs, n * 24, hipMemcpyHostToDevice, g.stream));
    rm::MarchCfg c = to_cfg(*cfg);
    c.full = 1;   // per-ray API always returns final_sdf, like MarchResult
    if (team) {
        // the teams of a few rays are all that runs: filler workgroups (two per compute unit in all) keep the chip at the
        // speed a frame's teams run at (KEEP BUSY, rm_kernels.h) -- rm_march_rays_team is how bench.py measures a chain
        if ((rc = g.busy.ensure(128))) return rc;
        HIP_TRY(hipMemsetAsync(g.busy.p, 0, sizeof(unsigned long long), g.stream));
        const long long nteams = (long long)((n + 63) / 64);
        const int fillers = (int)std::max<long long>(0, 2ll * g.prop.multiProcessorCount - nteams);
        HIP_TRY(rm::scene(scene_id)->march_rays_team(strategy_id, c, (const double*)g.in0.p, (const double*)g.in1.p, n, (uint8_t*)g.out0.p,
                                                     (double*)g.out1.p, (int32_t*)g.out2.p, (double*)g.out3.p,
                                                     (unsigned long long*)g.busy.p, fillers, g.stream));
    } else {
        HIP_TRY(rm::scene(scene_id)->march_rays(strategy_id, c, (const double*)g.in0.p, (const double*)g.in1.p, n, (uint8_t*)g.out0.p,
                                                (double*)g.out1.p, (int32_t*)g.out2.p, (double*)g.out3.p, g.stream));
    }
    HIP_TRY(hipMemcpyAsync(hit, g.out0.p, n, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(t, g.out1.p, n * 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(iters, g.out2.p, n * 4, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(final_sdf, g.out3.p, n * 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return RM_OK;
}

int rm_march_rays(int scene_id, int strategy_id, const RmMarchConfig* cfg, const double* origins, const double* dirs,
                  size_t n, uint8_t* hit, double* t, int32_t* iters, double* final_sdf)
{
    return march_rays_impl(false, scene_id, strategy_id, cfg, origins, dirs, n, hit, t, iters, final_sdf);
}

int rm_march_rays_team(int scene_id, int strategy_id, const RmMarchConfig* cfg, const double* origins, const double* dirs,
                       size_t n, uint8_t* hit, double* t, int32_t* iters, double* final_sdf)
{
    return march_rays_impl(true, scene_id, strategy_id, cfg, origins, dirs, n, hit, t, iters, final_sdf);
}

int rm_render_outputs(const RmFrameDesc* d, const RmOutputs* o, RmStats* stats, RmTiming* timing)
{
    int rc = check_ready();
    if (rc) return rc;
    if ((rc = check_desc(d))) return rc;
    if (!o || !o->depth || !o->iters || !o->hit) return fail(RM_E_BAD_ARG, "depth, iters and hit are required");
    if (o->final_sdf && !d->march.full) return fail(RM_E_BAD_ARG, "final_sdf requires march.full = 1");
    if (o->block_var && (d->row0 % 4) != 0) return fail(RM_E_BAD_ARG, "block_var requires row0 %% 4 == 0");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    const size_t n = (size_t)d->rows * (size_t)d->width;
    const size_t nblk = (size_t)(d->rows / 4) * (size_t)(d->width / 8);
    if ((rc = g.depth.ensure(n * 4 + 16)) || (rc = g.iters.ensure(n * 4 + 16)) || (rc = g.hit.ensure(n + 16))) return rc;
    if (o->t_raw && (rc = g.traw.ensure(n * 8 + 16))) return rc;
    if (o->final_sdf && (rc = g.fs.ensure(n * 8 + 16))) return rc;
    if (o->block_var && (rc = g.bvar.ensure(nblk * 8 + 16))) return rc;
    if (o->evals && (rc = g.evals.ensure(n * 4 + 16))) return rc;
    rm::KernelArgs a;
    int tile_h = 0, grid = 0;
    if ((rc = make_args(d, (float*)g.depth.p, (int32_t*)g.iters.p, (uint8_t*)g.hit.p, o->t_raw ? (double*)g.traw.p : nullptr,
                        o->final_sdf ? (double*)g.fs.p : nullptr, o->block_var ? (long long*)g.bvar.p : nullptr,
                        (unsigned long long*)g.stats.p, &a, &tile_h, &grid)))
        return rc;
    a.evals = o->evals ? (int32_t*)g.evals.p : nullptr;
    if (timing) {
        if ((rc = timed_launches(d, a, tile_h, grid, timing))) return rc;
    } else {
        if ((rc = launch(d, a, tile_h, grid, g.stream))) return rc;
    }
    if (n) {
        HIP_TRY(hipMemcpyAsync(o->depth, g.depth.p, n * 4, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipMemcpyAsync(o->iters, g.iters.p, n * 4, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipMemcpyAsync(o->hit, g.hit.p, n, hipMemcpyDeviceToHost, g.stream));
        if (o->t_raw) HIP_TRY(hipMemcpyAsync(o->t_raw, g.traw.p, n * 8, hipMemcpyDeviceToHost, g.stream));
        if (o->final_sdf) HIP_TRY(hipMemcpyAsync(o->final_sdf, g.fs.p, n * 8, hipMemcpyDeviceToHost, g.stream));
        if (o->block_var && nblk) HIP_TRY(hipMemcpyAsync(o->block_var, g.bvar.p, nblk * 8, hipMemcpyDeviceToHost, g.stream));
        if (o->evals) HIP_TRY(hipMemcpyAsync(o->evals, g.evals.p, n * 4, hipMemcpyDeviceToHost, g.stream));
    }
    unsigned long long w[rm::kStatsWords];
    HIP_TRY(hipMemcpyAsync(w, g.stats.p, kStatsBlockBytes, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    if ((rc = check_pipeline_error(w))) return rc;
    if (stats) decode_stats(w, stats);
    return RM_OK;
}

int rm_render(const RmFrameDesc* d, float* depth, int32_t* iters, uint8_t* hit, double* t_raw, double* final_sdf,
              int64_t* block_var, RmStats* stats, RmTiming* timing)
{
    RmOutputs o;
    memset(&o, 0, sizeof o);
    o.depth = depth; o.iters = iters; o.hit = hit; o.t_raw = t_raw; o.final_sdf = final_sdf; o.block_var = block_var;
    return rm_render_outputs(d, &o, stats, timing);
}

int rm_render_device(const RmFrameDesc* d, void* d_depth, void* d_iters, void* d_hit, void* d_stats, void* stream)
{
    int rc = check_ready();
    if (rc) return rc;
    if ((rc = check_desc(d))) return rc;
    if (!d_depth || !d_iters || !d_hit) return fail(RM_E_BAD_ARG, "device output pointers are required");
    std::lock_guard<std::mutex> lk(g_mu);     // the enqueue touches the shared workspace (queues, tile order, pass events)
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = check_stream(stream))) return rc;
    rm::KernelArgs a;
    int tile_h = 0, grid = 0;
    if ((rc = make_args(d, (float*)d_depth, (int32_t*)d_iters, (uint8_t*)d_hit, nullptr, nullptr, nullptr,
                        (unsigned long long*)(d_stats ? d_stats : g.stats.p), &a, &tile_h, &grid)))
        return rc;
    return launch(d, a, tile_h, grid, stream ? (hipStream_t)stream : g.stream);
}

int rm_read_stats(const void* d_stats, void* stream, RmStats* out)
{
    int rc = check_ready();
    if (rc) return rc;
    if (!out) return fail(RM_E_BAD_ARG, "out is NULL");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = check_stream(stream))) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : g.stream;
    unsigned long long w[rm::kStatsWords];
    HIP_TRY(hipMemcpyAsync(w, d_stats ? d_stats : g.stats.p, kStatsBlockBytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if ((rc = check_pipeline_error(w))) return rc;
    decode_stats(w, out);
    return RM_OK;
}

int rm_bench_device(const RmFrameDesc* d, void* d_depth, void* d_iters, void* d_hit, RmStats* stats, RmTiming* timing)
{
    int rc = check_ready();
    if (rc) return rc;
    if ((rc = check_desc(d))) return rc;
    if (!d_depth || !d_iters || !d_hit || !timing) return fail(RM_E_BAD_ARG, "device outputs and timing are required");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    rm::KernelArgs a;
    int tile_h = 0, grid = 0;
    if ((rc = make_args(d, (float*)d_depth, (int32_t*)d_iters, (uint8_t*)d_hit, nullptr, nullptr, nullptr,
                        (unsigned long long*)g.stats.p, &a, &tile_h, &grid)))
        return rc;
    if ((rc = timed_launches(d, a, tile_h, grid, timing))) return rc;
    if (stats) {
        unsigned long long w[rm::kStatsWords];
        HIP_TRY(hipMemcpy(w, g.stats.p, kStatsBlockBytes, hipMemcpyDeviceToHost));
        if ((rc = check_pipeline_error(w))) return rc;
        decode_stats(w, stats);
    }
    return RM_OK;
}

int rm_render_batch_outputs(const RmFrameDesc* shape, int32_t nframes, const double* cams, const RmMarchConfig* configs,
                            const RmOutputs* o, RmStats* stats, float* ms_total)
{
    if (!o) return fail(RM_E_BAD_ARG, "outputs record is NULL");
    if (o->t_raw || o->final_sdf || o->block_var) return fail(RM_E_BAD_ARG, "batches return depth, iters, hit and evals only");
    float* const depth = o->depth;
    int32_t* const iters = o->iters;
    uint8_t* const hit = o->hit;
    int32_t* const evals = o->evals;
    int rc = check_ready();
    if (rc) return rc;
    if ((rc = check_desc(shape))) return rc;
    if (nframes < 0 || (nframes > 0 && (!cams || !depth || !iters || !hit))) return fail(RM_E_BAD_ARG, "bad batch arguments");
    if (shape->tile_order_mode != 0) return fail(RM_E_BAD_ARG, "tile_order_mode is not supported for batches");
    if (nframes == 0) return RM_OK;
    const int full = configs ? (configs[0].full ? 1 : 0) : (shape->march.full ? 1 : 0);
    for (int f = 0; configs && f < nframes; ++f)
        if ((configs[f].full ? 1 : 0) != full) return fail(RM_E_BAD_ARG, "all frames of a batch must share march.full");
    const size_t n = (size_t)shape->rows * (size_t)shape->width;          // elements per frame
    const size_t total = n * (size_t)nframes;
    if (total > (size_t)1 << 31) return fail(RM_E_BAD_DIMS, "batch too large");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = ensure_events())) return rc;
    if ((rc = g.depth.ensure(total * 4 + 16)) || (rc = g.iters.ensure(total * 4 + 16)) || (rc = g.hit.ensure(total + 16)) ||
        (rc = g.bstats.ensure(sizeof(rm::FrameParams) * (size_t)nframes)))
        return rc;
    if (evals && (rc = g.evals.ensure(total * 4 + 16))) return rc;
    // the frame table: one camera + march configuration per frame
    std::vector<rm::FrameParams> fp((size_t)nframes);
    for (int f = 0; f < nframes; ++f) {
        for (int i = 0; i < 14; ++i) fp[f].cam.v[i] = cams[(size_t)f * 14 + i];
        fp[f].cfg = to_cfg(configs ? configs[f] : shape->march);
    }
    HIP_TRY(hipMemcpyAsync(g.bstats.p, fp.data(), sizeof(rm::FrameParams) * (size_t)nframes, hipMemcpyHostToDevice, g.stream));
    rm::KernelArgs a;
    int tile_h = 0, grid = 0;
    RmFrameDesc d = *shape;
    if (configs) {
        // the launch-wide scheduling policy (long-ray suspension) looks at the largest budget of the batch
        d.march = configs[0];
        for (int f = 1; f < nframes; ++f)
            if (configs[f].max_iterations > d.march.max_iterations) d.march = configs[f];
    }
    if ((rc = make_args(&d, (float*)g.depth.p, (int32_t*)g.iters.p, (uint8_t*)g.hit.p, nullptr, nullptr, nullptr,
                        (unsigned long long*)g.stats.p, &a, &tile_h, &grid)))
        return rc;
    a.frames = (const rm::FrameParams*)g.bstats.p;
    a.nframes = nframes;
    a.full = full;
    a.evals = evals ? (int32_t*)g.evals.p : nullptr;
    if (d.grid_waves <= 0) {      // the batch is one big launch: size the persistent grid for all its tiles
        const long long ntiles = (long long)a.tiles_per_frame * nframes;
        int per_cu = 0;
        if (rm::scene(d.scene_id)->occupancy(d.strategy_id, tile_h, a.interleave, 1, &per_cu) != hipSuccess || per_cu <= 0) per_cu = 2;
        per_cu = std::min(per_cu, 3);
        grid = (int)std::max<long long>(1, std::min<long long>((long long)g.prop.multiProcessorCount * per_cu,
                                                                (ntiles + rm::kWavesPerWG - 1) / rm::kWavesPerWG));
    }
    HIP_TRY(hipEventRecord(g.ev[0], g.stream));
    if ((rc = launch(&d, a, tile_h, grid, g.stream))) return rc;
    HIP_TRY(hipEventRecord(g.ev[1], g.stream));
    HIP_TRY(hipMemcpyAsync(depth, g.depth.p, total * 4, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(iters, g.iters.p, total * 4, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(hit, g.hit.p, total, hipMemcpyDeviceToHost, g.stream));
    if (evals) HIP_TRY(hipMemcpyAsync(evals, g.evals.p, total * 4, hipMemcpyDeviceToHost, g.stream));
    unsigned long long whead[rm::kStatsHead];
    HIP_TRY(hipMemcpyAsync(whead, g.stats.p, sizeof whead, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    if (g.last_was_pipeline && (rc = check_pipeline_error(whead))) return rc;      // a wait bound hit: the maps are incomplete
    if (ms_total) HIP_TRY(hipEventElapsedTime(ms_total, g.ev[0], g.ev[1]));
    if (stats) {
        // per-frame integer reduce of the returned maps (the in-kernel block aggregates the whole batch)
        for (int f = 0; f < nframes; ++f) {
            RmStats& st = stats[f];
            memset(&st, 0, sizeof st);
            st.total_rays = n;
            st.iter_min = n ? 0x7fffffff : 0;
            const int32_t* it = iters + (size_t)f * n;
            const uint8_t* h = hit + (size_t)f * n;
            for (size_t i = 0; i < n; ++i) {
                st.hit_count += h[i];
                st.sum_iters += (uint64_t)it[i];
                st.iter_max = std::max(st.iter_max, it[i]);
                st.iter_min = std::min(st.iter_min, it[i]);
                st.iter_hist[std::min<int32_t>(std::max<int32_t>(it[i], 0), RM_HIST_BINS - 1)] += 1;
            }
            if (evals)
                for (size_t i = 0; i < n; ++i) st.sum_evals += (uint64_t)evals[(size_t)f * n + i];
        }
    }
    return RM_OK;
}

int32_t rm_shard_rows(int32_t height, int32_t world_size)
{
    if (height <= 0 || world_size <= 0) return 0;
    const int nblk = (height + 3) / 4;
    return ((nblk + world_size - 1) / world_size) * 4;
}

int rm_comm_unique_id(uint8_t id[RM_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == RM_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) return fail(RM_E_BAD_ARG, "id is NULL");
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    RCCL_TRY(R.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return RM_OK;
}

int rm_comm_init(const uint8_t id[RM_COMM_ID_BYTES], int32_t world_size, int32_t rank)
{
    int rc = check_ready();
    if (rc) return rc;
    if (!id || world_size < 1 || rank < 0 || rank >= world_size) return fail(RM_E_BAD_ARG, "bad communicator arguments");
    std::lock_guard<std::mutex> lk(g_mu);
    if ((rc = rccl_load())) return rc;
    if (R.comm) return fail(RM_E_BAD_ARG, "a communicator exists already; call rm_comm_destroy() first");
    HIP_TRY(hipSetDevice(g.device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    RCCL_TRY(R.CommInitRank(&R.comm, world_size, u, rank));
    R.world = world_size;
    R.rank = rank;
    return RM_OK;
}

int rm_comm_destroy(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!R.comm) return RM_OK;
    if (g.ready) {
        (void)hipSetDevice(g.device);
        (void)hipDeviceSynchronize();
    }
    const ncclResult_t r = R.CommDestroy(R.comm);
    R.comm = nullptr;
    R.world = 0; R.rank = -1;
    for (Buf& b : R.gather) b.release();
    for (Buf& b : R.pad) b.release();
    if (r != ncclSuccess) return fail(RM_E_RCCL, "ncclCommDestroy failed: %s", R.GetErrorString(r));
    return RM_OK;
}

int rm_assemble_frame(int32_t world_size, int32_t height, int32_t width, int32_t rows_per_rank, int32_t cyclic, int32_t elem_bytes,
                      const void* d_gathered, void* d_full, void* stream)
{
    int rc = check_ready();
    if (rc) return rc;
    if (world_size < 1 || height < 0 || width <= 0 || rows_per_rank < 0 || (elem_bytes != 1 && elem_bytes != 4 && elem_bytes != 8) ||
        !d_gathered || !d_full)
        return fail(RM_E_BAD_ARG, "bad assemble arguments");
    if (cyclic && (height % (4 * world_size) != 0 || rows_per_rank != height / world_size))
        return fail(RM_E_BAD_DIMS, "band-cyclic plan needs height %% (4 * world_size) == 0 and rows_per_rank == height / world_size");
    if (!cyclic && (long long)rows_per_rank * world_size < height) return fail(RM_E_BAD_DIMS, "the shards do not cover the frame");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = check_stream(stream))) return rc;
    return assemble(world_size, height, width, rows_per_rank, cyclic ? 1 : 0, elem_bytes, d_gathered, d_full,
                    stream ? (hipStream_t)stream : g.stream);
}

// The plan a shard descriptor must follow on this communicator: rows every rank contributes (`per`), cyclic or not.
static int shard_plan(const RmFrameDesc* d, int* per_out, bool* cyclic_out)
{
    const int N = R.world, H = d->height;
    const bool cyclic = d->band_rows > 0 && d->band_stride > 1;
    int per;
    if (cyclic) {
        if (d->band_rows != 4 || d->band_stride != N || d->band_offset != R.rank || d->row0 != 0 || H % (4 * N) != 0 || d->rows != H / N)
            return fail(RM_E_BAD_DIMS, "band-cyclic shard does not match the communicator (4-row bands, stride = world size %d, "
                                       "offset = rank %d, rows = height / world size)", N, R.rank);
        per = d->rows;
    } else {
        per = rm_shard_rows(H, N);
        if (N == 1) per = H;
        const int r0 = std::min(R.rank * per, H), r1 = std::min((R.rank + 1) * per, H);
        if (d->row0 != r0 || d->rows != r1 - r0)
            return fail(RM_E_BAD_DIMS, "contiguous shard of rank %d must be rows [%d, %d)", R.rank, r0, r1);
    }
    *per_out = per;
    *cyclic_out = cyclic;
    return RM_OK;
}

int rm_gather_frame(const RmFrameDesc* d, const void* d_depth, const void* d_iters, const void* d_hit, void* d_full_depth,
                    void* d_full_iters, void* d_full_hit, void* stream)
{
    int rc = check_ready();
    if (rc) return rc;
    if ((rc = check_desc(d))) return rc;
    if (!d_depth || !d_iters || !d_hit || !d_full_depth || !d_full_iters || !d_full_hit) return fail(RM_E_BAD_ARG, "NULL buffer");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!R.comm) return fail(RM_E_RCCL, "no communicator: call rm_comm_init() first");
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = check_stream(stream))) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : g.stream;
    const int N = R.world, H = d->height, W = d->width;
    int per = 0;                                   // rows every rank contributes to the collective
    bool cyclic = false;
    if ((rc = shard_plan(d, &per, &cyclic))) return rc;
    const void* src[3] = { d_depth, d_iters, d_hit };
    void* dst[3] = { d_full_depth, d_full_iters, d_full_hit };
    const int eb[3] = { 4, 4, 1 };
    const void* send[3];
    for (int k = 0; k < 3; ++k) {
        const size_t shard_bytes = (size_t)per * W * eb[k];
        if ((rc = R.gather[k].ensure(shard_bytes * (size_t)N + 16))) return rc;
        send[k] = src[k];
        if (d->rows < per) {                        // short last shard of a contiguous plan: pad the send buffer
            if ((rc = R.pad[k].ensure(shard_bytes + 16))) return rc;
            HIP_TRY(hipMemsetAsync(R.pad[k].p, 0, shard_bytes, s));
            if (d->rows > 0) HIP_TRY(hipMemcpyAsync(R.pad[k].p, src[k], (size_t)d->rows * W * eb[k], hipMemcpyDeviceToDevice, s));
            send[k] = R.pad[k].p;
        }
    }
    // the frame's only exchange: three all-gathers in one group (direct xGMI links between the GPUs of a node)
    RCCL_TRY(R.GroupStart());
    for (int k = 0; k < 3; ++k) {
        const ncclResult_t r = R.AllGather(send[k], R.gather[k].p, (size_t)per * W * eb[k], ncclUint8, R.comm, s);
        if (r != ncclSuccess) {
            (void)R.GroupEnd();
            return fail(RM_E_RCCL, "ncclAllGather failed: %s", R.GetErrorString(r));
        }
    }
    RCCL_TRY(R.GroupEnd());
    for (int k = 0; k < 3; ++k)
        if ((rc = assemble(N, H, W, per, cyclic ? 1 : 0, eb[k], R.gather[k].p, dst[k], s))) return rc;
    return RM_OK;
}

int rm_gather_frame_root(const RmFrameDesc* d, const void* d_depth, const void* d_iters, const void* d_hit, void* d_full_depth,
                         void* d_full_iters, void* d_full_hit, int32_t root, void* stream)
{
    int rc = check_ready();
    if (rc) return rc;
    if ((rc = check_desc(d))) return rc;
    if (!d_depth || !d_iters || !d_hit) return fail(RM_E_BAD_ARG, "NULL shard buffer");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!R.comm) return fail(RM_E_RCCL, "no communicator: call rm_comm_init() first");
    if (root < 0 || root >= R.world) return fail(RM_E_BAD_ARG, "root %d outside the communicator of %d", root, R.world);
    const bool is_root = R.rank == root;
    if (is_root && (!d_full_depth || !d_full_iters || !d_full_hit)) return fail(RM_E_BAD_ARG, "the root needs the three full-frame buffers");
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = check_stream(stream))) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : g.stream;
    const int N = R.world, H = d->height, W = d->width;
    int per = 0;
    bool cyclic = false;
    if ((rc = shard_plan(d, &per, &cyclic))) return rc;
    const void* src[3] = { d_depth, d_iters, d_hit };
    void* dst[3] = { d_full_depth, d_full_iters, d_full_hit };
    const int eb[3] = { 4, 4, 1 };
    auto rows_of = [&](int r) { return cyclic ? per : std::max(0, std::min((r + 1) * per, H) - std::min(r * per, H)); };
    // Where rank r's rows land on the root: contiguous plan -> straight into the image (its rows are one block there: no
    // second pass over the frame); band-cyclic plan -> slot r of a rank-major landing buffer, placed by ONE pass of
    // assemble_rows_kernel afterwards (receiving every 4-row band into place would be H / (4 N) x 3 x (N - 1)
    // point-to-point operations per frame: 2835 at 7680x4320 on 8 ranks).
    if (is_root && cyclic)
        for (int k = 0; k < 3; ++k)
            if ((rc = R.gather[k].ensure((size_t)per * W * eb[k] * (size_t)N + 16))) return rc;
    auto slot = [&](int k, int r) -> char* {
        return cyclic ? (char*)R.gather[k].p + (size_t)r * per * W * eb[k] : (char*)dst[k] + (size_t)std::min(r * per, H) * W * eb[k];
    };
    RCCL_TRY(R.GroupStart());
    ncclResult_t err = ncclSuccess;
    for (int k = 0; k < 3 && err == ncclSuccess; ++k) {
        if (!is_root) {
            if (d->rows > 0) err = R.Send(src[k], (size_t)d->rows * W * eb[k], ncclUint8, root, R.comm, s);
        } else {
            for (int r = 0; r < N && err == ncclSuccess; ++r)
                if (r != root && rows_of(r) > 0) err = R.Recv(slot(k, r), (size_t)rows_of(r) * W * eb[k], ncclUint8, r, R.comm, s);
        }
    }
    if (err != ncclSuccess) {
        (void)R.GroupEnd();
        return fail(RM_E_RCCL, "ncclSend / ncclRecv failed: %s", R.GetErrorString(err));
    }
    RCCL_TRY(R.GroupEnd());
    if (is_root) {
        for (int k = 0; k < 3; ++k)
            if (d->rows > 0 && slot(k, root) != (const char*)src[k])
                HIP_TRY(hipMemcpyAsync(slot(k, root), src[k], (size_t)d->rows * W * eb[k], hipMemcpyDeviceToDevice, s));
        if (cyclic)
            for (int k = 0; k < 3; ++k)
                if ((rc = assemble(N, H, W, per, 1, eb[k], R.gather[k].p, dst[k], s))) return rc;
    }
    return RM_OK;
}

int rm_set_queue_capacity(int64_t entries)
{
    if (entries < 0) return fail(RM_E_BAD_ARG, "negative queue capacity");
    std::lock_guard<std::mutex> lk(g_mu);
    g_queue_cap = entries == 0 ? kQueueCapMax : std::min<long long>(entries, 1ll << 30);
    return RM_OK;
}

int rm_set_pass_timing(int enable)
{
    int rc = check_ready();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if (enable && !g.pev_ready) {
        for (auto& e : g.pev) HIP_TRY(hipEventCreate(&e));
        g.pev_ready = true;
    }
    g.pass_timing = enable != 0;
    g.pass_count = 0;
    return RM_OK;
}

int rm_get_pass_ms(void* stream, int32_t* npasses, float* ms)
{
    int rc = check_ready();
    if (rc) return rc;
    if (!npasses || !ms) return fail(RM_E_BAD_ARG, "NULL output");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if (!g.pass_timing || !g.pev_ready) return fail(RM_E_BAD_ARG, "pass timing is off (rm_set_pass_timing)");
    if ((rc = check_stream(stream))) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : g.stream;
    HIP_TRY(hipStreamSynchronize(s));
    *npasses = g.pass_count;
    for (int i = 0; i < g.pass_count; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], g.pev[i], g.pev[i + 1]));
    if (g.last_was_pipeline && g.pass_count == 1 && g.last_stats) {
        // one kernel: split its time at the marks its waves left in the stats block (100 MHz device clock)
        unsigned long long w[rm::kStatsHead];
        HIP_TRY(hipMemcpyAsync(w, g.last_stats, sizeof w, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const unsigned long long start = ~w[rm::kWMarkStart], tiles = ~w[rm::kWMarkTiles];
        if (w[rm::kWMarkStart] != 0 && w[rm::kWMarkTiles] != 0 && tiles >= start && w[rm::kWMarkFresh] >= tiles &&
            w[rm::kWMarkProd] >= w[rm::kWMarkFresh]) {
            const float total = ms[0];
            const float t_tiles = (float)((double)(tiles - start) * 1e-5);
            const float fresh = (float)((double)(w[rm::kWMarkFresh] - start) * 1e-5);
            const float prod = (float)((double)(w[rm::kWMarkProd] - start) * 1e-5);
            ms[0] = t_tiles; ms[1] = fresh - t_tiles; ms[2] = prod - fresh; ms[3] = std::max(0.f, total - prod);
            *npasses = 4;
            // times of the last push into / pop out of queue 1 since launch, for tools (rm_last_queue_marks)
            g.last_push_ms = w[rm::kWMarkPush] >= start ? (float)((double)(w[rm::kWMarkPush] - start) * 1e-5) : 0.f;
            g.last_pop_ms = w[rm::kWMarkPop] >= start ? (float)((double)(w[rm::kWMarkPop] - start) * 1e-5) : 0.f;
            g.long_marks[0] = w[rm::kWLongPushMin] ? (float)((double)(~w[rm::kWLongPushMin]) * 1e-5) : 0.f;
            g.long_marks[1] = (float)((double)w[rm::kWLongPushMax] * 1e-5);
            g.long_marks[2] = w[rm::kWLongTeamMin] ? (float)((double)(~w[rm::kWLongTeamMin]) * 1e-5) : 0.f;
            g.long_marks[3] = (float)((double)w[rm::kWLongTeamMax] * 1e-5);
        }
    }
    return RM_OK;
}

int rm_last_queue_marks(float* last_push_ms, float* last_pop_ms)
{
    if (!last_push_ms || !last_pop_ms) return fail(RM_E_BAD_ARG, "NULL output");
    std::lock_guard<std::mutex> lk(g_mu);
    *last_push_ms = g.last_push_ms;
    *last_pop_ms = g.last_pop_ms;
    return RM_OK;
}

int rm_long_ray_marks(float ms[4])
{
    if (!ms) return fail(RM_E_BAD_ARG, "NULL output");
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < 4; ++i) ms[i] = g.long_marks[i];
    return RM_OK;
}

int rm_render_batch(const RmFrameDesc* shape, int32_t nframes, const double* cams, const RmMarchConfig* configs,
                    float* depth, int32_t* iters, uint8_t* hit, RmStats* stats, float* ms_total)
{
    RmOutputs o;
    memset(&o, 0, sizeof o);
    o.depth = depth; o.iters = iters; o.hit = hit;
    return rm_render_batch_outputs(shape, nframes, cams, configs, &o, stats, ms_total);
}

int rm_alloc_frame(int32_t width, int32_t rows, void** d_depth, void** d_iters, void** d_hit)
{
    int rc = check_ready();
    if (rc) return rc;
    if (width <= 0 || rows <= 0 || !d_depth || !d_iters || !d_hit) return fail(RM_E_BAD_ARG, "bad arguments");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    const size_t n = (size_t)width * rows;
    HIP_TRY(hipMalloc(d_depth, n * 4));
    HIP_TRY(hipMalloc(d_iters, n * 4));
    HIP_TRY(hipMalloc(d_hit, n));
    return RM_OK;
}

int rm_free_frame(void* d_depth, void* d_iters, void* d_hit)
{
    int rc = check_ready();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if (d_depth) HIP_TRY(hipFree(d_depth));
    if (d_iters) HIP_TRY(hipFree(d_iters));
    if (d_hit) HIP_TRY(hipFree(d_hit));
    return RM_OK;
}

int rm_copy_frame_to_host(int32_t width, int32_t rows, const void* d_depth, const void* d_iters, const void* d_hit,
                          float* depth, int32_t* iters, uint8_t* hit)
{
    int rc = check_ready();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    const size_t n = (size_t)width * rows;
    HIP_TRY(hipStreamSynchronize(g.stream));
    if (depth) HIP_TRY(hipMemcpy(depth, d_depth, n * 4, hipMemcpyDeviceToHost));
    if (iters) HIP_TRY(hipMemcpy(iters, d_iters, n * 4, hipMemcpyDeviceToHost));
    if (hit) HIP_TRY(hipMemcpy(hit, d_hit, n, hipMemcpyDeviceToHost));
    return RM_OK;
}

int rm_debug_poison_queues(uint32_t word_offset, uint32_t* next_generation)
{
    int rc = check_ready();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(hipStreamSynchronize(g.stream));
    uint32_t next = g.generation + 1u;
    if (next == 0) next = 1;
    const uint32_t word = next + word_offset;
    for (int q = 0; q < rm::kQueues; ++q) {
        if (!g.queue[q].p || g.queue[q].cap == 0) continue;
        HIP_TRY(hipMemsetD32((hipDeviceptr_t)g.queue[q].p, (int)word, g.queue[q].cap / 4));
        g.qkey[q].writer = 3;                 // somebody else's layout: the next single launch must clear what it will poll
        g.qkey[q].used = g.queue[q].cap;
    }
    HIP_TRY(hipDeviceSynchronize());
    if (next_generation) *next_generation = next;
    return RM_OK;
}

int rm_debug_set_trace(int enable)
{
    int rc = check_ready();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    g.tracing = enable != 0;
    return RM_OK;
}

int rm_debug_get_trace(uint32_t* records, int64_t max_records, int64_t* nrecords, uint32_t* start_ticks, uint32_t* detach_ticks,
                       int64_t npix, uint32_t* launch_tick)
{
    int rc = check_ready();
    if (rc) return rc;
    if (!nrecords) return fail(RM_E_BAD_ARG, "nrecords is NULL");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(hipDeviceSynchronize());
    if (!g.trace.p || !g.last_stats) return fail(RM_E_BAD_ARG, "no traced single-launch frame (rm_debug_set_trace, then render)");
    uint32_t head[8];
    HIP_TRY(hipMemcpy(head, g.trace.p, sizeof head, hipMemcpyDeviceToHost));
    const int64_t n = std::min<int64_t>(head[0], 1 << 20);
    *nrecords = n;
    if (records && max_records > 0)
        HIP_TRY(hipMemcpy(records, (const uint32_t*)g.trace.p + 8, (size_t)std::min<int64_t>(n, max_records) * 32, hipMemcpyDeviceToHost));
    const size_t np = std::min<size_t>((size_t)std::max<int64_t>(npix, 0), g.trace_pix);
    if (start_ticks && np) HIP_TRY(hipMemcpy(start_ticks, g.trace_start.p, np * 4, hipMemcpyDeviceToHost));
    if (detach_ticks && np) HIP_TRY(hipMemcpy(detach_ticks, g.trace_detach.p, np * 4, hipMemcpyDeviceToHost));
    if (launch_tick) {
        unsigned long long w[rm::kStatsHead];
        HIP_TRY(hipMemcpy(w, g.last_stats, sizeof w, hipMemcpyDeviceToHost));
        *launch_tick = (uint32_t)(~w[rm::kWMarkStart]);
    }
    return RM_OK;
}

int rm_runtime_info(RmRuntimeInfo* out)
{
    if (!out) return fail(RM_E_BAD_ARG, "out is NULL");
    memset(out, 0, sizeof *out);
    // the address our own calls resolve to (a GOT entry in position-independent code), then the object that holds it
    Dl_info info;
    memset(&info, 0, sizeof info);
    void* const fn = reinterpret_cast<void*>(&hipStreamCreateWithFlags);
    if (dladdr(fn, &info) && info.dli_fname) snprintf(out->hip_runtime_path, sizeof out->hip_runtime_path, "%s", info.dli_fname);
    const HipCopies c = hip_runtimes_loaded();
    out->hip_runtimes_loaded = c.count;
    snprintf(out->other_runtime_path, sizeof out->other_runtime_path, "%s", c.other);
    int v = 0;
    if (hipRuntimeGetVersion(&v) == hipSuccess) out->hip_runtime_version = v;
    v = 0;
    if (hipDriverGetVersion(&v) == hipSuccess) out->hip_driver_version = v;
    return RM_OK;
}

int rm_stream_create(void** stream)
{
    int rc = check_ready();
    if (rc) return rc;
    if (!stream) return fail(RM_E_BAD_ARG, "stream is NULL");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    g_own_streams.push_back((void*)s);
    *stream = (void*)s;
    return RM_OK;
}

int rm_stream_synchronize(void* stream)
{
    int rc = check_ready();
    if (rc) return rc;
    hipStream_t s;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        HIP_TRY(hipSetDevice(g.device));
        if ((rc = check_stream(stream))) return rc;
        s = stream ? (hipStream_t)stream : g.stream;
    }
    HIP_TRY(hipStreamSynchronize(s));      // outside the lock: other threads may enqueue while this one waits
    return RM_OK;
}

int rm_stream_destroy(void* stream)
{
    int rc = check_ready();
    if (rc) return rc;
    if (!stream) return RM_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = check_stream(stream))) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (g.frame_ev_valid && g.frame_stream == (hipStream_t)stream) g.frame_stream = nullptr;   // its last frame has completed
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    g_own_streams.erase(std::remove(g_own_streams.begin(), g_own_streams.end(), stream), g_own_streams.end());
    g_seen_streams.erase(std::remove(g_seen_streams.begin(), g_seen_streams.end(), stream), g_seen_streams.end());
    return RM_OK;
}

int rm_bench_store_path(int32_t width, int32_t rows, void* d_depth, void* d_iters, void* d_hit, RmTiming* t)
{
    int rc = check_ready();
    if (rc) return rc;
    if (width <= 0 || rows <= 0 || !d_depth || !d_iters || !d_hit || !t) return fail(RM_E_BAD_ARG, "bad arguments");
    if (t->repeats < 1 || t->repeats > RM_MAX_TIMED || t->warmup < 0) return fail(RM_E_BAD_ARG, "bad timing request");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipSetDevice(g.device));
    if ((rc = ensure_events())) return rc;
    const int tiles_x = (width + 63) / 64, tiles_y = (rows + 3) / 4;
    const int ntiles = tiles_x * tiles_y;
    const int grid = std::min(ntiles, g.prop.multiProcessorCount * 32);
    for (int i = 0; i < t->warmup + t->repeats; ++i) {
        const int k = i - t->warmup;
        if (k >= 0) HIP_TRY(hipEventRecord(g.ev[2 * k], g.stream));
        hipLaunchKernelGGL(store_path_kernel, dim3(grid), dim3(64), 0, g.stream, (float*)d_depth, (int32_t*)d_iters,
                           (uint8_t*)d_hit, width, rows, tiles_x, ntiles);
        HIP_TRY(hipGetLastError());
        if (k >= 0) HIP_TRY(hipEventRecord(g.ev[2 * k + 1], g.stream));
    }
    HIP_TRY(hipStreamSynchronize(g.stream));
    for (int i = 0; i < t->repeats; ++i) HIP_TRY(hipEventElapsedTime(&t->ms_each[i], g.ev[2 * i], g.ev[2 * i + 1]));
    summarise(t);
    return RM_OK;
}

}  // extern "C"
