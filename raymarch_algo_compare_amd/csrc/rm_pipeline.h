// rm_pipeline.h -- the whole frame in ONE launch: producers, parked-ray consumers and wavefront teams
// running side by side.
//
// render_kernel / resume_kernel / resume_team_kernel (rm_kernels.h) finish a frame in three launches:
// every ray that crosses a trip threshold waits for the kernel boundary, so the dependent chain of the
// frame's longest ray (512 evaluations) only starts its fast form after the slow passes are over.
// Here the three roles share one persistent grid and two queues in HBM:
//
//   producer workgroups (4 independent waves)           team workgroups (3 waves carrying the same 64 rays)
//     fresh tiles --march--> finished: LDS-staged tile flush
//          |  still marching at suspend_after trips
//          v
//        queue 0 --(any producer wave with idle lanes)--> march on, results straight to the maps
//                      |  still marching at suspend_after2 trips
//                      v
//                    queue 1 ------------------------------> march to the end, three waves per 64 rays
//                                                            (and queue 0 entries while queue 1 is empty)
//
// A ray's chain therefore moves to the next form the moment it crosses a threshold.  The march state
// between two SDF evaluations is the strategy record, so every hand-over is bit-exact (same contract
// as resume_kernel).
//
// DEFAULT with teams ("detach", KernelArgs.q0_detach): queue 0 is not used.  At suspend_after trips a ray is only
// struck from its tile -- the tile is flushed without it and its slot is free again -- and marches on in its lane,
// writing its own result, until suspend_after2 trips hand it to queue 1.  The queue-0 stage measured 4-15 x slower
// (every producer wave polling one queue head) and is kept as a selectable, tested mode.
//
// LATE TEAMS (RmFrameDesc.late_teams, off by default): the grid may be larger than what is resident at once.  Workgroups
// [late_team_first, gridDim.x) are teams the dispatcher starts when a producer workgroup has left; producer workgroups
// [team_wgs, team_wgs + early_exit_wgs) leave early -- stop taking tiles, hand their struck rays to queue 1 at once -- when
// queue 1 holds more rays than the teams have taken (one exit_backlog per conversion under way).  A producer never waits
// for a team in detach mode, so no co-residency is assumed; a late team that finds nothing simply ends.
//
// Queue protocol (cdna_hip_programming.md Guideline 16, recipe R1; no order of dispatch, placement or
// co-residency is assumed):
//   push   lane 0 reserves slots with one agent-scope atomic add -> the entries are written with
//          WRITE-THROUGH stores (`global_store_dwordx4 sc1`, what a relaxed agent-scope atomic store compiles to, 16 bytes
//          wide: no release fence, so the frame's own dirty output lines are never written back early; every entry as ONE
//          contiguous run stored by a few lanes, see q_push) ->
//          s_waitcnt vmcnt(0) -> every pushing lane stores its entry's `ready` word = this launch's generation
//          tag (relaxed agent-scope store);
//   pop    lane 0 claims [taken, min(taken + n, reserved)) with a compare-and-swap -> every lane polls
//          the `ready` word of its entry (relaxed agent-scope loads; the writer is between its reserve
//          and its flag store, so the wait is bounded) -> ONE agent-scope acquire fence -> plain loads.
// Counters are only ever touched by agent-scope atomics (performed at the memory side, coherent).
// Termination (every wave reaches its exit): a producer wave leaves when it has no fresh work, queue 0
// is empty and either every producer wave has reported the end of its fresh work (nothing can be
// parked any more) or it has waited out a bounded number of polls (then the waves still marching consume
// what they park themselves -- results do not depend on who marches a ray); a team leaves when every
// producer wave has exited (nothing can be pushed any more) and both queues are handed out.
#pragma once

#include "rm_kernels.h"

namespace rm {

// The counters every wave of the launch hammers live in a control block of their own, ONE COUNTER PER 128-BYTE
// LINE (KernelArgs.ctl, zeroed before every launch): device-scope atomics are performed at the memory side and
// serialise per line (~12 ns each), so the tile counter, the two queue heads and tails and the progress counts
// must not share a line with one another or with the stats block the finished rays add to.
constexpr int kCtlStride = 16;    // u64 words per counter (128 bytes)
constexpr int kCTile = 0;         // next fresh tile
constexpr int kCReserved = 1;     // + q: entries reserved in queue q
constexpr int kCTaken = 3;        // + q: entries handed out of queue q
constexpr int kCFreshDone = 5;    // producer waves that have finished (or never had) fresh work
constexpr int kCProdExited = 6;   // producer waves that have left their loop
constexpr int kCLeft = 7;         // early-exit producer workgroups that have decided to leave (late teams take their place)
constexpr int kCLateStarted = 8;  // late team workgroups that have started
constexpr int kCTeamDone = 9;     // team workgroups that have left (KernelArgs.keep_busy: finished producers stay until all have)
constexpr int kCtlWords = 10 * kCtlStride;
__device__ __forceinline__ unsigned long long* ctl(const KernelArgs& a, int k) { return a.ctl + k * kCtlStride; }
// words of stats block 0 written once per wave (see the layout comment in rm_kernels.h)
constexpr int kWMarkStart = 13;   // ~min s_memrealtime at kernel entry            } 100 MHz device clock,
constexpr int kWMarkFresh = 14;   // max s_memrealtime when a wave reported fresh-done } for rm_get_pass_ms
constexpr int kWMarkProd = 15;    // max s_memrealtime when a producer wave exited  }
constexpr int kWMarkTiles = 17;   // ~min s_memrealtime when a wave found the tile counter exhausted
constexpr int kWMarkPush = 18;    // max s_memrealtime of a push into queue 1
constexpr int kWMarkPop = 19;     // max s_memrealtime of a pop out of queue 1
// tuning marks of the frame's longest rays (those that end with >= kLongRay iterations), 100 MHz ticks:
constexpr int kWLongPushMax = 20; // latest time since launch at which one of them entered queue 1
constexpr int kWLongTeamMax = 21; // longest time one of them spent between that push and its end
constexpr int kWLongTeamMin = 22; // ~shortest such time
constexpr int kWLongPushMin = 23; // ~earliest push
constexpr int kLongRay = 500;
constexpr int kWError = 16;       // != 0: a wait of the queue protocol ran into its bound (the host reports RM_E_HIP)
// No wait in this kernel is unbounded: a protocol bug or a lost workgroup must end in an error code, never in a
// hung device.  Bounds are far beyond anything a healthy launch reaches.
constexpr int kMaxReadyPolls = 4000000;                      // polls of one entry's `ready` word (~ seconds)
constexpr unsigned long long kMaxTeamWaitTicks = 3000000000ull;   // 30 s of the 100 MHz clock without any work for a team

__device__ __forceinline__ unsigned long long ld_relaxed(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Progress counters: everything this wave pushed before (write-through entry stores, flag stores, slot
// reservations) has drained when the add is issued, so whoever reads the new count finds it in memory.
__device__ __forceinline__ void add_after_drain(unsigned long long* p, unsigned long long v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }

// Entries available in queue q right now (lane-0 view; relaxed loads)
__device__ __forceinline__ long long q_available(const KernelArgs& a, int q)
{
    unsigned long long R = ld_relaxed(ctl(a, kCReserved + q));
    if (R > (unsigned long long)a.queue_cap) R = (unsigned long long)a.queue_cap;
    const unsigned long long T = ld_relaxed(ctl(a, kCTaken + q));
    return (long long)R - (long long)T;
}

// Lane 0 claims up to `want` entries of queue q: returns the count (0: none available) and the first index.
__device__ __forceinline__ int q_claim_lane0(const KernelArgs& a, int q, int want, unsigned int& base)
{
    unsigned long long R = ld_relaxed(ctl(a, kCReserved + q));
    if (R > (unsigned long long)a.queue_cap) R = (unsigned long long)a.queue_cap;
    unsigned long long T = ld_relaxed(ctl(a, kCTaken + q));
    for (int tries = 0; tries < 8 && T < R; ++tries) {
        const unsigned long long nt = (T + (unsigned long long)want < R) ? T + (unsigned long long)want : R;
        if (__hip_atomic_compare_exchange_strong(ctl(a, kCTaken + q), &T, nt, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)) {
            base = (unsigned int)T;
            return (int)(nt - T);
        }
        // T now holds the current value; another wave won the race
    }
    return 0;
}

// Wave-uniform: lanes with `mine` wait until their entry (index base + rank) is published, then the wave
// acquires.  The writer of an entry is between its slot reservation and its flag store: a bounded wait.
// Returns, per lane, whether its entry may be read: false for a lane whose entry never became ready within the bound
// (the error word is set and the lane must drop the entry -- an unpublished entry holds garbage, and its output index
// would steer stores out of bounds).
template <class Strat>
__device__ __forceinline__ bool q_wait_ready(const KernelArgs& a, int q, bool mine, unsigned int idx)
{
    const QEntry<Strat>* const e = (const QEntry<Strat>*)a.queue[q] + idx;
    bool ok = !mine;
    int polls = 0;
    while (!__all(ok)) {
        if (!ok) ok = __hip_atomic_load(&e->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.generation;
        if (!__all(ok)) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > kMaxReadyPolls) {                  // wave-uniform
                if (lane_id() == 0) atomicMax(&a.stats[kWError], 1ull);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

// Wave-uniform push with publication: lanes with `want` append their ray to queue q.  Returns per lane
// whether the ray was parked; a full queue leaves the ray where it is (it is not offered again).
template <class Strat>
__device__ __forceinline__ bool q_push(const KernelArgs& a, int q, bool want, uint32_t gi, const Strat& s, int nev)
{
    const unsigned long long m = __ballot(want);
    if (m == 0) return false;
    unsigned long long base = 0;
    if (lane_id() == 0)
        base = __hip_atomic_fetch_add(ctl(a, kCReserved + q), (unsigned long long)__popcll(m), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
    base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    const unsigned long long idx = base + (unsigned long long)rank_in_mask(m);      // 64-bit: never wraps
    const bool ok = want && idx < (unsigned long long)a.queue_cap;
    static_assert(sizeof(QEntry<Strat>) % 8 == 0 && alignof(QEntry<Strat>) == 8, "entries are copied as 8-byte words");
    constexpr int NW = (int)(sizeof(QEntry<Strat>) / 8);
    constexpr int NP = (NW + 1) / 2;                          // 16-byte pieces of an entry (the last one 8 bytes when NW is odd)
    unsigned long long* const dst = (unsigned long long*)((QEntry<Strat>*)a.queue[q] + (ok ? idx : 0ull));
    unsigned long long w[NW];
    {
        QEntry<Strat> e;
        e.gi = gi;
        e.nev = (uint32_t)nev;
        e.ready = 0u;                                         // never a generation tag: the entry stays unpublished until the flag store below
        e.pad = a.marks ? (uint32_t)(realtime() - ~ld_relaxed(&a.stats[kWMarkStart])) : 0u;      // push time since launch (tuning marks)
        e.s = s;
        __builtin_memcpy(w, &e, sizeof e);
    }
    // ONE RUN OF BYTES PER ENTRY.  A lane that stored its own entry word by word issued NW write-through stores of 8 bytes, and
    // the memory side counts every one of them as a request of its own: 377 bytes of HBM write traffic per 72-byte entry, 34 of
    // the 54 MB a 1080p Mandelbulb frame wrote (tools/traffic_variants.sh).  Instead the entry of every pushing lane travels
    // through scalar registers to lanes 0 .. NP-1, which store 16 bytes each: one contiguous run, one or two requests.
    const int lane = lane_id();
    for (unsigned long long mm = __ballot(ok); mm != 0; mm &= mm - 1) {      // wave-uniform
        const int src = (int)__builtin_ctzll(mm);
        const unsigned int ilo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)idx, src);
        const unsigned int ihi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(idx >> 32), src);
        unsigned int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned int a0 = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)w[2 * p], src);
            const unsigned int a1 = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(w[2 * p] >> 32), src);
            unsigned int a2 = 0, a3 = 0;
            if (2 * p + 1 < NW) {
                a2 = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)w[2 * p + 1], src);
                a3 = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(w[2 * p + 1] >> 32), src);
            }
            if (lane == p) { p0 = a0; p1 = a1; p2 = a2; p3 = a3; }
        }
        char* const d = (char*)((QEntry<Strat>*)a.queue[q] + (((unsigned long long)ihi << 32) | ilo)) + 16 * lane;
        // The result record of the strategy (StratBase::res, entry bytes 48 .. 71) is dead until the ray finishes unless the frame
        // asks for final_sdf (cfg.full: the tail evaluation runs AFTER the record is filled).  Its words are not stored then: piece
        // 3 (res.t, res.final_sdf) never, piece 4 neither where it is only the record's last word (a 72-byte entry).  The reader
        // copies whatever the slot holds into a field nobody reads before finish() overwrites it.
        static_assert(std::is_base_of<StratBase, Strat>::value && __builtin_offsetof(StratBase, res) == 32 && sizeof(Result) == 24 &&
                      sizeof(QEntry<StratBase>) - sizeof(StratBase) == 16, "entry pieces 3 and 4 hold the result record");
        const bool dead = !a.full && (lane == 3 || (NW == 9 && lane == 4));
        if (lane < NP && !dead) {
            if ((NW & 1) && lane == NP - 1) {
                __hip_atomic_store((unsigned long long*)d, ((unsigned long long)p1 << 32) | p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = { p0, p1, p2, p3 };
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d), "v"(v) : "memory");      // write-through, like the atomic stores
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every entry has left the wave
    if (ok) __hip_atomic_store((uint32_t*)(dst + 1), a.generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.marks && q == 1 && lane_id() == 0) atomicMax(&a.stats[kWMarkPush], realtime());
    return ok;
}

// Image coordinates of an output element (for the camera ray of a resumed ray).
__device__ __forceinline__ void element_pixel(const KernelArgs& a, uint32_t gi, uint32_t& frame, int& x, int& gy)
{
    const uint32_t frame_elems = (uint32_t)a.rows * (uint32_t)a.width;
    frame = gi / frame_elems;
    const uint32_t pix = gi - frame * frame_elems;
    const int y = (int)(pix / (uint32_t)a.width);
    x = (int)(pix - (uint32_t)y * (uint32_t)a.width);
    gy = a.band_rows > 0 ? a.row0 + ((y / a.band_rows) * a.band_stride + a.band_offset) * a.band_rows + (y % a.band_rows)
                         : a.row0 + y;
}

// Workgroup shape of the pipeline kernel (1080p Mandelbulb / Standard figures):
//  kPipeWaves 4, kTeamShare false   256-thread workgroups, two per compute unit: a team workgroup (three waves, the
//        fourth exits) shares its CU -- and every one of its SIMDs -- with a producer workgroup: 9.6-10.0 ms.
//  kPipeWaves 8, kTeamsPerWG 1 | 2, kTeamShare false   512-thread workgroups, one per CU, of which `team_grid` carry
//        one or two teams and nothing else: chains at the speed of an idle CU (13 us per evaluation against 17-20
//        next to producers), but 48-96 such CUs cannot absorb the rays that cross the threshold: 11.1-12.0 ms.
//  kPipeWaves 8, kTeamShare true    512-thread workgroups, one per CU; a team workgroup is waves {0,1,2} = the team,
//        wave 4 exits and waves {3,5,6,7} are producers.  Waves i and i + 4 of a workgroup share a SIMD, so the team's
//        critical wave (part 0: length -> divide -> acos -> sincos, the longest dependent chain of a trip) has its
//        SIMD to itself while the compute unit still renders tiles with four waves: 10.2-10.4 ms -- no better, so
//        what holds the longest rays back next to producers is not the issue slot they share.  The marks
//        (rm_get_pass_ms) say what is: when the producers are done the longest ray still has > 400 of its 464 team
//        evaluations ahead -- it sat in queue 1 behind the burst of rays that cross the threshold while the object's
//        tiles are rendered (90 000 at 48 trips, of which 133 run to 512 and nothing tells them apart).
// The first shape is built: it is the simplest and measured best.
constexpr int kPipeWaves = 4;
constexpr bool kTeamShare = false;

// Role-specific LDS of the pipeline kernel (one allocation: a workgroup has exactly one role).
template <int TILE_PIX>
struct PipeProducerLds {
    float depth[kPipeWaves][kSlots][TILE_PIX];
    uint32_t ih[kPipeWaves][kSlots][TILE_PIX];
};
// With kTeamsPerWG = 2 (512-thread workgroups only) a team workgroup carries two teams (waves {0,1,2} and {3,5,6};
// waves 4 and 7 exit): with the hardware's cyclic wave -> SIMD placement each team's critical wave (part 0: length
// -> divide -> acos -> sincos) is then alone on its SIMD.  Two teams are independent of one another, so they cannot
// use s_barrier (it spans the workgroup): a team then synchronises through an arrival counter in its own LDS block.
constexpr int kTeamsPerWG = 1;
static_assert(kTeamsPerWG == 1 || kPipeWaves == 8, "two teams need a 512-thread workgroup");
static_assert(!kTeamShare || (kPipeWaves == 8 && kTeamsPerWG == 1), "a shared team workgroup is 3 team + 1 idle + 4 producer waves");
// producer waves of a team workgroup when it is shared, and their rank among them
constexpr int kSharedProducers = 4;
struct PipeTeamLds {
    TeamXch xch;
    unsigned int hist[kHistBins];
    unsigned int bar;                     // arrivals of this team's waves, monotonic
    unsigned int base, count, queue, done;
    unsigned int ok_lo, ok_hi;            // lanes whose popped entry was published in time (part 0 polls for the team)
};

// Barrier of one team: LDS operations of a wave execute in order, so a wave's exchange writes are in LDS when
// its arrival is counted; `epoch` is the arrival count this barrier waits for (same value in the team's three waves).
__device__ __forceinline__ void team_barrier(PipeTeamLds& L, unsigned int& epoch)
{
    if constexpr (kTeamsPerWG == 1 && !kTeamShare) {   // the team is alone in its workgroup: the hardware barrier (it counts live waves only)
        __syncthreads();
        return;
    }
    epoch += (unsigned int)kTeam;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane_id() == 0) __hip_atomic_fetch_add(&L.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while ((int)(__hip_atomic_load(&L.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - epoch) < 0) {}
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// team_trip (rm_kernels.h) with the team's own barrier
template <class Scene>
__device__ __forceinline__ bool team_trip_lds(typename Scene::Eval& ev, bool go, int part, int lane, PipeTeamLds& L, int& turn,
                                              unsigned int& epoch)
{
    double o0 = 0.0, o1 = 0.0;
    if (go) Scene::trip_part(ev, part, o0, o1);
    double (*buf)[64] = L.xch.v[turn & 1];
    ++turn;
    buf[2 * part][lane] = o0;
    buf[2 * part + 1][lane] = o1;
    team_barrier(L, epoch);
    bool done = true;
    if (go) done = Scene::trip_join(ev, buf[0][lane], buf[1][lane], buf[2][lane], buf[3][lane], buf[4][lane], buf[5][lane]);
    return done;
}

// A finished ray whose result goes straight to the maps (a ray that was parked at least once).
__device__ __forceinline__ void store_direct(const KernelArgs& a, uint32_t gi, const Result& r, int nev, WaveAcc& acc,
                                             unsigned int* s_hist)
{
    const int it = r.iters, h = r.hit;
    a.depth[gi] = h ? (float)r.t : 0.0f;   // types.py:93
    a.iters[gi] = it;
    a.hit[gi] = (uint8_t)h;
    if (a.raw_outputs) store_raw(a, gi, r, nev);
    acc.evals += (unsigned)nev;
    acc.add(it, h);
    atomicAdd(&s_hist[min(it, a.hist_bins - 1)], 1u);
    if (a.tile_cost) {
        const uint32_t frame_elems = (uint32_t)a.rows * (uint32_t)a.width;
        const uint32_t frame = gi / frame_elems;
        const uint32_t pix = gi - frame * frame_elems;
        const uint32_t y = pix / (uint32_t)a.width, x = pix - y * (uint32_t)a.width;
        atomicMax(&a.tile_cost[frame * (uint32_t)a.tiles_per_frame + (y / (uint32_t)a.tile_h) * (uint32_t)a.tiles_x + (x >> 6)], it);
    }
}

// EARLY HAND-OVER (KernelArgs.early_handover, RmFrameDesc.early_handover).  A struck ray whose evaluation just took every trip
// of the scene's inner loop (Scene::eval_trips against KernelArgs.early_trips: all eight fractal iterations of the Mandelbulb
// without a bail-out, or six of them for the strategies that measured better so) is close to the
// surface: it costs a producer eight turns per evaluation -- 15-32 us where a team needs 10 -- and it is the kind of ray that
// runs long.  It goes to the teams at once instead of at suspend_after2; the others wait for that budget as before, so the
// teams are not flooded (the plain budgets 16 / 32 or 24 / 40 were: DESIGN.md section 3).  Scenes without such a measure: never.
template <class Scene, class E>
__device__ __forceinline__ auto early_handover(const KernelArgs& a, const E& ev, int i) -> decltype(Scene::eval_trips(ev), bool())
{
    return a.early_handover > 0 && i >= a.early_handover && Scene::eval_trips(ev) >= a.early_trips;
}
template <class Scene>
__device__ __forceinline__ bool early_handover(const KernelArgs&, const NoEval&, int) { return false; }
template <class Scene, class E>
__device__ __forceinline__ bool early_handover(const KernelArgs&, const E&, long) { return false; }

template <class Scene, class Strat, int TILE_H, bool INTERLEAVE, bool BATCH>
__global__ __launch_bounds__(64 * kPipeWaves, 2) void pipeline_kernel(const KernelArgs a)
{
    static_assert(!INTERLEAVE || SceneIterative<Scene>::value, "INTERLEAVE needs Scene::Eval");
    constexpr bool TEAMS = SceneIterative<Scene>::value;      // the scene has a team form (trip_part / trip_join)
    constexpr int TILE_PIX = kTileW * TILE_H;
    using Entry = QEntry<Strat>;
    using PLds = PipeProducerLds<TILE_PIX>;
    constexpr size_t kTeamBytes = sizeof(PipeTeamLds) * kTeamsPerWG;
    // a workgroup has one role (producer staging and team block overlap) unless team workgroups also render tiles
    constexpr size_t kRoleBytes = kTeamShare ? sizeof(PLds) + kTeamBytes : (sizeof(PLds) > kTeamBytes ? sizeof(PLds) : kTeamBytes);
    constexpr size_t kTeamOffset = kTeamShare ? sizeof(PLds) : 0;
    __shared__ __attribute__((aligned(16))) unsigned char s_role[kRoleBytes];
    __shared__ unsigned int s_hist[kHistBins];
    __shared__ unsigned int s_leave;          // an early-exit producer workgroup has decided to leave

    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) s_hist[b] = 0u;
    if (threadIdx.x == 0) s_leave = 0u;
    rm_load_tables<Scene>();
    __syncthreads();
    if (a.marks && threadIdx.x == 0) atomicMax(&a.stats[kWMarkStart], ~realtime());

    WaveAcc acc;
    // Teams: workgroups [0, team_wgs) from the start of the launch, and the LATE teams [late_team_first, gridDim.x): the
    // grid is larger than what is resident at once, the dispatcher starts a late team when an early-exit producer
    // workgroup has left (below) -- or, at the latest, when producers finish; a team that finds nothing simply ends.
    // A producer never waits for a team (detach mode), so no co-residency is assumed.
    const bool late_team = TEAMS && (int)blockIdx.x >= a.late_team_first;
    const bool team_wg = TEAMS && ((int)blockIdx.x < a.team_wgs || late_team);      // workgroup-uniform
    if (late_team && threadIdx.x == 0) __hip_atomic_fetch_add(ctl(a, kCLateStarted), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // wave roles in a team workgroup: 0..2 = part of team 0 (3..5: team 1), -1 = leave, -2 = producer
    constexpr int kRoleAlone[8] = { 0, 1, 2, kTeamsPerWG > 1 ? 3 : -1, -1, kTeamsPerWG > 1 ? 4 : -1, kTeamsPerWG > 1 ? 5 : -1, -1 };
    constexpr int kRoleShared[8] = { 0, 1, 2, -2, -1, -2, -2, -2 };
    constexpr int kProducerRank[8] = { 0, 0, 0, 0, 0, 1, 2, 3 };     // among the producer waves of a shared team workgroup
    const int wave_role = !team_wg ? -2 : (kTeamShare ? kRoleShared[wave & 7] : kRoleAlone[wave & 7]);
    const bool team_role = team_wg && wave_role >= 0;
    if (team_wg) {
        PipeTeamLds* const T = reinterpret_cast<PipeTeamLds*>(s_role + kTeamOffset);
        for (int t = 0; t < kTeamsPerWG; ++t) {
            for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) T[t].hist[b] = 0u;
            if (threadIdx.x == 0) { T[t].bar = 0u; T[t].base = T[t].count = T[t].queue = T[t].done = 0u; T[t].ok_lo = T[t].ok_hi = 0u; }
        }
        __syncthreads();                     // the last workgroup-wide barrier of a team workgroup
    }

    if constexpr (TEAMS) {
    if (team_wg && wave_role == -1) return;     // the wave that would share a SIMD with a team's critical wave
    if (team_role) {
        // =================================== TEAM ====================================================
        const int team = wave_role / kTeam;
        // a team carries the frame's critical chains: its waves win the issue arbitration on the SIMDs they share
        if (a.team_prio >= 3) __builtin_amdgcn_s_setprio(3);
        else if (a.team_prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (a.team_prio == 1) __builtin_amdgcn_s_setprio(1);
        PipeTeamLds& L = reinterpret_cast<PipeTeamLds*>(s_role + kTeamOffset)[team];
        const int part = wave_role % kTeam;
        unsigned int epoch = 0;               // this team's barrier count (identical in its three waves)
        bool active = false;
        uint32_t my_gi = 0;
        uint32_t my_push = 0;                 // when this ray entered the queue (ticks since launch; tuning marks)
        uint32_t my_pop = 0, nev_pop = 0;     // development trace: when it left the queue, evaluations it had then
        int nev = 0;
        vec3 origin = v3(0.0, 0.0, 0.0), dir = v3(0.0, 0.0, 0.0);
        MarchCfg lane_cfg = a.single.cfg;
        if constexpr (!BATCH) pin_cfg(lane_cfg);
        const MarchCfg& cfg = lane_cfg;
        Strat s;
        typename Scene::Eval ev;
        int turn = 0;
        int since_try = 0;                    // evaluations since the last look at the queues
        unsigned long long idle_since = 0;    // device clock when the team last had work (part 0, lane 0)

        for (;;) {
            const unsigned long long idle = __ballot(!active);
            const int nidle = __popcll(idle);
            // same decision in every wave of the team (identical state)
            const bool look = nidle == 64 || (nidle >= a.refill_min && since_try >= a.team_retry);
            if (look) {
                since_try = 0;
                if (part == 0) {
                    // part 0 claims AND waits for the publication of the claimed entries: one wave decides which lanes
                    // take a ray (a wait that runs into its bound drops the entry), the others follow its mask -- three
                    // waves polling on their own could disagree and leave the team's lock-step
                    unsigned int base = 0, done = 0;
                    int q = 1, cnt = 0;
                    if (lane == 0) {
                        cnt = q_claim_lane0(a, 1, nidle, base);
                        if (cnt == 0 && a.team_steal) { q = 0; cnt = q_claim_lane0(a, 0, nidle, base); }
                        if (cnt == 0 && nidle == 64) {
                            // nothing anywhere: finished once no producer can push any more (one load per idle poll) and
                            // both queues are handed out (looked at only then)
                            const unsigned long long ex = ld_relaxed(ctl(a, kCProdExited));      // read BEFORE the queue counters
                            if (ex >= (unsigned long long)a.producer_waves && q_available(a, 1) <= 0 && q_available(a, 0) <= 0) done = 1;
                            const unsigned long long now = realtime();
                            if (idle_since == 0) idle_since = now;
                            if (!done && now - idle_since > kMaxTeamWaitTicks) { atomicMax(&a.stats[kWError], 2ull); done = 1; }
                        } else if (cnt > 0) {
                            idle_since = 0;
                            if (a.marks && q == 1) atomicMax(&a.stats[kWMarkPop], realtime());
                        }
                    }
                    cnt = __builtin_amdgcn_readfirstlane(cnt);
                    q = __builtin_amdgcn_readfirstlane(q);
                    base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                    unsigned long long okm = 0;
                    if (cnt > 0) {
                        const int rank = rank_in_mask(idle);
                        const bool want = !active && rank < cnt;
                        const bool got = q_wait_ready<Strat>(a, q, want, base + (unsigned int)rank);
                        okm = __ballot(want && got);
                    }
                    if (lane == 0) {
                        L.base = base; L.count = (unsigned int)cnt; L.queue = (unsigned int)q; L.done = done;
                        L.ok_lo = (unsigned int)okm; L.ok_hi = (unsigned int)(okm >> 32);
                    }
                }
                team_barrier(L, epoch);
                const unsigned int base = L.base, cnt = L.count, q = L.queue, done = L.done;
                const unsigned long long okm = ((unsigned long long)L.ok_hi << 32) | (unsigned long long)L.ok_lo;
                team_barrier(L, epoch);       // the mailbox may be rewritten on the next look
                if (done) break;
                if (cnt > 0) {
                    if (part != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // part 0 acquired in q_wait_ready
                    const int rank = rank_in_mask(idle);
                    const bool mine = ((okm >> lane) & 1ull) != 0;
                    if (mine) {
                        const Entry* const e = (const Entry*)a.queue[q] + (base + (unsigned int)rank);
                        my_gi = e->gi;
                        nev = (int)e->nev;
                        s = e->s;
                        my_push = e->pad;
                        if (a.trace) { my_pop = (uint32_t)(realtime() - ~ld_relaxed(&a.stats[kWMarkStart])); nev_pop = (uint32_t)nev; }
                        uint32_t frame; int x, gy;
                        element_pixel(a, my_gi, frame, x, gy);
                        if constexpr (BATCH) {
                            const FrameParams& fp = a.frames[frame];
                            camera_ray(fp.cam, a.width, a.height, x, gy, origin, dir);   // recomputed: same bits
                            lane_cfg = fp.cfg;
                            lane_cfg.full = a.full;
                        } else {
                            camera_ray(a.single.cam, a.width, a.height, x, gy, origin, dir);
                        }
                        active = true;
                    }
                } else if (nidle == 64) {
                    // back off: one lane of one wave per team polls, a few microseconds apart (the pick-up delay is
                    // nothing against the hundreds of evaluations a parked ray has ahead of it)
                    __builtin_amdgcn_s_sleep(127);
                    __builtin_amdgcn_s_sleep(127);
                    continue;
                }
            }
            if (!__any(active)) continue;

            // ---- one whole SDF evaluation for every live ray, trips shared by the team -------------------
            ++since_try;
            bool ready = true;
            if (active) ready = Scene::begin(ev, origin + dir * s.te);   // ray.py:15-17
            while (__any(!ready)) {
                const bool fin = team_trip_lds<Scene>(ev, !ready, part, lane, L, turn, epoch);
                if (!ready) ready = fin;
            }
            const int live = a.trace ? __popcll(__ballot(active)) : 0;
            if (active) {
                ++nev;
                if (s.step(Scene::value(ev), cfg)) {
                    active = false;
                    if (part == 0) {
                        store_direct(a, my_gi, s.res, nev, acc, L.hist);
                        if (a.trace) {
                            const uint32_t k = atomicAdd(&a.trace[0], 1u);
                            if (k < a.trace_cap) {
                                uint32_t* const r = a.trace + 8 + 8 * (size_t)k;
                                r[0] = my_gi; r[1] = (uint32_t)s.res.iters; r[2] = my_push; r[3] = my_pop;
                                r[4] = (uint32_t)(realtime() - ~ld_relaxed(&a.stats[kWMarkStart]));
                                r[5] = nev_pop; r[6] = (uint32_t)nev; r[7] = (uint32_t)blockIdx.x | ((uint32_t)live << 16);
                            }
                        }
                        if (a.marks && s.res.iters >= kLongRay) {        // tuning marks of the frame's longest rays
                            const unsigned long long now = realtime() - ~ld_relaxed(&a.stats[kWMarkStart]);
                            atomicMax(&a.stats[kWLongPushMax], (unsigned long long)my_push);
                            atomicMax(&a.stats[kWLongPushMin], ~(unsigned long long)my_push);
                            atomicMax(&a.stats[kWLongTeamMax], now - my_push);
                            atomicMax(&a.stats[kWLongTeamMin], ~(now - my_push));
                        }
                    }
                }
            }
        }
        unsigned long long* const spart = stats_part(a.stats);
        if (part == 0) acc.flush(spart);
        team_barrier(L, epoch);               // part 0's histogram updates are in LDS
        for (int b = part * 64 + lane; b < kHistBins; b += 64 * kTeam) {
            const unsigned int c = L.hist[b];
            if (c) atomicAdd(&spart[kStatsHead + b], (unsigned long long)c);
        }
        if (part == 0 && lane == 0) __hip_atomic_fetch_add(ctl(a, kCTeamDone), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    }

    // ======================================= PRODUCER ================================================
    PLds& L = *reinterpret_cast<PLds*>(s_role);
    float (*const s_depth)[TILE_PIX] = L.depth[wave];
    uint32_t (*const s_ih)[TILE_PIX] = L.ih[wave];
    const int ntiles = a.tiles_per_frame * a.nframes;
    // this wave's index among the launch's producer waves (its first tile)
    const int team_wgs = TEAMS ? a.team_wgs : 0;
    const int pwave = (kTeamShare && team_wg) ? (int)blockIdx.x * kSharedProducers + kProducerRank[wave & 7]
                                              : (kTeamShare ? team_wgs * kSharedProducers : 0) + ((int)blockIdx.x - team_wgs) * kPipeWaves + wave;
    const int park0 = a.suspend_after;                                   // fresh rays -> queue 0
    const int park1 = (TEAMS && a.team_wgs > 0) ? a.suspend_after2 : 0;  // resumed rays -> queue 1 (teams)
    const bool queues = park0 > 0;
    // detach: a fresh ray that reaches park0 trips is not parked in queue 0 -- its pixel is struck from the tile
    // (the tile flushes without it, the slot is free for the next tile) and it marches on in its lane as a
    // "resumed" ray.  No queue traffic at all below park1; queue 0 stays empty.
    const bool detach = a.q0_detach != 0;
    // this workgroup may leave early to make room for a late team (wave 0 looks at the backlog, the others follow its flag)
    const bool may_leave = TEAMS && detach && park1 > 0 && ((int)blockIdx.x - team_wgs) < a.early_exit_wgs;
    bool leaving = false;                         // wave-uniform: no new tiles, detached rays go to queue 1 at once

    int slot_tile[kSlots], slot_out[kSlots];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) { slot_tile[k] = -1; slot_out[k] = 0; }
    int cur = 0;
    int pool_next = TILE_PIX;
    bool more_tiles = true;
    bool first_tile = true;
    TileGeom cg = { 0, 0, 0, 0, 0, 0, 0 };

    bool active = false;
    bool resumed = false;                         // the ray came out of queue 0: no tile slot, results go straight to the maps
    bool nopark = false;                          // a full queue refused this ray once: it stays where it is
    int my_slot = 0, my_pix = 0;
    uint32_t my_gi = 0;
    int nev = 0;
    vec3 origin = v3(0.0, 0.0, 0.0), dir = v3(0.0, 0.0, 0.0);
    MarchCfg lane_cfg = a.single.cfg;
    if constexpr (!BATCH) pin_cfg(lane_cfg);
    const MarchCfg& cfg = lane_cfg;
    int raw_out = a.raw_outputs;                  // parity outputs requested (kept per lane, see store_raw)
    pin_lane(raw_out);
    Strat s;
    typename EvalOf<Scene, INTERLEAVE>::type ev;
    bool ready = false;
    bool dirty = true;
    bool fresh_reported = false;                  // wave-uniform: this wave has reported the end of its fresh work
    int spins = 0;                                // polls of an empty queue 0 with nothing else to do
    int since_look = 0;                           // turns since the scheduler last ran
    int prio_level = 0;                           // current s_setprio level of this wave (age_prio)

    for (;;) {
        if (dirty) {
        dirty = false;
        since_look = 0;
        // ---- 1. flush every tile whose rays have all finished (and whose pool is handed out) -----
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            const bool pool_done = (k != cur) || pool_next >= TILE_PIX;
            if (slot_tile[k] >= 0 && slot_out[k] == 0 && pool_done) {   // wave-uniform
                wave_lds_fence();
                const TileGeom g = tile_geom<TILE_H>(a, slot_tile[k]);
                const int gx = g.x0 + lane;
                const bool col_ok = lane < g.tw;
                long long bs = 0, bq = 0;
                const bool want_bv = a.block_var != nullptr;      // (kernel-uniform)
                // the three output bases are read from the kernel arguments once per tile, not once per row
                float* o_depth = a.depth; int32_t* o_iters = a.iters; uint8_t* o_hit = a.hit;
                pin_lane(o_depth); pin_lane(o_iters); pin_lane(o_hit);
#pragma unroll
                for (int r = 0; r < TILE_H; ++r) {
                    if (r < g.th && col_ok) {
                        const int li = r * kTileW + lane;
                        const size_t gi = g.out0 + (size_t)(g.y0 + r) * (size_t)a.width + (size_t)gx;
                        const uint32_t ih = s_ih[k][li];
                        if (ih != kSuspended) {
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
                        long long S = bs, Q = bq;
                        S += __shfl_xor(S, 1); Q += __shfl_xor(Q, 1);
                        S += __shfl_xor(S, 2); Q += __shfl_xor(Q, 2);
                        S += __shfl_xor(S, 4); Q += __shfl_xor(Q, 4);
                        const int brow = g.y0 + (r - 3);
                        if (a.block_var && (lane & 7) == 0 && gx + 8 <= a.width && brow + 4 <= a.rows) {
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
                    if (lane == 0) atomicMax(&a.tile_cost[slot_tile[k]], tmax);
                }
                wave_lds_fence();
                slot_tile[k] = -1;
            }
        }

        // ---- 2. lane refill: parked rays (queue 0) and the next unassigned pixels -------------------
        unsigned long long idle = __ballot(!active);
        int nidle = __popcll(idle);
        if (nidle >= a.refill_min || nidle == 64) {
            bool fresh_left = more_tiles || pool_next < TILE_PIX;
            bool got_parked = false;
            // Idle lanes go, in this order, to (1) the pixels of the tile this wave has open -- every pixel of an open
            // tile starts at once, no long ray waits in a pixel pool behind lanes that older rays hold; (2) parked rays
            // (q0_first: before a NEW tile is opened, oldest work first; else only once no tile is left); (3) the next tile.
            if (queues && !detach && (a.q0_first ? pool_next >= TILE_PIX : !fresh_left) && (nidle >= a.q0_refill_min || nidle == 64)) {
                unsigned int base = 0;
                int cnt = 0;
                if (lane == 0) cnt = q_claim_lane0(a, 0, nidle, base);
                cnt = __builtin_amdgcn_readfirstlane(cnt);
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (cnt > 0) {
                    const int rank = rank_in_mask(idle);
                    const bool mine = !active && rank < cnt;
                    const bool got = q_wait_ready<Strat>(a, 0, mine, base + (unsigned int)rank);
                    if (mine && got) {
                        const Entry* const e = (const Entry*)a.queue[0] + (base + (unsigned int)rank);
                        my_gi = e->gi;
                        nev = (int)e->nev;
                        s = e->s;
                        uint32_t frame; int x, gy;
                        element_pixel(a, my_gi, frame, x, gy);
                        if constexpr (BATCH) {
                            const FrameParams& fp = a.frames[frame];
                            camera_ray(fp.cam, a.width, a.height, x, gy, origin, dir);   // recomputed: same bits as before
                            lane_cfg = fp.cfg;
                            lane_cfg.full = a.full;
                        } else {
                            camera_ray(a.single.cam, a.width, a.height, x, gy, origin, dir);
                        }
                        active = true;
                        resumed = true;
                        nopark = false;
                        if constexpr (INTERLEAVE) ready = Scene::begin(ev, origin + dir * s.te);
                    }
                    spins = 0;
                    dirty = true;
                    got_parked = true;
                    idle = __ballot(!active);
                    nidle = __popcll(idle);
                }
            }
            if (nidle > 0 && (!got_parked || nidle >= a.refill_min) && pool_next >= TILE_PIX && more_tiles) {
                int f = -1;
#pragma unroll
                for (int k = kSlots - 1; k >= 0; --k) f = (slot_tile[k] < 0) ? k : f;
                if (f >= 0 && may_leave && !first_tile) {
                    // the decision point of an early-exit workgroup: before it would take another tile.  Leave when queue 1
                    // holds more rays than the teams have taken -- exit_backlog for every conversion that is already under
                    // way (left but not yet replaced by a started late team) and one more.
                    int lv = 0;
                    if (lane == 0) {
                        lv = (int)__hip_atomic_load(&s_leave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (!lv && wave == 0) {
                            const long long left = (long long)ld_relaxed(ctl(a, kCLeft)), started = (long long)ld_relaxed(ctl(a, kCLateStarted));
                            const long long pending = left > started ? left - started : 0;
                            if (left < (long long)a.early_exit_wgs && q_available(a, 1) >= (long long)a.exit_backlog * (1 + pending)) {
                                __hip_atomic_fetch_add(ctl(a, kCLeft), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(&s_leave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                lv = 1;
                            }
                        }
                    }
                    if (__builtin_amdgcn_readfirstlane(lv)) {
                        leaving = true;
                        more_tiles = false;
                        dirty = true;
                        f = -1;
                    }
                }
                if (f >= 0) {
                    int tile = 0;
                    if (first_tile) {
                        tile = pwave;
                        first_tile = false;
                    } else {
                        if (lane == 0) tile = (int)atomicAdd(ctl(a, kCTile), 1ull) + a.producer_waves;
                        tile = __builtin_amdgcn_readfirstlane(tile);
                    }
                    if (tile < ntiles) {
                        if (a.tile_order) tile = __builtin_amdgcn_readfirstlane(a.tile_order[tile]);
                        cur = f;
                        pool_next = 0;
                        dirty = true;
                        cg = tile_geom<TILE_H>(a, tile);
#pragma unroll
                        for (int k = 0; k < kSlots; ++k) {
                            slot_tile[k] = (k == f) ? tile : slot_tile[k];
                            slot_out[k] = (k == f) ? 0 : slot_out[k];
                        }
                    } else {
                        more_tiles = false;
                        if (a.marks && lane == 0) atomicMax(&a.stats[kWMarkTiles], ~realtime());
                    }
                }
            }
            if (nidle > 0 && pool_next < TILE_PIX) {
                bool started = false;
                if (!active) {
                    const int id = pool_next + rank_in_mask(idle);
                    int px, py;
                    if constexpr (TILE_H == 4) {          // block-major: 32 consecutive ids form one 8x4 block
                        const int blk = id >> 5, within = id & 31;
                        px = (blk & 7) * 8 + (within & 7);
                        py = (blk >> 3) * 4 + (within >> 3);
                    } else {                              // one-row tiles: the 64 pixels of the tile start together
                        px = id & (kTileW - 1);
                        py = id >> 6;
                    }
                    if (id < TILE_PIX && px < cg.tw && py < cg.th) {
                        my_slot = cur;
                        my_pix = py * kTileW + px;
                        my_gi = (uint32_t)(cg.out0 + (size_t)(cg.y0 + py) * (size_t)a.width + (size_t)(cg.x0 + px));
                        if constexpr (BATCH) {
                            const FrameParams& fp = a.frames[cg.frame];
                            camera_ray(fp.cam, a.width, a.height, cg.x0 + px, cg.gy0 + py, origin, dir);
                            lane_cfg = fp.cfg;
                            lane_cfg.full = a.full;
                        } else {
                            camera_ray(a.single.cam, a.width, a.height, cg.x0 + px, cg.gy0 + py, origin, dir);
                        }
                        nev = 0;
                        resumed = false;
                        nopark = false;
                        if (s.start(cfg)) {
                            s_depth[cur][my_pix] = s.res.hit ? (float)s.res.t : 0.0f;
                            s_ih[cur][my_pix] = (uint32_t)s.res.iters | ((uint32_t)s.res.hit << 31);
                            if (raw_out) store_raw(a, my_gi, s.res, nev);
                            acc.evals += (unsigned)nev;
                        } else {
                            active = true;
                            started = true;
                            if (a.trace_start) a.trace_start[my_gi] = (uint32_t)realtime();
                            if constexpr (INTERLEAVE) ready = Scene::begin(ev, origin + dir * s.te);   // ray.py:15-17
                        }
                    }
                }
                const int nstarted = __popcll(__ballot(started));
#pragma unroll
                for (int k = 0; k < kSlots; ++k) slot_out[k] += (k == cur) ? nstarted : 0;
                pool_next += nidle;
                dirty = true;
            }
        }

        // ---- 2a. issue priority by age: a wave that carries an old ray (a candidate for the frame's longest chain)
        // wins the arbitration against the wave it shares its SIMD with; throughput-neutral among producers
        if (kAgePriority && a.age_prio > 0) {
            int age = active ? s.i : 0;
            for (int off = 32; off > 0; off >>= 1) age = max(age, __shfl_xor(age, off));
            const int lvl = age / a.age_prio;
            if (lvl != prio_level) {
                prio_level = lvl;
                if (lvl <= 0) __builtin_amdgcn_s_setprio(0);
                else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(2);
            }
        }

        // ---- 2b. the end of this wave's fresh work is reported once (queue 0 cannot grow through it any more)
        if (queues && !fresh_reported && !more_tiles && pool_next >= TILE_PIX) {
            bool in_flight = false;
#pragma unroll
            for (int k = 0; k < kSlots; ++k) in_flight = in_flight || (slot_tile[k] >= 0);
            if (!in_flight) {
                fresh_reported = true;
                if (lane == 0) {
                    add_after_drain(ctl(a, kCFreshDone), 1ull);
                    if (a.marks) atomicMax(&a.stats[kWMarkFresh], realtime());
                }
            }
        }
        }   // dirty

        // ---- 3. exit / idle turn ---------------------------------------------------------------------
        if (!__any(active)) {
            bool in_flight = false;
#pragma unroll
            for (int k = 0; k < kSlots; ++k) in_flight = in_flight || (slot_tile[k] >= 0);
            if (!more_tiles && !in_flight && pool_next >= TILE_PIX) {
                if (!queues || detach) break;              // nothing is ever parked in queue 0: done
                // no fresh work left here: this wave is a consumer of queue 0 until nothing can arrive any more
                int fin = 0;
                if (lane == 0) {
                    const unsigned long long fd = ld_relaxed(ctl(a, kCFreshDone));          // read BEFORE the queue counters
                    const bool empty = q_available(a, 0) <= 0;
                    if (empty && (fd >= (unsigned long long)a.producer_waves || spins > a.max_spins)) fin = 1;
                    else if (empty) fin = 2;               // wait
                }
                fin = __builtin_amdgcn_readfirstlane(fin);
                if (fin == 1) break;
                if (fin == 2) {
                    ++spins;
                    if (spins < 32) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(32);
                }
            }
            dirty = true;
            continue;
        }

        // ---- 4. one SDF evaluation (INTERLEAVE: one trip of it) for every live ray -----------------
        bool fin = false, park = false;
        bool freed = false;                           // a lane without a tile slot became idle
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
                if (resumed) {
                    store_direct(a, my_gi, s.res, nev, acc, s_hist);
                    freed = true;
                } else {
                    fin = true;
                    s_depth[my_slot][my_pix] = s.res.hit ? (float)s.res.t : 0.0f;   // types.py:93
                    s_ih[my_slot][my_pix] = (uint32_t)s.res.iters | ((uint32_t)s.res.hit << 31);
                    if (raw_out) store_raw(a, my_gi, s.res, nev);
                    acc.evals += (unsigned)nev;
                }
            } else if (!nopark && ((!resumed && park0 > 0 && s.i >= park0) || (resumed && park1 > 0 && (s.i >= park1 || leaving || early_handover<Scene>(a, ev, s.i))))) {
                park = true;
            } else if constexpr (INTERLEAVE) {
                ready = Scene::begin(ev, origin + dir * s.te);   // ray.py:15-17
            }
        }
        if (queues && __any(park)) {
            bool p0 = park && !resumed;
            const bool p1 = park && resumed;
            bool parked = false;
            if (detach) {
                if (p0) {                                          // struck from the tile, stays in the lane
                    if (a.trace_detach) a.trace_detach[my_gi] = (uint32_t)realtime();
                    s_ih[my_slot][my_pix] = kSuspended;
                    fin = true;
                    resumed = true;
                    park = false;
                    if constexpr (INTERLEAVE) ready = Scene::begin(ev, origin + dir * s.te);
                }
                p0 = false;
            }
            if (__any(p0)) parked = q_push<Strat>(a, 0, p0, my_gi, s, nev);
            if (__any(p1)) parked = q_push<Strat>(a, 1, p1, my_gi, s, nev) || parked;
            if (parked) {
                if (!resumed) { s_ih[my_slot][my_pix] = kSuspended; fin = true; }
                else freed = true;
                active = false;
            } else if (park) {
                nopark = true;                                     // queue full: march on where it is
                if constexpr (INTERLEAVE) ready = Scene::begin(ev, origin + dir * s.te);
            }
        }
        if (__any(fin)) {
#pragma unroll
            for (int k = 0; k < kSlots; ++k) slot_out[k] -= __popcll(__ballot(fin && my_slot == k));
            dirty = true;
        }
        if (__any(freed)) dirty = true;
        }
        // idle lanes and nothing finished lately: look at queue 0 again every few turns (entries arrive at any time)
        if (queues && !detach && ++since_look >= a.q0_retry) {
            since_look = 0;
            if (__popcll(__ballot(!active)) >= a.q0_refill_min) dirty = true;
        }
        if constexpr (INTERLEAVE) {
            if (active && !ready) ready = Scene::trip(ev);
        }
    }

    // ---- per-wave totals, exit report, histogram flush ---------------------------------------------
    unsigned long long* const part = stats_part(a.stats);
    acc.flush(part);
    if (queues && lane == 0) {
        if (!fresh_reported) add_after_drain(ctl(a, kCFreshDone), 1ull);     // (a wave that never entered 2b)
        add_after_drain(ctl(a, kCProdExited), 1ull);                           // after this wave's last push
        if (a.marks) atomicMax(&a.stats[kWMarkProd], realtime());
    }
    __syncthreads();      // (in a shared team workgroup this also waits for the team waves to have left)
    // the producer waves of this workgroup flush its histogram between them
    const bool shared_wg = kTeamShare && team_wg;
    const int fl_rank = shared_wg ? kProducerRank[wave & 7] * 64 + lane : (int)threadIdx.x;
    const int fl_stride = shared_wg ? kSharedProducers * 64 : (int)blockDim.x;
    for (int b = fl_rank; b < kHistBins; b += fl_stride) {
        const unsigned int c = s_hist[b];
        if (c) atomicAdd(&part[kStatsHead + b], (unsigned long long)c);
    }
    // a finished producer workgroup keeps its compute unit's lanes busy until the teams are through (KEEP BUSY, rm_kernels.h)
    if (TEAMS && a.keep_busy > 0 && !team_wg && a.team_wgs > 0) keep_busy_until(ctl(a, kCTeamDone), (unsigned long long)a.team_wgs, a.keep_busy);
}

}  // namespace rm
