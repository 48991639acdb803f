// rm_scene_tu.hip -- one translation unit per scene (compile with -DRM_SCENE_ID=<0..19>).
// Instantiates render / march_rays kernels for every strategy and the sdf_eval kernel
// of that scene, and exports their launchers through rm::scene_launchers_<id>().
#include "rm_kernels.h"
#include "rm_pipeline.h"

#ifndef RM_SCENE_ID
#error "compile with -DRM_SCENE_ID=<scene id>"
#endif

namespace rm {

template <int ID> struct SceneById;
#define RM_X(id, S) template <> struct SceneById<id> { using type = S; };
RM_SCENE_LIST(RM_X)
#undef RM_X
using SceneT = SceneById<RM_SCENE_ID>::type;

constexpr bool kIter = SceneIterative<SceneT>::value;

template <class Strat, int TH>
static hipError_t launch_render(const KernelArgs& a, int grid, hipStream_t s)
{
    const bool il = kIter && a.interleave;
    if (a.frames) {
        if (il) hipLaunchKernelGGL((render_kernel<SceneT, Strat, TH, kIter, true>), dim3(grid), dim3(64 * kWavesPerWG), 0, s, a);
        else hipLaunchKernelGGL((render_kernel<SceneT, Strat, TH, false, true>), dim3(grid), dim3(64 * kWavesPerWG), 0, s, a);
    } else {
        if (il) hipLaunchKernelGGL((render_kernel<SceneT, Strat, TH, kIter, false>), dim3(grid), dim3(64 * kWavesPerWG), 0, s, a);
        else hipLaunchKernelGGL((render_kernel<SceneT, Strat, TH, false, false>), dim3(grid), dim3(64 * kWavesPerWG), 0, s, a);
    }
    return hipGetLastError();
}

template <class Strat>
static hipError_t launch_resume(int level, const KernelArgs& a, int grid, hipStream_t s)
{
    const bool il = kIter && a.interleave;
    if (a.frames) {
        if (il) hipLaunchKernelGGL((resume_kernel<SceneT, Strat, kIter, true>), dim3(grid), dim3(64 * kWavesPerWG), 0, s, a, level);
        else hipLaunchKernelGGL((resume_kernel<SceneT, Strat, false, true>), dim3(grid), dim3(64 * kWavesPerWG), 0, s, a, level);
    } else {
        if (il) hipLaunchKernelGGL((resume_kernel<SceneT, Strat, kIter, false>), dim3(grid), dim3(64 * kWavesPerWG), 0, s, a, level);
        else hipLaunchKernelGGL((resume_kernel<SceneT, Strat, false, false>), dim3(grid), dim3(64 * kWavesPerWG), 0, s, a, level);
    }
    return hipGetLastError();
}

template <class Strat, int TH>
static hipError_t occ_render(int interleave, int batch, int* blocks)
{
    const bool il = kIter && interleave;
    if (batch) {
        if (il) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, render_kernel<SceneT, Strat, TH, kIter, true>, 64 * kWavesPerWG, 0);
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, render_kernel<SceneT, Strat, TH, false, true>, 64 * kWavesPerWG, 0);
    }
    if (il) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, render_kernel<SceneT, Strat, TH, kIter, false>, 64 * kWavesPerWG, 0);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, render_kernel<SceneT, Strat, TH, false, false>, 64 * kWavesPerWG, 0);
}

// One-row tiles (TILE_H = 1) are built for the scenes with a team form only: there a frame ends with its longest
// ray, and with 64x4 tiles that ray may sit in its tile's pixel pool for milliseconds behind lanes that older rays hold.
template <class Strat, int TH>
static hipError_t launch_pipeline_th(const KernelArgs& a, int grid, hipStream_t s)
{
    const bool il = kIter && a.interleave;
    const dim3 g(grid), b(64 * kPipeWaves);
    if (a.frames) {
        if (il) hipLaunchKernelGGL((pipeline_kernel<SceneT, Strat, TH, kIter, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((pipeline_kernel<SceneT, Strat, TH, false, true>), g, b, 0, s, a);
    } else {
        if (il) hipLaunchKernelGGL((pipeline_kernel<SceneT, Strat, TH, kIter, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((pipeline_kernel<SceneT, Strat, TH, false, false>), g, b, 0, s, a);
    }
    return hipGetLastError();
}

template <class Strat>
static hipError_t launch_pipeline(const KernelArgs& a, int grid, hipStream_t s)
{
    if constexpr (kIter) {
        if (a.tile_h == 1) return launch_pipeline_th<Strat, 1>(a, grid, s);
    }
    if (a.tile_h != 4) return hipErrorInvalidValue;
    return launch_pipeline_th<Strat, 4>(a, grid, s);
}

template <class Strat>
static hipError_t occ_pipeline(int interleave, int batch, int* blocks)
{
    const bool il = kIter && interleave;
    if (batch) {
        if (il) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, pipeline_kernel<SceneT, Strat, 4, kIter, true>, 64 * kPipeWaves, 0);
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, pipeline_kernel<SceneT, Strat, 4, false, true>, 64 * kPipeWaves, 0);
    }
    if (il) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, pipeline_kernel<SceneT, Strat, 4, kIter, false>, 64 * kPipeWaves, 0);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, pipeline_kernel<SceneT, Strat, 4, false, false>, 64 * kPipeWaves, 0);
}

static hipError_t pipeline(int strategy, const KernelArgs& a, int grid, hipStream_t s)
{
    switch (strategy) {
#define RM_X(id, S) \
    case id: return launch_pipeline<S>(a, grid, s);
        RM_STRATEGY_LIST(RM_X)
#undef RM_X
    }
    return hipErrorInvalidValue;
}

static hipError_t occupancy_pipeline(int strategy, int interleave, int batch, int* blocks)
{
    switch (strategy) {
#define RM_X(id, S) \
    case id: return occ_pipeline<S>(interleave, batch, blocks);
        RM_STRATEGY_LIST(RM_X)
#undef RM_X
    }
    return hipErrorInvalidValue;
}

static hipError_t render(int strategy, int tile_h, const KernelArgs& a, int grid, hipStream_t s)
{
    switch (strategy) {
#define RM_X(id, S) \
    case id: return launch_render<S, 4>(a, grid, s);
        RM_STRATEGY_LIST(RM_X)
#undef RM_X
    }
    return hipErrorInvalidValue;
}

static hipError_t resume(int strategy, int level, const KernelArgs& a, int grid, hipStream_t s)
{
    switch (strategy) {
#define RM_X(id, S) \
    case id: return launch_resume<S>(level, a, grid, s);
        RM_STRATEGY_LIST(RM_X)
#undef RM_X
    }
    return hipErrorInvalidValue;
}

template <bool ITER>
static hipError_t resume_team_impl(int strategy, int level, const KernelArgs& a, int grid, hipStream_t s)
{
    if constexpr (ITER) {
        switch (strategy) {
#define RM_X(id, S)                                                                                              \
    case id:                                                                                                     \
        if (a.frames) hipLaunchKernelGGL((resume_team_kernel<SceneT, S, true>), dim3(grid), dim3(64 * kTeam), 0, s, a, level);   \
        else hipLaunchKernelGGL((resume_team_kernel<SceneT, S, false>), dim3(grid), dim3(64 * kTeam), 0, s, a, level);           \
        return hipGetLastError();
            RM_STRATEGY_LIST(RM_X)
#undef RM_X
        }
    }
    return hipErrorInvalidValue;
}

static int entry_bytes(int strategy)
{
    switch (strategy) {
#define RM_X(id, S) \
    case id: return (int)sizeof(QEntry<S>);
        RM_STRATEGY_LIST(RM_X)
#undef RM_X
    }
    return 0;
}

static hipError_t occupancy(int strategy, int tile_h, int interleave, int batch, int* blocks)
{
    switch (strategy) {
#define RM_X(id, S) \
    case id: return occ_render<S, 4>(interleave, batch, blocks);
        RM_STRATEGY_LIST(RM_X)
#undef RM_X
    }
    return hipErrorInvalidValue;
}

static hipError_t sdf_eval(const double* xyz, size_t n, double* out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((sdf_eval_kernel<SceneT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, xyz, n, out);
    return hipGetLastError();
}

static hipError_t march_rays(int strategy, const MarchCfg& cfg, const double* o, const double* d, size_t n,
                             uint8_t* hit, double* t, int32_t* iters, double* fs, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    dim3 grid((unsigned)((n + 63) / 64)), block(64);
    switch (strategy) {
#define RM_X(id, S)                                                                                        \
    case id:                                                                                               \
        hipLaunchKernelGGL((march_rays_kernel<SceneT, S>), grid, block, 0, s, cfg, o, d, n, hit, t, iters, fs); \
        return hipGetLastError();
        RM_STRATEGY_LIST(RM_X)
#undef RM_X
    }
    return hipErrorInvalidValue;
}

template <bool ITER>
static hipError_t march_rays_team_impl(int strategy, const MarchCfg& cfg, const double* o, const double* d, size_t n,
                                       uint8_t* hit, double* t, int32_t* iters, double* fs, unsigned long long* busy, int fillers,
                                       hipStream_t s)
{
    if constexpr (ITER) {
        if (n == 0) return hipSuccess;
        dim3 grid((unsigned)((n + 63) / 64) + (unsigned)(busy ? fillers : 0)), block(64 * kTeam);
        switch (strategy) {
#define RM_X(id, S)                                                                                             \
    case id:                                                                                                    \
        hipLaunchKernelGGL((march_rays_team_kernel<SceneT, S>), grid, block, 0, s, cfg, o, d, n, hit, t, iters, fs, busy); \
        return hipGetLastError();
            RM_STRATEGY_LIST(RM_X)
#undef RM_X
        }
    }
    return hipErrorInvalidValue;
}

#define RM_CAT2(a, b) a##b
#define RM_CAT(a, b) RM_CAT2(a, b)
// a host function (not a const global: hipcc would try to emit that for the device too)
const SceneLaunchers* RM_CAT(scene_launchers_, RM_SCENE_ID)()
{
    static const SceneLaunchers l = { render, resume, kIter ? resume_team_impl<kIter> : nullptr, pipeline, occupancy_pipeline, kIter,
                                      entry_bytes, occupancy, sdf_eval, march_rays, kIter ? march_rays_team_impl<kIter> : nullptr };
    return &l;
}

}  // namespace rm
