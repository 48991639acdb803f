// Host-only check build of the product's kernel headers (scenes, strategy state
// machines, camera, math) -- compiled by g++ for tests ONLY, so the exact source
// the gfx950 kernels are built from can be diffed against the oracle and the
// reference goldens in a container without a GPU.  Never loaded by the product.
#include <stddef.h>
#include <stdint.h>
#include "../../raymarch_algo_compare_amd/csrc/rm_camera.h"
#include "../../raymarch_algo_compare_amd/csrc/rm_scenes.h"
#include "../../raymarch_algo_compare_amd/csrc/rm_strategies.h"

using namespace rm;

template <class Scene, class Strat>
static void render_t(const MarchCfg& cfg, const CameraParams& cam, int W, int H, int row0, int rows,
                     uint8_t* hit, double* t, int32_t* iters, double* fs)
{
    for (int r = 0; r < rows; ++r)
        for (int px = 0; px < W; ++px) {
            vec3 o, d;
            camera_ray(cam, W, H, px, row0 + r, o, d);
            Result res = march_one<Scene, Strat>(o, d, cfg);
            size_t k = (size_t)r * W + px;
            hit[k] = (uint8_t)res.hit; t[k] = res.t; iters[k] = res.iters; fs[k] = res.final_sdf;
        }
}

template <class Scene>
static int render_s(int strategy, const MarchCfg& cfg, const CameraParams& cam, int W, int H, int row0, int rows,
                    uint8_t* hit, double* t, int32_t* iters, double* fs)
{
    switch (strategy) {
#define RM_X(id, S) case id: render_t<Scene, S>(cfg, cam, W, H, row0, rows, hit, t, iters, fs); return 0;
        RM_STRATEGY_LIST(RM_X)
#undef RM_X
    }
    return -2;
}

extern "C" {

int rmh_render(int scene, int strategy, int max_iterations, double hit_threshold, double max_distance,
               double lipschitz, int full, const double* cam14, int W, int H, int row0, int rows,
               uint8_t* hit, double* t, int32_t* iters, double* fs, const double* prm18)
{
    MarchCfg cfg;
    cfg.hit_threshold = hit_threshold; cfg.max_distance = max_distance; cfg.lipschitz = lipschitz;
    cfg.max_iterations = max_iterations; cfg.full = full;
    cfg.prm = default_strat_params();
    if (prm18) {   // RmStrategyParams order, the four counts as doubles
        const double* prm16 = prm18;
        StratParams& p = cfg.prm;
        p.omega = prm16[0]; p.ar_omega_min = prm16[1]; p.ar_omega_max = prm16[2]; p.ar_smoothing = prm16[3];
        p.ar_growth_rate = prm16[4]; p.ar_decay_rate = prm16[5]; p.beta = prm16[6]; p.overstep_min_step = prm16[7];
        p.hybrid_stuck_step_ratio = prm16[8]; p.hybrid_min_step = prm16[9]; p.margin = prm16[10]; p.ar_omega_init = prm16[11];
        p.overstep_bisection_steps = (int32_t)prm16[12]; p.hybrid_stuck_threshold = (int32_t)prm16[13];
        p.segment_bisection_steps = (int32_t)prm16[14]; p.revaa_bisection_steps = (int32_t)prm16[15];
        p.step_scale = prm18[16]; p.dense_min_step = prm18[17];
    }
    CameraParams cam;
    for (int i = 0; i < 14; ++i) cam.v[i] = cam14[i];
    switch (scene) {
#define RM_X(id, S) case id: return render_s<S>(strategy, cfg, cam, W, H, row0, rows, hit, t, iters, fs);
        RM_SCENE_LIST(RM_X)
#undef RM_X
    }
    return -1;
}

int rmh_sdf_eval(int scene, const double* xyz, size_t n, double* out)
{
    switch (scene) {
#define RM_X(id, S) case id: for (size_t i = 0; i < n; ++i) out[i] = S::sdf(v3(xyz[3*i], xyz[3*i+1], xyz[3*i+2])); return 0;
        RM_SCENE_LIST(RM_X)
#undef RM_X
    }
    return -1;
}

}
