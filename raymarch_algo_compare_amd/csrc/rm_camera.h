// rm_camera.h -- primary-ray generation (core/camera.py:35-41 + core/ray.py:11-13).
//
// The per-frame basis (position, forward, right, up, half_width, half_height: 14
// doubles) is computed on the host by the Python expressions of Camera.__init__
// (raymarch_algo_compare_amd/camera.py) and handed to the kernel by value.
#pragma once

#include "rm_core.h"

namespace rm {

struct CameraParams {
    double v[14];  // pos(3) forward(3) right(3) up(3) half_width half_height
};

RM_HD void camera_ray(const CameraParams& cam, int width, int height, int px, int py, vec3& origin, vec3& dir)
{
    const double* c = cam.v;
    double u = (2.0 * (px + 0.5) / width - 1.0) * c[12];    // camera.py:37
    double w = (1.0 - 2.0 * (py + 0.5) / height) * c[13];   // camera.py:38 (row 0 = top)
    vec3 fwd = v3(c[3], c[4], c[5]), right = v3(c[6], c[7], c[8]), up = v3(c[9], c[10], c[11]);
    vec3 d = (fwd + right * u) + up * w;                     // camera.py:40
    origin = v3(c[0], c[1], c[2]);
    dir = normalized(d);                                     // ray.py:13
}

}  // namespace rm
