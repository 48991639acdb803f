// rm_strategies.h -- the 11 marching strategies of the CPU registry (+ the two that exist only in the reference's
// fragment shader) as resumable state machines.
//
// The reference strategies are Python loops that call sdf() at several places
// (strategies/*.py).  Here each strategy is a small per-lane state record with
//     bool start(cfg)        -> sets `te`, the ray parameter whose SDF value it needs
//     bool step(d, cfg)      -> consumes sdf(ray.at(te)); true when the ray is finished
// so a kernel has exactly ONE inlined SDF evaluation site inside its loop and a
// lane can retire a ray and pick up the next one between any two evaluations
// (rm_kernels.hip refills idle lanes of the 64-wide wavefront this way).
//
// Every arithmetic expression, comparison and counter update is the reference's;
// strategy ids follow the STRATEGIES dict order (strategies/__init__.py:16-28).
// `iterations` accounting quirks (Segment / Overstep-Bisect / Hybrid / RevAA) are
// reproduced on purpose -- SURVEY.md Appendix B.
//
// cfg.full == 0 skips evaluations whose only use is MarchResult.final_sdf (the
// reference's tail `d = sdf(ray.at(t))` on a miss and the midpoint re-evaluations
// of the bisection exits); hit / t / iterations are unaffected.
#pragma once

#include "rm_core.h"

namespace rm {

struct Result {
    double t;
    double final_sdf;
    int32_t iters;
    int32_t hit;
};

enum { PH_TAIL = 0, PH_MAIN = 1, PH_A = 2, PH_B = 3, PH_C = 4 };

struct StratBase {
    double te;  // evaluate the SDF at ray.at(te) next
    double t;
    int32_t i;      // index of the reference's `for i in range(...)`
    int32_t it;     // MarchResult.iterations
    int32_t phase;
    Result res;

    RM_HD bool finish(int hit, double th, double fs)
    {
        res.hit = hit; res.t = th; res.iters = it; res.final_sdf = fs;
        return true;
    }
    // hit whose final_sdf the reference re-evaluates at th
    RM_HD bool finish_hit_reeval(double th, const MarchCfg& c)
    {
        res.hit = 1; res.t = th; res.iters = it; res.final_sdf = 0.0;
        if (c.full) { te = th; phase = PH_TAIL; return false; }
        return true;
    }
    // miss at the current t; the reference evaluates sdf(ray.at(t)) once more for final_sdf
    RM_HD bool finish_miss(const MarchCfg& c)
    {
        res.hit = 0; res.t = t; res.iters = it; res.final_sdf = 0.0;
        if (c.full) { te = t; phase = PH_TAIL; return false; }
        return true;
    }
    // bottom of a `for i in range(max_iterations)` trip (also what `continue` reaches)
    RM_HD bool next_iter(const MarchCfg& c)
    {
        ++i;
        if (i >= c.max_iterations) return finish_miss(c);
        te = t; phase = PH_MAIN;
        return false;
    }
    RM_HD bool begin_loop(const MarchCfg& c)
    {
        t = 0.0; i = 0; it = 0;
        if (c.max_iterations <= 0) return finish_miss(c);
        te = t; phase = PH_MAIN;
        return false;
    }
};

// 0: strategies/standard_sphere.py:24-49
struct StratStandard : StratBase {
    RM_HD bool start(const MarchCfg& c) { return begin_loop(c); }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        it = i + 1;
        if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
        // `t += d`; the shader's standard() scales the step by its stepScale uniform (strategies.glsl:47; the
        // understep oracle of gpu/groundtruth.py:59-61 sets 0.6).  d * 1.0 is d: the default changes no bit.
        t += d * c.prm.step_scale;
        if (t > c.max_distance) return finish_miss(c);
        return next_iter(c);
    }
};

// 1: strategies/relaxed_sphere.py:28-70 (omega: constructor argument, :17)
struct StratRelaxed : StratBase {
    double prev_d, omega;
    RM_HD bool start(const MarchCfg& c) { prev_d = 0.0; omega = c.prm.omega; return begin_loop(c); }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        it = i + 1;
        if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
        if (d < 0.0) {
            t += d; omega = 1.0; prev_d = rm_fabs(d);
            return next_iter(c);
        }
        double stp = d * omega;
        if (i > 0 && (prev_d + d) < prev_d * omega) { stp = d; omega = 1.0; }
        t += stp;
        prev_d = d;
        if (t > c.max_distance) return finish_miss(c);
        return next_iter(c);
    }
};

// 2: strategies/auto_relaxed.py:38-90 (five constructor arguments, :21-23)
struct StratAutoRelaxed : StratBase {
    double prev_d, omega, ema;
    RM_HD bool start(const MarchCfg& c)
    {
        omega = c.prm.ar_omega_init; prev_d = __builtin_inf(); ema = 1.0;
        return begin_loop(c);
    }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        const double omega_min = c.prm.ar_omega_min, omega_max = c.prm.ar_omega_max, smoothing = c.prm.ar_smoothing,
                     growth = c.prm.ar_growth_rate, decay = c.prm.ar_decay_rate;
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        it = i + 1;
        if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
        if (prev_d > 1e-10 && i > 0) {
            double ratio = d / prev_d;
            ema = smoothing * ema + (1.0 - smoothing) * ratio;
            if (ema < 0.8) omega = py_max(omega_min, omega * decay);
            else if (ema > 1.0) omega = py_min(omega_max, omega * growth);
        }
        double stp = d * omega;
        if (d < 0.0) {
            t += d; omega = omega_min; prev_d = rm_fabs(d);
            return next_iter(c);
        }
        t += stp;
        prev_d = d;
        if (t > c.max_distance) return finish_miss(c);
        return next_iter(c);
    }
};

// 3: strategies/slope_auto_relaxed.py:41-115 (beta: constructor argument, :25)
struct StratSlope : StratBase {
    double r, z, m;
    RM_HD bool loop_head(const MarchCfg& c)
    {
        if (i >= c.max_iterations) return finish(0, t, r);   // :112-115, no tail evaluation
        it = i + 1;
        if (rm_fabs(r) < c.hit_threshold) return finish(1, t, r);
        if (t > c.max_distance) return finish(0, t, r);
        te = t + z; phase = PH_A;
        return false;
    }
    RM_HD bool start(const MarchCfg&)
    {
        t = 0.0; i = 0; it = 0;
        te = t; phase = PH_MAIN;   // pre-loop evaluation r = sdf(ray.at(0)) (:46-47)
        return false;
    }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        const double beta = c.prm.beta;
        if (phase == PH_MAIN) {
            r = d; z = r; m = -1.0;
            return loop_head(c);
        }
        double T = te, R = d;
        if (z <= r + rm_fabs(R)) {
            double denom = T - t;
            double M = (denom > 1e-12) ? (R - r) / denom : -1.0;
            m = (1.0 - beta) * m + beta * M;
            t = T;
            r = R;
        } else {
            m = -1.0;
        }
        double denom = 1.0 - m;
        if (denom < 1e-6) denom = 1e-6;
        z = (2.0 * r) / denom;
        if (z < 0.0) z = r;
        ++i;
        return loop_head(c);
    }
};

// 4: strategies/enhanced_sphere.py:32-86
struct StratEnhanced : StratBase {
    double prev_t, prev_d;
    RM_HD bool start(const MarchCfg& c)
    {
        prev_t = 0.0; prev_d = __builtin_inf();
        return begin_loop(c);
    }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        it = i + 1;
        if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
        double stp = d;
        if (i > 0 && prev_d > d && d > 0.0 && (prev_d - d) > 1e-10) {
            double dt = t - prev_t;
            double predicted = d * dt / (prev_d - d);
            if (0.0 < predicted && predicted < d * 3.0) stp = predicted;
        }
        if (d < 0.0) {
            t = (prev_t + t) * 0.5;
            prev_d = rm_fabs(d);
            return next_iter(c);
        }
        prev_t = t;
        prev_d = d;
        t += stp;
        if (t > c.max_distance) return finish_miss(c);
        return next_iter(c);
    }
};

// 5: strategies/curvature_auto_relaxed.py:23-89
struct StratCurvature : StratBase {
    double t1, t2, t3, d1, d2, d3;
    int32_t hist;
    RM_HD bool start(const MarchCfg& c)
    {
        t1 = t2 = t3 = 0.0; d1 = d2 = d3 = 0.0; hist = 0;
        return begin_loop(c);
    }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        it = i + 1;
        if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
        t1 = t2; t2 = t3; t3 = t;
        d1 = d2; d2 = d3; d3 = d;
        hist = (hist + 1 < 3) ? hist + 1 : 3;
        double stp = d;
        if (hist == 3) {
            if (rm_fabs(t3 - t2) > 1e-5 && rm_fabs(t2 - t1) > 1e-5) {
                // Python raises ZeroDivisionError for a zero divisor and the `except`
                // keeps step = d (:78-79); the divisors are +-(d1-d2), +-(d1-d3), +-(d2-d3).
                if ((d1 - d2) != 0.0 && (d1 - d3) != 0.0 && (d2 - d1) != 0.0 && (d2 - d3) != 0.0 &&
                    (d3 - d1) != 0.0 && (d3 - d2) != 0.0) {
                    double term1 = t1 * ((0.0 - d2) / (d1 - d2)) * ((0.0 - d3) / (d1 - d3));
                    double term2 = t2 * ((0.0 - d1) / (d2 - d1)) * ((0.0 - d3) / (d2 - d3));
                    double term3 = t3 * ((0.0 - d1) / (d3 - d1)) * ((0.0 - d2) / (d3 - d2));
                    double t_pred = term1 + term2 + term3;
                    double pred_step = t_pred - t;
                    if (d < d2 && 0.0 < pred_step && pred_step < 3.0 * d) stp = pred_step;
                }
            }
        }
        t += stp;
        if (t > c.max_distance) return finish_miss(c);
        return next_iter(c);
    }
};

// 6: strategies/overstep_bisect.py:30-121 (min_step_factor, bisection_steps: constructor arguments, :18)
struct StratOverstepBisect : StratBase {
    double t_near, t_far, t_mid;
    int32_t j;
    RM_HD bool bis_head(const MarchCfg& c)
    {
        t_mid = (t_near + t_far) * 0.5;
        te = t_mid;
        if (j >= c.prm.overstep_bisection_steps) { phase = PH_B; return false; }   // :104-112 final midpoint decides the hit
        it += 1;
        phase = PH_A;
        return false;
    }
    RM_HD bool after_phase1(const MarchCfg& c)
    {
        if (t_far > 0.0) { j = 0; return bis_head(c); }
        return finish_miss(c);
    }
    RM_HD bool start(const MarchCfg& c)
    {
        t = 0.0; i = 0; it = 0; t_near = 0.0; t_far = -1.0; j = 0; t_mid = 0.0;
        if (c.max_iterations - c.prm.overstep_bisection_steps <= 0) return after_phase1(c);
        te = t; phase = PH_MAIN;
        return false;
    }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        if (phase == PH_MAIN) {
            it = i + 1;
            if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
            if (d > 0.0) {
                t_near = t;
                t += py_max(d, c.prm.overstep_min_step);
            } else {
                t_far = t;
                return after_phase1(c);
            }
            if (t > c.max_distance) return finish_miss(c);
            ++i;
            if (i >= c.max_iterations - c.prm.overstep_bisection_steps) return after_phase1(c);
            te = t;
            return false;
        }
        if (phase == PH_A) {
            if (rm_fabs(d) < c.hit_threshold) return finish(1, t_mid, d);
            if (d > 0.0) t_near = t_mid; else t_far = t_mid;
            if ((t_far - t_near) < c.hit_threshold) {
                t_mid = (t_near + t_far) * 0.5;
                return finish_hit_reeval(t_mid, c);
            }
            ++j;
            return bis_head(c);
        }
        // PH_B
        return finish(rm_fabs(d) < c.hit_threshold * 10.0, t_mid, d);
    }
};

// 7: strategies/skipping_spheres.py:25-72 (margin: a literal there, :30; a uniform of the GLSL seam)
struct StratSkipping : StratBase {
    int32_t j, coarse, fine;
    RM_HD bool fine_head(const MarchCfg& c)
    {
        if (j >= fine) return finish_miss(c);
        it += 1;
        te = t; phase = PH_A;
        return false;
    }
    RM_HD bool start(const MarchCfg& c)
    {
        t = 0.0; i = 0; it = 0; j = 0;
        coarse = (c.max_iterations * 2) / 3;
        fine = c.max_iterations - coarse;
        if (coarse <= 0) return fine_head(c);
        te = t; phase = PH_MAIN;
        return false;
    }
    RM_HD bool step(double d_raw, const MarchCfg& c)
    {
        if (phase == PH_TAIL) { res.final_sdf = d_raw; return true; }
        if (phase == PH_MAIN) {
            it = i + 1;
            double d = d_raw - c.prm.margin;
            if (d < c.hit_threshold) return fine_head(c);
            t += d;
            if (t > c.max_distance) return fine_head(c);
            ++i;
            if (i >= coarse) return fine_head(c);
            te = t;
            return false;
        }
        double d = d_raw;
        if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
        if (d < 0.0) {
            t = py_max(0.0, t + d);
            ++j;
            return fine_head(c);
        }
        t += d;
        if (t > c.max_distance) return finish_miss(c);
        ++j;
        return fine_head(c);
    }
};

// 8: strategies/rev_affine.py:37-97
struct StratRevAA : StratBase {
    double d_lo, next_t, a, b, mid;
    int32_t j;
    RM_HD bool bis_head(const MarchCfg& c)
    {
        mid = 0.5 * (a + b);
        if (j >= c.prm.revaa_bisection_steps) return finish_hit_reeval(mid, c);
        te = mid; phase = PH_B;
        return false;
    }
    RM_HD bool start(const MarchCfg& c) { j = 0; return begin_loop(c); }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        if (phase == PH_MAIN) {
            it = i + 1;
            if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
            if (d < 0.0) {
                t = py_max(0.0, t + d);
                return next_iter(c);
            }
            d_lo = d;
            next_t = t + py_max(d, c.hit_threshold);
            te = next_t; phase = PH_A;
            return false;
        }
        if (phase == PH_A) {
            double lo = py_min(d_lo, d), hi = py_max(d_lo, d);
            if (lo <= 0.0 && hi >= 0.0) {
                a = t; b = next_t; j = 0;
                return bis_head(c);
            }
            t = next_t;
            if (t > c.max_distance) return finish_miss(c);
            return next_iter(c);
        }
        // PH_B
        it += 1;
        if (rm_fabs(d) < c.hit_threshold) return finish(1, mid, d);
        if (d > 0.0) a = mid; else b = mid;
        ++j;
        return bis_head(c);
    }
};

// 9: strategies/adaptive_hybrid.py:35-149 (stuck_threshold, stuck_step_ratio, min_step_factor: constructor arguments, :17-19)
struct StratHybrid : StratBase {
    double t_near, t_far, t_mid;
    int32_t mode, small_cnt;   // mode 0 sphere, 1 overstep, 2 bisect
    RM_HD bool start(const MarchCfg& c)
    {
        mode = 0; small_cnt = 0; t_near = 0.0; t_far = -1.0; t_mid = 0.0;
        return begin_loop(c);
    }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        if (phase == PH_MAIN) {
            it = i + 1;
            if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
            if (mode == 0) {
                if (d > 0.0 && d < c.prm.hybrid_stuck_step_ratio * py_max(t, 1.0)) small_cnt += 1; else small_cnt = 0;
                if (small_cnt >= c.prm.hybrid_stuck_threshold) {
                    mode = 1; t_near = t; t_far = -1.0; small_cnt = 0;
                    return next_iter(c);
                }
                if (d < 0.0) {
                    t_far = t; t_near = py_max(0.0, t + d); mode = 2;
                    return next_iter(c);
                }
                t += d;
            } else if (mode == 1) {
                if (d > 0.0) {
                    t_near = t;
                    t += py_max(d, c.prm.hybrid_min_step);
                } else {
                    t_far = t; mode = 2;
                    return next_iter(c);
                }
            } else {
                if (t_far < 0.0) { mode = 0; return next_iter(c); }
                t_mid = (t_near + t_far) * 0.5;
                te = t_mid; phase = PH_A;
                return false;
            }
            if (t > c.max_distance) return finish_miss(c);
            return next_iter(c);
        }
        // PH_A: midpoint evaluation of a bisect trip (:112-137)
        if (rm_fabs(d) < c.hit_threshold) return finish(1, t_mid, d);
        if (d > 0.0) t_near = t_mid; else t_far = t_mid;
        if ((t_far - t_near) < c.hit_threshold) {
            t = (t_near + t_far) * 0.5;
            return finish_hit_reeval(t, c);
        }
        t = t_mid;
        return next_iter(c);
    }
};

// 10: strategies/segment_tracing.py:37-113
struct StratSegment : StratBase {
    double cand, t_lo, t_hi, t_mid;
    int32_t k;
    RM_HD bool bis_head(const MarchCfg& c)
    {
        if (k >= c.prm.segment_bisection_steps) {
            t = (t_lo + t_hi) * 0.5;
            return finish_hit_reeval(t, c);
        }
        it += 1;
        t_mid = (t_lo + t_hi) * 0.5;
        te = t_mid; phase = PH_B;
        return false;
    }
    RM_HD bool start(const MarchCfg& c) { k = 0; cand = 0.0; return begin_loop(c); }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        const double L = c.lipschitz;
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        if (phase == PH_MAIN) {
            it = i + 1;
            if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
            if (d < 0.0) {
                t -= rm_fabs(d) * 0.5;
                return next_iter(c);
            }
            cand = d / L;
            te = t + cand; phase = PH_A;
            return false;
        }
        if (phase == PH_A) {
            it += 1;
            if (rm_fabs(d) < c.hit_threshold) {
                t += cand;
                return finish(1, t, d);
            }
            if (d < 0.0) {
                t_lo = t; t_hi = t + cand; k = 0;
                return bis_head(c);
            }
            double extended = cand + d / L;
            t += extended;
            if (t > c.max_distance) return finish_miss(c);
            return next_iter(c);
        }
        // PH_B
        if (rm_fabs(d) < c.hit_threshold) return finish(1, t_mid, d);
        if (d > 0.0) t_lo = t_mid; else t_hi = t_mid;
        ++k;
        return bis_head(c);
    }
};

// ---- the two strategies that exist only in the reference's fragment shader --------------------------------------
// No Python statement of them exists; these are the shader's control flow on the CPU path's
// arithmetic (binary64, this engine's camera): PARITY UNPINNED -- nothing of the reference can check them here
// (moderngl is absent and the shader computes in fp32).  The oracle restates the same text, so GPU == oracle holds.
// A miss reports the LAST evaluated distance as final_sdf, as the shader does (no tail evaluation).

// 11: gpu/shaders/strategies.glsl:508-541  safe_relaxed (Keinert et al. 2014), uniform `omega`
struct StratSafeRelaxed : StratBase {
    double omega_eff, prev_radius, step_length, last_d;
    RM_HD bool start(const MarchCfg& c)
    {
        omega_eff = c.prm.omega; prev_radius = 0.0; step_length = 0.0; last_d = 0.0;
        t = 0.0; i = 0; it = 0;
        if (c.max_iterations <= 0) return finish(0, t, last_d);
        te = t; phase = PH_MAIN;
        return false;
    }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        it = i + 1;
        last_d = d;
        const double radius = rm_fabs(d);
        // disjoint-sphere test: the spheres at the previous and the current point must cover the step just taken
        const bool sor_fail = (omega_eff > 1.0) && ((radius + prev_radius) < step_length);
        if (sor_fail) {
            step_length -= omega_eff * step_length;       // undo the over-relaxation
            omega_eff = 1.0;
        } else {
            step_length = d * omega_eff;
        }
        prev_radius = radius;
        if (!sor_fail && radius < c.hit_threshold) return finish(1, t, d);
        t += step_length;
        if (t > c.max_distance) return finish(0, t, d);
        if (t < 0.0) t = 0.0;
        ++i;
        if (i >= c.max_iterations) return finish(0, t, d);
        te = t;
        return false;
    }
};

// 12: gpu/shaders/strategies.glsl:559-593  dense_march (calibration oracle), uniforms `stepScale`, `minStep`
struct StratDenseMarch : StratBase {
    double prev_t, prev_d, lo, hi, d_cur;
    int32_t k;
    RM_HD bool start(const MarchCfg& c)
    {
        t = 0.0; i = 0; it = 1; d_cur = 0.0;
        te = 0.0; phase = PH_A;                            // the sample at t = 0
        (void)c;
        return false;
    }
    RM_HD bool advance(const MarchCfg& c)                  // top of `for (int i = 1; i < maxIterations; i++)`
    {
        ++i;
        if (i >= c.max_iterations) return finish(0, t, d_cur);
        it = i + 1;
        prev_t = t; prev_d = d_cur;
        const double sc = d_cur * c.prm.step_scale;
        t += (sc < c.prm.dense_min_step) ? c.prm.dense_min_step : sc;      // max(d * stepScale, minStep)
        if (t > c.max_distance) return finish(0, t, d_cur);
        te = t; phase = PH_MAIN;
        return false;
    }
    RM_HD bool step(double d, const MarchCfg& c)
    {
        if (phase == PH_TAIL) { res.final_sdf = d; return true; }
        if (phase == PH_A) {
            d_cur = d;
            if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
            i = 0;
            return advance(c);
        }
        if (phase == PH_MAIN) {
            d_cur = d;
            if (prev_d > 0.0 && d <= 0.0) {               // entered the surface between prev_t and t: bisect
                lo = prev_t; hi = t; k = 0;
                te = 0.5 * (lo + hi); phase = PH_B;
                return false;
            }
            if (rm_fabs(d) < c.hit_threshold) return finish(1, t, d);
            return advance(c);
        }
        // PH_B: 30 halvings of [lo (outside), hi (inside)]
        if (d > 0.0) lo = te; else hi = te;
        if (++k < 30) { te = 0.5 * (lo + hi); return false; }
        return finish_hit_reeval(0.5 * (lo + hi), c);      // final_sdf = map at the midpoint (one more evaluation)
    }
};

#define RM_NUM_STRATEGIES 11

// X(id, functor) in registry order
#if defined(RM_DEV_STRATEGIES)
// development builds (make DEV=1): kernels for Standard and Enhanced only -- a scene's translation unit compiles in a
// fraction of the time; every other strategy id is refused by the launchers.  Never shipped (the Makefile writes
// librm_hip_dev.so, which nothing loads unless RM_HIP_LIB points at it).
#define RM_STRATEGY_LIST(X) X(0, StratStandard) X(4, StratEnhanced)
#else
#define RM_STRATEGY_LIST(X)                                                                 \
    X(0, StratStandard) X(1, StratRelaxed) X(2, StratAutoRelaxed) X(3, StratSlope)          \
    X(4, StratEnhanced) X(5, StratCurvature) X(6, StratOverstepBisect) X(7, StratSkipping)  \
    X(8, StratRevAA) X(9, StratHybrid) X(10, StratSegment)                                  \
    X(11, StratSafeRelaxed) X(12, StratDenseMarch)
#endif
// ids [0, RM_NUM_STRATEGIES) are the CPU registry; [RM_NUM_STRATEGIES, RM_NUM_STRATEGY_KERNELS) the shader-only two
#define RM_NUM_STRATEGY_KERNELS 13

// March one ray to completion (no lane refill): used by rm_march_rays and the host check.
template <class Scene, class Strat>
RM_HD Result march_one(vec3 o, vec3 dir, const MarchCfg& c)
{
    Strat s;
    bool done = s.start(c);
    while (!done) {
        double d = Scene::sdf(o + dir * s.te);   // ray.py:15-17
        done = s.step(d, c);
    }
    return s.res;
}

}  // namespace rm
