// Micro-benchmark (developer tool): where the time of a Mandelbulb TEAM trip goes (rm_kernels.h "wavefront teams").
// One wavefront alone on a SIMD runs, on near-surface points, n times: the whole single-wave trip, each of the three
// parts, the join, and begin + value + one strategy step; then one team workgroup (three waves, LDS exchange + barrier)
// runs whole trips.  Prints s_memtime ticks (100 MHz? no: shader clock on gfx9; printed next to wall time) per call.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-cgp-select2branch \
//        -I raymarch_algo_compare_amd/csrc tools/ubench/trip_latency.hip -o tools/ubench/trip_latency.exe
#include "rm_kernels.h"
#include <cstdio>
#include <vector>
using namespace rm;
using E = SceneMandelbulb::Eval;

// every kernel: lanes [0, live) carry points, the others are idle (exec-masked by the early return pattern the real kernels have)
template <int WHAT>
__global__ __launch_bounds__(256) void k(const double* __restrict__ pts, int live, int n, double* out, long long* cyc)
{
    __shared__ TeamXch xch;
    rm_load_tables<SceneMandelbulb>();
    const int lane = lane_id();
    const int part = (int)(threadIdx.x >> 6);
    if (lane == 0) cyc[4 + part] = (long long)((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3);
    if (part >= 3) return;                       // a fourth wave only holds the fourth SIMD's slot at dispatch (as in pipeline_kernel)
    const bool on = lane < live;
    vec3 p = v3(pts[3 * lane], pts[3 * lane + 1], pts[3 * lane + 2]);
    E ev;
    SceneMandelbulb::begin(ev, p);
    double acc = 0.0;
    int turn = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (WHAT == 6) {
        // every lane of the three waves takes the same n trips (one workgroup barrier each); only the live lanes compute
        for (int i = 0; i < n; ++i) {
            ev.z = v3(p.x + acc * 1e-300, p.y, p.z);
            ev.r = length_a(ev.z) * (1.0 + acc * 1e-300);
            ev.dr = 1.0; ev.i = 0;
            const bool f = team_trip<SceneMandelbulb>(ev, on, part, lane, xch, turn);
            acc += ev.r + (f ? 1.0 : 0.0);
        }
    } else if (on) {
        for (int i = 0; i < n; ++i) {
            // keep the state on the same near-surface point (a real ray's evaluation restarts from p every time) while
            // making every iteration depend on the previous one
            ev.z = v3(p.x + acc * 1e-300, p.y, p.z);
            ev.r = length_a(ev.z) * (1.0 + acc * 1e-300);
            ev.dr = 1.0; ev.i = 0;
            if constexpr (WHAT == 0) { SceneMandelbulb::trip(ev); acc += ev.r; }
            else if constexpr (WHAT >= 1 && WHAT <= 3) { double a, b; SceneMandelbulb::trip_part(ev, WHAT - 1, a, b); acc += a + b; }
            else if constexpr (WHAT == 4) { SceneMandelbulb::trip_join(ev, 0.3 + acc * 1e-300, 0.4, 0.5, 0.6, 1.1, 1.2); acc += ev.r; }
            else if constexpr (WHAT == 5) { acc += SceneMandelbulb::value(ev); }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

// the team trip with only some of the three parts computing (mask bit p: wave p evaluates its part); everything else
// -- exchange, barrier, join in all three waves -- as in team_trip
__global__ __launch_bounds__(192) void team_mask_kernel(const double* __restrict__ pts, int live, int n, int mask, int do_join, double* out, long long* cyc, int same_part)
{
    __shared__ TeamXch xch;
    rm_load_tables<SceneMandelbulb>();
    const int lane = lane_id();
    const int part = (int)(threadIdx.x >> 6);
    const bool on = lane < live;
    const bool mine = ((mask >> part) & 1) != 0;
    vec3 p = v3(pts[3 * lane], pts[3 * lane + 1], pts[3 * lane + 2]);
    E ev;
    SceneMandelbulb::begin(ev, p);
    double acc = 0.0;
    __syncthreads();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        ev.z = v3(p.x + acc * 1e-300, p.y, p.z);
        ev.r = length_a(ev.z) * (1.0 + acc * 1e-300);
        ev.dr = 1.0; ev.i = 0;
        double o0 = 0.3, o1 = 0.4;
        if (on && mine) SceneMandelbulb::trip_part(ev, same_part >= 0 ? same_part : part, o0, o1);
        double (*buf)[64] = xch.v[i & 1];
        buf[2 * part][lane] = o0;
        buf[2 * part + 1][lane] = o1;
        __syncthreads();
        if (on) {
            if (do_join) SceneMandelbulb::trip_join(ev, buf[0][lane], buf[1][lane], buf[2][lane], buf[3][lane], buf[4][lane], buf[5][lane]);
            else ev.r = buf[0][lane] + buf[1][lane] + buf[2][lane] + buf[3][lane] + buf[4][lane] + buf[5][lane];
        }
        acc += ev.r;
    }
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = r1 - r0;
}

// which SIMD each wave of a workgroup runs on (HW_ID register), and the cost of the team exchange alone
__global__ void where_kernel(unsigned* out)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: wave [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
__global__ __launch_bounds__(256) void exchange_kernel(int n, int nwaves_in_team, double* out, long long* cyc)
{
    __shared__ TeamXch xch;
    const int lane = lane_id();
    const int part = (int)(threadIdx.x >> 6);
    if (part >= nwaves_in_team) return;                                   // the pipeline kernel's fourth wave leaves like this
    double acc = 1.0 + lane;
    __syncthreads();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        double (*buf)[64] = xch.v[i & 1];
        buf[2 * part][lane] = acc;
        buf[2 * part + 1][lane] = acc + 1.0;
        __syncthreads();
        acc = buf[0][lane] + buf[1][lane] + buf[2][lane] + buf[3][lane] + buf[4][lane] + buf[5][lane];
        acc = acc * 1e-3 + 1.0;
    }
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = r1 - r0;
}

template <int WHAT>
void run(const char* name, const double* d_pts, int live, int threads)
{
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 64);
    const int n = 20000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<WHAT>), dim3(1), dim3(threads), 0, 0, d_pts, live, n, out, cyc);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-28s live lanes %2d, %3d threads (SIMDs", name, live, threads);
    for (int w = 0; w < threads / 64; ++w) printf(" %lld", h[4 + w]);
    printf("): %8.1f shader ticks = %7.1f ns per call\n", (double)h[0] / n, (double)h[1] * 10.0 / n);
    hipFree(out); hipFree(cyc);
}

int main()
{
    // points just outside the bulb's surface along +x / diagonal directions (|p| ~ 1.1: evaluations there take all 8 trips)
    std::vector<double> pts(64 * 3);
    for (int i = 0; i < 64; ++i) {
        const double a = 0.1 + 0.045 * i, b = 0.3 + 0.02 * i;
        const double r = 1.02 + 0.001 * i;
        pts[3 * i] = r * cos(a) * cos(b); pts[3 * i + 1] = r * sin(a) * cos(b); pts[3 * i + 2] = r * sin(b);
    }
    double* d_pts; hipMalloc(&d_pts, pts.size() * 8);
    hipMemcpy(d_pts, pts.data(), pts.size() * 8, hipMemcpyHostToDevice);
    for (int live : { 1, 4, 16, 64 }) {
        run<0>("single-wave trip", d_pts, live, 64);
        run<1>("part 0 (acos, sincos)", d_pts, live, 64);
        run<2>("part 1 (atan2, sincos)", d_pts, live, 64);
        run<3>("part 2 (pow 7, pow 8)", d_pts, live, 64);
        run<4>("join (z, length)", d_pts, live, 64);
        run<5>("value (log)", d_pts, live, 64);
        run<6>("team trip (3 waves)", d_pts, live, 192);
    }
    {
        unsigned* d; hipMalloc(&d, 64 * 8 * 4); hipMemset(d, 0xff, 64 * 8 * 4);
        for (int threads : { 192, 256 }) {
            hipLaunchKernelGGL(where_kernel, dim3(64), dim3(threads), 0, 0, d);
            hipDeviceSynchronize();
            std::vector<unsigned> h(64 * 8); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
            printf("workgroups of %d threads: SIMD of each wave (cu:simd), first 12 workgroups\n", threads);
            for (int b = 0; b < 12; ++b) {
                printf("  wg %2d:", b);
                for (int w = 0; w < threads / 64; ++w) printf(" %x:%u", (h[b * 8 + w] >> 8) & 0xf, (h[b * 8 + w] >> 4) & 3);
                printf("\n");
            }
        }
        double* out; long long* cyc; hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 8);
        for (int live : { 1, 16 })
            for (int do_join : { 1, 0 })
                for (int mask : { 0, 1, 2, 4, 3, 5, 6, 7 }) {
                    const int n = 20000;
                    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(team_mask_kernel, dim3(1), dim3(192), 0, 0, d_pts, live, n, mask, do_join, out, cyc, -1);
                    hipDeviceSynchronize();
                    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
                    printf("team trip, live %2d, parts computing mask %d%d%d, join %d: %7.1f ns per trip\n", live, mask & 1, (mask >> 1) & 1, (mask >> 2) & 1, do_join,
                           (double)hc * 10.0 / n);
                }
        printf("no barrier, no exchange: one, two, three waves of a workgroup each run the SAME independent loop (time of wave 0)\n");
        for (int threads : { 64, 128, 192, 256 }) {
            run<6>("team trip (3 waves)", d_pts, 1, threads >= 192 ? threads : 192);
            run<1>("part 0 (acos, sincos)", d_pts, 1, threads);
            run<2>("part 1 (atan2, sincos)", d_pts, 1, threads);
            run<3>("part 2 (pow 7, pow 8)", d_pts, 1, threads);
            run<0>("single-wave trip", d_pts, 1, threads);
        }
        for (int same : { 0, 1, 2 })
            for (int mask : { 1, 3, 7 }) {
                const int n = 20000;
                for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(team_mask_kernel, dim3(1), dim3(192), 0, 0, d_pts, 1, n, mask, 1, out, cyc, same);
                hipDeviceSynchronize();
                long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
                printf("team trip, live  1, EVERY computing wave runs part %d, waves computing mask %d%d%d: %7.1f ns per trip\n", same, mask & 1, (mask >> 1) & 1,
                       (mask >> 2) & 1, (double)hc * 10.0 / n);
            }
        for (int threads : { 192, 256 }) {
            const int n = 20000;
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(exchange_kernel, dim3(1), dim3(threads), 0, 0, n, 3, out, cyc);
            hipDeviceSynchronize();
            long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
            printf("exchange alone (2 writes, barrier, 6 reads), workgroup of %d threads: %.1f ns per trip\n", threads, (double)hc * 10.0 / n);
        }
    }
    hipFree(d_pts);
    return 0;
}
