// bf_local.hip -- the contrast-score evaluation of OptimizerLocal::iteration_step
// (optimizer_sampler.cpp:120-153,192-205) for gfx950.
//
//   L1  k_local_project_count : Event::project (event.h:65-70,164-168) + the point form of the
//       saturating s x s splat (:129-146): one 32-bit atomic per accepted event at its centre pixel.
//   L2  k_local_blur_score    : per 16 x 64 tile, box-sum the points (== the splat), saturate at 255
//       (each `if (< 255) ++` of :141-143 is order independent: min(255, total)), the 8-bit Gaussian
//       (ksize = scale, this build's own stated kernel -- defined in include/bf_accel.h), and
//       the non-zero sum / count of get_event_score (:192-205) as two integer atomics: exact whatever
//       the order.  It also clears the OTHER point plane (consumed by the previous evaluation).
//
// HBM-bound integer work: 8 B read per event + one random 4-byte atomic; 4 B read + 1 B written per pixel.
#include <hip/hip_runtime.h>
#include <limits.h>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {

__global__ __launch_bounds__(kThreads) void k_local_project_count(const uint32_t* __restrict__ xy,
                                                                  const int32_t* __restrict__ t, long long n,
                                                                  LocalGeom g, uint32_t* __restrict__ plane) {
    const long long i0 = ((long long)blockIdx.x * kThreads + threadIdx.x) * kEvPerThread;
    if (i0 >= n) return;
    uint32_t vxy[kEvPerThread];
    int32_t vt[kEvPerThread];
    if (i0 + kEvPerThread <= n) {   // 16-byte loads (the arrays are padded to a multiple of 4 events)
        const uint4 a = *reinterpret_cast<const uint4*>(xy + i0);
        const int4 b = *reinterpret_cast<const int4*>(t + i0);
        vxy[0] = a.x; vxy[1] = a.y; vxy[2] = a.z; vxy[3] = a.w;
        vt[0] = b.x; vt[1] = b.y; vt[2] = b.z; vt[3] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < kEvPerThread; ++k) {
            vxy[k] = i0 + k < n ? xy[i0 + k] : 0u;
            vt[k] = i0 + k < n ? t[i0 + k] : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < kEvPerThread; ++k) {
        if (i0 + k >= n) break;
        // event.h:164-168 with the uniform kx, ky = float(n) / nz computed once on the host
        const float ft = (float)vt[k];
        const double pr_x = pr_from_p(vxy[k] & 0xffffu, g.kx * ft);
        const double pr_y = pr_from_p(vxy[k] >> 16, g.ky * ft);
        int X = trunc_x86(pr_x * (double)g.scale + g.x_shift);   // optimizer_sampler.cpp:130-131
        int Y = trunc_x86(pr_y * (double)g.scale + g.y_shift);
        if ((X >= g.wsx) || (X < 0) || (Y >= g.wsy) || (Y < 0)) continue;   // :133
        X += g.scale / 2;
        Y += g.scale / 2;
        atomicAdd(&plane[(size_t)X * (size_t)g.C + (size_t)Y], 1u);
    }
}

__device__ __forceinline__ int reflect101(int i, int n) {   // cv::BORDER_REFLECT_101
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

template <int HS>
__global__ __launch_bounds__(kThreads) void k_local_blur_score(const uint32_t* __restrict__ plane,
                                                               uint32_t* __restrict__ zero_plane, LocalGeom g,
                                                               unsigned long long* __restrict__ score /* [2] */,
                                                               uint8_t* __restrict__ img_out) {
    constexpr int TR = kTileR, TC = kTileC;
    constexpr int PR = TR + 4 * HS, PC = TC + 4 * HS;   // point tile: halo 2 HS
    constexpr int CR = TR + 2 * HS, CC = TC + 2 * HS;   // count tile: halo HS
    __shared__ uint32_t s_pts[PR * PC];
    __shared__ uint16_t s_cnt[CR * CC];
    __shared__ unsigned long long s_red[2 * (kThreads / 64)];
    const int R = g.R, C = g.C, tid = threadIdx.x;
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * TC;
    for (int idx = tid; idx < PR * PC; idx += kThreads) {
        const int pr = idx / PC, pc = idx - pr * PC;
        const int gr = r0 - 2 * HS + pr, gc = c0 - 2 * HS + pc;
        s_pts[idx] = (gr >= 0 && gr < R && gc >= 0 && gc < C) ? plane[(size_t)gr * C + gc] : 0u;
    }
    __syncthreads();
    for (int idx = tid; idx < CR * CC; idx += kThreads) {
        const int cr = idx / CC, cc = idx - cr * CC;
        uint32_t acc = 0;
#pragma unroll
        for (int da = 0; da <= 2 * HS; ++da)
#pragma unroll
            for (int db = 0; db <= 2 * HS; ++db) acc += s_pts[(cr + da) * PC + (cc + db)];
        s_cnt[idx] = (uint16_t)(acc < 255u ? acc : 255u);   // optimizer_sampler.cpp:141-143
    }
    __syncthreads();
    // binomial taps of one pass (OpenCV's table for sigma <= 0), sum 4 / 16 / 64
    constexpr int kTap[4][7] = {{1, 0, 0, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0, 0}, {1, 4, 6, 4, 1, 0, 0}, {2, 7, 14, 18, 14, 7, 2}};
    constexpr int kNorm[4] = {1, 4, 16, 64};
    constexpr int n2 = kNorm[HS] * kNorm[HS];
    unsigned long long nz_sum = 0, nz_cnt = 0;
#pragma unroll
    for (int k = 0; k < (TR * TC) / kThreads; ++k) {
        const int pidx = tid + k * kThreads;
        const int lr = pidx / TC, lc = pidx - lr * TC;
        const int gr = r0 + lr, gc = c0 + lc;
        if (gr < R && gc < C) {
            int acc = 0;
#pragma unroll
            for (int a = -HS; a <= HS; ++a) {
                const int rr = reflect101(gr + a, R) - (r0 - HS);
                int row = 0;
#pragma unroll
                for (int b = -HS; b <= HS; ++b) row += kTap[HS][b + HS] * (int)s_cnt[rr * CC + (reflect101(gc + b, C) - (c0 - HS))];
                acc += kTap[HS][a + HS] * row;
            }
            const uint32_t v = (uint32_t)((acc + n2 / 2) / n2);
            if (img_out) img_out[(size_t)gr * C + gc] = (uint8_t)v;
            if (v) { nz_sum += v; nz_cnt += 1; }
            zero_plane[(size_t)gr * C + gc] = 0u;
        }
    }
    nz_sum = (unsigned long long)wave_total_dpp((long long)nz_sum);
    nz_cnt = (unsigned long long)wave_total_dpp((long long)nz_cnt);
    if ((tid & 63) == 63) { s_red[2 * (tid >> 6)] = nz_sum; s_red[2 * (tid >> 6) + 1] = nz_cnt; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long a = 0, b = 0;
        for (int w = 0; w < kThreads / 64; ++w) { a += s_red[2 * w]; b += s_red[2 * w + 1]; }
        if (b) { atomicAdd(&score[0], a); atomicAdd(&score[1], b); }
    }
}

// ---------------------------------------------------------------------------------------
// A GRID of OptimizerLocal windows (bf_local_run_tiles): the window-centred constructor (optimizer_sampler.h:31-34) once per
// sensor tile, each with the tile's own events as its cloud, and the WHOLE run() (optimizer_sampler.cpp:4-38: coordinate descent
// on (nx, ny), step halved and reversed whenever the contrast score does not rise) by ONE work-group per window, on chip: the
// tile's events in registers, the point plane and the saturated count image in LDS, every evaluation the arithmetic of
// k_local_project_count + k_local_blur_score above -- Event::project, the saturating s x s splat as a box sum of points, this
// build's 8-bit Gaussian, the non-zero average as an exact integer ratio -- and the step logic of compute_new_nx / compute_new_ny
// (:90-117) on one lane.  No launch and no host round trip per evaluation (a window of 1M events alone costs 81 us per
// evaluation through bf_local_run; a ~1000-event tile here a few microseconds).  Every quantity is an integer or one IEEE
// operation of the reference's own expression: the states come out bit for bit as the oracle's.
// ---------------------------------------------------------------------------------------
struct LocalTileArgs {
    const uint32_t* xy;
    const int32_t* t;
    const uint32_t* tile_start;
    bf_local_state* states;   // one per tile (out)
    int32_t* rcs;             // 0 ran, 1 window guard (optimizer_sampler.cpp:9-13), BF_ERR_NOCONV evaluation cap
    TileGrid g;
    int32_t scale, wsz, guard_res_x, guard_res_y;
    long long max_evaluations;
};

template <int HS>
__global__ __launch_bounds__(kThreads) void k_local_tile_optimizer(LocalTileArgs a) {
    extern __shared__ uint32_t s_loc[];
    const int tid = threadIdx.x, tile = blockIdx.x;
    const int s = a.scale;
    const int ws = s * a.wsz, R = ws + s, P = R * R;   // metric_wsize, scale_img (square window, optimizer_sampler.h:32)
    uint32_t* s_pts = s_loc;                                       // points (the splat's centres)
    uint16_t* s_cnt = reinterpret_cast<uint16_t*>(s_loc + P);      // saturated counts
    __shared__ unsigned long long s_red[2 * (kThreads / 64)];
    __shared__ double s_nxny[2];
    __shared__ int s_go;
    const uint32_t beg = a.tile_start[tile], end = a.tile_start[tile + 1];
    // the window's centre event: the middle of the tile's rows / columns (tile (tr, tc) holds fr_x * rows / res_x == tr), t = 0
    const int tr = tile / a.g.cols, tc = tile - tr * a.g.cols;
    const int x_lo = (tr * a.g.res_x + a.g.rows - 1) / a.g.rows, x_hi = ((tr + 1) * a.g.res_x + a.g.rows - 1) / a.g.rows - 1;
    const int y_lo = (tc * a.g.res_y + a.g.cols - 1) / a.g.cols, y_hi = ((tc + 1) * a.g.res_y + a.g.cols - 1) / a.g.cols - 1;
    const int c_fr_x = (x_lo + x_hi) / 2, c_fr_y = (y_lo + y_hi) / 2;
    // events in registers (a fuller tile streams the rest)
    constexpr int kUR = 8;
    uint32_t rxy[kUR];
    int32_t rt[kUR];
#pragma unroll
    for (int k = 0; k < kUR; ++k) {
        const uint32_t i = beg + (uint32_t)(k * kThreads + tid);
        rxy[k] = i < end ? a.xy[i] : 0u;
        rt[k] = i < end ? a.t[i] : 0;
    }
    const uint32_t stream_beg = beg + (uint32_t)(kUR * kThreads);
    // run(): optimizer_sampler.cpp:4-38 (state on lane 0 of wave 0, published through LDS)
    bf_local_state st;
    st.nx = 0; st.ny = 0; st.last_score = 0; st.dnx = 0.01; st.dny = 0.01; st.evaluations = 0;
    st.dn_th = (127 * 1 * 1000.0) / (double)(10ull * (unsigned long long)s * 100000000ull);
    int rc = 0;
    if ((R < s * a.guard_res_x / 15) && (R < s * a.guard_res_y / 15)) {   // :9-13
        if (tid == 0) { a.states[tile] = st; a.rcs[tile] = 1; }
        return;
    }
    int phase = 0;   // 0: the first evaluation (:16), 1: compute_new_nx, 2: compute_new_ny
    double ex = 0.0, ey = 0.0;   // the (nx, ny) being evaluated
    for (;;) {
        // ---- iteration_step(ex, ey), :120-153 ----
        for (int i = tid; i < P; i += kThreads) s_pts[i] = 0u;
        const float kx = (float)((double)(float)ex / 127.0), ky = (float)((double)(float)ey / 127.0);   // event.h:164-165
        const double cpx = pr_from_p((uint32_t)c_fr_x, kx * 0.0f), cpy = pr_from_p((uint32_t)c_fr_y, ky * 0.0f);   // event_c.project, t = 0
        const double x_shift = -cpx * (double)s + (double)ws / 2.0, y_shift = -cpy * (double)s + (double)ws / 2.0;   // :126-127
        __syncthreads();
        auto one = [&](uint32_t v, int32_t ti) {
            const float ft = (float)ti;
            const double pr_x = pr_from_p(v & 0xffffu, kx * ft), pr_y = pr_from_p(v >> 16, ky * ft);
            int X = trunc_x86(pr_x * (double)s + x_shift), Y = trunc_x86(pr_y * (double)s + y_shift);   // :130-131
            if ((X >= ws) || (X < 0) || (Y >= ws) || (Y < 0)) return;   // :133
            atomicAdd(&s_pts[(X + s / 2) * R + (Y + s / 2)], 1u);
        };
#pragma unroll
        for (int k = 0; k < kUR; ++k)
            if (beg + (uint32_t)(k * kThreads + tid) < end) one(rxy[k], rt[k]);
        for (uint32_t i = stream_beg + tid; i < end; i += kThreads) one(a.xy[i], a.t[i]);
        __syncthreads();
        // the saturating splat (:137-146) == min(255, box sum of the points)
        for (int i = tid; i < P; i += kThreads) {
            const int r = i / R, c = i - r * R;
            uint32_t acc = 0;
#pragma unroll
            for (int da = -HS; da <= HS; ++da)
#pragma unroll
                for (int db = -HS; db <= HS; ++db) {
                    const int rr = r + da, cc = c + db;
                    if (rr >= 0 && rr < R && cc >= 0 && cc < R) acc += s_pts[rr * R + cc];
                }
            s_cnt[i] = (uint16_t)(acc < 255u ? acc : 255u);
        }
        __syncthreads();
        // this build's 8-bit Gaussian (ksize = scale, BORDER_REFLECT_101; k_local_blur_score) and get_event_score (:192-205)
        constexpr int kTap[4][7] = {{1, 0, 0, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0, 0}, {1, 4, 6, 4, 1, 0, 0}, {2, 7, 14, 18, 14, 7, 2}};
        constexpr int kNorm[4] = {1, 4, 16, 64};
        constexpr int n2 = kNorm[HS] * kNorm[HS];
        unsigned long long nz_sum = 0, nz_cnt = 0;
        for (int i = tid; i < P; i += kThreads) {
            const int r = i / R, c = i - r * R;
            int acc = 0;
#pragma unroll
            for (int da = -HS; da <= HS; ++da) {
                const int rr = reflect101(r + da, R);
                int row = 0;
#pragma unroll
                for (int db = -HS; db <= HS; ++db) row += kTap[HS][db + HS] * (int)s_cnt[rr * R + reflect101(c + db, R)];
                acc += kTap[HS][da + HS] * row;
            }
            const uint32_t v = (uint32_t)((acc + n2 / 2) / n2);
            if (v) { nz_sum += v; nz_cnt += 1; }
        }
        nz_sum = (unsigned long long)wave_total_dpp((long long)nz_sum);
        nz_cnt = (unsigned long long)wave_total_dpp((long long)nz_cnt);
        if ((tid & 63) == 63) { s_red[2 * (tid >> 6)] = nz_sum; s_red[2 * (tid >> 6) + 1] = nz_cnt; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long sa = 0, sb = 0;
            for (int w = 0; w < kThreads / 64; ++w) { sa += s_red[2 * w]; sb += s_red[2 * w + 1]; }
            const double score = sb == 0 ? 0.0 : (double)sa / (double)sb;
            st.evaluations += 1;
            int go = 1;
            if (phase == 0) {
                st.last_score = score;   // :16
            } else if (phase == 1) {     // compute_new_nx, :90-102
                const double dscore = score - st.last_score;
                st.last_score = score;
                if (dscore <= 0) st.dnx = -st.dnx / 2.0;
                st.nx = ex;
            } else {                     // compute_new_ny, :105-117
                const double dscore = score - st.last_score;
                st.last_score = score;
                if (dscore <= 0) st.dny = -st.dny / 2.0;
                st.ny = ey;
                if (a.max_evaluations > 0 && st.evaluations >= a.max_evaluations) { rc = BF_ERR_NOCONV; go = 0; }
            }
            // the next evaluation: after the first one and after every ny step the loop condition (:20) decides
            if (go && phase != 1 && !(hypot(st.dnx, st.dny) > st.dn_th)) go = 0;
            if (go) {
                if (phase == 1) { s_nxny[0] = st.nx; s_nxny[1] = st.ny + st.dny; }
                else { s_nxny[0] = st.nx + st.dnx; s_nxny[1] = st.ny; }
            }
            s_go = go;
        }
        __syncthreads();
        if (!s_go) break;
        ex = s_nxny[0]; ey = s_nxny[1];
        phase = phase == 1 ? 2 : 1;
    }
    if (tid == 0) { a.states[tile] = st; a.rcs[tile] = rc; }
}

int launch_local_tile_optimizer(const uint32_t* xy, const int32_t* t, const uint32_t* tile_start, bf_local_state* states, int32_t* rcs,
                                const TileGrid& g, int scale, int wsz, int guard_res_x, int guard_res_y, long long max_evaluations,
                                hipStream_t s) {
    LocalTileArgs a;
    a.xy = xy; a.t = t; a.tile_start = tile_start; a.states = states; a.rcs = rcs; a.g = g;
    a.scale = scale; a.wsz = wsz; a.guard_res_x = guard_res_x; a.guard_res_y = guard_res_y; a.max_evaluations = max_evaluations;
    const int R = scale * wsz + scale;
    const size_t lds = (size_t)R * R * (4 + 2) + 16;
    void (*k)(LocalTileArgs) = scale / 2 == 0 ? k_local_tile_optimizer<0> : (scale / 2 == 1 ? k_local_tile_optimizer<1>
                              : (scale / 2 == 2 ? k_local_tile_optimizer<2> : k_local_tile_optimizer<3>));
    if (scale / 2 > 3) return -1;
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(k)) != hipSuccess) return -2;
    if (lds + at.sharedSizeBytes > 160 * 1024) return -3;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)at.sharedSizeBytes) != hipSuccess) return -2;
    hipLaunchKernelGGL(k, dim3(g.rows * g.cols), dim3(kThreads), lds, s, a);
    return 0;
}

void launch_local_project_count(const uint32_t* xy, const int32_t* t, long long n, const LocalGeom& g, uint32_t* plane,
                                hipStream_t s) {
    if (n <= 0) return;
    const long long per = (long long)kThreads * kEvPerThread;
    hipLaunchKernelGGL(k_local_project_count, dim3((unsigned)((n + per - 1) / per)), dim3(kThreads), 0, s, xy, t, n, g, plane);
}

int launch_local_blur_score(const uint32_t* plane, uint32_t* zero_plane, const LocalGeom& g, unsigned long long* score,
                            uint8_t* img_out, hipStream_t s) {
    const dim3 grid((g.C + kTileC - 1) / kTileC, (g.R + kTileR - 1) / kTileR);
    switch (g.scale / 2) {
        case 0: hipLaunchKernelGGL(k_local_blur_score<0>, grid, dim3(kThreads), 0, s, plane, zero_plane, g, score, img_out); break;
        case 1: hipLaunchKernelGGL(k_local_blur_score<1>, grid, dim3(kThreads), 0, s, plane, zero_plane, g, score, img_out); break;
        case 2: hipLaunchKernelGGL(k_local_blur_score<2>, grid, dim3(kThreads), 0, s, plane, zero_plane, g, score, img_out); break;
        case 3: hipLaunchKernelGGL(k_local_blur_score<3>, grid, dim3(kThreads), 0, s, plane, zero_plane, g, score, img_out); break;
        default: return -1;   // this build states its Gaussian for ksize <= 7 only
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// EventFile::projection_img (event_file.h:460-515): the motion-compensated event image on the full sensor.
// P1 point scatter at pr * scale (current projected positions) or fr * scale; then k_local_blur_score (box sum ==
// the clamped splat: with x < scale (RES - 1) the box never leaves the image, saturation, Gaussian, non-zero sum /
// count); P2 cv::convertScaleAbs with alpha = 127 / nonzero average: saturate_u8(rint((float)v * (float)alpha)).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_proj_count(const uint32_t* __restrict__ xy, const float2* __restrict__ p,
                                                         const uint8_t* __restrict__ noise, long long n, int scale,
                                                         int res_x, int res_y, int show_final,
                                                         uint32_t* __restrict__ plane) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    if (noise && noise[i]) return;                                        // :472
    const uint32_t v = xy[i];
    const uint32_t fx = v & 0xffffu, fy = v >> 16;
    int X, Y;
    if (show_final) {                                                     // :485-488
        X = (int)fx * scale;
        Y = (int)fy * scale;
    } else {                                                              // :482-483
        const float2 q = p[i];
        X = trunc_x86(pr_from_p(fx, q.x) * (double)scale);
        Y = trunc_x86(pr_from_p(fy, q.y) * (double)scale);
    }
    if ((X >= scale * (res_x - 1)) || (X < 0) || (Y >= scale * (res_y - 1)) || (Y < 0)) return;   // :490
    X += scale / 2;
    Y += scale / 2;
    atomicAdd(&plane[(size_t)X * (size_t)(res_y * scale) + (size_t)Y], 1u);
}

__global__ __launch_bounds__(kThreads) void k_proj_scale(uint8_t* __restrict__ img, long long n,
                                                         const unsigned long long* __restrict__ score) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    // EventFile::nonzero_average (event_file.cpp:282-294) and img_scale = 127.0 / it (:512)
    const double avg = score[1] == 0 ? 0.0 : (double)score[0] / (double)score[1];
    const float a = (float)(127.0 / avg);
    const float v = fabsf((float)img[i] * a);
    img[i] = (uint8_t)((v != v) ? 0 : (v >= 255.0f ? 255 : (int)rintf(v)));   // cvRound + saturate_cast<uchar>
}

void launch_proj_count(const uint32_t* xy, const float2* p, const uint8_t* noise, long long n, int scale, int res_x,
                       int res_y, int show_final, uint32_t* plane, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_proj_count, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, xy, p, noise, n,
                       scale, res_x, res_y, show_final, plane);
}

void launch_proj_scale(uint8_t* img, long long n, const unsigned long long* score, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_proj_scale, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, img, n, score);
}

// ---------------------------------------------------------------------------------------
// EventFile::color_time_img (event_file.h:649-747): the colour-coded time image on the full sensor -- hue = mean
// phase of the events' time within the slice, saturation = coherence of that phase, value = 255.
// C1 point scatter of (1, cos, sin) at pr * scale + shift (or fr * scale + shift); the reference adds the f32
// cos / sin in event order, here they are 2^-32 fixed point in 64-bit integer planes, so the image does not depend on
// the order of the events.  C2 gathers the scale x scale box (== the splat :699-705), forms hue / saturation
// (:712-723) and converts HSV -> BGR (the 8-bit convention of cv::cvtColor, stated in include/bf_accel.h).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_color_point(const uint32_t* __restrict__ xy, const int32_t* __restrict__ t,
                                                          const float2* __restrict__ p, const uint8_t* __restrict__ noise,
                                                          long long n, ColorGeom g, uint32_t* __restrict__ cnt,
                                                          unsigned long long* __restrict__ pc,
                                                          unsigned long long* __restrict__ ps) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    if (noise && noise[i]) return;                                        // :679
    const uint32_t v = xy[i];
    const uint32_t fx = v & 0xffffu, fy = v >> 16;
    int X, Y;
    if (g.show_final) {                                                   // :684-687
        X = trunc_x86((double)(fx * (uint32_t)g.scale) + g.x_shift);
        Y = trunc_x86((double)(fy * (uint32_t)g.scale) + g.y_shift);
    } else {                                                              // :681-682
        const float2 q = p[i];
        X = trunc_x86(pr_from_p(fx, q.x) * (double)g.scale + g.x_shift);
        Y = trunc_x86(pr_from_p(fy, q.y) * (double)g.scale + g.y_shift);
    }
    if ((X >= g.mx) || (X < 0) || (Y >= g.my) || (Y < 0)) return;         // :689-692
    // :694 float angle = 2 * 3.14 * (double(e.t - t_min) / double(t_max - t_min))
    const double num = (double)((long long)t[i] - g.t_min);
    const double ratio = g.t_range > 0 ? num / (double)g.t_range : 0.0;
    const float angle = (float)(2 * 3.14 * ratio);
    const size_t at = (size_t)X * (size_t)g.C + (size_t)Y;
    atomicAdd(&cnt[at], 1u);
    atomicAdd(&pc[at], (unsigned long long)llrint(cos((double)angle) * 4294967296.0));   // two's complement sums
    atomicAdd(&ps[at], (unsigned long long)llrint(sin((double)angle) * 4294967296.0));
}

__device__ __forceinline__ uint8_t unit_to_u8(float x) {
    const float v = x * 255.0f;
    return (uint8_t)(v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (int)rintf(v)));
}

// HSV (H in [0, 180), S, V in [0, 255]) -> B, G, R; float32, no contraction.
__device__ __forceinline__ void hsv_to_bgr_u8(int H, int S, int V, uint8_t* bgr) {
    const float s = (float)S * (1.0f / 255.0f), v = (float)V * (1.0f / 255.0f);
    float h = (float)H * (6.0f / 180.0f);
    int sector = (int)floorf(h);
    h -= (float)sector;
    sector = ((sector % 6) + 6) % 6;
    float tab[4];
    tab[0] = v;
    tab[1] = v * (1.0f - s);
    tab[2] = v * (1.0f - s * h);
    tab[3] = v * (1.0f - s * (1.0f - h));
    int ib, ig, ir;
    switch (sector) {
        case 0: ib = 1; ig = 3; ir = 0; break;
        case 1: ib = 1; ig = 0; ir = 2; break;
        case 2: ib = 3; ig = 0; ir = 1; break;
        case 3: ib = 0; ig = 2; ir = 1; break;
        case 4: ib = 0; ig = 1; ir = 3; break;
        default: ib = 2; ig = 1; ir = 0; break;
    }
    float b = tab[0], gg = tab[0], r = tab[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        b = ib == k ? tab[k] : b;
        gg = ig == k ? tab[k] : gg;
        r = ir == k ? tab[k] : r;
    }
    bgr[0] = unit_to_u8(b); bgr[1] = unit_to_u8(gg); bgr[2] = unit_to_u8(r);
}

__global__ __launch_bounds__(kThreads) void k_color_final(const uint32_t* __restrict__ cnt,
                                                          const unsigned long long* __restrict__ pc,
                                                          const unsigned long long* __restrict__ ps, ColorGeom g,
                                                          uint8_t* __restrict__ bgr) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (long long)g.R * g.C) return;
    const int jx = (int)(i / g.C), jy = (int)(i % g.C);
    // a point at (X, Y) covers rows X .. X + 2 (scale / 2), so pixel (jx, jy) gathers the box ending at itself
    const int bw = 2 * (g.scale / 2) + 1;
    unsigned int c = 0;
    long long sc = 0, ss = 0;
    for (int a = jx - bw + 1; a <= jx; ++a) {
        if (a < 0) continue;
        for (int b = jy - bw + 1; b <= jy; ++b) {
            if (b < 0) continue;
            const size_t at = (size_t)a * (size_t)g.C + (size_t)b;
            c += cnt[at]; sc += (long long)pc[at]; ss += (long long)ps[at];
        }
    }
    uint8_t out[3] = {0, 0, 0};                                           // HSV (0, 0, 0) -> black
    if (c >= 1) {                                                         // :710
        const float fc = (float)c;
        const float vx = (float)((double)sc * (1.0 / 4294967296.0)) / fc;   // :712-713
        const float vy = (float)((double)ss * (1.0 / 4294967296.0)) / fc;
        const double speed = hypot((double)vx, (double)vy);               // :715
        double angle = 0;
        if (speed != 0) angle = (atan2((double)vy, (double)vx) + 3.1416) * 180 / 3.1416;   // :717-718
        const int H = (int)(unsigned char)trunc_x86(angle / 2);           // :720-722 (double -> uchar)
        const int S = (int)(unsigned char)trunc_x86(speed * 255);
        hsv_to_bgr_u8(H, S, 255, out);
    }
    bgr[3 * i + 0] = out[0]; bgr[3 * i + 1] = out[1]; bgr[3 * i + 2] = out[2];
}

void launch_color_time(const uint32_t* xy, const int32_t* t, const float2* p, const uint8_t* noise, long long n,
                       const ColorGeom& g, uint32_t* cnt, unsigned long long* pc, unsigned long long* ps, uint8_t* bgr,
                       hipStream_t s) {
    if (n > 0)
        hipLaunchKernelGGL(k_color_point, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, xy, t, p,
                           noise, n, g, cnt, pc, ps);
    const long long px = (long long)g.R * g.C;
    hipLaunchKernelGGL(k_color_final, dim3((unsigned)((px + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, cnt, pc, ps, g,
                       bgr);
}


}  // namespace bf
