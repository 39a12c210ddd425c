// bf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the motion-compensation
// path.  Built with -ffp-contract=off: the reference is compiled for baseline x86-64 and
// none of its float/double operations are fused (better_flow_core/CMakeLists.txt:4,10);
// contraction changes the Scharr planes bitwise.
//
// K1  k_warp_scatter  per-event warp (event.h:99-110,164-168) fused with the point scatter
//                     of the time / event-count image (accel_lib.h:147-166).  HBM-bound:
//                     16-byte coalesced loads of xy / t / p, one 64-bit integer atomic per
//                     event (PACKED) or a 64-bit + a 32-bit one (SPLIT).
// K3  k_stencil       LDS-tiled: s x s box sum of the point planes (== the reference's
//                     s x s splat, exactly, because the planes are integers), normalise
//                     (accel_lib.h:168-175), 3x3 gated Scharr (accel_lib.h:513-615), the
//                     centre-of-mass and moment sums (object_model.cpp:4-39,103-126) with
//                     wave64 shuffle reductions, and zeroing of the other plane buffer.
// K4  k_update        one work-group: deterministic reduction of the per-group partials,
//                     ObjectModel::update_accumulators (object_model.h:48-53), the glue of
//                     iteration_step (optimizer_rolling.h:328-346) and the loop control of
//                     run() (optimizer_rolling.h:61-101) -- entirely on the device.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "bf_device.h"
#include "bf_kernels.h"

namespace bf {

// `int x = <double>` on x86-64 is cvttsd2si: NaN / out-of-range -> INT_MIN (then rejected by
// the bounds test of accel_lib.h:157).  v_cvt_i32_f64 would saturate / give 0 instead.
__device__ __forceinline__ int trunc_x86(double v) {
    return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN;
}

// f32 form of the reference's `p > 0.000001` (float against a double literal): 1e-6f is the
// largest float below 1e-6, so (double)p > 1e-6  <=>  p > 1e-6f.
__device__ __forceinline__ bool valid_px(float p) { return p > 1e-6f; }

// Previous / new projected position from the stored f32 product (event.h:167-168):
//   pr = float(fr) - (kx * float(t)) / 10000.0      (f32 product, f64 divide and subtract)
__device__ __forceinline__ double pr_from_p(uint32_t fr, float prod) {
    return (double)(float)fr - (double)prod / 10000.0;
}

template <bool WARP, bool SCATTER, bool WRITE_N, bool PACKED>
__device__ __forceinline__ void event_body(uint32_t xy, int32_t t, float2& p, bool live,
                                           bool noise, double2* nxny_out,
                                           unsigned long long* plane, uint32_t* cplane,
                                           const DevState* st, const WarpParams& wp) {
    const uint32_t fx = xy & 0xffffu, fy = xy >> 16;
    double pr_x = pr_from_p(fx, p.x);
    double pr_y = pr_from_p(fy, p.y);
    if (WARP) {
        // event.h:100-108
        const double rx = pr_x - wp.cx, ry = pr_y - wp.cy;
        const double qx = wp.c * rx - wp.s * ry;
        const double qy = wp.s * rx + wp.c * ry;
        const double nx = ((-qx) * wp.div + (qx - rx)) + wp.dnx;
        const double ny = ((-qy) * wp.div + (qy - ry)) + wp.dny;
        // event.h:164-165: float kx = float(nx) / nz.  The double division rounded to float
        // equals the correctly rounded f32 division (double has >= 2*24+2 digits).
        const float kx = (float)nx / 127.0f;
        const float ky = (float)ny / 127.0f;
        const float ft = (float)t;   // round-to-nearest int -> f32, as float(sll t)
        p.x = kx * ft;
        p.y = ky * ft;
        pr_x = pr_from_p(fx, p.x);
        pr_y = pr_from_p(fy, p.y);
        if (WRITE_N && live) *nxny_out = make_double2(nx, ny);
    }
    if (SCATTER) {
        if (live && !noise) {
            const int s = st->scale;
            // accel_lib.h:154-158
            const int X = trunc_x86(pr_x * (double)s + (double)st->x_sh);
            const int Y = trunc_x86(pr_y * (double)s + (double)st->y_sh);
            const int hs = s / 2;
            if (!((X >= st->wsx + hs) || (X < hs) || (Y >= st->wsy + hs) || (Y < hs))) {
                const size_t k = (size_t)X * (size_t)st->C + (size_t)Y;
                const unsigned long long dt = (unsigned long long)((long long)t - st->tmin);
                if (PACKED) {
                    atomicAdd(&plane[k], (1ull << st->tbits) + dt);
                } else {
                    atomicAdd(&plane[k], dt);
                    atomicAdd(&cplane[k], 1u);
                }
            }
        }
    }
}

template <bool WARP, bool SCATTER, bool WRITE_N, bool PACKED>
__global__ __launch_bounds__(kThreads) void k_warp_scatter(
    const uint32_t* __restrict__ xy, const int32_t* __restrict__ t, float2* __restrict__ p,
    const uint8_t* __restrict__ noise, double2* __restrict__ nxny,
    unsigned long long* __restrict__ plane, uint32_t* __restrict__ cplane,
    const DevState* __restrict__ st, long long n, int check_done) {
    if (check_done && st->done) return;
    const long long base = ((long long)blockIdx.x * kThreads + threadIdx.x) * kEvPerThread;
    if (base >= n) return;
    const WarpParams wp = st->wp;
    // arrays are padded to a multiple of kEvPerThread * kThreads elements
    const uint4 vxy = *reinterpret_cast<const uint4*>(xy + base);
    const int4 vt = *reinterpret_cast<const int4*>(t + base);
    float4 pa = *reinterpret_cast<const float4*>(p + base);
    float4 pb = *reinterpret_cast<const float4*>(p + base + 2);
    float2 p0 = make_float2(pa.x, pa.y), p1 = make_float2(pa.z, pa.w);
    float2 p2 = make_float2(pb.x, pb.y), p3 = make_float2(pb.z, pb.w);
    bool z0 = false, z1 = false, z2 = false, z3 = false;
    if (SCATTER && noise) {
        const uchar4 vn = *reinterpret_cast<const uchar4*>(noise + base);
        z0 = vn.x; z1 = vn.y; z2 = vn.z; z3 = vn.w;
    }
    event_body<WARP, SCATTER, WRITE_N, PACKED>(vxy.x, vt.x, p0, base + 0 < n, z0, nxny + base + 0,
                                               plane, cplane, st, wp);
    event_body<WARP, SCATTER, WRITE_N, PACKED>(vxy.y, vt.y, p1, base + 1 < n, z1, nxny + base + 1,
                                               plane, cplane, st, wp);
    event_body<WARP, SCATTER, WRITE_N, PACKED>(vxy.z, vt.z, p2, base + 2 < n, z2, nxny + base + 2,
                                               plane, cplane, st, wp);
    event_body<WARP, SCATTER, WRITE_N, PACKED>(vxy.w, vt.w, p3, base + 3 < n, z3, nxny + base + 3,
                                               plane, cplane, st, wp);
    if (WARP) {
        *reinterpret_cast<float4*>(p + base) = make_float4(p0.x, p0.y, p1.x, p1.y);
        *reinterpret_cast<float4*>(p + base + 2) = make_float4(p2.x, p2.y, p3.x, p3.y);
    }
}

// ---------------------------------------------------------------------------------------
// Slice staging: pack fr_x / fr_y, copy t, reset p, and reduce min / max / sum statistics.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_min(int v) {
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__global__ __launch_bounds__(kThreads) void k_prepare(const int32_t* __restrict__ fr_x,
                                                      const int32_t* __restrict__ fr_y,
                                                      const int32_t* __restrict__ t_in,
                                                      uint32_t* __restrict__ xy,
                                                      int32_t* __restrict__ t_out,
                                                      float2* __restrict__ p, long long n,
                                                      long long n_pad, SliceStats* stats) {
    int xmin = INT_MAX, xmax = INT_MIN, ymin = INT_MAX, ymax = INT_MIN;
    int tmin = INT_MAX, tmax = INT_MIN;
    long long tsum = 0;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n_pad;
         i += (long long)gridDim.x * kThreads) {
        if (i < n) {
            const int x = fr_x[i], y = fr_y[i], t = t_in[i];
            xmin = min(xmin, x); xmax = max(xmax, x);
            ymin = min(ymin, y); ymax = max(ymax, y);
            tmin = min(tmin, t); tmax = max(tmax, t);
            tsum += t;
            xy[i] = ((uint32_t)x & 0xffffu) | ((uint32_t)y << 16);
            if (t_out != t_in) t_out[i] = t;
        } else {
            xy[i] = 0;
            t_out[i] = 0;
        }
        p[i] = make_float2(0.f, 0.f);   // Event::reset (event.h:54-59): pr <- fr
    }
    xmin = wave_min(xmin); xmax = wave_max(xmax);
    ymin = wave_min(ymin); ymax = wave_max(ymax);
    tmin = wave_min(tmin); tmax = wave_max(tmax);
    tsum = wave_sum(tsum);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&stats->xmin, xmin); atomicMax(&stats->xmax, xmax);
        atomicMin(&stats->ymin, ymin); atomicMax(&stats->ymax, ymax);
        atomicMin(&stats->tmin, tmin); atomicMax(&stats->tmax, tmax);
        atomicAdd(reinterpret_cast<unsigned long long*>(&stats->tsum), (unsigned long long)tsum);
    }
}

// ---------------------------------------------------------------------------------------
// K3: box sum + normalise + Scharr + moments, LDS tiled.
// ---------------------------------------------------------------------------------------
struct Sums {
    long long n, sci, scj;
    double sgx, sgy, sigx, sigy, sjgx, sjgy;
};

// SRC 0: packed u64 plane; 1: split u64 sum + u32 count planes; 2: f32 time image.
template <int SRC>
__global__ __launch_bounds__(kThreads) void k_stencil(StencilArgs a) {
    if (a.check_done && a.st->done) return;
    constexpr int TR = kTileR, TC = kTileC;
    constexpr int MAXH = kMaxHalfScale + 1;
    __shared__ unsigned long long s_pt[(TR + 2 * MAXH) * (TC + 2 * MAXH)];
    __shared__ uint32_t s_pc[SRC == 1 ? (TR + 2 * MAXH) * (TC + 2 * MAXH) : 1];
    __shared__ float s_time[(TR + 2) * (TC + 2)];
    __shared__ Sums s_red[kThreads / 64];

    const int R = a.R, C = a.C;
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * TC;
    constexpr int TW = TC + 2, TH = TR + 2;

    if (SRC == 2) {
        for (int idx = tid; idx < TH * TW; idx += kThreads) {
            const int tr = idx / TW, tc = idx - tr * TW;
            const int gr = r0 - 1 + tr, gc = c0 - 1 + tc;
            float v = 0.f;
            if (gr >= 0 && gr < R && gc >= 0 && gc < C) v = a.time_in[(size_t)gr * C + gc];
            s_time[idx] = v;
        }
    } else {
        const int hs = a.scale / 2;
        const int H = hs + 1;
        const int PC = TC + 2 * H, PR = TR + 2 * H;
        for (int idx = tid; idx < PR * PC; idx += kThreads) {
            const int pr = idx / PC, pc = idx - pr * PC;
            const int gr = r0 - H + pr, gc = c0 - H + pc;
            unsigned long long v = 0;
            uint32_t cv = 0;
            if (gr >= 0 && gr < R && gc >= 0 && gc < C) {
                v = a.plane[(size_t)gr * C + gc];
                if (SRC == 1) cv = a.cplane[(size_t)gr * C + gc];
            }
            s_pt[idx] = v;
            if (SRC == 1) s_pc[idx] = cv;
        }
        __syncthreads();
        const int tbits = a.tbits;
        const unsigned long long tmask = (tbits >= 64) ? ~0ull : ((1ull << tbits) - 1ull);
        for (int idx = tid; idx < TH * TW; idx += kThreads) {
            const int tr = idx / TW, tc = idx - tr * TW;
            const int gr = r0 - 1 + tr, gc = c0 - 1 + tc;
            float tv = 0.f;
            uint32_t cnt = 0;
            if (gr >= 0 && gr < R && gc >= 0 && gc < C) {
                // s x s box sum == the s x s splat of accel_lib.h:160-165 on integer planes.
                unsigned long long acc = 0;
                uint32_t cacc = 0;
                for (int da = 0; da <= 2 * hs; ++da)
                    for (int db = 0; db <= 2 * hs; ++db) {
                        acc += s_pt[(tr + da) * PC + (tc + db)];
                        if (SRC == 1) cacc += s_pc[(tr + da) * PC + (tc + db)];
                    }
                long long tsum;
                if (SRC == 0) {
                    cnt = (uint32_t)(acc >> tbits);
                    tsum = (long long)(acc & tmask);
                } else {
                    cnt = cacc;
                    tsum = (long long)acc;
                }
                if (cnt > 0) {
                    // accel_lib.h:162,172: f32 sum of seconds, then f32 divide by the count.
                    // The integer-ns sum is exact; it is rounded to f32 seconds once.
                    const long long ts = tsum + (long long)cnt * a.tmin;
                    const float sum_s = (float)((double)ts / 1000000000.0);
                    tv = sum_s / (float)cnt;
                }
                const bool own = (tr >= 1 && tr <= TR && tc >= 1 && tc <= TC);
                if (own) {
                    if (a.time_out) a.time_out[(size_t)gr * C + gc] = tv;
                    if (a.count_out) a.count_out[(size_t)gr * C + gc] = cnt;
                }
            }
            s_time[idx] = tv;
        }
    }
    __syncthreads();

    Sums sm;
    sm.n = sm.sci = sm.scj = 0;
    sm.sgx = sm.sgy = sm.sigx = sm.sigy = sm.sjgx = sm.sjgy = 0.0;
    const int hR = R / 2, hC = C / 2;
#pragma unroll
    for (int k = 0; k < (TR * TC) / kThreads; ++k) {
        const int pidx = tid + k * kThreads;
        const int lr = pidx / TC, lc = pidx - lr * TC;
        const int gr = r0 + lr, gc = c0 + lc;
        if (gr < R && gc < C) {
            const float* tp = &s_time[(lr + 1) * TW + (lc + 1)];
            const float ctr = tp[0];
            float gx = 0.f, gy = 0.f;
            const bool v = valid_px(ctr);
            if (v && gr >= 1 && gr < R - 1 && gc >= 1 && gc < C - 1) {
                // accel_lib.h:594-604: k = column offset (outer), l = row offset (inner),
                // idx = 3k + l; sharr_x = {3,0,-3,10,0,-10,3,0,-3},
                // sharr_y = {3,10,3,0,0,0,-3,-10,-3}; any tap <= 1e-6 -> gradient stays 0.
                const float t00 = tp[-TW - 1], t10 = tp[-1], t20 = tp[TW - 1];
                const float t01 = tp[-TW], t21 = tp[TW];
                const float t02 = tp[-TW + 1], t12 = tp[1], t22 = tp[TW + 1];
                const bool all = valid_px(t00) && valid_px(t10) && valid_px(t20) &&
                                 valid_px(t01) && valid_px(t21) && valid_px(t02) &&
                                 valid_px(t12) && valid_px(t22);
                if (all) {
                    float dx = 0.f, dy = 0.f;
                    // k = 0 (column c-1): l = 0,1,2 (rows r-1, r, r+1)
                    dx = dx + t00 * 3.f;   dy = dy + t00 * 3.f;
                    dx = dx + t10 * 0.f;   dy = dy + t10 * 10.f;
                    dx = dx + t20 * -3.f;  dy = dy + t20 * 3.f;
                    // k = 1 (column c)
                    dx = dx + t01 * 10.f;  dy = dy + t01 * 0.f;
                    dx = dx + ctr * 0.f;   dy = dy + ctr * 0.f;
                    dx = dx + t21 * -10.f; dy = dy + t21 * 0.f;
                    // k = 2 (column c+1)
                    dx = dx + t02 * 3.f;   dy = dy + t02 * -3.f;
                    dx = dx + t12 * 0.f;   dy = dy + t12 * -10.f;
                    dx = dx + t22 * -3.f;  dy = dy + t22 * -3.f;
                    gx = dx;
                    gy = dy;
                }
            }
            if (a.gx_out) {
                a.gx_out[(size_t)gr * C + gc] = gx;
                a.gy_out[(size_t)gr * C + gc] = gy;
            }
            if (v) {
                // object_model.cpp:22-30 and :112-116 in one pass, centred coordinates
                const int ci = gr - hR, cj = gc - hC;
                sm.n += 1;
                sm.sci += ci;
                sm.scj += cj;
                const double gxd = (double)gx, gyd = (double)gy;
                sm.sgx += gxd;
                sm.sgy += gyd;
                sm.sigx += (double)ci * gxd;
                sm.sigy += (double)ci * gyd;
                sm.sjgx += (double)cj * gxd;
                sm.sjgy += (double)cj * gyd;
            }
            if (a.zero_plane) {
                a.zero_plane[(size_t)gr * C + gc] = 0ull;
                if (a.zero_cplane) a.zero_cplane[(size_t)gr * C + gc] = 0u;
            }
        }
    }
    if (a.partials) {
        sm.n = wave_sum(sm.n); sm.sci = wave_sum(sm.sci); sm.scj = wave_sum(sm.scj);
        sm.sgx = wave_sum(sm.sgx); sm.sgy = wave_sum(sm.sgy);
        sm.sigx = wave_sum(sm.sigx); sm.sigy = wave_sum(sm.sigy);
        sm.sjgx = wave_sum(sm.sjgx); sm.sjgy = wave_sum(sm.sjgy);
        if ((tid & 63) == 0) s_red[tid >> 6] = sm;
        __syncthreads();
        if (tid == 0) {
            Sums t = s_red[0];
            for (int w = 1; w < kThreads / 64; ++w) {
                t.n += s_red[w].n; t.sci += s_red[w].sci; t.scj += s_red[w].scj;
                t.sgx += s_red[w].sgx; t.sgy += s_red[w].sgy;
                t.sigx += s_red[w].sigx; t.sigy += s_red[w].sigy;
                t.sjgx += s_red[w].sjgx; t.sjgy += s_red[w].sjgy;
            }
            Partial& o = a.partials[blockIdx.y * gridDim.x + blockIdx.x];
            o.n = t.n; o.sci = t.sci; o.scj = t.scj;
            o.sgx = t.sgx; o.sgy = t.sgy;
            o.sigx = t.sigx; o.sigy = t.sigy; o.sjgx = t.sjgx; o.sjgy = t.sjgy;
        }
    }
}

// ---------------------------------------------------------------------------------------
// K4: deterministic reduction + model + accumulators + loop control.
// mode 0: model only (AccelLib::fast_model).  mode 1: full iteration_step / run() glue.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_update(DevState* st, const Partial* __restrict__ partials,
                                                     int nblocks, bf_trace_rec* trace, int mode) {
    if (mode == 1 && st->done) return;
    __shared__ Sums s_red[kThreads / 64];
    const int tid = threadIdx.x;
    Sums sm;
    sm.n = sm.sci = sm.scj = 0;
    sm.sgx = sm.sgy = sm.sigx = sm.sigy = sm.sjgx = sm.sjgy = 0.0;
    for (int i = tid; i < nblocks; i += kThreads) {
        const Partial q = partials[i];
        sm.n += q.n; sm.sci += q.sci; sm.scj += q.scj;
        sm.sgx += q.sgx; sm.sgy += q.sgy;
        sm.sigx += q.sigx; sm.sigy += q.sigy; sm.sjgx += q.sjgx; sm.sjgy += q.sjgy;
    }
    sm.n = wave_sum(sm.n); sm.sci = wave_sum(sm.sci); sm.scj = wave_sum(sm.scj);
    sm.sgx = wave_sum(sm.sgx); sm.sgy = wave_sum(sm.sgy);
    sm.sigx = wave_sum(sm.sigx); sm.sigy = wave_sum(sm.sigy);
    sm.sjgx = wave_sum(sm.sjgx); sm.sjgy = wave_sum(sm.sjgy);
    if ((tid & 63) == 0) s_red[tid >> 6] = sm;
    __syncthreads();
    if (tid != 0) return;
    Sums t = s_red[0];
    for (int w = 1; w < kThreads / 64; ++w) {
        t.n += s_red[w].n; t.sci += s_red[w].sci; t.scj += s_red[w].scj;
        t.sgx += s_red[w].sgx; t.sgy += s_red[w].sgy;
        t.sigx += s_red[w].sigx; t.sigy += s_red[w].sigy;
        t.sjgx += s_red[w].sjgx; t.sjgy += s_red[w].sjgy;
    }

    bf_model m = st->model;
    const int R = st->R, C = st->C;
    const double dn = (double)t.n;   // cnt == 0 -> 0/0 = NaN, as in the reference (assert off)
    // object_model.cpp:103-126: cx = (sum of rows) / cnt, exact integer numerator
    m.cx = (double)(t.sci + t.n * (long long)(R / 2)) / dn;
    m.cy = (double)(t.scj + t.n * (long long)(C / 2)) / dn;
    const double cxc = (double)t.sci / dn, cyc = (double)t.scj / dn;
    // object_model.cpp:26-38 with r = (ci - cxc, cj - cyc)
    m.dx = t.sgx / dn;
    m.dy = t.sgy / dn;
    m.rot = ((t.sigy - cxc * t.sgy) - (t.sjgx - cyc * t.sgx)) / dn;
    m.div = ((t.sigx - cxc * t.sgx) + (t.sjgy - cyc * t.sgy)) / dn;
    m.cnt = (uint32_t)t.n;
    if (mode == 0) {
        st->model = m;
        return;
    }
    // object_model.h:48-53 via optimizer_rolling.h:328
    m.total_rot += m.rot / (double)st->rot_div;
    m.total_div += m.div / (double)st->div_div;
    m.total_dx += m.dx / (double)st->x_div;
    m.total_dy += m.dy / (double)st->y_div;
    // optimizer_rolling.h:330-331,340-346
    const double cxs = (m.cx - st->x_shift) / (double)st->scale;
    const double cys = (m.cy - st->y_shift) / (double)st->scale;
    WarpParams wp;
    wp.dnx = -m.total_dx; wp.dny = -m.total_dy;
    wp.cx = cxs; wp.cy = cys;
    wp.div = m.total_div;
    const double crl = -m.total_rot;
    wp.c = cos(crl);
    wp.s = sin(crl);
    m.cx = cxs;
    m.cy = cys;
    st->wp = wp;
    st->model = m;

    // ---- run(), optimizer_rolling.h:73-101, as a state machine after each step ----
    const int it = st->it + 1;
    st->it = it;
    float xd = st->x_div, yd = st->y_div, rd = st->rot_div, dd = st->div_div;
    int done = 0, rc = 0;
    if (it > 1) {
        if (st->max_iter > 0 && it > st->max_iter) {   // :94-96 (before the sign flips)
            done = 1;
        } else {                                       // :98-101
            if (m.dx * (double)st->old_dx < 0) xd *= 2;
            if (m.dy * (double)st->old_dy < 0) yd *= 2;
            if (m.rot * (double)st->old_rot < 0) rd *= 2;
            if (m.div * (double)st->old_div < 0) dd *= 2;
            st->x_div = xd; st->y_div = yd; st->rot_div = rd; st->div_div = dd;
            if (st->hard_cap > 0 && it >= st->hard_cap) { done = 1; rc = BF_ERR_NOCONV; }
        }
    }
    if (trace && it <= st->trace_cap) {
        bf_trace_rec& r = trace[it - 1];
        r.model = m;
        r.x_divider = xd; r.y_divider = yd; r.rot_divider = rd; r.div_divider = dd;
        r.iteration = it;
    }
    if (!done) {
        if (!(xd < 32 * 10 || yd < 32 * 10 || rd < 32 * 1000 || dd < 32 * 1000)) {   // :76-79
            done = 1;
        } else if (fabs(m.dx / (double)xd) < 1e-5 && fabs(m.dy / (double)yd) < 1e-5 &&
                   fabs(m.rot / (double)rd) < 1e-4 && fabs(m.div / (double)dd) < 1e-1) {   // :81-84
            done = 1;
        } else {                                                                         // :86-89
            st->old_dx = (float)m.dx; st->old_dy = (float)m.dy;
            st->old_rot = (float)m.rot; st->old_div = (float)m.div;
        }
    }
    if (done) {
        st->rc = rc;
        st->done = 1;
    }
}

// Event::compute_uv, event.h:135-142.
__global__ __launch_bounds__(kThreads) void k_compute_uv(const double2* __restrict__ nxny,
                                                         double2* __restrict__ uv, long long n) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const double2 v = nxny[i];
    const double len = hypot(v.x, v.y);
    const double speed = len / (127.0 / (double)(1000000000 / (1 * 10000)));
    double2 o;
    o.x = (len == 0) ? 0 : speed * v.x / len;
    o.y = (len == 0) ? 0 : speed * v.y / len;
    uv[i] = o;
}

// pr / n read-back helper for writeout_events (accel_lib.h:310-329).
__global__ __launch_bounds__(kThreads) void k_expand_pr(const uint32_t* __restrict__ xy,
                                                        const float2* __restrict__ p,
                                                        double2* __restrict__ pr, long long n) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = xy[i];
    const float2 q = p[i];
    pr[i] = make_double2(pr_from_p(v & 0xffffu, q.x), pr_from_p(v >> 16, q.y));
}

__global__ __launch_bounds__(kThreads) void k_copy_f4(const float4* __restrict__ src,
                                                      float4* __restrict__ dst, long long n4) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4;
         i += (long long)gridDim.x * kThreads)
        dst[i] = src[i];
}

// Host -> device state hand-over.  The struct travels as a kernel argument (captured at
// launch), so no pinned staging buffer has to stay alive until the copy executes.
__global__ void k_set_state(DevState* st, DevState v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *st = v;
}

__global__ void k_init_stats(SliceStats* s) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        s->xmin = s->ymin = s->tmin = INT_MAX;
        s->xmax = s->ymax = s->tmax = INT_MIN;
        s->tsum = 0;
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
void launch_set_state(DevState* st, const DevState& v, hipStream_t s) {
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(64), 0, s, st, v);
}

void launch_init_stats(SliceStats* st, hipStream_t s) {
    hipLaunchKernelGGL(k_init_stats, dim3(1), dim3(64), 0, s, st);
}

template <bool W, bool S, bool N>
static void launch_ws(const WarpScatterArgs& a, hipStream_t s, dim3 grid) {
    if (a.packed)
        hipLaunchKernelGGL((k_warp_scatter<W, S, N, true>), grid, dim3(kThreads), 0, s, a.xy, a.t, a.p,
                           a.noise, a.nxny, a.plane, a.cplane, a.st, a.n, a.check_done);
    else
        hipLaunchKernelGGL((k_warp_scatter<W, S, N, false>), grid, dim3(kThreads), 0, s, a.xy, a.t, a.p,
                           a.noise, a.nxny, a.plane, a.cplane, a.st, a.n, a.check_done);
}

void launch_warp_scatter(const WarpScatterArgs& a, bool warp, bool scatter, bool write_n,
                         hipStream_t s) {
    const long long per_block = (long long)kThreads * kEvPerThread;
    dim3 grid((unsigned)((a.n + per_block - 1) / per_block));
    if (grid.x == 0) return;
    if (warp && scatter && !write_n) launch_ws<true, true, false>(a, s, grid);
    else if (!warp && scatter) launch_ws<false, true, false>(a, s, grid);
    else if (warp && !scatter && write_n) launch_ws<true, false, true>(a, s, grid);
    else if (warp && !scatter && !write_n) launch_ws<true, false, false>(a, s, grid);
    else if (warp && scatter && write_n) launch_ws<true, true, true>(a, s, grid);
}

void launch_prepare(const int32_t* fr_x, const int32_t* fr_y, const int32_t* t_in, uint32_t* xy,
                    int32_t* t_out, float2* p, long long n, long long n_pad, SliceStats* stats,
                    hipStream_t s) {
    long long blocks = (n_pad + kThreads - 1) / kThreads;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)blocks), dim3(kThreads), 0, s, fr_x, fr_y, t_in, xy,
                       t_out, p, n, n_pad, stats);
}

void stencil_grid(int R, int C, int* gx, int* gy) {
    *gx = (C + kTileC - 1) / kTileC;
    *gy = (R + kTileR - 1) / kTileR;
}

void launch_stencil(const StencilArgs& a, int src, hipStream_t s) {
    int gx, gy;
    stencil_grid(a.R, a.C, &gx, &gy);
    dim3 grid(gx, gy);
    if (src == 0) hipLaunchKernelGGL(k_stencil<0>, grid, dim3(kThreads), 0, s, a);
    else if (src == 1) hipLaunchKernelGGL(k_stencil<1>, grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL(k_stencil<2>, grid, dim3(kThreads), 0, s, a);
}

void launch_update(DevState* st, const Partial* partials, int nblocks, bf_trace_rec* trace, int mode,
                   hipStream_t s) {
    hipLaunchKernelGGL(k_update, dim3(1), dim3(kThreads), 0, s, st, partials, nblocks, trace, mode);
}

void launch_compute_uv(const double2* nxny, double2* uv, long long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_compute_uv, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       s, nxny, uv, n);
}

void launch_expand_pr(const uint32_t* xy, const float2* p, double2* pr, long long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_expand_pr, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       s, xy, p, pr, n);
}

void launch_copy(const void* src, void* dst, long long bytes, hipStream_t s) {
    hipLaunchKernelGGL(k_copy_f4, dim3(2048), dim3(kThreads), 0, s, (const float4*)src, (float4*)dst,
                       bytes / 16);
}

}  // namespace bf
