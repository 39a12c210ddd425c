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
//                     The sums go to exact accumulators (bf_device.h: MomentAcc); the last work-group to arrive
//                     runs ObjectModel::update_accumulators (object_model.h:48-53), the glue of
//                     iteration_step (optimizer_rolling.h:328-346) and the loop control of
//                     run() (optimizer_rolling.h:61-101) -- entirely on the device.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {

// Event::compute_uv, event.h:135-142.
__device__ __forceinline__ double2 uv_from_n(double2 v) {
    const double len = hypot(v.x, v.y);
    const double speed = len / (127.0 / (double)(1000000000 / (1 * 10000)));
    double2 o;
    o.x = (len == 0) ? 0 : speed * v.x / len;
    o.y = (len == 0) ? 0 : speed * v.y / len;
    return o;
}

template <bool WARP, bool SCATTER, bool WRITE_N, bool PACKED>
__device__ __forceinline__ void event_body(uint32_t xy, int32_t t, float2& p, bool live,
                                           bool noise, double2* nxny, double2* uv, const uint32_t* perm,
                                           long long slot,
                                           unsigned long long* plane, uint32_t* cplane,
                                           const HotState& st, const WarpParams& wp) {
    const uint32_t fx = xy & 0xffffu, fy = xy >> 16;
    double pr_x = pr_from_p(fx, p.x);
    double pr_y = pr_from_p(fy, p.y);
    if (WARP) {
        double nx, ny;
        warp_products(wp, pr_x, pr_y, t, p, nx, ny);
        pr_x = pr_from_p(fx, p.x);
        pr_y = pr_from_p(fy, p.y);
        if (WRITE_N && live) {
            const long long o = perm ? (long long)perm[slot] : slot;
            nxny[o] = make_double2(nx, ny);
            if (uv) uv[o] = uv_from_n(make_double2(nx, ny));   // compute_uv fused into the final warp
        }
    }
    if (SCATTER) {
        if (live && !noise) {
            const int s = st.scale;
            // accel_lib.h:154-158
            const int X = trunc_scatter(pr_x * (double)s + (double)st.x_sh);
            const int Y = trunc_scatter(pr_y * (double)s + (double)st.y_sh);
            const int hs = s / 2;
            if (!((X >= st.wsx + hs) || (X < hs) || (Y >= st.wsy + hs) || (Y < hs))) {
                const size_t k = (size_t)X * (size_t)st.C + (size_t)Y;
                const unsigned long long dt = (unsigned long long)((long long)t - st.tmin);
                if (PACKED) {
                    atomicAdd(&plane[k], (1ull << st.tbits) + dt);
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
    const uint32_t* xy, const int32_t* t, float2* p,
    const uint8_t* __restrict__ noise, double2* __restrict__ nxny, double2* __restrict__ uv,
    const uint32_t* perm,
    unsigned long long* __restrict__ plane, uint32_t* __restrict__ cplane,
    const DevState* __restrict__ st, long long n, int check_done, EvSets sets, int pick_set, int sorted_out) {
    const HotState hs = st->hot;   // one burst of scalar loads, then the branch
    // check_done 1: a loop kernel, idle once the loop is done; 2: the final warp enqueued ahead of the poll,
    // runs only if the loop IS done
    if (check_done == 1 && hs.done) return;
    if (check_done == 2 && !hs.done) return;
    const long long base = ((long long)blockIdx.x * kThreads + threadIdx.x) * kEvPerThread;
    if (base >= n) return;
    if (pick_set) {   // tile-binned loop: the device knows which set holds the (sorted) events
        const EvSetPtrs e = (hs.cs ^ hs.flip) ? sets.s[1] : sets.s[0];
        xy = e.xy; t = e.t; p = e.p; perm = e.perm;
    }
    if (sorted_out) perm = nullptr;   // outputs in slot order (coalesced); the host un-permutes them when they are read back
    const WarpParams& wp = hs.wp;
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
    event_body<WARP, SCATTER, WRITE_N, PACKED>(vxy.x, vt.x, p0, base + 0 < n, z0, nxny, uv, perm, base + 0,
                                               plane, cplane, hs, wp);
    event_body<WARP, SCATTER, WRITE_N, PACKED>(vxy.y, vt.y, p1, base + 1 < n, z1, nxny, uv, perm, base + 1,
                                               plane, cplane, hs, wp);
    event_body<WARP, SCATTER, WRITE_N, PACKED>(vxy.z, vt.z, p2, base + 2 < n, z2, nxny, uv, perm, base + 2,
                                               plane, cplane, hs, wp);
    event_body<WARP, SCATTER, WRITE_N, PACKED>(vxy.w, vt.w, p3, base + 3 < n, z3, nxny, uv, perm, base + 3,
                                               plane, cplane, hs, wp);
    if (WARP) {
        *reinterpret_cast<float4*>(p + base) = make_float4(p0.x, p0.y, p1.x, p1.y);
        *reinterpret_cast<float4*>(p + base + 2) = make_float4(p2.x, p2.y, p3.x, p3.y);
    }
}

// AccelLib::project_4param (accel_lib.h:275-281) -> Event::project_4param (event.h:88-96): the INCREMENTAL warp -- the same dn as
// project_4param_reinit from the previous pr, but ADDED to the event's (nx, ny) (Event::project_dn, event.h:72-76) before
// apply_project.  Dead in the reference (its only call is commented out, optimizer_rolling.h:333-339); exported for signature
// completeness.  nxny: the events' current (nx, ny), indexed by upload order (perm) -- zero when have_n == 0 (Event::reset).
__global__ __launch_bounds__(kThreads) void k_project_dn(const uint32_t* __restrict__ xy, const int32_t* __restrict__ t, float2* __restrict__ p,
                                                         double2* __restrict__ nxny, const uint32_t* __restrict__ perm, int have_n,
                                                         const DevState* __restrict__ st, long long n) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const WarpParams wp = st->hot.wp;
    const uint32_t v = xy[i];
    float2 q = p[i];
    const double pr_x = pr_from_p(v & 0xffffu, q.x), pr_y = pr_from_p(v >> 16, q.y);
    const long long o = perm ? (long long)perm[i] : i;
    const double2 old = have_n ? nxny[o] : make_double2(0.0, 0.0);
    // event.h:89-95 (the expression of warp_products, the sum then ADDED: project_dn)
    const double rx = pr_x - wp.cx, ry = pr_y - wp.cy;
    const double qx = wp.c * rx - wp.s * ry;
    const double qy = wp.s * rx + wp.c * ry;
    const double nx = old.x + (((-qx) * wp.div + (qx - rx)) + wp.dnx);
    const double ny = old.y + (((-qy) * wp.div + (qy - ry)) + wp.dny);
    const float ft = (float)t[i];
    q.x = div_127((float)nx) * ft;   // apply_project, event.h:164-168
    q.y = div_127((float)ny) * ft;
    p[i] = q;
    nxny[o] = make_double2(nx, ny);
}

void launch_project_dn(const uint32_t* xy, const int32_t* t, float2* p, double2* nxny, const uint32_t* perm, int have_n,
                       const DevState* st, long long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_project_dn, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, xy, t, p, nxny, perm, have_n, st, n);
}

// The last project_4param_reinit of a run (optimizer_rolling.h:340-344) with Event::compute_uv (event.h:135-142) fused:
// n, (u, v) and the new products for every event, outputs in slot order.  ONE event per thread: the general kernel above
// takes four consecutive events per thread (16-byte loads), which leaves 3900 waves for 1M events -- four per SIMD, each
// running ~1000 dependent f64 instructions (hypot and three IEEE divisions per event); this form has four times the waves
// to interleave.  Same arithmetic per event, same bits.
__global__ __launch_bounds__(kThreads) void k_final_warp(WarpScatterArgs a) {
    const HotState hs = sload(&a.st->hot);
    if (a.check_done == 1 && hs.done) return;
    if (a.check_done == 2 && !hs.done) return;
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t* xy = a.xy;
    const int32_t* t = a.t;
    float2* p = a.p;
    const float2* src = a.p;
    if (a.pick_set) {   // tile-binned loop: the device knows which set holds the (sorted) events
        const EvSetPtrs e = (hs.cs ^ hs.flip) ? a.sets.s[1] : a.sets.s[0];
        xy = e.xy; t = e.t; p = e.p;
        if (hs.pp) src = e.p2;   // one-kernel iteration: the current products may sit in the set's second array;
        else src = e.p;          // the final ones always go to the first
    }
    const uint32_t v = xy[i];
    const int32_t ti = t[i];
    float2 q = src[i];
    double nx, ny;
    warp_products(hs.wp, pr_from_p(v & 0xffffu, q.x), pr_from_p(v >> 16, q.y), ti, q, nx, ny);
    p[i] = q;
    a.nxny[i] = make_double2(nx, ny);
    if (a.uv) a.uv[i] = uv_from_n(make_double2(nx, ny));
}
void launch_final_warp(const WarpScatterArgs& a, hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(k_final_warp, dim3((unsigned)((a.n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, a);
}

// ---------------------------------------------------------------------------------------
// Slice staging: pack fr_x / fr_y, copy t, reset p, and reduce min / max / sum statistics.
// ---------------------------------------------------------------------------------------
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
    // one SliceStats per work-group, no atomics: the host folds the <= kPrepBlocks records at
    // bf_set_cloud (it synchronises there anyway).  Per-wave atomics on seven shared words made
    // this kernel 660 us per 1M events.
    __shared__ SliceStats s_w[kThreads / 64];
    xmin = wave_min(xmin); xmax = wave_max(xmax);
    ymin = wave_min(ymin); ymax = wave_max(ymax);
    tmin = wave_min(tmin); tmax = wave_max(tmax);
    tsum = wave_sum(tsum);
    if ((threadIdx.x & 63) == 0) {
        SliceStats& w = s_w[threadIdx.x >> 6];
        w.xmin = xmin; w.xmax = xmax; w.ymin = ymin; w.ymax = ymax; w.tmin = tmin; w.tmax = tmax;
        w.tsum = tsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        SliceStats r = s_w[0];
        for (int k = 1; k < kThreads / 64; ++k) {
            r.xmin = min(r.xmin, s_w[k].xmin); r.xmax = max(r.xmax, s_w[k].xmax);
            r.ymin = min(r.ymin, s_w[k].ymin); r.ymax = max(r.ymax, s_w[k].ymax);
            r.tmin = min(r.tmin, s_w[k].tmin); r.tmax = max(r.tmax, s_w[k].tmax);
            r.tsum += s_w[k].tsum;
        }
        stats[blockIdx.x] = r;
    }
}

// SRC 0: packed u64 plane; 1: split u64 sum + u32 count planes; 2: f32 time image.
// (The tile-binned source has its own kernel, k_stencil_binned in bf_stencil.hip.)
template <int SRC, int NT>
__global__ __launch_bounds__(NT) void k_stencil(StencilArgs a) {
    if (a.check_done && a.st->hot.done) return;
    constexpr int TR = kTileR, TC = kTileC;
    constexpr int MAXH = kMaxHalfScale + 1;
    __shared__ unsigned long long s_pt[(TR + 2 * MAXH) * (TC + 2 * MAXH)];
    __shared__ uint32_t s_pc[(SRC == 1) ? (TR + 2 * MAXH) * (TC + 2 * MAXH) : 1];
    __shared__ float s_time[(TR + 2) * (TC + 2)];
    __shared__ Sums s_red[NT / 64];

    const int R = a.R, C = a.C;
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * TC;
    constexpr int TW = TC + 2, TH = TR + 2;

    if (SRC == 2) {
        for (int idx = tid; idx < TH * TW; idx += NT) {
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
        for (int idx = tid; idx < PR * PC; idx += NT) {
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
        for (int idx = tid; idx < TH * TW; idx += NT) {
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
                tv = time_from_sums(cnt, tsum, a.tmin);
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

    const bool do_zero = a.zero_plane != nullptr;   // the other plane buffer is cleared here
    stencil_tail<TR, TC, NT>(a, s_time, s_red, r0, c0, do_zero);
}

__global__ __launch_bounds__(kThreads) void k_compute_uv(const double2* __restrict__ nxny,
                                                         double2* __restrict__ uv, long long n) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    uv[i] = uv_from_n(nxny[i]);
}

// Per-event outputs written in slot (tile-sorted) order -> upload order: dst[perm[i]] = src[i].
__global__ __launch_bounds__(kThreads) void k_unpermute(const double2* __restrict__ src, const uint32_t* __restrict__ perm,
                                                        double2* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) dst[perm[i]] = src[i];
}

// pr / n read-back helper for writeout_events (accel_lib.h:310-329).
__global__ __launch_bounds__(kThreads) void k_expand_pr(const uint32_t* __restrict__ xy,
                                                        const float2* __restrict__ p,
                                                        const uint32_t* __restrict__ perm,
                                                        double2* __restrict__ pr, long long n) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = xy[i];
    const float2 q = p[i];
    pr[perm ? (long long)perm[i] : i] = make_double2(pr_from_p(v & 0xffffu, q.x), pr_from_p(v >> 16, q.y));
}

// Streaming-copy probe (bf_copy_bandwidth): every thread keeps four 16-byte loads in flight per round (consecutive
// threads touch consecutive 16-byte words, the four loads of a thread are one work-group stride apart), then stores
// them -- NT: with non-temporal stores, which do not allocate the destination lines in the L2.
typedef float bf_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_copy_f4(const bf_v4f* __restrict__ src,
                                                      bf_v4f* __restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * kThreads * 4;
    for (long long base = (long long)blockIdx.x * kThreads * 4 + threadIdx.x; base < n4; base += stride) {
        bf_v4f v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = base + (long long)k * kThreads;
            if (i < n4) v[k] = NT ? __builtin_nontemporal_load(&src[i]) : src[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = base + (long long)k * kThreads;
            if (i < n4) {
                if (NT) __builtin_nontemporal_store(v[k], &dst[i]);
                else dst[i] = v[k];
            }
        }
    }
}

// Host -> device state hand-over.  The struct travels as a kernel argument (captured at
// launch), so no pinned staging buffer has to stay alive until the copy executes.
__global__ void k_set_state(DevState* st, DevState v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *st = v;
}

// Event::set_local_time (event.h:61-63) for a slice handed over with absolute 64-bit timestamps:
// t = ts > t0 ? ts - t0 : -(t0 - ts), as the 32-bit slice-local time of the device layout.  A time that does
// not fit is stored as INT32_MIN, which bf_set_cloud reports (the host front end rejects such slices too).
__global__ __launch_bounds__(kThreads) void k_local_time(const unsigned long long* __restrict__ ts,
                                                         unsigned long long t0, int32_t* __restrict__ t_out,
                                                         long long n) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const unsigned long long v = ts[i];
    const long long t = v > t0 ? (long long)(v - t0) : -(long long)(t0 - v);
    t_out[i] = (t > (long long)INT_MAX || t <= (long long)INT_MIN) ? INT_MIN : (int32_t)t;
}

// The same for a ring with 16-bit addresses (bf_upload_ring16_async): one pass widens row / column for the staging kernel.
// TS32 (bf_upload_ring16t32_async): only the low 32 bits of every timestamp came over the link; the difference of the low
// halves, taken modulo 2^32 and read as a signed number, IS the slice-local time while |timestamp - t0| < 2^31 ns.
template <bool TS32>
__global__ __launch_bounds__(kThreads) void k_local_time16(const unsigned long long* __restrict__ ts,
                                                           const uint16_t* __restrict__ row, const uint16_t* __restrict__ col,
                                                           unsigned long long t0, int32_t* __restrict__ x_out,
                                                           int32_t* __restrict__ y_out, int32_t* __restrict__ t_out, long long n) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    if (TS32) {
        t_out[i] = (int32_t)(reinterpret_cast<const uint32_t*>(ts)[i] - (uint32_t)t0);
    } else {
        const unsigned long long v = ts[i];
        const long long t = v > t0 ? (long long)(v - t0) : -(long long)(t0 - v);
        t_out[i] = (t > (long long)INT_MAX || t <= (long long)INT_MIN) ? INT_MIN : (int32_t)t;
    }
    x_out[i] = (int32_t)row[i];
    y_out[i] = (int32_t)col[i];
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
void launch_local_time16(const unsigned long long* ts, bool ts32, const uint16_t* row, const uint16_t* col, unsigned long long t0,
                         int32_t* x_out, int32_t* y_out, int32_t* t_out, long long n, hipStream_t s) {
    if (n <= 0) return;
    const dim3 grid((unsigned)((n + kThreads - 1) / kThreads));
    if (ts32) hipLaunchKernelGGL(k_local_time16<true>, grid, dim3(kThreads), 0, s, ts, row, col, t0, x_out, y_out, t_out, n);
    else hipLaunchKernelGGL(k_local_time16<false>, grid, dim3(kThreads), 0, s, ts, row, col, t0, x_out, y_out, t_out, n);
}

void launch_local_time(const unsigned long long* ts, unsigned long long t0, int32_t* t_out, long long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_local_time, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, ts, t0, t_out, n);
}

LaunchTimer& launch_timer() {
    static thread_local LaunchTimer t;
    return t;
}

void launch_set_state(DevState* st, const DevState& v, hipStream_t s) {
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(64), 0, s, st, v);
}

template <bool W, bool S, bool N>
static void launch_ws(const WarpScatterArgs& a, hipStream_t s, dim3 grid) {
    if (a.packed)
        hipLaunchKernelGGL((k_warp_scatter<W, S, N, true>), grid, dim3(kThreads), 0, s, a.xy, a.t, a.p,
                           a.noise, a.nxny, a.uv, a.perm, a.plane, a.cplane, a.st, a.n, a.check_done, a.sets, a.pick_set, a.sorted_out);
    else
        hipLaunchKernelGGL((k_warp_scatter<W, S, N, false>), grid, dim3(kThreads), 0, s, a.xy, a.t, a.p,
                           a.noise, a.nxny, a.uv, a.perm, a.plane, a.cplane, a.st, a.n, a.check_done, a.sets, a.pick_set, a.sorted_out);
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
    hipLaunchKernelGGL(k_prepare, dim3(kPrepBlocks), dim3(kThreads), 0, s, fr_x, fr_y, t_in, xy,
                       t_out, p, n, n_pad, stats);
}

void stencil_grid(int R, int C, int* gx, int* gy) {
    *gx = (C + kTileC - 1) / kTileC;
    *gy = (R + kTileR - 1) / kTileR;
}

void launch_stencil(const StencilArgs& a, int src, hipStream_t s, int n_cus) {
    int gx, gy;
    stencil_grid(a.R, a.C, &gx, &gy);
    dim3 grid(gx, gy);
    if (src == 3) { launch_stencil_binned(a, grid, s, n_cus); return; }
    if (src == 0) hipLaunchKernelGGL((k_stencil<0, kThreads>), grid, dim3(kThreads), 0, s, a);
    else if (src == 1) hipLaunchKernelGGL((k_stencil<1, kThreads>), grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((k_stencil<2, kThreads>), grid, dim3(kThreads), 0, s, a);
}

void launch_compute_uv(const double2* nxny, double2* uv, long long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_compute_uv, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       s, nxny, uv, n);
}

void launch_unpermute(const double2* src, const uint32_t* perm, double2* dst, long long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unpermute, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, src, perm, dst, n);
}

void launch_expand_pr(const uint32_t* xy, const float2* p, const uint32_t* perm, double2* pr, long long n,
                      hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_expand_pr, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       s, xy, p, perm, pr, n);
}

// The device loops' sin / cos of the warp's rotation angle (sincos_small; sincos_small_tab with its coefficients from an LDS
// table, as the persistent loop kernel evaluates it), for arbitrary arguments: what bf_eval_sincos measures against libm.
__global__ __launch_bounds__(kThreads) void k_eval_sincos(const double* __restrict__ x, long long n, int table,
                                                          double* __restrict__ sn, double* __restrict__ cs) {
    __shared__ double s_tab[14];
    if (threadIdx.x == 0) sincos_table_fill(s_tab);
    __syncthreads();
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        double a, b;
        if (table) sincos_small_tab(x[i], s_tab, &a, &b);
        else sincos_small(x[i], &a, &b);
        sn[i] = a;
        cs[i] = b;
    }
}
void launch_eval_sincos(const double* x, long long n, int table, double* sn, double* cs, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_eval_sincos, dim3(1024), dim3(kThreads), 0, s, x, n, table, sn, cs);
}

void launch_copy(const void* src, void* dst, long long bytes, int blocks, bool nontemporal, hipStream_t s) {
    if (nontemporal)
        hipLaunchKernelGGL(k_copy_f4<true>, dim3(blocks), dim3(kThreads), 0, s, (const bf_v4f*)src, (bf_v4f*)dst, bytes / 16);
    else
        hipLaunchKernelGGL(k_copy_f4<false>, dim3(blocks), dim3(kThreads), 0, s, (const bf_v4f*)src, (bf_v4f*)dst, bytes / 16);
}

}  // namespace bf
