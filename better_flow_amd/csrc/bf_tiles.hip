// bf_tiles.hip -- BASELINE config 4: a grid of independent local optimizers over one slice
// (the reference's multi-object task queue, dvs_flow.h:200-231: one OptimizerRolling per
// (events, model) pair).  MI355X-native form: events are counting-sorted by sensor tile and ONE
// WORK-GROUP PER TILE runs the whole OptimizerRolling::run loop on chip -- the tile's time /
// count image lives in LDS (LDS atomics for the scatter, same integer accumulators), the
// stencil and moment reduction are work-group local, and the same model_update() code as the
// global path advances the model.  No grid-wide synchronisation, no host round trip: 1024 tiles
// = 1024 concurrent gradient-descent loops.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdlib.h>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {

__device__ __forceinline__ int tile_of(uint32_t xy, const TileGrid& g) {
    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
    int tr = (int)(((long long)x * g.rows) / g.res_x), tc = (int)(((long long)y * g.cols) / g.res_y);
    tr = min(max(tr, 0), g.rows - 1);
    tc = min(max(tc, 0), g.cols - 1);
    return tr * g.cols + tc;
}

__global__ __launch_bounds__(kThreads) void k_tile_count(const uint32_t* __restrict__ xy, long long n, TileGrid g,
                                                         uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t s_h[];
    const int nt = g.rows * g.cols;
    for (int i = threadIdx.x; i < nt; i += kThreads) s_h[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads)
        atomicAdd(&s_h[tile_of(xy[i], g)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += kThreads)
        if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

__global__ __launch_bounds__(1024) void k_tile_scan(uint32_t* __restrict__ hist, int nt, uint32_t* __restrict__ start,
                                                    uint32_t* __restrict__ cursor) {
    __shared__ uint32_t s_sum[1024];
    const int tid = threadIdx.x;
    const int per = (nt + 1023) / 1024;
    uint32_t local = 0;
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < nt) local += hist[b];
    }
    s_sum[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = (tid >= off) ? s_sum[tid - off] : 0u;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[tid] - local;
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < nt) {
            start[b] = run;
            run += hist[b];
            cursor[b] = 0;
            hist[b] = 0;
        }
    }
    if (tid == 1023) start[nt] = s_sum[1023];
}

__global__ __launch_bounds__(kThreads) void k_tile_scatter(const uint32_t* __restrict__ xy, const int32_t* __restrict__ t,
                                                           const uint32_t* __restrict__ perm_in, long long n, TileGrid g,
                                                           const uint32_t* __restrict__ start, uint32_t* __restrict__ cursor,
                                                           uint32_t* __restrict__ oxy, int32_t* __restrict__ ot,
                                                           float2* __restrict__ op, uint32_t* __restrict__ operm) {
    extern __shared__ uint32_t s_u[];
    const int nt = g.rows * g.cols;
    uint32_t* s_cnt = s_u;
    uint32_t* s_base = s_u + nt;
    for (int i = threadIdx.x; i < nt; i += kThreads) s_cnt[i] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * kThreads * 4;
    uint32_t rank[4];
    int bin[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * kThreads + threadIdx.x;
        bin[k] = -1;
        if (i < n) {
            bin[k] = tile_of(xy[i], g);
            rank[k] = atomicAdd(&s_cnt[bin[k]], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += kThreads)
        if (s_cnt[i]) s_base[i] = start[i] + atomicAdd(&cursor[i], s_cnt[i]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * kThreads + threadIdx.x;
        if (bin[k] >= 0) {
            const uint32_t o = s_base[bin[k]] + rank[k];
            oxy[o] = xy[i];
            ot[o] = t[i];
            op[o] = make_float2(0.f, 0.f);   // Event::reset
            operm[o] = perm_in ? perm_in[i] : (uint32_t)i;
        }
    }
}

// One OptimizerRolling (set_cloud, set_time already applied, run) per work-group.
// THREADS: a tile is a few hundred to a few thousand events and ~1000 pixels, and the slice is done when
// its SLOWEST tile is (3047 iterations against a mean of 91 on the config-4 slice): what counts is the latency of one
// tile's iteration.  Measured: more waves per tile do not shorten it (256 / 512 / 1024 threads: 24.1 / 23.1 / 23.3 ms --
// the reduction across sixteen waves eats what the shorter loops give), so 256 it is.
// HS: scale / 2 when it is 0, 1 or 2 (the box sum's eighteen LDS reads per pixel are then issued together instead of one
// by one from a loop with run-time bounds: 4.1 -> us of a 7.8 us iteration), -1: any scale.
template <int THREADS, int HS>
__device__ __forceinline__ void tile_optimizer_body(const TileArgs& a, const int tile, unsigned long long* s_dyn) {
    __shared__ DevState s_st;
    __shared__ int s_box[4];
    const int tid = threadIdx.x;
    const uint32_t beg = a.tile_start[tile], end = a.tile_start[tile + 1];
    const int n = (int)(end - beg);
    unsigned long long* s_ts = s_dyn;                                         // sum of t (i64 as u64)
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_dyn + a.max_px);          // event count
    float* s_time = reinterpret_cast<float*>(s_cnt + a.max_px);               // time image

    // ---- OptimizerRolling::set_cloud + set_scale (optimizer_rolling.h:248-283) ----
    int xmin = INT_MAX, xmax = INT_MIN, ymin = INT_MAX, ymax = INT_MIN;
    for (uint32_t i = beg + tid; i < end; i += THREADS) {
        const uint32_t v = a.xy[i];
        const int x = (int)(v & 0xffffu), y = (int)(v >> 16);
        xmin = min(xmin, x); xmax = max(xmax, x);
        ymin = min(ymin, y); ymax = max(ymax, y);
    }
    xmin = wave_min(xmin); xmax = wave_max(xmax); ymin = wave_min(ymin); ymax = wave_max(ymax);
    if (tid == 0) { s_box[0] = INT_MAX; s_box[1] = INT_MIN; s_box[2] = INT_MAX; s_box[3] = INT_MIN; }
    __syncthreads();
    if ((tid & 63) == 0) {
        atomicMin(&s_box[0], xmin); atomicMax(&s_box[1], xmax);
        atomicMin(&s_box[2], ymin); atomicMax(&s_box[3], ymax);
    }
    __syncthreads();
    if (tid == 0) {
        DevState& st = s_st;
        st = a.states[tile];   // zero model / loop-control template prepared by the host
        const int s = a.scale;
        int x_min = a.seed_res_x, y_min = a.seed_res_y, x_max = 0, y_max = 0;   // :252-253
        if (n > 0) {
            x_max = max(x_max, s_box[1]); y_max = max(y_max, s_box[3]);
            x_min = min(x_min, s_box[0]); y_min = min(y_min, s_box[2]);
        }
        const int wsx = s * (x_max - x_min), wsy = s * (y_max - y_min);
        st.hot.scale = s;
        st.hot.wsx = wsx; st.hot.wsy = wsy;
        st.hot.R = wsx + s; st.hot.C = wsy + s;
        st.x_shift = -(double)((x_max - x_min) / 2 + x_min) * (double)s + (double)wsx / 2.0 + (double)(s / 2);
        st.y_shift = -(double)((y_max - y_min) / 2 + y_min) * (double)s + (double)wsy / 2.0 + (double)(s / 2);
        st.hot.x_sh = (int)st.x_shift;
        st.hot.y_sh = (int)st.y_shift;
        st.hot.tmin = 0;
        st.n_events = (uint32_t)n;
        // ---- guards of run() (optimizer_rolling.h:49-58), with run-time RES / minimum ----
        int rc = 0, done = 0;
        if ((st.hot.R < s * a.guard_res_x / 15) && (st.hot.C < s * a.guard_res_y / 15)) { rc = BF_SKIPPED; done = 1; }
        else if (n < a.min_events) { rc = BF_SKIPPED; done = 1; }
        else if (st.hot.R <= 0 || st.hot.C <= 0 || (long long)st.hot.R * st.hot.C > a.max_px) { rc = BF_ERR_CAPACITY; done = 1; }
        st.rc = rc;
        st.hot.done = done;
    }
    __syncthreads();
    const int R = s_st.hot.R, C = s_st.hot.C, P = R * C;
    const int s = a.scale, hsc = s / 2;
    // A tile's events live in registers for the whole loop (up to kTileUR per thread; a fuller tile streams the rest
    // from global memory as before): the loop body then touches global memory only to stream that rest.
    constexpr int kTileUR = 4096 / THREADS;
    uint32_t rxy[kTileUR];
    int32_t rt[kTileUR];
    float2 rp[kTileUR];
#pragma unroll
    for (int k = 0; k < kTileUR; ++k) {
        uint32_t i = beg + (uint32_t)(k * THREADS + tid);
        i = i < end ? i : (end > beg ? beg : 0u);
        rxy[k] = (end > beg) ? a.xy[i] : 0u;
        rt[k] = (end > beg) ? a.t[i] : 0;
        rp[k] = (end > beg) ? a.p[i] : make_float2(0.f, 0.f);
    }
    const uint32_t stream_beg = beg + (uint32_t)(kTileUR * THREADS);
    __shared__ unsigned long long s_rpart[kSumFields * (THREADS / 64)];
    const int x_sh = s_st.hot.x_sh, y_sh = s_st.hot.y_sh, wsx = s_st.hot.wsx, wsy = s_st.hot.wsy;
    const int hR = R / 2, hC = C / 2;
    // pixel index -> row by one multiply: (i + 0.5) / C is at least 0.5 / C away from an integer, the f32 error of
    // (i + 0.5) * (1 / C) is below 1e-5 for the image sizes that fit the LDS (i < 10^4, C < 10^3)
    const float rC = 1.0f / (float)(C > 0 ? C : 1);

#ifdef BF_TIMELINE
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TK(i_) do { if (tid == 0 && s_st.hot.it == 100 && n > 300 && n < 700) tk[i_] = wall_clock64(); } while (0)
#else
#define TK(i_) do { } while (0)
#endif
    // (the accumulator planes are cleared once here and then by the Scharr pass of every iteration, which is the first
    // phase after the box sums have read them: one pass over the pixels and one barrier less per iteration)
    for (int i = tid; i < P; i += THREADS) { s_ts[i] = 0; s_cnt[i] = 0; }
    __syncthreads();
    while (!s_st.hot.done) {
        TK(0);
        const WarpParams wp = s_st.hot.wp;
        const bool warp = s_st.hot.it > 0;
        // ---- warp (event.h:99-110,164-168) + point scatter (accel_lib.h:151-166) ----
        auto one_event = [&](uint32_t v, int32_t ti, float2& q) {
            const uint32_t fx = v & 0xffffu, fy = v >> 16;
            double pr_x = pr_from_p(fx, q.x), pr_y = pr_from_p(fy, q.y);
            if (warp) {
                double nx, ny;
                warp_products(wp, pr_x, pr_y, ti, q, nx, ny);
                pr_x = pr_from_p(fx, q.x);
                pr_y = pr_from_p(fy, q.y);
            }
            const int X = trunc_scatter(pr_x * (double)s + (double)x_sh);
            const int Y = trunc_scatter(pr_y * (double)s + (double)y_sh);
            if (!((X >= wsx + hsc) || (X < hsc) || (Y >= wsy + hsc) || (Y < hsc))) {
                atomicAdd(&s_ts[X * C + Y], (unsigned long long)(long long)ti);
                atomicAdd(&s_cnt[X * C + Y], 1u);
            }
        };
        // (opaque per iteration: keeps the compiler from hoisting per-event conversions out of the loop into registers)
#pragma unroll
        for (int k = 0; k < kTileUR; ++k) asm volatile("" : "+v"(rxy[k]), "+v"(rt[k]));
#pragma unroll
        for (int k = 0; k < kTileUR; ++k)
            if (beg + (uint32_t)(k * THREADS + tid) < end) one_event(rxy[k], rt[k], rp[k]);
        for (uint32_t i = stream_beg + tid; i < end; i += THREADS) {
            float2 q = a.p[i];
            one_event(a.xy[i], a.t[i], q);
            if (warp) a.p[i] = q;
        }
        TK(1);
        __syncthreads();
        TK(2);
        // ---- s x s box sum + normalise (accel_lib.h:160-175) ----
        for (int i = tid; i < P; i += THREADS) {
            const int r = (int)(((float)i + 0.5f) * rC), c = i - r * C;   // exact: see rC
            long long ts = 0;
            uint32_t cn = 0;
            if (HS >= 0) {
#pragma unroll
                for (int dr = -HS; dr <= HS; ++dr)
#pragma unroll
                    for (int dc = -HS; dc <= HS; ++dc) {
                        const int rr = r + dr, cc = c + dc;
                        const bool in = rr >= 0 && rr < R && cc >= 0 && cc < C;
                        const int k = in ? rr * C + cc : i;   // (a valid address either way: the load is unconditional)
                        const long long tv = (long long)s_ts[k];
                        const uint32_t cv = s_cnt[k];
                        ts += in ? tv : 0ll;
                        cn += in ? cv : 0u;
                    }
            } else {
                for (int dr = -hsc; dr <= hsc; ++dr)
                    for (int dc = -hsc; dc <= hsc; ++dc) {
                        const int rr = r + dr, cc = c + dc;
                        if (rr >= 0 && rr < R && cc >= 0 && cc < C) {
                            ts += (long long)s_ts[rr * C + cc];
                            cn += s_cnt[rr * C + cc];
                        }
                    }
            }
            s_time[i] = time_from_sums(cn, ts, 0);
        }
        __syncthreads();
        TK(3);
        // ---- gated Scharr (accel_lib.h:513-615) + centre of mass + moments (object_model.cpp) ----
        Sums sm;
        sums_zero(sm);
        for (int i = tid; i < P; i += THREADS) {
            const int r = (int)(((float)i + 0.5f) * rC), c = i - r * C;   // exact: see rC
            const float ctr = s_time[i];
            s_ts[i] = 0; s_cnt[i] = 0;   // for the next iteration's scatter (nobody reads them any more in this one)
            if (!valid_px(ctr)) continue;
            float gx = 0.f, gy = 0.f;
            if (r >= 1 && r < R - 1 && c >= 1 && c < C - 1) {
                const float* tp = &s_time[i];
                const float t00 = tp[-C - 1], t10 = tp[-1], t20 = tp[C - 1];
                const float t01 = tp[-C], t21 = tp[C];
                const float t02 = tp[-C + 1], t12 = tp[1], t22 = tp[C + 1];
                if (valid_px(t00) && valid_px(t10) && valid_px(t20) && valid_px(t01) && valid_px(t21) &&
                    valid_px(t02) && valid_px(t12) && valid_px(t22)) {
                    float dx = 0.f, dy = 0.f;
                    dx = dx + t00 * 3.f;   dy = dy + t00 * 3.f;
                    dx = dx + t10 * 0.f;   dy = dy + t10 * 10.f;
                    dx = dx + t20 * -3.f;  dy = dy + t20 * 3.f;
                    dx = dx + t01 * 10.f;  dy = dy + t01 * 0.f;
                    dx = dx + ctr * 0.f;   dy = dy + ctr * 0.f;
                    dx = dx + t21 * -10.f; dy = dy + t21 * 0.f;
                    dx = dx + t02 * 3.f;   dy = dy + t02 * -3.f;
                    dx = dx + t12 * 0.f;   dy = dy + t12 * -10.f;
                    dx = dx + t22 * -3.f;  dy = dy + t22 * -3.f;
                    gx = dx;
                    gy = dy;
                }
            }
            const int ci = r - hR, cj = c - hC;
            sm.n += 1; sm.sci += ci; sm.scj += cj;
            const double gxd = (double)gx, gyd = (double)gy;
            sm.sgx += gxd; sm.sgy += gyd;
            sm.sigx += (double)ci * gxd; sm.sigy += (double)ci * gyd;
            sm.sjgx += (double)cj * gxd; sm.sjgy += (double)cj * gyd;
        }
        TK(4);
        const Sums tot = block_reduce_sums<THREADS>(sm, s_rpart, tid);   // DPP wave totals + one LDS hop
        TK(5);
        if (tid < 64) {   // update_accumulators + iteration_step glue + run() control (state in LDS), by one wave
            model_update_wave(&s_st, sums_lane_word(tot, tid), tid, 1);
            if (tid == 0) model_update_rest(&s_st, nullptr, 0);
        }
        TK(6);
        __syncthreads();
#ifdef BF_TIMELINE
        if (tid == 0 && tk[0] && s_st.hot.it == 101)
            printf("tile %d n %d P %d: events %llu barrier %llu boxsum %llu scharr %llu reduce %llu update %llu (x10 ns)\n", tile, n, P,
                   tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5]);
#endif
    }
    // ---- final state: the last project_4param_reinit of the loop, n for compute_uv ----
    const WarpParams wp = s_st.hot.wp;
    const bool ran = s_st.rc == 0 && s_st.hot.it > 0;
    auto final_event = [&](uint32_t i, uint32_t v, int32_t ti, float2 q) {
        double nx = 0.0, ny = 0.0;
        if (ran) {
            const double pr_x = pr_from_p(v & 0xffffu, q.x), pr_y = pr_from_p(v >> 16, q.y);
            warp_products(wp, pr_x, pr_y, ti, q, nx, ny);
        }
        a.p[i] = q;
        a.nxny[a.perm[i]] = make_double2(nx, ny);
    };
#pragma unroll
    for (int k = 0; k < kTileUR; ++k) {
        const uint32_t i = beg + (uint32_t)(k * THREADS + tid);
        if (i < end) final_event(i, rxy[k], rt[k], rp[k]);
    }
    for (uint32_t i = stream_beg + tid; i < end; i += THREADS) final_event(i, a.xy[i], a.t[i], a.p[i]);
    if (tid == 0) a.states[tile] = s_st;
}

template <int THREADS, int HS>
__global__ __launch_bounds__(THREADS) void k_tile_optimizer(TileArgs a) {
    extern __shared__ unsigned long long s_dyn[];
    tile_optimizer_body<THREADS, HS>(a, blockIdx.x, s_dyn);
}

// The tile grids of SEVERAL slices in one launch (bf_run_tiles_many): a grid of resident work-groups, each claiming
// (slice, tile) pairs from one device counter until none are left.  A slice's grid alone lasts as long as its slowest tile
// (3047 iterations against a mean of 91 on the config-4 slice) while its bulk is done in under a millisecond; here the
// stragglers of slices 0 .. k run under the bulk of slices k + 1 ..., in ONE stream -- no extra hardware queues, no
// environment variable (a grid per slice context needed GPU_MAX_HW_QUEUES=16 before the first HIP call to get past four
// grids in flight).  Items are claimed slice by slice, a slice's tiles in index order; an item is the same computation as
// a work-group of k_tile_optimizer -- the same function, the same bits.
template <int THREADS, int HS>
__global__ __launch_bounds__(THREADS) void k_tile_optimizer_many(const TileArgs* __restrict__ slices, int nslices, int ntiles,
                                                                 uint32_t* __restrict__ counter) {
    extern __shared__ unsigned long long s_dyn[];
    __shared__ uint32_t s_item;
    const uint32_t total = (uint32_t)nslices * (uint32_t)ntiles;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(counter, 1u);
        __syncthreads();
        const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_item);
        if (item >= total) return;
        const uint32_t sl = item / (uint32_t)ntiles;
        const TileArgs a = slices[sl];   // (uniform index: scalar loads)
        tile_optimizer_body<THREADS, HS>(a, (int)(item - sl * (uint32_t)ntiles), s_dyn);
        __syncthreads();   // (the item's LDS -- state, planes, s_item -- is free again)
    }
}

__global__ void k_fill_states(DevState* states, DevState tmpl, int nt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt) states[i] = tmpl;
}

void launch_fill_states(DevState* states, const DevState& tmpl, int nt, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_states, dim3((nt + 63) / 64), dim3(64), 0, s, states, tmpl, nt);
}

void launch_tile_sort(const uint32_t* xy, const int32_t* t, const uint32_t* perm_in, long long n, const TileGrid& g,
                      uint32_t* hist, uint32_t* start, uint32_t* cursor, uint32_t* oxy, int32_t* ot, float2* op,
                      uint32_t* operm, hipStream_t s) {
    const int nt = g.rows * g.cols;
    long long blocks = (n + kThreads * 8 - 1) / (kThreads * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_tile_count, dim3((unsigned)blocks), dim3(kThreads), (size_t)nt * 4, s, xy, n, g, hist);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, hist, nt, start, cursor);
    if (n > 0) {
        const long long per = (long long)kThreads * 4;
        hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)((n + per - 1) / per)), dim3(kThreads), (size_t)nt * 8, s, xy, t,
                           perm_in, n, g, start, cursor, oxy, ot, op, operm);
    }
}

// (256 threads per tile optimizer; 512 / 1024: measured, no gain -- experiments/rounds_1_to_3.md)
constexpr int kTileThreads = 256;

template <class K>
static int tile_kernel_lds(K k) {
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(k)) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - (int)at.sharedSizeBytes) != hipSuccess) return -1;
    return 0;
}

int launch_tile_optimizer(const TileArgs& a, int ntiles, hipStream_t s) {
    const size_t lds = (size_t)a.max_px * (8 + 4 + 4);
    const int hs = a.scale / 2;
    void (*k)(TileArgs) = hs == 0 ? k_tile_optimizer<kTileThreads, 0>
                        : (hs == 1 ? k_tile_optimizer<kTileThreads, 1> : (hs == 2 ? k_tile_optimizer<kTileThreads, 2> : k_tile_optimizer<kTileThreads, -1>));
    if (tile_kernel_lds(k) != 0) return -1;
    hipLaunchKernelGGL(k, dim3(ntiles), dim3(kTileThreads), lds, s, a);
    return 0;
}

// `slices`: nslices TileArgs in DEVICE memory (same scale and max_px in all); `counter`: one zeroed word.
int launch_tile_optimizer_many(const TileArgs* slices, int nslices, int ntiles, int scale, int max_px, uint32_t* counter, int n_cus,
                               hipStream_t s) {
    const size_t lds = (size_t)max_px * (8 + 4 + 4);
    const int hs = scale / 2;
    void (*k)(const TileArgs*, int, int, uint32_t*) =
        hs == 0 ? k_tile_optimizer_many<kTileThreads, 0>
                : (hs == 1 ? k_tile_optimizer_many<kTileThreads, 1> : (hs == 2 ? k_tile_optimizer_many<kTileThreads, 2> : k_tile_optimizer_many<kTileThreads, -1>));
    if (tile_kernel_lds(k) != 0) return -1;
    // as many work-groups as the device holds at once (more would only queue and find the counter exhausted)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), kTileThreads, lds) != hipSuccess || per_cu < 1)
        per_cu = 1;
    long long grid = (long long)per_cu * (n_cus > 0 ? n_cus : 256);
    const long long items = (long long)nslices * ntiles;
    if (grid > items) grid = items;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kTileThreads), lds, s, slices, nslices, ntiles, counter);
    return 0;
}

}  // namespace bf
