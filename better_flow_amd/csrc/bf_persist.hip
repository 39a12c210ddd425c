// bf_persist.hip -- the whole gradient-descent loop of OptimizerRolling::run()
// (optimizer_rolling.h:48-125,305-347) as ONE cooperative launch for gfx950.
//
// Why: with one launch per stage an iteration is a chain of two kernels whose boundaries
// (end-of-kernel write-back, dispatch, first scalar loads) and whose single reducing work-group
// cost more than the work itself (~30 us per iteration for ~6 us of HBM traffic at 1M events).
// Here every image tile of the tile-binned scatter (bf_binned.hip) is owned by one resident
// work-group for the whole run:
//
//   * its events stay in REGISTERS (xy, t and the two f32 products of event.h:164-168; up to
//     kPersistUR per thread, the rest of an over-full bin is streamed from global memory);
//   * phase A: warp + LDS scatter exactly as k_bin_warp_scatter, then only the RING of the LDS
//     tile that neighbouring tiles need (width D + H) goes to the tile's slab;
//   * grid barrier; phase B: add the neighbours' rings (and the overflow planes if anything took
//     the overflow path) -> s x s box sum -> time image in LDS -> gated Scharr + moment sums; the
//     64 x 64 core is processed as the four 16 x 64 tiles k_stencil_binned would use, in the same
//     order, so the partial sums are BIT-IDENTICAL to the multi-kernel path;
//   * grid barrier; phase C: EVERY work-group reduces all partials in the fixed order of
//     stencil_tail and runs the model / loop update on its own LDS copy of the state -- no
//     broadcast, no third barrier; work-group 0 records the trace and writes the state back.
//
// The launch returns when the loop is done or the update asks for a re-bin (events drifted
// towards the edge of their tiles); the host runs the device-gated re-bin and launches again.
// All cross-work-group data uses write-through stores and agent-scope loads (cdna_hip_
// programming.md, Guideline 16); barriers are two-level arrival counters that only count up.
#include <hip/hip_runtime.h>
#include <limits.h>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {

namespace {

constexpr int kPTS = 64;    // tile size the kernel is written for (== kTileC, 4 x kTileR)

// Grid barrier `epoch` (1, 2, ...) of this launch.  Returns false if it timed out (a work-group
// of the grid never arrived: the launch was not co-resident) -- the caller leaves the kernel.
__device__ __forceinline__ bool grid_barrier(unsigned int* bar, int nwg, int wg, unsigned int epoch, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores have landed
    __syncthreads();
    if (threadIdx.x == 0) {
        const int grp = wg % kTicketGroups;
        const unsigned int n_groups = nwg < kTicketGroups ? nwg : kTicketGroups;
        const unsigned int grp_size = nwg / kTicketGroups + (grp < nwg % kTicketGroups ? 1 : 0);
        const unsigned int v = __hip_atomic_fetch_add(&bar[16 * (1 + grp)], 1u, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
        if (v + 1u == epoch * grp_size)
            __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = epoch * n_groups;
        int ok = 1;
        unsigned int spins = 0;
        unsigned long long t0 = 0;
        while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                const unsigned long long now = wall_clock64();   // 100 MHz
                if (t0 == 0) t0 = now;
                else if (now - t0 > 100000000ull) { ok = 0; break; }   // 1 s
            }
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

__device__ __forceinline__ void tlp(unsigned long long* tl, int it, int wg, int nwg, int slot) {
#ifdef BF_TIMELINE
    if (!tl || it >= kTlLaunches || threadIdx.x != 0) return;
    int g = -1;
    if (wg == 0) g = 0;
    else if (wg == nwg / 2) g = 1;
    if (g < 0) return;
    tl[((size_t)it * 2 + g) * 16 + slot] = wall_clock64();
#else
    (void)tl; (void)it; (void)wg; (void)nwg; (void)slot;
#endif
}

// Everything one event needs from the loop state.
struct ScatterCtx {
    WarpParams wp;
    long long tmin;
    int s, x_sh, y_sh, hsc, wsx, wsy, C, tbits, X0, Y0, L, bin_ok;
};

// Warp (event.h:100-108,164-168) + scatter (accel_lib.h:154-158) of one event -- the arithmetic
// of k_bin_warp_scatter.  q holds the event's two f32 products and is updated in place.
template <bool WARP>
__device__ __forceinline__ void warp_scatter_one(const ScatterCtx& c, uint32_t v, int32_t ti, float2& q,
                                                 unsigned long long* s_tile, unsigned long long* ovf_plane,
                                                 uint32_t* ovf_cplane, uint32_t& n_ovf) {
    const uint32_t fx = v & 0xffffu, fy = v >> 16;
    double pr_x = pr_from_p(fx, q.x);
    double pr_y = pr_from_p(fy, q.y);
    if (WARP) {
        double nx, ny;
        warp_products(c.wp, pr_x, pr_y, ti, q, nx, ny);
        pr_x = pr_from_p(fx, q.x);
        pr_y = pr_from_p(fy, q.y);
    }
    const int X = trunc_x86(pr_x * (double)c.s + (double)c.x_sh);
    const int Y = trunc_x86(pr_y * (double)c.s + (double)c.y_sh);
    if (!((X >= c.wsx + c.hsc) || (X < c.hsc) || (Y >= c.wsy + c.hsc) || (Y < c.hsc))) {
        const unsigned long long dt = (unsigned long long)((long long)ti - c.tmin);
        const int lx = X - c.X0, ly = Y - c.Y0;
        if (c.bin_ok && lx >= 0 && lx < c.L && ly >= 0 && ly < c.L) {
            atomicAdd(&s_tile[lx * c.L + ly], (1ull << c.tbits) + dt);
        } else {   // drifted out of this bin's tile: exact, slow path
            const size_t kk = (size_t)X * (size_t)c.C + (size_t)Y;
            atomicAdd(&ovf_plane[kk], dt);
            atomicAdd(&ovf_cplane[kk], 1u);
            ++n_ovf;
        }
    }
}

}  // namespace

template <int THREADS, int HS>
__global__ __launch_bounds__(THREADS) void k_persist(PersistArgs a) {
    constexpr int H = HS + 1;              // halo of the merged region: box sum (HS) + Scharr (1)
    constexpr int TS = kPTS;
    constexpr int NW = TS + 2 * H;         // merged accumulator region
    constexpr int TW = TS + 2;             // time tile (halo 1)
    constexpr int SUBS = THREADS / kThreads;   // 256-thread sub-groups, one 16 x 64 stencil tile each
    constexpr int UR = kPersistUR * 1024 / THREADS;
    static_assert(SUBS == 4 || SUBS == 2, "1024 or 512 threads");
    extern __shared__ unsigned long long s_mem[];
    const BinGrid g = a.g;
    const int L = g.L, LL = L * L, D = g.D;
    // LDS: [s_tile LL u64 | s_ts NW^2 u64 | s_cnt NW^2 u32 | s_time TW^2 f32 | s_rpart | s_state | flags]
    unsigned long long* s_tile = s_mem;
    unsigned long long* s_ts = s_tile + ((LL + 1) & ~1);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_ts + NW * NW);
    float* s_time = reinterpret_cast<float*>(s_cnt + NW * NW);
    unsigned long long* s_rpart = reinterpret_cast<unsigned long long*>(s_time + ((TW * TW + 1) & ~1));
    DevState* s_state = reinterpret_cast<DevState*>(s_rpart + kSumFields * (THREADS / 64));
    int* s_flag = reinterpret_cast<int*>(s_state + 1);

    const int b = blockIdx.x, nwg = gridDim.x, tid0 = threadIdx.x;
    const int tid = tid0;
    const int mybr = b / g.nbc, mybc = b - mybr * g.nbc;
    if (tid == 0) *s_state = *a.st;
    __syncthreads();
    if (s_state->hot.done || s_state->hot.need_rebin) return;

    const uint32_t beg = a.bin_start[b], end = a.bin_start[b + 1];
    const EvSetPtrs ev = (s_state->hot.cs ^ s_state->hot.flip) ? a.sets.s[1] : a.sets.s[0];
    // resident events
    uint32_t rxy[UR];
    int32_t rt[UR];
    float2 rp[UR];
#pragma unroll
    for (int k = 0; k < UR; ++k) {
        // (unconditional loads from a clamped index: slots past the bin's end are never used)
        uint32_t i = beg + (uint32_t)(k * THREADS + tid);
        i = i < end ? i : beg;
        rxy[k] = ev.xy[i];
        rt[k] = ev.t[i];
        rp[k] = ev.p[i];
    }
    const uint32_t stream_beg = beg + (uint32_t)(UR * THREADS);   // events beyond the resident ones

    ScatterCtx sc;
    sc.s = s_state->hot.scale; sc.x_sh = s_state->hot.x_sh; sc.y_sh = s_state->hot.y_sh;
    sc.hsc = s_state->hot.scale / 2;
    sc.wsx = s_state->hot.wsx; sc.wsy = s_state->hot.wsy; sc.C = s_state->hot.C;
    sc.tmin = s_state->hot.tmin;
    sc.X0 = mybr * TS - D; sc.Y0 = mybc * TS - D; sc.L = L;
    const int R = s_state->hot.R, C = s_state->hot.C;
    const int r0 = mybr * TS, c0 = mybc * TS;
    const int nblk = a.gx * a.gy;
    const long long tmin = sc.tmin;

    int cur = a.cur0;
    uint32_t prev_ovf = s_state->hot.ovf_cnt[0];
    if (cur == 0) prev_ovf = s_state->hot.ovf_cnt[1];   // dirtiness of plane buffer cur ^ 1
    bool nowarp = a.first_nowarp != 0;
    unsigned int epoch = 0;
    bool ok = true;

    for (int li = 0; li < a.max_iters; ++li) {
        const int it = s_state->hot.it;
        // Opaque per-iteration copy of the thread index: almost all index / address arithmetic below is
        // loop invariant, and hoisted out of the iteration loop it would stay live for the whole run
        // (hundreds of bytes of spills per lane).
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        tlp(a.tl, it, b, nwg, 0);
        // ---------------- phase A: warp + scatter into the LDS tile ----------------
        {
            ulonglong2* z = reinterpret_cast<ulonglong2*>(s_tile);
            for (int i = tid; i < LL / 2; i += THREADS) z[i] = make_ulonglong2(0ull, 0ull);
        }
        sc.wp = s_state->hot.wp;
        sc.tbits = s_state->hot.bin_tbits;
        sc.bin_ok = s_state->hot.bin_ok;
        // (static indices only: a runtime index would push the argument struct into scratch memory)
        unsigned long long* ovf_plane = cur ? a.ovf_plane[1] : a.ovf_plane[0];
        uint32_t* ovf_cplane = cur ? a.ovf_cplane[1] : a.ovf_cplane[0];
        unsigned int* ovf_ctr = cur ? &a.st->hot.ovf_cnt[1] : &a.st->hot.ovf_cnt[0];
        unsigned int* ovf_ctr_other = cur ? &a.st->hot.ovf_cnt[0] : &a.st->hot.ovf_cnt[1];
        uint32_t n_ovf = 0;
        // The resident events are made opaque once per iteration: otherwise the compiler hoists every
        // loop-invariant conversion of every event (f64 fr, f32 t, t - tmin: ~7 more registers per
        // event) out of the iteration loop and the kernel spills.
#pragma unroll
        for (int k = 0; k < UR; ++k) asm volatile("" : "+v"(rxy[k]), "+v"(rt[k]));
        __syncthreads();
        if (nowarp) {
#pragma unroll
            for (int k = 0; k < UR; ++k)
                if (beg + (uint32_t)(k * THREADS + tid) < end) {
                    warp_scatter_one<false>(sc, rxy[k], rt[k], rp[k], s_tile, ovf_plane, ovf_cplane, n_ovf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            for (uint32_t i = stream_beg + tid; i < end; i += THREADS) {
                float2 q = ev.p[i];
                warp_scatter_one<false>(sc, ev.xy[i], ev.t[i], q, s_tile, ovf_plane, ovf_cplane, n_ovf);
            }
        } else {
#pragma unroll
            for (int k = 0; k < UR; ++k)
                if (beg + (uint32_t)(k * THREADS + tid) < end) {
                    warp_scatter_one<true>(sc, rxy[k], rt[k], rp[k], s_tile, ovf_plane, ovf_cplane, n_ovf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            for (uint32_t i = stream_beg + tid; i < end; i += THREADS) {
                float2 q = ev.p[i];
                warp_scatter_one<true>(sc, ev.xy[i], ev.t[i], q, s_tile, ovf_plane, ovf_cplane, n_ovf);
                ev.p[i] = q;   // private to this work-group
            }
        }
        if (n_ovf) atomicAdd(ovf_ctr, n_ovf);
        tlp(a.tl, it, b, nwg, 1);
        __syncthreads();
        {   // the ring of width D + H is all a neighbour ever reads of this tile
            unsigned long long* dst = a.slabs + (size_t)b * (size_t)LL;
            const int lo = D + H, hi = L - D - H;
            for (int i = tid; i < LL; i += THREADS) {
                const int lx = i / L, ly = i - lx * L;
                if (lx < lo || lx >= hi || ly < lo || ly >= hi)
                    __hip_atomic_store(&dst[i], s_tile[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        tlp(a.tl, it, b, nwg, 2);
        if (!(ok = grid_barrier(a.bar, nwg, b, ++epoch, s_flag))) break;
        tlp(a.tl, it, b, nwg, 3);

        // ---------------- phase B: merge, box sum, time image, Scharr + moments ----------------
        const uint32_t ovf_now = __hip_atomic_load(ovf_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int bt = sc.tbits;
        const unsigned long long bm = (1ull << bt) - 1ull;
        // (opaque copy of the thread index: the pixel -> slab address arithmetic below is loop
        // invariant, and hoisted out of the iteration loop it would hold ~60 registers for the whole run)
        const int tid_b = tid;
        {
            constexpr int NPX = (NW * NW + THREADS - 1) / THREADS;
            unsigned long long w[NPX][4];
            unsigned long long ov[NPX];
            uint32_t oc[NPX];
#pragma unroll
            for (int c = 0; c < NPX; ++c) {
                const int idx = tid_b + c * THREADS;
                const int mr = idx / NW, mc = idx - mr * NW;
                const int gr = r0 - H + mr, gc = c0 - H + mc;
                const bool in = idx < NW * NW && gr >= 0 && gr < R && gc >= 0 && gc < C;
                const int brl = max(gr - D, 0) >> g.lg, brh = min((gr + D) >> g.lg, g.nbr - 1);
                const int bcl = max(gc - D, 0) >> g.lg, bch = min((gc + D) >> g.lg, g.nbc - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int br = (q & 2) ? brh : brl, bc = (q & 1) ? bch : bcl;
                    const bool use = in && (!(q & 2) || brh > brl) && (!(q & 1) || bch > bcl) &&
                                     !(br == mybr && bc == mybc);
                    const int lx = gr - ((br << g.lg) - D), ly = gc - ((bc << g.lg) - D);
                    w[c][q] = use ? __hip_atomic_load(&a.slabs[(size_t)(br * g.nbc + bc) * (size_t)LL + (size_t)(lx * L + ly)],
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : 0ull;
                }
                const bool o = in && ovf_now != 0;
                ov[c] = o ? __hip_atomic_load(&ovf_plane[(size_t)gr * C + gc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                oc[c] = o ? __hip_atomic_load(&ovf_cplane[(size_t)gr * C + gc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            }
#pragma unroll
            for (int c = 0; c < NPX; ++c) {
                const int idx = tid_b + c * THREADS;
                if (idx < NW * NW) {
                    const int mr = idx / NW, mc = idx - mr * NW;
                    const unsigned long long own = s_tile[(D - H + mr) * L + (D - H + mc)];
                    unsigned long long ts = ov[c] + (own & bm);
                    uint32_t cn = oc[c] + (uint32_t)(own >> bt);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ts += w[c][q] & bm;
                        cn += (uint32_t)(w[c][q] >> bt);
                    }
                    s_ts[idx] = ts;
                    s_cnt[idx] = cn;
                }
            }
        }
        if (prev_ovf) {   // plane buffer cur ^ 1 took overflow events last iteration: clear this tile's share
            unsigned long long* zp = cur ? a.ovf_plane[0] : a.ovf_plane[1];
            uint32_t* zc = cur ? a.ovf_cplane[0] : a.ovf_cplane[1];
            for (int i = tid; i < TS * TS; i += THREADS) {
                const int gr = r0 + i / TS, gc = c0 + (i & (TS - 1));
                if (gr < R && gc < C) {
                    __hip_atomic_store(&zp[(size_t)gr * C + gc], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&zc[(size_t)gr * C + gc], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (b == 0 && tid == 0)
            __hip_atomic_store(ovf_ctr_other, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tlp(a.tl, it, b, nwg, 4);
        __syncthreads();
        for (int idx = tid; idx < TW * TW; idx += THREADS) {
            const int tr = idx / TW, tc = idx - tr * TW;
            const int gr = r0 - 1 + tr, gc = c0 - 1 + tc;
            float tv = 0.f;
            if (gr >= 0 && gr < R && gc >= 0 && gc < C) {
                // s x s box sum == the s x s splat of accel_lib.h:160-165 on integer planes
                unsigned long long acc = 0;
                uint32_t cacc = 0;
#pragma unroll
                for (int da = 0; da <= 2 * HS; ++da)
#pragma unroll
                    for (int db = 0; db <= 2 * HS; ++db) {
                        acc += s_ts[(tr + da) * NW + (tc + db)];
                        cacc += s_cnt[(tr + da) * NW + (tc + db)];
                    }
                tv = time_from_sums(cacc, (long long)acc, tmin);
            }
            s_time[idx] = tv;
        }
        __syncthreads();
        tlp(a.tl, it, b, nwg, 5);
        {
            const int sub = tid / kThreads, tsub = tid - sub * kThreads;
            Sums sm;
            sums_zero(sm);
            const int hR = R / 2, hC = C / 2;
#pragma unroll
            for (int rep = 0; rep < 4 / SUBS; ++rep) {
                const int tile = sub + rep * SUBS;   // 16 x 64 tile `tile` of the 64 x 64 core
#pragma unroll
                for (int k = 0; k < (kTileR * kTileC) / kThreads; ++k) {
                    const int pidx = tsub + k * kThreads;
                    const int lr = pidx / kTileC, lc = pidx - lr * kTileC;
                    const int cr = tile * kTileR + lr;
                    const int gr = r0 + cr, gc = c0 + lc;
                    if (gr < R && gc < C) {
                        float gx, gy;
                        stencil_px<TW>(&s_time[(cr + 1) * TW + (lc + 1)], gr, gc, R, C, hR, hC, sm, gx, gy);
                    }
                }
                const Sums blk = block_reduce_sums<kThreads>(sm, s_rpart + sub * kSumFields * (kThreads / 64), tsub);
                const int ty = mybr * (TS / kTileR) + tile;
                if (tsub == 0 && ty < a.gy) publish_partial(a.partials, nblk, ty * a.gx + mybc, blk);
                if (rep + 1 < 4 / SUBS) {
                    sums_zero(sm);
                    __syncthreads();
                }
            }
        }
        tlp(a.tl, it, b, nwg, 6);
        if (!(ok = grid_barrier(a.bar, nwg, b, ++epoch, s_flag))) break;
        tlp(a.tl, it, b, nwg, 7);

        // ---------------- phase C: every work-group reduces and updates its own copy ----------------
        {
            const int sub = tid / kThreads, tsub = tid - sub * kThreads;
            Sums acc;
            sums_zero(acc);
            if (sub == 0) acc = gather_partials(a.partials, nblk, tsub);
            tlp(a.tl, it, b, nwg, 8);
            const Sums tot = block_reduce_sums<kThreads>(acc, s_rpart + sub * kSumFields * (kThreads / 64), tsub);
            if (tid == 0) {
                if (cur) { s_state->hot.ovf_cnt[1] = ovf_now; } else { s_state->hot.ovf_cnt[0] = ovf_now; }
                model_update_local(s_state, tot, b == 0 ? a.trace : nullptr, 1, cur);
            }
        }
        __syncthreads();
        tlp(a.tl, it, b, nwg, 9);
        prev_ovf = ovf_now;
        cur ^= 1;
        nowarp = false;
        if (s_state->hot.done || s_state->hot.need_rebin) break;
    }

    // ---------------- exit: products back to global memory, state by work-group 0 ----------------
#pragma unroll
    for (int k = 0; k < UR; ++k) {
        const uint32_t i = beg + (uint32_t)(k * THREADS + tid);
        if (i < end) ev.p[i] = rp[k];
    }
    if (!ok && tid == 0) __hip_atomic_store(&a.bar[16 * (kTicketGroups + 2)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b == 0 && tid == 0 && ok) *a.st = *s_state;
}

// ---------------------------------------------------------------------------------------
size_t persist_lds_bytes(const BinGrid& g, int scale, int threads) {
    const int H = scale / 2 + 1, NW = kPTS + 2 * H, TW = kPTS + 2;
    const size_t LL = ((size_t)g.L * g.L + 1) & ~(size_t)1;
    size_t bytes = LL * 8 + (size_t)NW * NW * 8 + (size_t)NW * NW * 4 + (((size_t)TW * TW + 1) & ~(size_t)1) * 4 +
                   (size_t)kSumFields * (threads / 64) * 8 + sizeof(DevState) + 64;
    return bytes;
}

template <int T, int HS>
static const void* persist_fn() { return reinterpret_cast<const void*>(&k_persist<T, HS>); }

static const void* persist_pick(int threads, int hs) {
    if (threads >= 1024) {
        switch (hs) {
            case 0: return persist_fn<1024, 0>();
            case 1: return persist_fn<1024, 1>();
            case 2: return persist_fn<1024, 2>();
            case 3: return persist_fn<1024, 3>();
            default: return persist_fn<1024, 4>();
        }
    }
    switch (hs) {
        case 0: return persist_fn<512, 0>();
        case 1: return persist_fn<512, 1>();
        case 2: return persist_fn<512, 2>();
        case 3: return persist_fn<512, 3>();
        default: return persist_fn<512, 4>();
    }
}

// Largest co-resident grid of the kernel for this geometry (0: cannot run), raising the dynamic
// LDS limit on the way.
int persist_max_groups(const BinGrid& g, int scale, int threads, int device) {
    if (g.TS != kPTS || g.TSR != kPTS || g.D < scale / 2 + 1 || g.D > g.TS / 2) return 0;   // square 64 x 64 tiles only
    const size_t lds = persist_lds_bytes(g, scale, threads);
    if (lds > 160 * 1024) return 0;
    int coop = 0, cus = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) != hipSuccess || !coop) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    const void* fn = persist_pick(threads, scale / 2);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess) return 0;
    return per_cu * cus;
}

hipError_t launch_persist(const PersistArgs& a, int scale, int threads, hipStream_t s) {
    PersistArgs args = a;
    void* params[] = {&args};
    const size_t lds = persist_lds_bytes(a.g, scale, threads);
    return hipLaunchCooperativeKernel(persist_pick(threads, scale / 2), dim3(a.g.nbins), dim3(threads), params,
                                      (unsigned int)lds, s);
}

}  // namespace bf
