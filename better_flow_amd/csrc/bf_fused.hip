// bf_fused.hip -- the one-kernel iteration (k_fused_pass): warp + scatter + stencil + moments of one image tile per work-group.
#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <limits.h>
#include <cstdlib>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {

// ---- the one-kernel iteration -----------------------------------------------------------------------------------------
// K1 and K3 in ONE launch for a slice context that has the GPU to itself (the latency regime: one iteration is a chain of
// two dependent launches there, ~20 us at 346x260 however few events the slice holds).  One work-group per image tile of
// TSR x 64 scaled pixels (TSR = 16 NSUB):
//   head     the pending model / loop update, by every work-group for itself (as k_bin_warp_scatter);
//   scatter  the tile's own events AND those of the neighbouring tiles' edge strips that face it (FusedTab: ten ranges of
//            the arrays sorted by (tile, zone)) are warped and added to an LDS tile with halo H = scale / 2 + 1.  Only the
//            owner of an event stores its new products -- into the OTHER product array (EvSetPtrs::p2, hot.pp), so that
//            the neighbours read the previous positions whatever the order the work-groups run in;
//   stencil  each 256-thread sub-group takes one 16 x 64 sub-tile -- the tile of k_stencil_binned, same thread -> pixel
//            mapping, same reduction tree: the f64 partial of a sub-tile, and with it the exact accumulators, carry the
//            bits of the two-kernel loop.
// No slabs, no overflow planes.  Exactness rests on every event that lands inside a tile's halo window being in that
// tile's ranges: true while no event has moved more than D since the sort.  The owner of an event CHECKS that from the
// zone it was sorted into and where it lands now; a violation raises `lost`, the next pass (or the re-bin, whichever comes
// first) sees it, the sums of that pass are dropped, and the pass is repeated on fresh bins (hot.redo) before the update
// runs -- late, never wrong.  The predictive re-bin (drift_limit) keeps that path rare.
// Packing: the scan sizes count << tbits | time sum for the fullest list; when even that does not fit 64 bits
// (hot.bin_ok == 0) the tile keeps separate u64 time sums and u32 counts.
// (pre_*: what the head's loads go through, as leading scalar arguments the command processor preloads -- see k_bin_warp_scatter)
template <int HS, int NSUB, int U>
__global__ __launch_bounds__(256 * NSUB) void k_fused_pass(const uint32_t* __restrict__ pre_ftab, const DevState* pre_st_in, MomentAcc* pre_acc_in,
                                                           const uint32_t* pre_lost, int pre_j, FusedArgs a) {
    constexpr int THREADS = 256 * NSUB;
    constexpr int TR = kTileR, TC = kTileC;
    constexpr int H = HS + 1;
    constexpr int TSR = TR * NSUB;
    constexpr int AR = TSR + 2 * H, AC = TC + 2 * H;   // the LDS tile
    constexpr int PC = AC;
    constexpr int TH = TR + 2, TW = TC + 2;
    extern __shared__ unsigned long long s_dyn[];       // (above 64 KiB for the 64-row tile: dynamic, see launch_fused_pass)
    unsigned long long* const s_acc = s_dyn;                                         // [AR * AC]
    float (*const s_time)[TH * TW] = reinterpret_cast<float (*)[TH * TW]>(s_dyn + AR * AC);   // [NSUB][TH * TW]
    uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(&s_time[NSUB][0]);           // [AR * AC]   (bin_ok == 0 only)
    __shared__ unsigned long long s_rpart[NSUB][kSumFields * 4];
    __shared__ DevState s_state;
    const int b = blockIdx.x, tid = threadIdx.x;
    tl_stamp(a.tl, a.j, 0);
    // scalar loads first (see k_bin_warp_scatter), then the vector loads of the head, nothing consumed in between
    const HotState h0 = sload(&pre_st_in->hot);
    // (`lost`: three words by launch number mod 3 -- this pass reads its predecessor's, raises its own, clears its
    // successor's.  One shared word was read by late work-groups of a pass AFTER early ones of the same pass had raised it.)
    const uint32_t lost_prev = sload(pre_lost + (pre_j + 2) % 3);
    const FusedTab ft = sload(reinterpret_cast<const FusedTab*>(pre_ftab) + b);
    // running index -> global index: off[0] plus the offset STEPS of the ranges the index has passed.  (A chain of
    // selects among the offsets themselves was turned into a select among ADDRESSES of a scratch copy of the table: a
    // scratch load in front of every event load.)
    uint32_t off_step[kFusedRanges];
#pragma unroll
    for (int r = 1; r < kFusedRanges; ++r) off_step[r] = ft.off[r] - ft.off[r - 1];
    unsigned long long accv[kAccPerLane];
    if (tid < 64) acc_load_wave<false, false>(pre_acc_in, tid, accv);
    unsigned long long state_word = 0;
    if (tid < kStateWords) state_word = reinterpret_cast<const unsigned long long*>(pre_st_in)[tid];
    auto store_state = [&]() {   // work-group 0, after a barrier
        if (b == 0 && tid < kStateWords) {
            const unsigned long long v = reinterpret_cast<const unsigned long long*>(&s_state)[tid];
            reinterpret_cast<unsigned long long*>(a.st_out)[tid] = v;
            if (a.snap) reinterpret_cast<unsigned long long*>(a.snap)[tid] = v;
        }
    };
    auto clear_next_acc = [&]() {   // the accumulators the NEXT pass adds to (last read two passes ago), and its `lost` word
        if (b == 0 && tid < 64) {
            for (int i = tid; i < kAccGroups * 16; i += 64) (&a.acc_zero[0].f[0])[i] = 0ull;
            if (tid == 0) a.lost[(a.j + 1) % 3] = 0u;
        }
    };
    // A pass that does nothing: the loop is over, or it waits for a re-bin (need_rebin == 2), or the previous pass has
    // lost events -- then this one raises the request.
    const bool stalled = h0.need_rebin == 2;
    const bool trip = !h0.done && !stalled && lost_prev != 0u;
    if (h0.done || stalled || trip) {
        if (b != 0) return;
        if (tid < kStateWords) reinterpret_cast<unsigned long long*>(&s_state)[tid] = state_word;
        __syncthreads();
        if (tid == 0) {
            if (trip) { s_state.hot.need_rebin = 2; s_state.hot.redo = 1; s_state.hot.pend = 0; s_state.ovf_total += 1; }
            s_state.last_j = a.j;
        }
        __syncthreads();
        store_state();
        clear_next_acc();
        return;
    }
    const bool redo = h0.redo != 0;
    const bool pending = !redo && h0.pend != 0;
    const bool do_warp = a.warp && !redo;
    const EvSetPtrs ev = a.sets.s[h0.cs ^ h0.flip];
    const uint32_t* __restrict__ xy = ev.xy;
    const int32_t* __restrict__ t = ev.t;
    const float2* __restrict__ p_in = h0.pp ? ev.p2 : ev.p;
    float2* __restrict__ p_out = h0.pp ? ev.p : ev.p2;
    const int br = b / a.nbc, bc = b - br * a.nbc;
    const int X0 = br * TSR - H, Y0 = bc * TC - H;
    const uint32_t M = ft.total, own = ft.pre[1];
    uint32_t vxy[U], vi[U];
    int32_t vt[U];
    float2 vp[U];
    uint32_t base = 0;
    auto load_pass = [&]() {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            uint32_t v = base + k * THREADS + tid;
            v = v < M ? v : 0u;
            uint32_t off = ft.off[0];
#pragma unroll
            for (int r = 1; r < kFusedRanges; ++r) off += v >= ft.pre[r] ? off_step[r] : 0u;
            const uint32_t i = v + off;
            vi[k] = i;
            vxy[k] = xy[i];
            vt[k] = t[i];
            vp[k] = p_in[i];
        }
    };
    if (M) load_pass();
    asm volatile("" ::: "memory");
    {
        ulonglong2* z = reinterpret_cast<ulonglong2*>(s_acc);
        for (int i = tid; i < AR * AC / 2; i += THREADS) z[i] = make_ulonglong2(0ull, 0ull);
        if (!h0.bin_ok)
            for (int i = tid; i < AR * AC; i += THREADS) s_cnt[i] = 0u;
    }
    if (tid < kStateWords) reinterpret_cast<unsigned long long*>(&s_state)[tid] = state_word;
    double ppx[U], ppy[U];
    auto previous_positions = [&]() {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            ppx[k] = pr_from_p(vxy[k] & 0xffffu, vp[k].x);
            ppy[k] = pr_from_p(vxy[k] >> 16, vp[k].y);
        }
    };
    tl_stamp(a.tl, a.j, 1);
    if (pending && tid < 64) {
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long word = acc_reduce_wave(accv);
        __builtin_amdgcn_wave_barrier();
        tl_stamp(a.tl, a.j, 2);
        model_update_wave(&s_state, word, tid, 1);
        __builtin_amdgcn_s_setprio(0);
        tl_stamp(a.tl, a.j, 3);
    } else {
        previous_positions();
    }
    __syncthreads();
    tl_stamp(a.tl, a.j, 4);
    const ScatterHot hs = scatter_hot(&s_state);
    if (b == 0 && tid == 0) {   // bookkeeping only the stored state needs (fields the scatter does not read)
        if (pending) model_update_rest(&s_state, a.trace, 0, 0u);
        else if (s_state.hot.flip) { s_state.hot.cs ^= 1; s_state.hot.flip = 0; }   // (the commit model_update_rest would make)
        if (do_warp && !hs.done) s_state.hot.pp ^= 1;   // (a pass that finds the loop finished stores nothing)
        s_state.hot.redo = 0;
        s_state.hot.pend = hs.done ? 0 : 1;   // this pass's sums, unless the update has just ended the loop
        s_state.last_j = a.j;
    }
    if (hs.done) {
        __syncthreads();
        store_state();
        return;
    }
    if (pending && tid < 64) previous_positions();
    const int hsc = hs.scale / 2;
    const bool packed = hs.bin_ok != 0;
    bool lost_here = false;
    for (;;) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t v = base + k * THREADS + tid;
            if (v >= M) continue;
            const bool mine = v < own;
            double px = ppx[k], py = ppy[k];
            if (do_warp) {
                float2 q;
                double nx, ny;
                warp_products(hs.wp, px, py, vt[k], q, nx, ny);
                if (mine)
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&p_out[vi[k]]),
                                       ((unsigned long long)__float_as_uint(q.y) << 32) | (unsigned long long)__float_as_uint(q.x),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                px = pr_from_p(vxy[k] & 0xffffu, q.x);
                py = pr_from_p(vxy[k] >> 16, q.y);
            }
            const int X = trunc_scatter(px * (double)hs.scale + (double)hs.x_sh);   // accel_lib.h:154-158
            const int Y = trunc_scatter(py * (double)hs.scale + (double)hs.y_sh);
            if ((X >= hs.wsx + hsc) || (X < hsc) || (Y >= hs.wsy + hsc) || (Y < hsc)) continue;
            const int lx = X - X0, ly = Y - Y0;
            if (lx >= 0 && lx < AR && ly >= 0 && ly < AC) {
                const unsigned long long dt = (unsigned long long)((long long)vt[k] - hs.tmin);
                if (packed) {
                    atomicAdd(&s_acc[lx * AC + ly], (1ull << hs.bin_tbits) + dt);
                } else {
                    atomicAdd(&s_acc[lx * AC + ly], dt);
                    atomicAdd(&s_cnt[lx * AC + ly], 1u);
                }
            }
            if (mine) {
                // Does every tile whose halo window holds (X, Y) read this event?  The tile of its sort key does; the
                // neighbours read the key's edge strips.  dx, dy: the landing pixel relative to the key's tile.
                const int dx = lx - H, dy = ly - H;
                if (dx < H || dx >= TSR - H || dy < H || dy >= TC - H) {
                    int z = 0;
#pragma unroll
                    for (int q = 0; q < kFusedZones - 1; ++q) z += v >= ft.zone[q] ? 1 : 0;
                    const bool top = (0x00eu >> z) & 1u, right = (0x038u >> z) & 1u, bottom = (0x0e0u >> z) & 1u, left = (0x182u >> z) & 1u;
                    const bool ok = (dx >= H || top) && (dx < TSR - H || bottom) && dx >= H - TSR && dx < 2 * TSR - H &&
                                    (dy >= H || left) && (dy < TC - H || right) && dy >= H - TC && dy < 2 * TC - H;
                    lost_here |= !ok;
                }
            }
        }
        base += THREADS * U;
        if (base >= M) break;
        load_pass();
        previous_positions();
    }
    if (lost_here) __hip_atomic_store(a.lost + a.j % 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tl_stamp(a.tl, a.j, 5);
    __syncthreads();
    tl_stamp(a.tl, a.j, 6);
    store_state();
    // ---- the stencil of k_stencil_binned, one 16 x 64 sub-tile per 256-thread sub-group, on the LDS tile ----
    const int g = tid >> 8, lt = tid & 255;
    const int R = a.R, C = a.C;
    const int r0 = br * TSR + g * TR, c0 = bc * TC;
    const int bt = hs.bin_tbits;
    const unsigned long long bm = (1ull << bt) - 1ull;
    const unsigned long long* win = s_acc + (g * TR) * AC;   // rows r0 - H .. of this sub-tile
    const uint32_t* cwin = s_cnt + (g * TR) * AC;
    for (int idx = lt; idx < TH * TW; idx += 256) {
        const int tr = idx / TW, tc = idx - tr * TW;
        const int gr = r0 - 1 + tr, gc = c0 - 1 + tc;
        float tv = 0.f;
        if (gr >= 0 && gr < R && gc >= 0 && gc < C) {
            // s x s box sum == the s x s splat of accel_lib.h:160-165 on integer planes
            unsigned long long pk = 0;
            uint32_t cacc = 0;
#pragma unroll
            for (int da = 0; da <= 2 * HS; ++da)
#pragma unroll
                for (int db = 0; db <= 2 * HS; ++db) {
                    pk += win[(tr + da) * PC + (tc + db)];
                    if (!packed) cacc += cwin[(tr + da) * PC + (tc + db)];
                }
            unsigned long long acc = pk;
            if (packed) { acc = pk & bm; cacc = (uint32_t)(pk >> bt); }
            tv = time_from_sums(cacc, (long long)acc, hs.tmin);
        }
        s_time[g][idx] = tv;
    }
    tl_stamp(a.tl, a.j, 7);
    __syncthreads();
    tl_stamp(a.tl, a.j, 8);
    SumsT smt;   // (a thread's own pixels: 32-bit integer sums, see bf_device_fns.h)
    sums_zero(smt);
    const int hR = R / 2, hC = C / 2;
#pragma unroll
    for (int k = 0; k < (TR * TC) / 256; ++k) {
        const int pidx = lt + k * 256;
        const int lr = pidx / TC, lc = pidx - lr * TC;
        const int gr = r0 + lr, gc = c0 + lc;
        if (gr < R && gc < C) {
            float gx, gy;
            stencil_px<TW>(&s_time[g][(lr + 1) * TW + (lc + 1)], gr, gc, R, C, hR, hC, smt, gx, gy);
        }
    }
    const Sums sm = sums_widen(smt);
    constexpr bool kPack = TR * TC <= 1024 && TR <= 64 && TC <= 64;
    tl_stamp(a.tl, a.j, 9);
    block_reduce_publish<256, kPack>(sm, s_rpart[g], lt, r0 - hR, c0 - hC);
    tl_stamp(a.tl, a.j, 10);
    if (tid < 64) clear_next_acc();
    if (lt >= 64 || r0 >= R) return;   // (a sub-tile below the image has nothing to add)
    const Sums blk = block_reduce_total<256, kPack>(s_rpart[g], r0 - hR, c0 - hC);
    acc_add(a.acc_out, (b * NSUB + g) % kAccGroups, blk, lt);
    tl_stamp(a.tl, a.j, 11);
}

// One pass of the one-kernel iteration (k_fused_pass).  rows_per_tile: 32 or 64.
template <int HS, int NSUB>
static hipError_t launch_fused2(const FusedArgs& a, hipStream_t s) {
    constexpr int U = 4;
    constexpr int H = HS + 1, AR = 16 * NSUB + 2 * H, AC = kTileC + 2 * H;
    constexpr size_t lds = (size_t)AR * AC * 12 + (size_t)NSUB * (kTileR + 2) * (kTileC + 2) * 4;
    static_assert(lds + 4096 <= (size_t)kBinTileLdsMax, "the tile fits a CU");
    static std::atomic<unsigned long long> raised{0ull};   // (per device: see launch_bws2)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const unsigned long long dev_bit = 1ull << (dev & 63);
    if (lds > 48 * 1024 && !(raised.load(std::memory_order_acquire) & dev_bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_pass<HS, NSUB, U>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kBinTileLdsMax);
        if (e != hipSuccess) return e;
        raised.fetch_or(dev_bit, std::memory_order_release);
    }
    launch_timed(k_fused_pass<HS, NSUB, U>, dim3(a.nbr * a.nbc), dim3(256 * NSUB), lds, s, a.ftab, a.st_in, a.acc_in, a.lost, a.j, a);
    return hipSuccess;
}
template <int HS>
static hipError_t launch_fused1(const FusedArgs& a, int rows_per_tile, hipStream_t s) {
    return rows_per_tile == 64 ? launch_fused2<HS, 4>(a, s) : launch_fused2<HS, 2>(a, s);
}
hipError_t launch_fused_pass(const FusedArgs& a, int half_scale, int rows_per_tile, hipStream_t s) {
    switch (half_scale) {
        case 0: return launch_fused1<0>(a, rows_per_tile, s);
        case 1: return launch_fused1<1>(a, rows_per_tile, s);
        case 2: return launch_fused1<2>(a, rows_per_tile, s);
        case 3: return launch_fused1<3>(a, rows_per_tile, s);
        case 4: return launch_fused1<4>(a, rows_per_tile, s);
        default: return hipErrorInvalidValue;
    }
}


}  // namespace bf
