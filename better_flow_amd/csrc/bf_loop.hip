// bf_loop.hip -- the persistent form of the one-kernel iteration: MANY iterations of OptimizerRolling::run
// (optimizer_rolling.h:48-125,305-347) per launch, for a slice context that has the GPU to itself.
//
// Why.  k_fused_pass (bf_fused.hip) made an iteration one launch, and for the reference's own operating point -- the
// compiled-in ring of 50 000 events on a 240x180 sensor, ~115 iterations per warm-started slice
// (bf_motion_compensator.cpp:6-10,135) -- that launch IS the iteration: 11.7 us of which 3.6 are the gap between two
// dependent launches and 2.0 the head's reload of state, accumulators and events.  Here the work-groups stay resident
// (one per image tile, all co-resident: the grid is checked against the occupancy query), keep their events in REGISTERS, and
// replace the launch boundary by an all-to-all exchange of the moment sums through memory:
//
//   pass     as k_fused_pass: the tile's own events and the neighbouring tiles' edge strips are warped and added to an LDS
//            tile; each 256-thread sub-group runs K3's box sum / time image / Scharr / moments on one 16 x 64 sub-tile
//            -- same thread -> pixel mapping and reduction tree, so every f64 sub-tile partial carries the bits of the
//            other loops;
//   publish  the sub-tile's sums as the fifteen exact integer words of MomentAcc (sums_lane_word) -- not added to shared
//            accumulators with atomics (a memory-side atomic is ~1.1 us away from its reader) but WRITTEN, one record per
//            sub-tile, sixteen lanes x 16 bytes (payload, tag) with write-through stores.  A 16-byte store of one lane is
//            never torn, the tag (run, pass) says which pass the word belongs to: no flag, no fence, no atomic;
//   reduce   sixteen reducer waves (wave 1 of work-groups 0 .. 15) each poll 1/16 of the records and publish their integer
//            sum the same way; wave 0 of EVERY work-group polls the sixteen reduced records.  Integer sums: the total does
//            not depend on who adds what in which order, so it is the total the accumulators would have held;
//   update   every work-group runs the model / loop update on its own LDS copy of the state (model_update_wave +
//            model_update_rest: identical inputs, identical results), and the next pass starts -- no launch, no reload.
//
// Two buffers of records by pass parity suffice: a work-group writes the records of pass j + 2 only after it has read the
// reduced records of pass j + 1, which exist only once every reducer has finished with pass j.
//
// The kernel returns when the loop is over, when a re-bin is due (the predictive request of model_update_rest, or events
// that outran their bins: `lost`, carried in lane 15 of the records), or after `max_passes`; the owners then store their
// events' products and work-group 0 the state.  It always leaves with the update of its last pass applied (hot.pend == 0),
// so nothing but the state crosses the launch boundary.  Lists longer than a thread's registers (THREADS x U events) take
// the multi-pass path: products kept in memory, the strips' in per-reader private arrays (`scratch`), so that no
// work-group ever reads what another one writes during the loop.
//
// A waiter gives up after 0.2 s without progress (a work-group that never became resident: another process's kernels hold
// part of the CUs).  The launch then UNDOES itself: nobody stores products or state -- the events and the state are as the
// launch found them --, work-group 0 only marks the state (hot.spare_ < 0: work-groups that start late leave at once), and
// bf_run carries on with one launch per iteration.  Never a hang, never a wrong sum.
// Whether a launch is kept or undone is ONE decision, taken by a compare-and-swap on a verdict word (verdict_decide): a
// work-group that has timed out proposes ABORT, a work-group that is about to leave with its results proposes COMMIT, the
// first proposal wins and everybody follows it.  (Every waiter runs its own clock: without the arbiter a record arriving
// right at the deadline could be accepted by most work-groups and time out in one, which then stored nothing while the
// others stored their products and the final state.)  A work-group that timed out but finds COMMIT reads the reduced records
// again -- they exist, somebody accepted them -- and catches up.
#include <hip/hip_runtime.h>
#include <atomic>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {

#define BF_LOOP_U 4   // events a thread keeps in registers
constexpr int kLoopReducers = 16;
constexpr int kRecWords = 32;   // a record: 16 lanes x (payload u64, tag u64)

__device__ __forceinline__ void xchg_store(unsigned long long* slot, unsigned long long payload, unsigned long long tag) {
    bf_u32x4 v;
    v.x = (unsigned int)payload; v.y = (unsigned int)(payload >> 32);
    v.z = (unsigned int)tag;     v.w = (unsigned int)(tag >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(slot), "v"(v) : "memory");
}
__device__ __forceinline__ bf_u32x4 xchg_load_issue(const unsigned long long* slot) {
    bf_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(slot) : "memory");
    return v;
}

// The launch's verdict: one 64-bit word per context, (launch id << 2) | decision.  Launch ids grow, so the word needs no reset:
// whoever finds an older id in it proposes; the first compare-and-swap that lands decides for everybody.
constexpr int kVerdictCommit = 1, kVerdictAbort = 2;
__device__ __forceinline__ int verdict_decide(unsigned long long* w, unsigned long long id /* low two bits clear */, int mine,
                                              bool* won = nullptr) {
    unsigned long long old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if ((old >> 2) == (id >> 2)) return (int)(old & 3ull);
        if (__hip_atomic_compare_exchange_strong(w, &old, id | (unsigned long long)mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)) {
            if (won) *won = true;
            return mine;
        }
    }
}

// Lane-parallel poll: every lane waits for its NS slots (16-byte units at base + 2 * idx[k]; idx < 0: no slot) to carry
// `tag`, all of a wave's loads in flight together, the whole wave retrying until every lane is served.  Returns the sum of
// the payloads; false on time-out.
template <int NS>
__device__ __forceinline__ bool xchg_poll_sum(const unsigned long long* base, const int (&idx)[NS], unsigned long long tag,
                                              unsigned long long& sum, unsigned long long limit = 20000000ull /* 0.2 s of the 100 MHz clock */) {
    bf_u32x4 v[NS];
    unsigned long long t0 = 0;
    for (unsigned tries = 0;; ++tries) {
#pragma unroll
        for (int k = 0; k < NS; ++k) v[k] = xchg_load_issue(base + 2 * (size_t)(idx[k] < 0 ? 0 : idx[k]));
        if constexpr (NS == 4)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const unsigned long long tg = ((unsigned long long)v[k].w << 32) | v[k].z;
            ok = ok && (idx[k] < 0 || tg == tag);
        }
        if (__ballot(!ok) == 0ull) break;
        __builtin_amdgcn_s_sleep(2);   // (~50 ns: a poller must not crowd the memory side the records travel through)
        if (tries >= 32u) {
            if (tries == 32u) t0 = wall_clock64();
            else if ((tries & 255u) == 0u && wall_clock64() - t0 > limit) return false;
        }
    }
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k)
        if (idx[k] >= 0) s += ((unsigned long long)v[k].y << 32) | v[k].x;
    sum = s;
    return true;
}

template <int HS, int NSUB, int U>
__global__ __launch_bounds__(256 * NSUB, NSUB == 2 ? 4 : 2) void k_fused_loop(FusedLoopArgs a) {
    constexpr int THREADS = 256 * NSUB;
    constexpr int TR = kTileR, TC = kTileC;
    constexpr int H = HS + 1;
    constexpr int TSR = TR * NSUB;
    constexpr int AR = TSR + 2 * H, AC = TC + 2 * H;   // the LDS tile
    constexpr int PC = AC;
    constexpr int TH = TR + 2, TW = TC + 2;
    extern __shared__ unsigned long long s_dyn[];
    unsigned long long* const s_acc = s_dyn;                                                    // [AR * AC]
    float (*const s_time)[TH * TW] = reinterpret_cast<float (*)[TH * TW]>(s_dyn + AR * AC);    // [NSUB][TH * TW]
    uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(&s_time[NSUB][0]);                      // [AR * AC]   (bin_ok == 0 only)
    __shared__ unsigned long long s_rpart[NSUB][kSumFields * 4];
    __shared__ DevState s_state;
    __shared__ int s_lost, s_exit, s_abort, s_rest;
    __shared__ double s_sctab[14];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ntiles = a.nbr * a.nbc, nrec = ntiles * NSUB;
    const int nred = ntiles < kLoopReducers ? ntiles : kLoopReducers;

    // ---- entry: state -> LDS (written by the previous kernel: plain loads), range table, events -> registers ----
    const FusedTab ft = sload(reinterpret_cast<const FusedTab*>(a.ftab) + b);
    uint32_t off_step[kFusedRanges];
#pragma unroll
    for (int r = 1; r < kFusedRanges; ++r) off_step[r] = ft.off[r] - ft.off[r - 1];
    if (tid < kStateWords) reinterpret_cast<unsigned long long*>(&s_state)[tid] = reinterpret_cast<const unsigned long long*>(a.st)[tid];
    if (tid == 0) { s_lost = 0; s_exit = 0; s_abort = 0; s_rest = 0; sincos_table_fill(s_sctab); }
    __syncthreads();
    auto store_state = [&]() {   // work-group 0, after a barrier: both state buffers and the host's snapshot
        if (b == 0 && tid < kStateWords) {
            const unsigned long long v = reinterpret_cast<const unsigned long long*>(&s_state)[tid];
            reinterpret_cast<unsigned long long*>(a.st)[tid] = v;
            reinterpret_cast<unsigned long long*>(a.st_other)[tid] = v;
            if (a.snap) reinterpret_cast<unsigned long long*>(a.snap)[tid] = v;
        }
    };
    {
        const int done0 = lds_sreg(&s_state.hot.done), rebin0 = lds_sreg(&s_state.hot.need_rebin);
        if (lds_sreg(&s_state.hot.spare_) < 0) return;   // this launch has given up (see above): nothing was and nothing is changed
        if (done0 || rebin0 == 2) {   // the loop is over, or it waits for a re-bin nobody enqueued: nothing to do
            __syncthreads();
            if (b == 0 && tid == 0) s_state.hot.spare_ += 1;
            __syncthreads();
            store_state();
            return;
        }
    }
    const int live_set = lds_sreg(&s_state.hot.cs) ^ lds_sreg(&s_state.hot.flip);
    __syncthreads();
    // A re-bin moved the events to the other set (hot.flip): commit it now, as every pass of k_fused_pass does at its head.
    // (model_update_rest commits it too, but a pass that lost events has no update -- and the NEXT re-bin's scatter kernel
    // reads set `cs`: left uncommitted, it would re-sort the stale set with the fresh one's bin ids.)
    if (tid == 0 && s_state.hot.flip) { s_state.hot.cs ^= 1; s_state.hot.flip = 0; }
    const EvSetPtrs ev = a.sets.s[live_set];
    const uint32_t* __restrict__ xy = ev.xy;
    const int32_t* __restrict__ t = ev.t;
    float2* const p_cur = lds_sreg(&s_state.hot.pp) ? ev.p2 : ev.p;
    const int br = b / a.nbc, bc = b - br * a.nbc;
    const int X0 = br * TSR - H, Y0 = bc * TC - H;
    const uint32_t M = ft.total, own = ft.pre[1];
    const bool single = M <= (uint32_t)(THREADS * U);   // the whole list lives in registers
    uint32_t vxy[U], vi[U];
    int32_t vt[U];
    float2 vp[U];
    // products of list entry v during the loop (multi-pass lists): private arrays -- the owner's copy in scratch[3], a strip
    // event's in the array of the reader's direction (N / S -> 0, W / E -> 1, diagonal -> 2: no two readers of an event
    // share a direction class)
    auto entry_of = [&](uint32_t v, uint32_t& i, float2*& parr) {
        uint32_t off = ft.off[0];
#pragma unroll
        for (int r = 1; r < kFusedRanges; ++r) off += v >= ft.pre[r] ? off_step[r] : 0u;
        i = v + off;
        const int slot = (v >= ft.pre[1] ? 1 : 0) + (v >= ft.pre[3] ? 1 : 0) + (v >= ft.pre[6] ? 1 : 0);
        parr = slot == 0 ? a.scratch[3] : (slot == 1 ? a.scratch[0] : (slot == 2 ? a.scratch[1] : a.scratch[2]));
    };
    auto load_pass = [&](uint32_t base, bool from_global) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            uint32_t v = base + k * THREADS + tid;
            v = v < M ? v : 0u;
            uint32_t i;
            float2* parr;
            entry_of(v, i, parr);
            vi[k] = i;
            vxy[k] = xy[i];
            vt[k] = t[i];
            vp[k] = from_global ? p_cur[i] : parr[i];
        }
    };
    if (M && single) load_pass(0u, true);
    if (M && !single) {   // copy-in: the list's products into this work-group's private arrays (its own events' too: the slice's
                          // array is only written when the launch ends in order)
        for (uint32_t v = tid; v < M; v += THREADS) {
            uint32_t i;
            float2* parr;
            entry_of(v, i, parr);
            parr[i] = p_cur[i];
        }
    }
    const int R = a.R, C = a.C;
    const int hR = R / 2, hC = C / 2;
    const int g = tid >> 8, lt = tid & 255;
    const int r0 = br * TSR + g * TR, c0 = bc * TC;
    const unsigned long long run_hi = (unsigned long long)(unsigned int)s_state.run_tag << 32;
    // this launch's id in the verdict word: (run, launches of the kernel completed in this run); low two bits: the decision
    const unsigned long long launch_id = run_hi | ((unsigned long long)((unsigned int)s_state.hot.spare_ & 0x3fffffffu) << 2);
    int j = s_state.last_j + 1;
    bool first_of_run = s_state.last_j < 0;
    int passes = 0;
    {   // the LDS tile, zeroed once here and then under every pass's exchange
        ulonglong2* z = reinterpret_cast<ulonglong2*>(s_acc);
        for (int i = tid; i < AR * AC / 2; i += THREADS) z[i] = make_ulonglong2(0ull, 0ull);
        for (int i = tid; i < AR * AC; i += THREADS) s_cnt[i] = 0u;
        __syncthreads();
    }
    for (;;) {
        // (opaque copies of the thread's indices: everything derived from them inside a pass would otherwise be hoisted out
        // of the pass loop as loop-invariant -- ~100 registers of addresses and pixel coordinates, twice the budget of two
        // work-groups per CU)
        int tid_ = tid, lt_ = lt;
        asm volatile("" : "+v"(tid_), "+v"(lt_));
        tl_stamp(a.tl, j, 0);
        // ---- scatter ----
        const ScatterHot hs = scatter_hot(&s_state);
        const bool redo = lds_sreg(&s_state.hot.redo) != 0;
        const bool do_warp = (first_of_run ? a.first_warp != 0 : true) && !redo;
        const int hsc = hs.scale / 2;
        const bool packed = hs.bin_ok != 0;
        // (the LDS tile is zero: cleared before the loop, and by the idle waves under the previous pass's exchange)
        if (tid_ == THREADS - 1 && s_rest) {   // bookkeeping of the last update, off the critical path: a thread of the last wave
            model_update_rest(&s_state, b == 0 ? a.trace : nullptr, 0, 0u);
            s_rest = 0;
        }
        tl_stamp(a.tl, j, 1);
        bool lost_here = false;
        for (uint32_t base = 0; base < M; base += THREADS * U) {
            if (!single) load_pass(base, false);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const uint32_t v = base + k * THREADS + tid_;
                if (v >= M) continue;
                const bool mine = v < own;
                double px = pr_from_p(vxy[k] & 0xffffu, vp[k].x), py = pr_from_p(vxy[k] >> 16, vp[k].y);
                if (do_warp) {
                    float2 q;
                    double nx, ny;
                    warp_products(hs.wp, px, py, vt[k], q, nx, ny);
                    vp[k] = q;
                    if (!single) {
                        uint32_t i;
                        float2* parr;
                        entry_of(v, i, parr);
                        parr[i] = q;
                    }
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
                if (mine) {   // does every tile whose halo window holds (X, Y) read this event?  (see k_fused_pass)
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
        }
        if (lost_here) s_lost = 1;
        tl_stamp(a.tl, j, 2);
        __syncthreads();
        tl_stamp(a.tl, j, 3);
        // ---- the stencil of k_stencil_binned, one 16 x 64 sub-tile per 256-thread sub-group, on the LDS tile ----
        const int bt = hs.bin_tbits;
        const unsigned long long bm = (1ull << bt) - 1ull;
        const unsigned long long* win = s_acc + (g * TR) * AC;
        const uint32_t* cwin = s_cnt + (g * TR) * AC;
        for (int idx = lt_; idx < TH * TW; idx += 256) {
            const int tr = idx / TW, tc = idx - tr * TW;
            const int gr = r0 - 1 + tr, gc = c0 - 1 + tc;
            float tv = 0.f;
            if (gr >= 0 && gr < R && gc >= 0 && gc < C) {
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
        tl_stamp(a.tl, j, 4);
        __syncthreads();
        tl_stamp(a.tl, j, 5);
        SumsT smt;   // (a thread's own pixels: 32-bit integer sums, see bf_device_fns.h)
        sums_zero(smt);
#pragma unroll
        for (int k = 0; k < (TR * TC) / 256; ++k) {
            const int pidx = lt_ + k * 256;
            const int lr = pidx / TC, lc = pidx - lr * TC;
            const int gr = r0 + lr, gc = c0 + lc;
            if (gr < R && gc < C) {
                float gx, gy;
                stencil_px<TW>(&s_time[g][(lr + 1) * TW + (lc + 1)], gr, gc, R, C, hR, hC, smt, gx, gy);
            }
        }
        const Sums sm = sums_widen(smt);
        constexpr bool kPack = TR * TC <= 1024 && TR <= 64 && TC <= 64;
        tl_stamp(a.tl, j, 6);
        block_reduce_publish<256, kPack>(sm, s_rpart[g], lt_, r0 - hR, c0 - hC);   // (work-group barrier inside)
        tl_stamp(a.tl, j, 7);
        // ---- publish: this sub-tile's record of pass j ----
        const unsigned long long tag = run_hi | (unsigned long long)(unsigned int)(j + 1);
        unsigned long long* const rec = a.rec + (size_t)(j & 1) * (size_t)nrec * kRecWords;
        unsigned long long* const red = a.red + (size_t)(j & 1) * (size_t)kLoopReducers * kRecWords;
        if (lt_ < 64) {
            const Sums blk = block_reduce_total<256, kPack>(s_rpart[g], r0 - hR, c0 - hC);
            if (lt_ < 16) {
                unsigned long long w = (r0 < R) ? sums_lane_word(blk, lt_) : 0ull;   // (a sub-tile below the image adds nothing)
                if (lt_ == 15) w = (g == 0 && s_lost) ? 1ull : 0ull;                  // lane 15: events outran their bins
                const bool muted = a.debug_mute >= 0 && j >= a.debug_mute && b == ntiles - 1;   // (test hook)
                if (!muted) xchg_store(rec + (size_t)(b * NSUB + g) * kRecWords + 2 * lt_, w, tag);
            }
        }
        tl_stamp(a.tl, j, 8);
        if (tid_ >= 128) {   // waves that take no part in the exchange clear the tile for the next pass meanwhile
            ulonglong2* z = reinterpret_cast<ulonglong2*>(s_acc);
            for (int i = tid_ - 128; i < AR * AC / 2; i += THREADS - 128) z[i] = make_ulonglong2(0ull, 0ull);
            if (!packed)
                for (int i = tid_ - 128; i < AR * AC; i += THREADS - 128) s_cnt[i] = 0u;
        }
        // ---- reduce: wave 1 of the first work-groups adds up its share of the records ----
        if (tid_ >= 64 && tid_ < 128 && b < nred) {
            const int lane = tid_ - 64, f = lane & 15, sub = lane >> 4;
            unsigned long long tot = 0;
            bool good = true;
            // this reducer's records: b, b + nred, b + 2 nred, ...; lane group `sub` takes every fourth of them, eight per
            // round (the trip count is the same for the whole wave)
            for (int q0 = 0; b + nred * q0 < nrec && good; q0 += 32) {
                int idx[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int k = b + nred * (q0 + sub + 4 * m);
                    idx[m] = k < nrec ? k * 16 + f : -1;
                }
                unsigned long long part = 0;
                good = xchg_poll_sum<8>(rec, idx, tag, part);
                tot += part;
            }
            tot += __shfl_xor(tot, 16, 64);
            tot += __shfl_xor(tot, 32, 64);
            // A reducer that timed out publishes NOTHING: its partial total under a valid tag would be taken for the sum by
            // every work-group's first wave (they look at the reduced records every microsecond, at their own clock only
            // every few hundred) -- an update on wrong sums.  Without the record they all time out as well and the launch
            // undoes itself as a whole.
            // (It does not decide anything either: this work-group's first wave waits for the same missing record and proposes
            // the ABORT.)
            const bool timed_out = __ballot(!good) != 0ull;
            if (lane < 16 && !timed_out) xchg_store(red + (size_t)b * kRecWords + 2 * lane, tot, tag);
        }
        // ---- wave 0: the reduced records -> the total of field lane % 16 in every lane; the update ----
        if (tid_ < 64) {
            const int f = tid_ & 15, sub = tid_ >> 4;
            int idx[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int q = sub + 4 * m;
                idx[m] = q < nred ? q * 16 + f : -1;
            }
            unsigned long long word = 0;
            bool good = xchg_poll_sum<4>(red, idx, tag, word);
            if (a.debug_abort >= 0 && j >= a.debug_abort) good = false;   // (test hook: every work-group "times out" at that pass)
            if (a.debug_split >= 0 && j == a.debug_split && b == ntiles - 1) {   // (test hook: ONE work-group "times out" on the launch's
                good = false;                                                    //  last pass -- at once, or when the others have committed)
                if (a.debug_split_late)
                    for (int q = 0; q < 64; ++q) __builtin_amdgcn_s_sleep(64);   // ~100 us
            }
            tl_stamp(a.tl, j, 9);
            bool aborted = false;
            if (!good) {
                // Timed out on this work-group's own clock.  ABORT -- unless the launch has been COMMITted meanwhile: then the
                // reduced records of this pass exist (whoever committed had accepted them) and are read again.
                int v = 0;
                if (tid_ == 0) {
                    bool won = false;
                    v = verdict_decide(a.verdict, launch_id, kVerdictAbort, &won);
                    if (won) {   // the mark, in both state buffers: work-groups that start late leave at entry, the host falls back
                        __hip_atomic_store(&a.st->hot.spare_, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&a.st_other->hot.spare_, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                v = __builtin_amdgcn_readfirstlane(v);
                if (v == kVerdictCommit) {
                    good = xchg_poll_sum<4>(red, idx, tag, word, 400000000ull);   // (4 s: cannot fail -- if it does, say so loudly)
                    if (!good && tid_ == 0 && a.broken) *reinterpret_cast<volatile int*>(a.broken) = 1;
                }
                aborted = !good;
            }
            word += __shfl_xor(word, 16, 64);
            word += __shfl_xor(word, 32, 64);
            const bool lost = lane_i64(word, 15) != 0;
            if (aborted) {
                if (tid_ == 0) { s_abort = 1; }
            } else if (lost) {   // sums of a pass that cannot vouch for them: dropped; the pass is repeated on fresh bins
                if (tid_ == 0) {
                    s_state.hot.need_rebin = 2; s_state.hot.redo = 1; s_state.hot.pend = 0; s_state.ovf_total += 1;
                    s_state.last_j = j;
                }
            } else {
                __builtin_amdgcn_s_setprio(3);
                model_update_wave(&s_state, word, tid_, 1, s_sctab);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_wave_barrier();
                if (tid_ == 0) {
                    // (model_update_rest -- the drift bound that asks for a re-bin, the commit of a re-bin's flip, the trace
                    // record -- runs under the next pass's scatter: a re-bin request is acted on one pass later)
                    s_rest = 1;
                    s_state.hot.redo = 0; s_state.hot.pend = 0;
                    s_state.last_j = j;
                }
            }
            __builtin_amdgcn_wave_barrier();
            tl_stamp(a.tl, j, 10);
            if (tid_ == 0) {
                const bool out = s_abort || s_state.hot.done || s_state.hot.need_rebin || passes + 1 >= a.max_passes;
                // leaving with results: COMMIT the launch -- or learn that somebody has given up on it
                if (out && !s_abort) {
                    if (a.debug_split >= 0 && !a.debug_split_late && j == a.debug_split && b != ntiles - 1)
                        for (int q = 0; q < 64; ++q) __builtin_amdgcn_s_sleep(64);   // (test hook: the straggler's ABORT lands first)
                    if (verdict_decide(a.verdict, launch_id, kVerdictCommit) != kVerdictCommit) s_abort = 1;
                }
                s_exit = out ? 1 : 0;
                s_lost = 0;
                if (b == 0 && a.snap && !out) {   // progress for the host's watchdog: (done, it), one 8-byte store
                    const unsigned long long w0 = (unsigned long long)(unsigned int)s_state.hot.done |
                                                  ((unsigned long long)(unsigned int)s_state.hot.it << 32);
                    *reinterpret_cast<volatile unsigned long long*>(a.snap) = w0;
                }
            }
        }
        __syncthreads();
        tl_stamp(a.tl, j, 11);
        ++passes;
        ++j;
        first_of_run = false;
        if (s_exit) break;
    }
    // ---- exit: the owners' products, then the state -- unless the launch gave up: then nothing is stored ----
    if (s_abort) return;   // (whoever's ABORT decided the launch has marked the state: hot.spare_ < 0)
    if (single && M) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t v = k * THREADS + tid;
            if (v < own) p_cur[vi[k]] = vp[k];
        }
    } else if (M) {
        for (uint32_t v = tid; v < own; v += THREADS) {
            uint32_t i;
            float2* parr;
            entry_of(v, i, parr);
            p_cur[i] = parr[i];
        }
    }
    if (tid == 0 && s_rest) model_update_rest(&s_state, b == 0 ? a.trace : nullptr, 0, 0u);
    if (b == 0 && tid == 0) s_state.hot.spare_ += 1;
    __syncthreads();
    store_state();
}

// Raises the dynamic-LDS limit of one instantiation (once per device) and returns how many of its work-groups fit a CU.
template <int HS, int NSUB>
static hipError_t loop_setup(int* per_cu, size_t* lds_out) {
    constexpr int U = BF_LOOP_U;
    constexpr int H = HS + 1, AR = 16 * NSUB + 2 * H, AC = kTileC + 2 * H;
    constexpr size_t lds = (size_t)AR * AC * 12 + (size_t)NSUB * (kTileR + 2) * (kTileC + 2) * 4;
    const void* fn = reinterpret_cast<const void*>(&k_fused_loop<HS, NSUB, U>);
    static std::atomic<unsigned long long> raised{0ull};   // (per device: see launch_bws2)
    static std::atomic<int> cached_per_cu{-1};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const unsigned long long dev_bit = 1ull << (dev & 63);
    if (lds > 48 * 1024 && !(raised.load(std::memory_order_acquire) & dev_bit)) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kBinTileLdsMax);
        if (e != hipSuccess) return e;
        raised.fetch_or(dev_bit, std::memory_order_release);
    }
    int n = cached_per_cu.load(std::memory_order_acquire);   // (the devices of one node are alike)
    if (n < 0) {
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256 * NSUB, lds);
        if (e != hipSuccess) return e;
        cached_per_cu.store(n, std::memory_order_release);
    }
    *per_cu = n;
    *lds_out = lds;
    return hipSuccess;
}
template <int HS, int NSUB>
static hipError_t launch_loop2(const FusedLoopArgs& a, int n_cus, hipStream_t s) {
    int per_cu = 0;
    size_t lds = 0;
    const hipError_t e = loop_setup<HS, NSUB>(&per_cu, &lds);
    if (e != hipSuccess) return e;
    const int ntiles = a.nbr * a.nbc;
    if ((long long)per_cu * n_cus < ntiles) return hipErrorCooperativeLaunchTooLarge;
    // A plain launch: the grid was checked against the occupancy query above, which is all hipLaunchCooperativeKernel adds
    // (at 15-19 us of host time per launch on this stack); residency is the same either way, and should the hardware admit
    // fewer work-groups than the query says, the kernel's waiters time out and the launch undoes itself.
    hipLaunchKernelGGL((k_fused_loop<HS, NSUB, BF_LOOP_U>), dim3(ntiles), dim3(256 * NSUB), lds, s, a);
    return hipGetLastError();
}
// Only the 32-row tiles (NSUB = 2) are instantiated: bf_set_cloud takes 64-row tiles when the nine sort keys per 32-row tile
// would not fit the counting sort (> 910 tiles), and that many tiles are never all resident -- the persistent form of the
// 64-row tile could only run when forced by an option, which went in round 5.
template <int HS>
static hipError_t launch_loop1(const FusedLoopArgs& a, int rows_per_tile, int n_cus, hipStream_t s) {
    return rows_per_tile == 32 ? launch_loop2<HS, 2>(a, n_cus, s) : hipErrorInvalidValue;
}
hipError_t launch_fused_loop(const FusedLoopArgs& a, int half_scale, int rows_per_tile, int n_cus, hipStream_t s) {
    switch (half_scale) {
        case 0: return launch_loop1<0>(a, rows_per_tile, n_cus, s);
        case 1: return launch_loop1<1>(a, rows_per_tile, n_cus, s);
        case 2: return launch_loop1<2>(a, rows_per_tile, n_cus, s);
        case 3: return launch_loop1<3>(a, rows_per_tile, n_cus, s);
        case 4: return launch_loop1<4>(a, rows_per_tile, n_cus, s);
        default: return hipErrorInvalidValue;
    }
}
template <int HS>
static bool loop_resident1(int rows_per_tile, int n_cus, int ntiles) {
    if (rows_per_tile != 32) return false;
    int per_cu = 0;
    size_t lds = 0;
    const hipError_t e = loop_setup<HS, 2>(&per_cu, &lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return (long long)per_cu * n_cus >= ntiles;
}
bool fused_loop_resident(int half_scale, int rows_per_tile, int n_cus, int ntiles) {
    switch (half_scale) {
        case 0: return loop_resident1<0>(rows_per_tile, n_cus, ntiles);
        case 1: return loop_resident1<1>(rows_per_tile, n_cus, ntiles);
        case 2: return loop_resident1<2>(rows_per_tile, n_cus, ntiles);
        case 3: return loop_resident1<3>(rows_per_tile, n_cus, ntiles);
        case 4: return loop_resident1<4>(rows_per_tile, n_cus, ntiles);
        default: return false;
    }
}

}  // namespace bf
