// bf_scatter.hip -- tile-binned form of the warp+scatter kernel (K1) for gfx950.
//
// Why: one random 64-bit global atomic per event costs ~48 us per 1M events on MI355X no
// matter how small the footprint is (scripts/micro/atomics.hip: random 48 us, coalesced 9 us,
// LDS-accumulate + dense store flush 9-14 us).  So the scatter is made local:
//
//   * events are counting-sorted by the image tile (TS x TS scaled pixels) their CURRENT
//     target falls into (k_bin_count / k_bin_scan / k_bin_scatter, once per slice and again
//     only when the model has drifted by more than the margin D);
//   * k_bin_warp_scatter: one work-group per bin.  It owns an LDS tile of (TS+2D)^2 packed
//     64-bit accumulators placed over its image tile, warps its events (coalesced loads of
//     xy / t / p), adds them with LDS atomics, and writes the tile with plain 16-byte stores
//     to its private slab -- no global atomics, nothing to zero, deterministic;
//   * an event that lands outside its bin's LDS tile (drift > D) takes an exact overflow
//     path (global atomics into the double-buffered overflow planes) and is counted; the
//     update kernel raises `need_rebin` when that count is large;
//   * the stencil kernel (k_stencil<3>, bf_kernels.hip) sums the <= 9 slabs that overlap
//     each pixel while it loads its LDS tile.
//
// All accumulators are integers (count << tbits | sum(t - tmin)), so the result is exactly
// the reference's s x s splat (accel_lib.h:147-166) whatever the event order.
#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <limits.h>
#include <cstdlib>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {

// Element idx of an array at a uniform base: scalar base register pair + 32-bit per-thread byte offset (no 64-bit vector address
// arithmetic per access).  The event arrays of a tile-binned slice stay below 2^32 bytes (bf_set_cloud: < 2^29 events).
template <class T>
__device__ __forceinline__ T ld_idx(const T* base, uint32_t idx) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + idx * (uint32_t)sizeof(T));
}

constexpr int kMaxTileRows = 192;   // LR = TSR + 2 D <= 128 + 64

// One event of the tile-binned scatter, from its previous projected position: warp (event.h:100-108,164-168 -- same
// arithmetic as k_warp_scatter), store of the new products, splat centre (accel_lib.h:154-158), LDS accumulate or --
// drifted out of this bin's tile -- the exact overflow path.
struct ScatterGeo {
    int X0, Y0, L, LR;
    int zw;   // event lists: width of the two column zones along the tile's left / right edge (BinGrid::zw; 0: one zone)
};
// The warp of one event and its splat centre (accel_lib.h:154-158); false: the event falls outside the window.
template <bool WARP>
__device__ __forceinline__ bool event_target(const ScatterHot& hs, float2* p, uint32_t i, uint32_t v, int32_t ti,
                                             double pr_x, double pr_y, int& X, int& Y) {
    const uint32_t fx = v & 0xffffu, fy = v >> 16;
    if (WARP) {
        float2 q;
        double nx, ny;
        warp_products(hs.wp, pr_x, pr_y, ti, q, nx, ny);
        // write-through as well (see the slab flush)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p) + i * 8u),
                           ((unsigned long long)__float_as_uint(q.y) << 32) | (unsigned long long)__float_as_uint(q.x),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pr_x = pr_from_p(fx, q.x);
        pr_y = pr_from_p(fy, q.y);
    }
    const int s = hs.scale, hsc = hs.scale / 2;
    // accel_lib.h:154-158.  The x86 conversion turns NaN into INT_MIN (rejected below); the hardware's turns it into 0 -- which
    // the window test rejects as well whenever scale / 2 >= 1 (0 < hsc), so only scale 1 needs the two-instruction fix-up
    // per coordinate (trunc_scatter); the branch is uniform.
    if (hsc > 0) {
        X = __double2int_rz(pr_x * (double)s + (double)hs.x_sh);
        Y = __double2int_rz(pr_y * (double)s + (double)hs.y_sh);
    } else {
        X = trunc_scatter(pr_x * (double)s + (double)hs.x_sh);
        Y = trunc_scatter(pr_y * (double)s + (double)hs.y_sh);
    }
    // accel_lib.h:157-158: hsc <= X < wsx + hsc and the same for Y -- one unsigned compare each
    return (unsigned)(X - hsc) < (unsigned)hs.wsx && (unsigned)(Y - hsc) < (unsigned)hs.wsy;
}
// the exact slow path: straight into the overflow planes
__device__ __forceinline__ void overflow_add(const ScatterHot& hs, const BinScatterArgs& a, int X, int Y, unsigned long long dt) {
    const size_t kk = (size_t)X * (size_t)hs.C + (size_t)Y;
    atomicAdd(&a.ovf_plane[kk], dt);
    atomicAdd(&a.ovf_cplane[kk], 1u);
    // (the stencil kernel reads the planes only around pixels flagged here)
    atomicOr(&a.ovf_bits[(size_t)X * (size_t)a.ovf_pitch + (size_t)(Y >> 5) + 1], 1u << (Y & 31));
}

// Dense slabs: accumulate in the bin's LDS tile.
template <bool WARP>
__device__ __forceinline__ void scatter_event(const ScatterHot& hs, const ScatterGeo& sg, unsigned long long* s_tile,
                                              const BinScatterArgs& a, float2* p, uint32_t i, uint32_t v, int32_t ti,
                                              double pr_x, double pr_y, uint32_t& n_ovf) {
    int X, Y;
    if (!event_target<WARP>(hs, p, i, v, ti, pr_x, pr_y, X, Y)) return;
    const unsigned long long dt = (unsigned long long)((long long)ti - hs.tmin);
    const int lx = X - sg.X0, ly = Y - sg.Y0;
    if (hs.bin_ok && (unsigned)lx < (unsigned)sg.LR && (unsigned)ly < (unsigned)sg.L) {
        atomicAdd(&s_tile[__mul24(lx, sg.L) + ly], (1ull << hs.bin_tbits) + dt);
    } else {   // drifted out of this bin's tile: exact, slow path
        overflow_add(hs, a, X, Y, dt);
        ++n_ovf;
    }
}

// ---- event lists (the compact form) ----------------------------------------------------------------------------------
// Sparse slices (fewer than one event per four pixels: a 1280x720 sensor at scale 3 has 8.3M pixels for 1M events): the
// bin's LDS tile would be ~100 KB -- one work-group per CU, four rounds of a latency chain per launch -- to merge events
// that almost never meet at a pixel.  Instead every event becomes one ENTRY (tile-local pixel index, packed accumulator
// of one event) of the bin's list, SORTED BY TILE ROW (counting sort: per-row counts in LDS, an exclusive scan, cursors),
// with the first entry of every row in `crow` (LR + 1 words per bin): the stencil kernel reads exactly the rows it needs
// and splats entries with LDS atomics, so duplicates simply add.  With column ZONES (BinGrid::zw > 0; round 6) the sort key
// is zone * LR + row, zone = 0 / 1 / 2 for the tile's first zw columns / the middle / its last zw columns, and `crow` holds
// 3 LR + 1 words: a stencil tile takes from a NEIGHBOURING bin column only the zone that faces it (zw = D + scale / 2 + 1
// columns of the bin's TS + 2 D: what a box sum at the tile's edge can reach) instead of the bin's full width -- at 1280x720
// a 16 x 64 stencil tile gathered the rows of three 80-column bins (240 columns) for a 68-column window; now 80 + 10 + 10.  No LDS tile: occupancy is set by registers, all bins of
// a 1280x720 slice are resident at once.  The order inside a row is whatever the atomics make it (integers).  A list
// holds LL entries (the slab's size); a bin with more events sends the surplus down the overflow path.
constexpr uint32_t kNoEntry = 0xffffffffu;
// pixel of one event -> entry code (row << 16 | index inside the tile fits: LL <= 65536), or kNoEntry (outside the window,
// or outside the bin's tile: overflow path, taken here unless `count_only`)
template <bool WARP>
__device__ __forceinline__ uint32_t list_event(const ScatterHot& hs, const ScatterGeo& sg, const BinScatterArgs& a, float2* p,
                                               uint32_t i, uint32_t v, int32_t ti, double pr_x, double pr_y, bool take_overflow,
                                               int& row, uint32_t& n_ovf) {
    int X, Y;
    row = 0;
    if (!event_target<WARP>(hs, p, i, v, ti, pr_x, pr_y, X, Y)) return kNoEntry;
    const int lx = X - sg.X0, ly = Y - sg.Y0;
    if (hs.bin_ok && (unsigned)lx < (unsigned)sg.LR && (unsigned)ly < (unsigned)sg.L) {
        // (the sort key: [zone *] LR + row)
        row = lx + (sg.zw ? (ly < sg.zw ? 0 : (ly >= sg.L - sg.zw ? 2 * sg.LR : sg.LR)) : 0);
        return (uint32_t)(__mul24(lx, sg.L) + ly);
    }
    if (take_overflow) {
        overflow_add(hs, a, X, Y, (unsigned long long)((long long)ti - hs.tmin));
        ++n_ovf;
    }
    return kNoEntry;
}
// exclusive scan of the key counts s_row[1 .. LR] (LR here: the number of sort keys -- rows, or 3 x rows with column zones;
// one wave, 64 keys per step): cursors in LDS, key starts in `crow`
__device__ __forceinline__ void list_row_scan(uint32_t* s_row, int LR, uint32_t* crow, int lane) {
    uint32_t carry = 0;
    for (int r0 = 0; r0 < LR; r0 += 64) {
        const int r = r0 + lane;
        const uint32_t v = r < LR ? s_row[1 + r] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (r < LR) {
            s_row[1 + r] = carry + incl - v;
            crow[r] = carry + incl - v;
        }
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) crow[LR] = carry;
}
// one entry to its slot (the row's cursor); a full list -> overflow path
__device__ __forceinline__ void list_put(const ScatterHot& hs, const ScatterGeo& sg, const BinScatterArgs& a, uint32_t* s_row,
                                         uint32_t code, int row, int32_t ti, uint32_t LL, unsigned long long* vals,
                                         uint16_t* cidx, uint32_t& n_ovf) {
    const unsigned long long dt = (unsigned long long)((long long)ti - hs.tmin);
    const uint32_t slot = atomicAdd(&s_row[1 + row], 1u);
    if (slot < LL) {
        // (plain stores: the slots of a wave are scattered over the list, a write-through store would send each 8-byte
        // entry to memory on its own -- measured 17.7 against 16.5 us per launch at 1280x720)
        vals[slot] = (1ull << hs.bin_tbits) + dt;
        cidx[slot] = (uint16_t)code;
    } else {
        const int lx = row >= 2 * sg.LR ? row - 2 * sg.LR : (row >= sg.LR ? row - sg.LR : row);   // (key -> tile row; without zones key < LR)
        overflow_add(hs, a, sg.X0 + lx, sg.Y0 + (int)code - __mul24(lx, sg.L), dt);
        ++n_ovf;
    }
}

// The whole tile goes to the bin's slab (nothing to zero, no atomics).  WRITE-THROUGH stores (agent-scope relaxed =
// global_store ... sc1): with plain stores the ~15 MB of slabs (+ 8 MB of p) sat dirty in the L2s until the end of the
// kernel, and their write-back stretched the kernel boundary to ~5.6 us (measured; "B / 6 TB/s" in the MI355X notes).
// (16 bytes per lane: L is even, so LL is, and a slab starts on a 16-byte boundary)
template <int THREADS>
__device__ __forceinline__ void flush_tile(const unsigned long long* s_tile, unsigned long long* slab, int LL, int tid) {
    const bf_u32x4* src4 = reinterpret_cast<const bf_u32x4*>(s_tile);
    for (int i = tid; i < LL / 2; i += THREADS) {
        const bf_u32x4 v = src4[i];
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(slab + 2 * i), "v"(v) : "memory");
    }
}
// ---- interior + margin format (FMT 3) ---------------------------------------------------------------------------------
// The dense slab of a bin is its whole LDS tile, margin included: (TS + 2 D)(TSR + 2 D) / (TS TSR) = 1.56 .. 1.67 x the
// image's pixels are written every iteration, and the stencil kernel merges up to 2 x 2 slabs per pixel (four loads and
// the row / column -> bin arithmetic for each).  But the margin of a tile is EMPTY right after a re-bin and fills only as
// the model drifts: 0.01 .. 2 % of the events (measured along cold runs).  So the bin writes its own TSR x TS pixels --
// which no other bin writes -- with plain stores into a tiled image (`slabs`, TS * TSR words per bin), and ADDS what its
// events left in the margin to a margin plane (image-linear, packed like the slab words: the packing bound of k_bin_scan
// covers the events of four bins, and a pixel hears from at most its own bin and three neighbours), with device atomics,
// one per touched margin pixel.  The stencil kernel reads one tiled-image word per pixel, plus the margin-plane word for
// pixels within D of a boundary of their bin.  The margin plane is double buffered like the overflow planes (buffer `cur`
// of the iteration); a bin lists the pixels it added to and clears exactly those in the other buffer at its next
// executed launch (the list is per bin, and the buffer it clears is not the one anybody adds to in that launch).
// (The first list entry of every thread is REQUESTED behind the first pass's events and consumed after the scatter loop: a
// dependent load + store at either end would put a memory round trip on every work-group's chain; ahead of the events,
// the scatter loop's header -- which waits, vmcnt(0), for the registers of its previous pass -- waited for it too.)
__device__ __forceinline__ uint32_t margin_preload(const BinScatterArgs& a, int b, uint32_t n_prev, int tid) {
    return (uint32_t)tid < n_prev ? a.mlist[(size_t)b * (size_t)a.mcap + tid] : 0xffffffffu;
}
__device__ __forceinline__ void margin_clear(const BinScatterArgs& a, int b, uint32_t n_prev, uint32_t e0, int tid, int threads) {
    if (e0 != 0xffffffffu) a.m_prev[e0] = 0ull;
    const uint32_t* lst = a.mlist + (size_t)b * (size_t)a.mcap;
    for (uint32_t i = tid + threads; i < n_prev; i += threads) a.m_prev[lst[i]] = 0ull;
}
template <int THREADS>
__device__ __forceinline__ void flush_split(const unsigned long long* s_tile, const BinScatterArgs& a, int b, int X0, int Y0, int C,
                                            uint32_t* s_mcnt /* [0] entries, [1] waves done */, int tid) {
    const BinGrid& g = a.g;
    const bf_u32x4* src4 = reinterpret_cast<const bf_u32x4*>(s_tile);
    const int half = g.L >> 1, hD = g.D >> 1, lgh = g.lg - 1;   // 16-byte pairs per tile row / per margin / per interior row (log2)
    // the bin's own pixels: rows D .. D + TSR of the tile, pairs D/2 .. D/2 + TS/2 of each -- a plain copy
    unsigned long long* own = a.slabs + (size_t)b * (size_t)(g.TS * g.TSR);
    const int nown = g.TSR << lgh;
    for (int i = tid; i < nown; i += THREADS) {
        const int r = i >> lgh, q = i - (r << lgh);
        const bf_u32x4 v = src4[(r + g.D) * half + hD + q];
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(own + 2 * i), "v"(v) : "memory");
    }
    // the margin: D full rows above and below (half pairs each), D/2 pairs left and right of the TSR rows in between
    // (D is a power of two here: bf_set_cloud)
    uint32_t* lst = a.mlist + (size_t)b * (size_t)a.mcap;
    const int nfull = 2 * g.D * half, nmar = nfull + g.TSR * g.D;   // (2 sides x D/2 pairs per row)
    const int lgD = 31 - __clz(g.D);
    for (int j = tid; j < nmar; j += THREADS) {
        int lx, lp;   // tile row, pair inside the row
        if (j < nfull) {
            const int r = (int)__umulhi((uint32_t)j, g.mul_h);   // j / half
            lp = j - r * half;
            lx = r < g.D ? r : r + g.TSR;
        } else {
            const int k = j - nfull, r = k >> lgD, q = k & (g.D - 1);
            lx = g.D + r;
            lp = q < hD ? q : q + (g.TS >> 1);
        }
        const bf_u32x4 v = src4[lx * half + lp];
        const unsigned long long w0 = ((unsigned long long)v.y << 32) | v.x, w1 = ((unsigned long long)v.w << 32) | v.z;
        if (w0 | w1) {
            const uint32_t px = (uint32_t)((X0 + lx) * C + Y0 + 2 * lp);
            if (w0) {
                __hip_atomic_fetch_add(&a.m_cur[px], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lst[atomicAdd(&s_mcnt[0], 1u)] = px;
            }
            if (w1) {
                __hip_atomic_fetch_add(&a.m_cur[px + 1], w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lst[atomicAdd(&s_mcnt[0], 1u)] = px + 1;
            }
        }
    }
    // The last wave to get here publishes the length of the list: no work-group barrier (it would hold every wave until its
    // write-through stores and atomics have drained).  LDS operations of a wave complete in order, so the count this wave
    // reads includes every entry of the waves that arrived before it.
    uint32_t arrived = 0;
    if ((tid & 63) == 0) arrived = atomicAdd(&s_mcnt[1], 1u);
    arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
    if (arrived == THREADS / 64 - 1 && (tid & 63) == 0) a.mcount[b] = s_mcnt[0];
}
__global__ __launch_bounds__(256) void k_margin_clean(unsigned long long* mplane, const uint32_t* mlist, uint32_t* mcount, int mcap) {
    const int b = blockIdx.x;
    const uint32_t n = mcount[b];
    const uint32_t* lst = mlist + (size_t)b * (size_t)mcap;
    for (uint32_t i = threadIdx.x; i < n; i += 256) mplane[lst[i]] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) mcount[b] = 0u;
}
void launch_margin_clean(unsigned long long* mplane, const uint32_t* mlist, uint32_t* mcount, int nbins, int mcap, hipStream_t s) {
    if (nbins > 0) hipLaunchKernelGGL(k_margin_clean, dim3(nbins), dim3(256), 0, s, mplane, mlist, mcount, mcap);
}

// K1 (binned): [pending update] + warp + LDS scatter + slab flush, one work-group per bin.
//
// The model / loop update of the tile-binned loop runs HERE.  The stencil kernel of iteration j - 1 only adds its
// moment sums to the exact accumulators; the total, ObjectModel::update_accumulators, the iteration_step glue and the
// run() loop control (optimizer_rolling.h:61-101,328-346) are done at the head of launch j (= number of stencil launches
// completed before it) by every work-group for itself, on an LDS copy of the state written by launch j - 1 (st_in), if
// the update is still pending (state.it < j; a k_finish_update may have done it).  Work-group 0 stores the new state
// to st_out, the buffer the stencil kernel of this iteration and launch j + 1 read (ping-pong: a work-group that starts
// late must still find the OLD state in st_in), and to a pinned host snapshot the host polls (no copy commands in the
// stream).  The critical part of the update is ~0.6 us of serial f64 arithmetic on one lane, and it is hidden: the
// accumulators are the first thing requested, the events of the first pass the second, and while the first wave forms
// the total and updates, the other fifteen turn their events' stored f32 products into the previous positions (the
// model-independent third of the per-event arithmetic); the first wave catches up after the barrier.
// (pre_*: the three pointers the kernel's FIRST loads go through, repeated ahead of the argument block as scalar parameters:
// the command processor preloads leading scalar arguments into SGPRs before a wave starts -- `-mllvm
// -amdgpu-kernarg-preload-count`, Makefile -- so those loads leave together with the fetch of the argument block instead
// of behind it: 7.87 -> 7.66 us per launch alone at config 2, same box, libraries alternating.)
template <bool WARP, int THREADS, int U, int FMT>
__global__ __launch_bounds__(THREADS) void k_bin_warp_scatter(const uint32_t* __restrict__ pre_bin_start, const DevState* pre_st_in,
                                                              MomentAcc* pre_acc, BinScatterArgs a) {
    constexpr bool COMPACT = FMT == 2;   // event lists / (0) dense slabs   (FMT 1, lists merged per pixel in the LDS tile, went in round 5: no BASELINE configuration took it)
    constexpr bool SPLIT = FMT == 3;                        // interior + margin (see flush_split)
    extern __shared__ unsigned long long s_tile[];   // (dense slabs only)
    __shared__ DevState s_state;
    __shared__ uint32_t s_row[FMT == 2 ? 1 + 3 * kMaxTileRows : 1];   // lists: entries per sort key ([zone,] tile row), then the keys' cursors
    __shared__ uint32_t s_mcnt[2];
    if (FMT == 2)
        for (int r = threadIdx.x; r <= 3 * kMaxTileRows; r += THREADS) s_row[r] = 0;
    if (SPLIT && threadIdx.x < 2) s_mcnt[threadIdx.x] = 0;
    const BinGrid& g = a.g;
    const int L = g.L, LR = g.LR, LL = g.LR * g.L;
    const int b = blockIdx.x, tid = threadIdx.x;
    tl_stamp(a.tl, a.j, 0);
    // Everything the block needs from global memory is requested up front, in one burst, and NOTHING is consumed
    // before the last request is out (vector loads complete in order: consuming the state copy early would also wait
    // for the accumulators, ~1.1 us away -- they were last written by atomics at the memory side).
    // (The scalar loads come first in program order: placed after the lane-conditional vector loads, the compiler
    // carried the state pointer through a vector register and turned them into vector loads -- which complete in order
    // behind the accumulators.)
    const uint32_t beg = sload(pre_bin_start + b), end = sload(pre_bin_start + b + 1);
    const int done0 = sload(&pre_st_in->hot.done), it0 = sload(&pre_st_in->hot.it);
    const int live_set = sload(&pre_st_in->hot.cs) ^ sload(&pre_st_in->hot.flip);
    const uint32_t m_prev_n = SPLIT ? sload(a.mcount + b) : 0u;
    unsigned long long accv[kAccPerLane];
    if (pre_acc && tid < 64) acc_load_wave<false, false>(pre_acc, tid, accv);
    const uint32_t ovf_prev_part = (b == 0 && a.acc && tid < 64) ? ovf_part(a.ovf_prev, tid) : 0u;   // (work-group 0 books it below)
    unsigned long long state_word = 0;
    if (tid < kStateWords) state_word = reinterpret_cast<const unsigned long long*>(pre_st_in)[tid];
    const int X0 = (b / g.nbc) * g.TSR - g.D, Y0 = (b - (b / g.nbc) * g.nbc) * g.TS - g.D;
    auto store_state = [&]() {   // work-group 0, after a barrier: the state for the next launches and for the host
        if (b == 0 && tid < kStateWords) {
            const unsigned long long v = reinterpret_cast<const unsigned long long*>(&s_state)[tid];
            reinterpret_cast<unsigned long long*>(a.st_out)[tid] = v;
            if (a.snap) reinterpret_cast<unsigned long long*>(a.snap)[tid] = v;
        }
    };
    if (done0) {   // the loop is over: keep both state buffers identical, do nothing else
        if (tid < kStateWords) reinterpret_cast<unsigned long long*>(&s_state)[tid] = state_word;
        __syncthreads();
        store_state();
        return;
    }
    const bool pending = a.acc && it0 < a.j;
    const EvSetPtrs ev = a.sets.s[live_set];   // (the update's commit of a re-bin flip keeps cs ^ flip)
    const uint32_t* __restrict__ xy = ev.xy;
    const int32_t* __restrict__ t = ev.t;
    float2* __restrict__ p = ev.p;
    // Events in flight per thread: all loads of a pass are issued first.  U * THREADS covers a whole
    // bin of the usual size in ONE pass: a second pass would wait (vmcnt) for the first pass's
    // write-through stores of p before it sees its own loads (~2 us per extra pass, measured).  U is chosen by the host
    // from the events per bin: on a large image a bin holds a few hundred events, and the registers of eight events per
    // thread only cost occupancy there (1 work-group per CU instead of 2).
    uint32_t vxy[U];
    int32_t vt[U];
    float2 vp[U];
    uint32_t base = beg;
    auto load_pass = [&]() {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            // unconditional loads from a clamped index (no branch per load); dead slots are skipped below
            uint32_t i = base + k * THREADS + tid;
            i = i < end ? i : beg;
            vxy[k] = ld_idx(xy, i);
            vt[k] = ld_idx(t, i);
            vp[k] = ld_idx(p, i);
        }
    };
    const uint32_t m_e0 = SPLIT ? margin_preload(a, b, m_prev_n, tid) : 0xffffffffu;   // (ahead of the events: see the lean form)
    load_pass();
    asm volatile("" ::: "memory");   // (keep the requests above ahead of everything below)
    if (!COMPACT) {   // zero the LDS tile, 16 bytes per lane (overlaps the loads above)
        ulonglong2* z = reinterpret_cast<ulonglong2*>(s_tile);
        for (int i = tid; i < LL / 2; i += THREADS) z[i] = make_ulonglong2(0ull, 0ull);
    }
    if (tid < kStateWords) reinterpret_cast<unsigned long long*>(&s_state)[tid] = state_word;
    // previous projected positions (event.h:167-168 re-derived from the stored products): independent of the model
    double ppx[U], ppy[U];
    auto previous_positions = [&]() {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            ppx[k] = pr_from_p(vxy[k] & 0xffffu, vp[k].x);
            ppy[k] = pr_from_p(vxy[k] >> 16, vp[k].y);
        }
    };
    tl_stamp(a.tl, a.j, 5);
    if (pending && tid < 64) {
        // (this wave shares its SIMD with three others that are busy with their events: without priority it gets a
        // quarter of the issue slots and the update takes 1.4 us instead of ~0.5)
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long word = acc_reduce_wave(accv);
        __builtin_amdgcn_wave_barrier();   // (LDS operations of one wave complete in order: the state copy is in place)
        tl_stamp(a.tl, a.j, 6);
        model_update_wave(&s_state, word, tid, 1);
        __builtin_amdgcn_s_setprio(0);
        tl_stamp(a.tl, a.j, 7);
    } else {
        previous_positions();
    }
    __syncthreads();
    tl_stamp(a.tl, a.j, 1);
    const ScatterHot hs = scatter_hot(&s_state);
    if (b == 0 && pending && tid < 64) {   // bookkeeping only the stored state needs (fields the scatter does not read)
        const uint32_t ovf_prev = ovf_total_wave(ovf_prev_part);
        if (tid == 0) model_update_rest(&s_state, a.trace, a.cur ^ 1, ovf_prev);
    }
    if (hs.done) {
        __syncthreads();
        store_state();
        return;
    }
    if (pending && tid < 64) previous_positions();
    const ScatterGeo sg = {X0, Y0, L, LR, FMT == 2 ? g.zw : 0};
    uint32_t n_ovf = 0;
    if constexpr (COMPACT) {
        // Event lists.  Pass A: warp, store the products, count the entries per tile row (a bin of the usual size is one
        // pass and keeps its entries in registers).  Scan.  Pass B: every entry to its row's cursor -- from the registers,
        // or, for a bin of several passes, recomputed from the products just stored (the cheap half of the arithmetic).
        const bool single = end - beg <= (uint32_t)(THREADS * U);
        uint32_t code[U];
        int row[U];
        for (;;) {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const uint32_t i = base + k * THREADS + tid;
                code[k] = kNoEntry;
                if (i >= end) continue;
                code[k] = list_event<WARP>(hs, sg, a, p, i, vxy[k], vt[k], ppx[k], ppy[k], true, row[k], n_ovf);
                if (code[k] != kNoEntry) atomicAdd(&s_row[1 + row[k]], 1u);
            }
            base += THREADS * U;
            if (base >= end) break;
            load_pass();
            previous_positions();
        }
        tl_stamp(a.tl, a.j, 2);
        __syncthreads();
        tl_stamp(a.tl, a.j, 3);
        store_state();
        if (tid < 64) list_row_scan(s_row, (g.zw ? 3 : 1) * LR, a.chdr + (size_t)b * (size_t)((g.zw ? 3 : 1) * LR + 1), tid);
        __syncthreads();
        unsigned long long* vals = a.slabs + (size_t)b * (size_t)LL;
        uint16_t* cidx = a.cidx + (size_t)b * (size_t)LL;
        if (single) {
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (code[k] != kNoEntry) list_put(hs, sg, a, s_row, code[k], row[k], vt[k], (uint32_t)LL, vals, cidx, n_ovf);
        } else {
            for (base = beg; base < end; base += THREADS * U) {
                load_pass();   // (p: the products this thread stored in pass A)
                previous_positions();
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t i = base + k * THREADS + tid;
                    if (i >= end) continue;
                    int r;
                    const uint32_t cd = list_event<false>(hs, sg, a, p, i, vxy[k], vt[k], ppx[k], ppy[k], false, r, n_ovf);
                    if (cd != kNoEntry) list_put(hs, sg, a, s_row, cd, r, vt[k], (uint32_t)LL, vals, cidx, n_ovf);
                }
            }
        }
        if (n_ovf) { atomicAdd(ovf_counter(a.ovf_cur, b), n_ovf); a.ovf_cur[0] = 1u; }   // (count on the bin's line, flag: bf_device_fns.h)
        tl_stamp(a.tl, a.j, 4);
        return;
    }
    for (;;) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t i = base + k * THREADS + tid;
            if (i >= end) continue;
            scatter_event<WARP>(hs, sg, s_tile, a, p, i, vxy[k], vt[k], ppx[k], ppy[k], n_ovf);
        }
        base += THREADS * U;
        if (base >= end) break;
        load_pass();
        previous_positions();
    }
    if (n_ovf) { atomicAdd(ovf_counter(a.ovf_cur, b), n_ovf); a.ovf_cur[0] = 1u; }   // (count on the bin's line, flag: bf_device_fns.h)
    if (SPLIT) margin_clear(a, b, m_prev_n, m_e0, tid, THREADS);
    tl_stamp(a.tl, a.j, 2);
    __syncthreads();
    tl_stamp(a.tl, a.j, 3);
    store_state();
    if (SPLIT) flush_split<THREADS>(s_tile, a, b, X0, Y0, hs.C, s_mcnt, tid);
    else flush_tile<THREADS>(s_tile, a.slabs + (size_t)b * (size_t)LL, LL, tid);
    tl_stamp(a.tl, a.j, 4);
}

// K1 (binned), lean form: no update at the head -- the state it reads is final (the stencil kernel's last work-group
// updated it: "co_schedule", the throughput mode with several slice contexts per GPU).  No barrier between the loads
// and the scatter, so the waves of a work-group drift apart and overlap each other's memory latency.  Work-group 0
// still carries the state to the other buffer and to the host snapshot.
template <bool WARP, int THREADS, int U, int FMT>
__global__ __launch_bounds__(THREADS) void k_bin_warp_scatter_lean(const uint32_t* __restrict__ pre_bin_start, const DevState* pre_st_in,
                                                                   MomentAcc* /* pre_acc: unused here, same signature */, BinScatterArgs a) {
    constexpr bool COMPACT = FMT == 2;   // event lists / (0) dense slabs   (FMT 1, lists merged per pixel in the LDS tile, went in round 5: no BASELINE configuration took it)
    constexpr bool SPLIT = FMT == 3;                        // interior + margin (see flush_split)
    extern __shared__ unsigned long long s_tile[];   // (dense slabs only)
    __shared__ uint32_t s_row[FMT == 2 ? 1 + 3 * kMaxTileRows : 1];   // lists: entries per sort key ([zone,] tile row), then the keys' cursors
    __shared__ uint32_t s_mcnt[2];
    if (FMT == 2)
        for (int r = threadIdx.x; r <= 3 * kMaxTileRows; r += THREADS) s_row[r] = 0;
    if (SPLIT && threadIdx.x < 2) s_mcnt[threadIdx.x] = 0;
    const BinGrid& g = a.g;
    const int L = g.L, LR = g.LR, LL = g.LR * g.L;
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t beg = sload(pre_bin_start + b), end = sload(pre_bin_start + b + 1);
    const HotState h0 = sload(&pre_st_in->hot);   // one burst of scalar loads
    const uint32_t m_prev_n = SPLIT ? sload(a.mcount + b) : 0u;
    const int X0 = (b / g.nbc) * g.TSR - g.D, Y0 = (b - (b / g.nbc) * g.nbc) * g.TS - g.D;
    if (!COMPACT) {   // zero the LDS tile, 16 bytes per lane (overlaps the loads above)
        ulonglong2* z = reinterpret_cast<ulonglong2*>(s_tile);
        for (int i = tid; i < LL / 2; i += THREADS) z[i] = make_ulonglong2(0ull, 0ull);
    }
    // The state is NOT carried to the other buffer here while the loop runs: the stencil kernel's last work-group writes its
    // update there (and to the host snapshot) itself.  (Work-group 0 used to copy it -- a vector load that the scatter loop's
    // header, which waits vmcnt(0) for the registers of its previous pass, waited for before the first event load, on the
    // work-group that also finishes last.)  Only once the loop is over does every launch keep both buffers identical.
    if (h0.done) {
        if (b == 0 && tid < kStateWords) {
            const unsigned long long state_word = reinterpret_cast<const unsigned long long*>(a.st_in)[tid];
            reinterpret_cast<unsigned long long*>(a.st_out)[tid] = state_word;
            if (a.snap) reinterpret_cast<unsigned long long*>(a.snap)[tid] = state_word;
        }
        return;
    }
    ScatterHot hs;
    hs.done = h0.done; hs.bin_tbits = h0.bin_tbits; hs.bin_ok = h0.bin_ok; hs.fmt = h0.fmt; hs.scale = h0.scale; hs.C = h0.C;
    hs.wsx = h0.wsx; hs.wsy = h0.wsy; hs.x_sh = h0.x_sh; hs.y_sh = h0.y_sh; hs.tmin = h0.tmin; hs.wp = h0.wp;
    const EvSetPtrs ev = a.sets.s[h0.cs ^ h0.flip];
    const uint32_t* __restrict__ xy = ev.xy;
    const int32_t* __restrict__ t = ev.t;
    float2* __restrict__ p = ev.p;
    const ScatterGeo sg = {X0, Y0, L, LR, FMT == 2 ? g.zw : 0};
    uint32_t n_ovf = 0;
    __syncthreads();
    uint32_t vxy[U];
    int32_t vt[U];
    float2 vp[U];
    auto load_pass = [&](uint32_t base) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            uint32_t i = base + k * THREADS + tid;
            i = i < end ? i : beg;
            vxy[k] = ld_idx(xy, i);
            vt[k] = ld_idx(t, i);
            vp[k] = ld_idx(p, i);
        }
    };
    if constexpr (COMPACT) {   // event lists: see k_bin_warp_scatter
        const bool single = end - beg <= (uint32_t)(THREADS * U);
        uint32_t code[U];
        int row[U];
#pragma unroll
        for (int k = 0; k < U; ++k) code[k] = kNoEntry;   // (an empty bin does not enter the loop)
        for (uint32_t base = beg; base < end; base += THREADS * U) {
            load_pass(base);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const uint32_t i = base + k * THREADS + tid;
                code[k] = kNoEntry;
                if (i >= end) continue;
                code[k] = list_event<WARP>(hs, sg, a, p, i, vxy[k], vt[k], pr_from_p(vxy[k] & 0xffffu, vp[k].x),
                                           pr_from_p(vxy[k] >> 16, vp[k].y), true, row[k], n_ovf);
                if (code[k] != kNoEntry) atomicAdd(&s_row[1 + row[k]], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) list_row_scan(s_row, (g.zw ? 3 : 1) * LR, a.chdr + (size_t)b * (size_t)((g.zw ? 3 : 1) * LR + 1), tid);
        __syncthreads();
        unsigned long long* vals = a.slabs + (size_t)b * (size_t)LL;
        uint16_t* cidx = a.cidx + (size_t)b * (size_t)LL;
        if (single) {
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (code[k] != kNoEntry) list_put(hs, sg, a, s_row, code[k], row[k], vt[k], (uint32_t)LL, vals, cidx, n_ovf);
        } else {
            for (uint32_t base = beg; base < end; base += THREADS * U) {
                load_pass(base);   // (p: the products this thread stored in the first pass)
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t i = base + k * THREADS + tid;
                    if (i >= end) continue;
                    int r;
                    const uint32_t cd = list_event<false>(hs, sg, a, p, i, vxy[k], vt[k], pr_from_p(vxy[k] & 0xffffu, vp[k].x),
                                                          pr_from_p(vxy[k] >> 16, vp[k].y), false, r, n_ovf);
                    if (cd != kNoEntry) list_put(hs, sg, a, s_row, cd, r, vt[k], (uint32_t)LL, vals, cidx, n_ovf);
                }
            }
        }
        if (n_ovf) { atomicAdd(ovf_counter(a.ovf_cur, b), n_ovf); a.ovf_cur[0] = 1u; }   // (count on the bin's line, flag: bf_device_fns.h)
        return;
    }
    auto scatter_pass = [&](uint32_t base) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t i = base + k * THREADS + tid;
            if (i >= end) continue;
            scatter_event<WARP>(hs, sg, s_tile, a, p, i, vxy[k], vt[k], pr_from_p(vxy[k] & 0xffffu, vp[k].x),
                                pr_from_p(vxy[k] >> 16, vp[k].y), n_ovf);
        }
    };
    if constexpr (SPLIT) {
        // The first list entry of every thread is the OLDEST vector load of the kernel and is consumed after the scatter
        // loop, whose waits for the events' (younger) loads have covered it by then.  Requested behind the events it
        // would be waited for with vmcnt(0) -- i.e. together with the write-through stores of the products, ~1.7 us --,
        // and ahead of a loop whose header waits for the previous pass's registers it is waited for there: hence the
        // first pass outside the loop.
        const uint32_t m_e0 = margin_preload(a, b, m_prev_n, tid);
        uint32_t base = beg;
        load_pass(base);
        for (;;) {
            scatter_pass(base);
            base += THREADS * U;
            if (base >= end) break;
            load_pass(base);
        }
        if (n_ovf) { atomicAdd(ovf_counter(a.ovf_cur, b), n_ovf); a.ovf_cur[0] = 1u; }   // (count on the bin's line, flag: bf_device_fns.h)
        margin_clear(a, b, m_prev_n, m_e0, tid, THREADS);
    } else {
        for (uint32_t base = beg; base < end; base += THREADS * U) {
            load_pass(base);
            scatter_pass(base);
        }
        if (n_ovf) { atomicAdd(ovf_counter(a.ovf_cur, b), n_ovf); a.ovf_cur[0] = 1u; }   // (count on the bin's line, flag: bf_device_fns.h)
    }
    __syncthreads();
    if (SPLIT) flush_split<THREADS>(s_tile, a, b, X0, Y0, hs.C, s_mcnt, tid);
    else flush_tile<THREADS>(s_tile, a.slabs + (size_t)b * (size_t)LL, LL, tid);
}

// The pending update outside a warp+scatter launch (a warm start's gated final warp needs `done` of the batch's last
// iteration; nothing else does): one work-group, state updated in place.
__global__ __launch_bounds__(64) void k_finish_update(DevState* st, MomentAcc* acc, const uint32_t* ovf_prev, int j,
                                                            int cur_prev, bf_trace_rec* trace, DevState* snap, const uint32_t* lost,
                                                            unsigned long long* snap_seq, unsigned long long seq) {
    __shared__ DevState s_state;
    const int tid = threadIdx.x;
    const int done = st->hot.done, it = st->hot.it;
    // one-kernel iteration: sums that are not to be applied -- the last pass lost events, or the loop waits for a re-bin
    // or for the repeat of a pass (k_fused_pass)
    const bool trip = lost && *lost != 0u && st->hot.need_rebin != 2;
    const bool apply = lost ? st->hot.pend != 0 : it < j;   // (one-kernel iteration: the state says whether sums are waiting)
    if (tid < kStateWords)
        reinterpret_cast<unsigned long long*>(&s_state)[tid] = reinterpret_cast<const unsigned long long*>(st)[tid];
    if (trip && !done) {
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) { s_state.hot.need_rebin = 2; s_state.hot.redo = 1; s_state.hot.pend = 0; }
        __builtin_amdgcn_wave_barrier();
        if (tid < kStateWords)
            reinterpret_cast<unsigned long long*>(st)[tid] = reinterpret_cast<const unsigned long long*>(&s_state)[tid];
    } else if (!done && apply) {
        const uint32_t ovf = ovf_total_wave(ovf_part(ovf_prev, tid));
        unsigned long long accv[kAccPerLane];
        acc_load_wave<false, false>(acc, tid, accv);
        const unsigned long long word = acc_reduce_wave(accv);
        __builtin_amdgcn_wave_barrier();   // (LDS operations of one wave complete in order: the state copy is in place)
        model_update_wave(&s_state, word, tid, 1);
        if (tid == 0) {
            model_update_rest(&s_state, trace, cur_prev, ovf);
            s_state.hot.pend = 0;
        }
        __builtin_amdgcn_wave_barrier();
        if (tid < kStateWords)
            reinterpret_cast<unsigned long long*>(st)[tid] = reinterpret_cast<const unsigned long long*>(&s_state)[tid];
    }
    // the state after this batch, straight to the host's pinned copy (no copy command behind the batch)
    if (snap && tid < kStateWords)
        reinterpret_cast<unsigned long long*>(snap)[tid] = reinterpret_cast<const unsigned long long*>(&s_state)[tid];
    // ... and, once every word of it is out (one wave: its own stores, fenced at system scope), the sequence number the host
    // spins on: the host reads the model as soon as it exists, without waiting for the final warp behind this kernel and the
    // completion signal behind that (~12 us of a warm-started slice)
    if (snap_seq) {
        __threadfence_system();
        if (tid == 0) __hip_atomic_store(snap_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The scatter kernel's instantiations are the ones the host really picks (bf_run), not the full product: the update's home
// fixes the form -- HEAD: every work-group applies the pending update itself (a context that has the GPU to itself), lean: the
// stencil kernel's last work-group did ("co_schedule") -- and form + format fix the work-group sizes:
//     dense slabs / own pixels + margin plane:  head 1024 threads (bins of >= 1536 events) or 512, lean 512
//     event lists:                              256 (thousands of small bins) or 512, either form
// times 1, 2, 4 or 8 events per thread (dense slabs on 512 threads: also 10 and 12) and warp / no warp (the first pass of a cold run): 88 kernels, where the full product
// of the knobs that used to be options (3 sizes x 4 formats, both forms) was 192.  (The 512-thread head form is what a
// context alone runs at 640x480 -- BASELINE config 3: 540-690 bins of ~1500 events -- 11.7 us per launch against 17.4 with
// 1024 threads: measured when round 5's pruning first took it out.)
template <bool HEAD, int THREADS, int U, int FMT>
static hipError_t launch_bws2(const BinScatterArgs& a, bool warp, hipStream_t s) {
    // dynamic LDS: the bin's tile; event lists: none
    const size_t lds = FMT == 2 ? 0 : (size_t)a.g.LR * a.g.L * sizeof(unsigned long long) + 16;
    // LDS tiles above 64 KiB need the dynamic-LDS attribute raised (160 KiB per CU on gfx950).  The attribute belongs to
    // the (function, device) pair, so it is raised once per device the instantiation is launched on: a bit per device
    // ordinal, set after the calls succeeded (two threads racing here both make the calls, which is harmless).
    static std::atomic<unsigned long long> raised{0ull};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const unsigned long long dev_bit = 1ull << (dev & 63);
    const void* fns[2] = {HEAD ? reinterpret_cast<const void*>(&k_bin_warp_scatter<true, THREADS, U, FMT>)
                               : reinterpret_cast<const void*>(&k_bin_warp_scatter_lean<true, THREADS, U, FMT>),
                          HEAD ? reinterpret_cast<const void*>(&k_bin_warp_scatter<false, THREADS, U, FMT>)
                               : reinterpret_cast<const void*>(&k_bin_warp_scatter_lean<false, THREADS, U, FMT>)};
    if (!(raised.load(std::memory_order_acquire) & dev_bit)) {
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kBinTileLdsMax);
            if (e != hipSuccess) return e;
        }
        raised.fetch_or(dev_bit, std::memory_order_release);
    }
    if constexpr (HEAD) {
        if (warp) launch_timed(k_bin_warp_scatter<true, THREADS, U, FMT>, dim3(a.g.nbins), dim3(THREADS), lds, s, a.bin_start, a.st_in, a.acc, a);
        else launch_timed(k_bin_warp_scatter<false, THREADS, U, FMT>, dim3(a.g.nbins), dim3(THREADS), lds, s, a.bin_start, a.st_in, a.acc, a);
    } else {
        if (warp) launch_timed(k_bin_warp_scatter_lean<true, THREADS, U, FMT>, dim3(a.g.nbins), dim3(THREADS), lds, s, a.bin_start, a.st_in, a.acc, a);
        else launch_timed(k_bin_warp_scatter_lean<false, THREADS, U, FMT>, dim3(a.g.nbins), dim3(THREADS), lds, s, a.bin_start, a.st_in, a.acc, a);
    }
    return hipSuccess;
}
template <bool HEAD, int THREADS, int FMT>
static hipError_t launch_bws(const BinScatterArgs& a, bool warp, int per_thread, hipStream_t s) {
    if (per_thread <= 1) return launch_bws2<HEAD, THREADS, 1, FMT>(a, warp, s);
    if (per_thread <= 2) return launch_bws2<HEAD, THREADS, 2, FMT>(a, warp, s);
    if (per_thread <= 4) return launch_bws2<HEAD, THREADS, 4, FMT>(a, warp, s);
    return launch_bws2<HEAD, THREADS, 8, FMT>(a, warp, s);
}

// `threads`: bin_scatter_threads()'s answer for this slice; `per_thread`: events a thread keeps in flight (1, 2, 4 or 8; dense slabs on 512 threads also 10 or 12).
// a.acc != NULL: the head form (the pending update's sums), else the lean one.
int bin_scatter_threads(int fmt, bool head, bool many_small_bins, double events_per_bin) {
    if (fmt == 2) return many_small_bins ? 256 : 512;
    return (head && events_per_bin >= 1536.0) ? 1024 : 512;
}
hipError_t launch_bin_warp_scatter(const BinScatterArgs& a, bool warp, int threads, int per_thread, hipStream_t s) {
    const bool head = a.acc != nullptr;
    if (a.compact == 2) {
        if (threads <= 256) return head ? launch_bws<true, 256, 2>(a, warp, per_thread, s) : launch_bws<false, 256, 2>(a, warp, per_thread, s);
        return head ? launch_bws<true, 512, 2>(a, warp, per_thread, s) : launch_bws<false, 512, 2>(a, warp, per_thread, s);
    }
    const bool wide = head && threads >= 1024;
    if (a.compact == 3)
        return wide ? launch_bws<true, 1024, 3>(a, warp, per_thread, s)
                    : (head ? launch_bws<true, 512, 3>(a, warp, per_thread, s) : launch_bws<false, 512, 3>(a, warp, per_thread, s));
    if (a.compact != 0) return hipErrorInvalidValue;
    // Dense bins of several thousand events on 512 threads: a pass that covers the FULLEST bin, not the average one (config 2:
    // 272 bins, 3673 events on average, 4912 at most -- with 8 per thread two thirds of the bins took a second pass)
    if (!wide && per_thread >= 12) return head ? launch_bws2<true, 512, 12, 0>(a, warp, s) : launch_bws2<false, 512, 12, 0>(a, warp, s);
    if (!wide && per_thread >= 10) return head ? launch_bws2<true, 512, 10, 0>(a, warp, s) : launch_bws2<false, 512, 10, 0>(a, warp, s);
    return wide ? launch_bws<true, 1024, 0>(a, warp, per_thread, s)
                : (head ? launch_bws<true, 512, 0>(a, warp, per_thread, s) : launch_bws<false, 512, 0>(a, warp, per_thread, s));
}


void launch_finish_update(DevState* st, MomentAcc* acc, const uint32_t* ovf_prev, int j, int cur_prev, bf_trace_rec* trace,
                          DevState* snap, hipStream_t s, const uint32_t* lost, unsigned long long* snap_seq, unsigned long long seq) {
    hipLaunchKernelGGL(k_finish_update, dim3(1), dim3(64), 0, s, st, acc, ovf_prev, j, cur_prev, trace, snap, lost, snap_seq, seq);
}

}  // namespace bf
