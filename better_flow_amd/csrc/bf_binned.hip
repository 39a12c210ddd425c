// bf_binned.hip -- tile-binned form of the warp+scatter kernel (K1) for gfx950.
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


// Row -> bin row: exact division by the (not necessarily power-of-two) tile height.
constexpr int kStencilSgprs = 74;
__device__ __forceinline__ int row_bin(int row, const BinGrid& g) { return (int)__umulhi((uint32_t)row, g.mul_r); }

// 64-bit load at (uniform base) + (uniform byte offset) + (per-thread byte offset): a BUFFER load -- base in a resource
// descriptor (four scalar registers), the row part of the offset in a scalar register, the column part in one vector register
// that is the same for every row of the thread.  No vector instruction per load for its address: as plain global loads the
// compiler formed every address with a 64-bit vector add (the zero-extended column offset is computed in another basic block
// than the load, so its scalar-base + 32-bit-offset form was not matched): 26 of the stencil kernel's ~620 vector
// instructions per wave.  (num_records = 2^32 - 1 bytes: no clamping is relied on; word 3: 32-bit raw data format, gfx9.)
typedef unsigned int bf_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_of(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ unsigned long long buf_ld_u64(__amdgpu_buffer_rsrc_t r, uint32_t thread_bytes, uint32_t uniform_bytes) {
    const bf_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)thread_bytes, (int)uniform_bytes, 0);
    return ((unsigned long long)v.y << 32) | (unsigned long long)v.x;
}

// Element idx of an array at a uniform base: scalar base register pair + 32-bit per-thread byte offset (no 64-bit vector address
// arithmetic per access).  The event arrays of a tile-binned slice stay below 2^32 bytes (bf_set_cloud: < 2^29 events).
template <class T>
__device__ __forceinline__ T ld_idx(const T* base, uint32_t idx) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + idx * (uint32_t)sizeof(T));
}

// Image tile of the current target of one event (clamped into the grid: events whose target
// is outside the image are rejected by the scatter but still need a home bin).
__device__ __forceinline__ int bin_of(uint32_t xy, float2 p, const HotState& hs, const BinGrid& g) {
    const double pr_x = pr_from_p(xy & 0xffffu, p.x);
    const double pr_y = pr_from_p(xy >> 16, p.y);
    int X = trunc_scatter(pr_x * (double)hs.scale + (double)hs.x_sh);
    int Y = trunc_scatter(pr_y * (double)hs.scale + (double)hs.y_sh);
    X = min(max(X, 0), hs.R - 1);
    Y = min(max(Y, 0), hs.C - 1);
    const int br = row_bin(X, g), bc = Y >> g.lg;
    const int b = br * g.nbc + bc;
    if (!g.fz) return b;
    // one-kernel iteration: the sort key is (tile, zone) -- see kFusedZones
    const int dx = X - br * g.TSR, dy = Y - (bc << g.lg);
    const int zr = dx < g.fz ? 0 : (dx >= g.TSR - g.fz ? 2 : 1);
    const int zc = dy < g.fz ? 0 : (dy >= g.TS - g.fz ? 2 : 1);
    // (zr, zc) -> C 0, TL 1, T 2, TR 3, R 4, BR 5, B 6, BL 7, L 8
    const int z = (zr == 0) ? (1 + zc) : (zr == 1 ? (zc == 0 ? 8 : (zc == 1 ? 0 : 4)) : (7 - zc));
    return b * kFusedZones + z;
}

// The re-bin kernels are enqueued by the host at a fixed cadence and run only when the update
// asked for it (hot.need_rebin): no host round trip sits between "drifted" and "re-sorted".
//
// R1: per-bin event count; remembers each event's bin.  PREWARP: the warm-start warp
// of OptimizerRolling::set_model (optimizer_rolling.h:294-298) is applied on the way (the events must be
// sorted by where that warp puts them), saving a pass over the events.  A launch that has nothing to do
// also disarms the scatter kernel (see k_bin_scatter).
// Work-groups of 1024 threads x 4 events (the same kBsEvents consecutive events per work-group as in k_bin_scatter), every
// load of a thread's events issued before the first is used; the local histogram is flushed with atomics into ONE OF
// kHistCopies copies of the global histogram (work-group b -> copy b % kHistCopies; the scan kernel adds the copies up):
// a slice in upload order has events of every bin in every work-group, and with one copy ~500 atomics queued on each
// address at ~30 ns apiece -- 15.6 us for this kernel; now 7-10.  (A (work-groups x bins) histogram matrix with column
// prefixes in the scan kernel, i.e. no atomics at all, was tried: the one-work-group column scan cost what the atomics
// had, 10-14 us against 5.)
constexpr int kBsThreads = 1024;
constexpr int kBsPerThread = 4;
constexpr int kBsEvents = kBsThreads * kBsPerThread;   // 4096 events per work-group, here and in k_bin_scatter
template <bool PREWARP>
__global__ __launch_bounds__(kBsThreads) void k_bin_count(EvSets sets, long long n,
                                                          const DevState* __restrict__ st, BinGrid g,
                                                          uint16_t* __restrict__ binid,
                                                          uint32_t* __restrict__ hist_cnt,
                                                          uint32_t* __restrict__ armed, WarpParams prewarp) {
    const HotState hs = st->hot;
    if (!hs.need_rebin || hs.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *armed = 0;
        return;
    }
    const EvSetPtrs e = sets.s[hs.cs ^ hs.flip];
    float2* const ep = hs.pp ? e.p2 : e.p;   // (the current products: see EvSetPtrs::p2)
    extern __shared__ uint32_t s_cnt[];
    for (int i = threadIdx.x; i < g.nbins; i += kBsThreads) s_cnt[i] = 0;
    __syncthreads();
    // every load of a thread's events is issued before the first is used
    const long long base = (long long)blockIdx.x * kBsEvents;
    uint32_t v[kBsPerThread];
    int32_t ti[kBsPerThread];
    float2 q[kBsPerThread];
#pragma unroll
    for (int k = 0; k < kBsPerThread; ++k) {
        long long i = base + k * kBsThreads + threadIdx.x;
        i = i < n ? i : base;
        v[k] = e.xy[i];
        if (PREWARP) ti[k] = e.t[i];
        q[k] = ep[i];
    }
#pragma unroll
    for (int k = 0; k < kBsPerThread; ++k) {
        const long long i = base + k * kBsThreads + threadIdx.x;
        if (i >= n) continue;
        if (PREWARP) {
            double nx, ny;
            warp_products(prewarp, pr_from_p(v[k] & 0xffffu, q[k].x), pr_from_p(v[k] >> 16, q[k].y), ti[k], q[k], nx, ny);
            ep[i] = q[k];
        }
        const int b = bin_of(v[k], q[k], hs, g);
        binid[i] = (uint16_t)b;
        atomicAdd(&s_cnt[b], 1u);
    }
    __syncthreads();
    uint32_t* copy = hist_cnt + (size_t)(blockIdx.x % kHistCopies) * (size_t)g.nbins;
    for (int i = threadIdx.x; i < g.nbins; i += kBsThreads)
        if (s_cnt[i]) atomicAdd(&copy[i], s_cnt[i]);
}

// R2: the copies of the histogram added up, exclusive scan of the counts -> bin_start.
constexpr int kMaxGridBins = 8192;   // (bf_set_cloud keeps the bin grid below this)
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* __restrict__ hist_cnt, int nbins,
                                                   uint32_t* __restrict__ bin_start,
                                                   uint32_t* __restrict__ cursor, DevState* st,
                                                   uint32_t* __restrict__ armed, int pack_limit,
                                                   int f_nbr, int f_nbc, uint32_t* __restrict__ ftab, uint32_t* __restrict__ lost) {
    if (!st->hot.need_rebin || st->hot.done) return;
    __shared__ uint32_t s_tot[kMaxGridBins + 1];
    __shared__ uint32_t s_wsum[16], s_wmax[16];
    __shared__ uint32_t s_fmax;
    if (threadIdx.x == 0) s_fmax = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = tid; b < nbins; b += 1024) {   // (consecutive threads -> consecutive bins: coalesced)
        uint32_t c[kHistCopies];
#pragma unroll
        for (int q = 0; q < kHistCopies; ++q) c[q] = hist_cnt[(size_t)q * (size_t)nbins + (size_t)b];
        uint32_t tot = 0;
#pragma unroll
        for (int q = 0; q < kHistCopies; ++q) {
            tot += c[q];
            if (c[q]) hist_cnt[(size_t)q * (size_t)nbins + (size_t)b] = 0;   // ready for the next re-bin
        }
        s_tot[b] = tot;
        cursor[b] = 0;
    }
    __syncthreads();
    const int per = (nbins + 1023) / 1024;
    uint32_t local = 0, maxc = 0;
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < nbins) { local += s_tot[b]; maxc = max(maxc, s_tot[b]); }
    }
    // inclusive scan of `local` / maximum over the work-group: wave scan by shuffles, then the 16 wave totals
    uint32_t incl = local, wmax = maxc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
        wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, off, 64));
    }
    if (lane == 63) { s_wsum[wave] = incl; s_wmax[wave] = wmax; }
    __syncthreads();
    uint32_t wbase = 0, total = 0, allmax = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t v = s_wsum[w];
        if (w < wave) wbase += v;
        total += v;
        allmax = max(allmax, s_wmax[w]);
    }
    incl += wbase;
    uint32_t run = incl - local;   // exclusive prefix of this thread's first bin
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < nbins) {
            const uint32_t cnt = s_tot[b];
            bin_start[b] = run;
            s_tot[b] = run;   // (the counts are no longer needed: the fused table below wants the starts)
            run += cnt;
        }
    }
    if (f_nbr > 0) {
        // One-kernel iteration: per image tile, the ten ranges of the sorted arrays its work-group reads (kFusedRanges) as
        // prefix sums of their lengths and `start - prefix` offsets -- a thread turns its running index v into the
        // global index v + offset[r], r the last range with prefix[r] <= v -- and the total.  The fullest list sizes
        // the packing of the LDS tile (every event of a list can meet at one pixel; a box sum adds each event once).
        if (tid == 0) s_tot[nbins] = total;
        __syncthreads();
        const int ntiles = f_nbr * f_nbc;
        for (int b = tid; b < ntiles; b += 1024) {
            const int br = b / f_nbc, bc = b - br * f_nbc;
            uint32_t lo[kFusedRanges], hi[kFusedRanges];
            auto rng = [&](int r, int dr, int dc, int z0, int z1) {
                const int nr = br + dr, nc = bc + dc;
                lo[r] = hi[r] = 0;
                if (nr < 0 || nr >= f_nbr || nc < 0 || nc >= f_nbc) return;
                const int k = (nr * f_nbc + nc) * kFusedZones;
                lo[r] = s_tot[k + z0];
                hi[r] = s_tot[k + z1];
            };
            rng(0, 0, 0, 0, 9);     // the tile's own events: all nine zones
            rng(1, -1, 0, 5, 8);    // north neighbour: BR, B, BL
            rng(2, 1, 0, 1, 4);     // south: TL, T, TR
            rng(3, 0, -1, 3, 6);    // west: TR, R, BR
            rng(4, 0, 1, 7, 9);     // east: BL, L ...
            rng(5, 0, 1, 1, 2);     // ... and TL
            rng(6, -1, -1, 5, 6);   // north-west: BR
            rng(7, -1, 1, 7, 8);    // north-east: BL
            rng(8, 1, -1, 3, 4);    // south-west: TR
            rng(9, 1, 1, 1, 2);     // south-east: TL
            uint32_t* row = ftab + (size_t)b * kFusedTabWords;
            uint32_t pre = 0;
#pragma unroll
            for (int r = 0; r < kFusedRanges; ++r) {
                row[r] = pre;
                row[kFusedRanges + r] = lo[r] - pre;
                pre += hi[r] - lo[r];
            }
            row[2 * kFusedRanges] = pre;
#pragma unroll
            for (int z = 1; z < kFusedZones; ++z) row[2 * kFusedRanges + z] = s_tot[b * kFusedZones + z] - lo[0];
            atomicMax(&s_fmax, pre);
        }
        __syncthreads();
    }
    if (tid == 1023) {
        bin_start[nbins] = total;
        if (f_nbr > 0) {
            allmax = (s_fmax + 3u) / 4u;   // (the bound below is written for "four bins": m4 = 4 allmax >= the fullest list)
            // a pass that lost events (see k_fused_pass) is repeated on the new bins, whether or not a later pass
            // has noticed the flag yet
            if (*lost || st->hot.need_rebin == 2) {
                if (st->hot.need_rebin != 2) st->ovf_total += 1;   // (bf_run_info::overflow_events counts the repeated passes of this loop)
                st->hot.redo = 1; st->hot.pend = 0;
            }
            *lost = 0u;
        }
        // Packing of the per-bin tiles (count << tbits | time sum).  Whatever is summed in packed form downstream -- a
        // tile pixel, the <= 2 x 2 slabs merged at a pixel, the s x s box around it -- is a sum over events of at most
        // four bins, each adding 1 and at most t_span: the fields need bits(4 maxc) and bits(4 maxc t_span), with maxc
        // the fullest bin.  (A slice-wide bound -- bits(N) + bits(sum of all times) -- stops fitting 64 bits just above
        // 1M events x 30 ms.)  If even this does not fit (nearly all events in one bin), bin_ok = 0 sends every event
        // down the exact overflow path (unpacked u64 + u32 planes).
        const unsigned long long m4 = 4ull * (unsigned long long)allmax;
        int cb = 0, tb = 0;
        for (unsigned long long v = m4; v; v >>= 1) ++cb;
        const unsigned long long span = (unsigned long long)(st->t_span > 0 ? st->t_span : 1);
        // bits(m4 * span) without overflowing 64 bits: bits(a b) <= bits(a) + bits(b)
        int sb = 0;
        for (unsigned long long v = span; v; v >>= 1) ++sb;
        tb = cb + sb;
        if (tb < 1) tb = 1;
        st->hot.bin_tbits = tb;
        st->hot.bin_ok = (tb + cb <= pack_limit) ? 1 : 0;   // (pack_limit: 64; lower only to test the fallback)
        st->hot.need_rebin = 0;
        st->hot.flip = 1;            // k_bin_scatter (next kernel) moves the events to set cs^1
        st->hot.rebins += 1;
        st->ref_wp = st->hot.wp;     // drift is measured from the model the bins were built for
        *armed = 1;
    }
}

// R3: move every event to its bin's range (order inside a bin is irrelevant: integer sums).
// Runs right after k_bin_scan set hot.flip; `armed` (set by the scan, cleared by the next
// sequence's idle k_bin_count) guards a second launch before the update has committed the flip.
//
// A work-group takes kBsEvents consecutive events, sorts them by bin INSIDE LDS (local counting sort:
// rank by LDS atomics, exclusive scan of the local histogram) and then writes them out in sorted
// order, so that consecutive lanes write consecutive addresses of a bin's range.  Writing each event
// straight to its slot (one 4 / 8-byte store per lane to ~64 different cache lines per instruction)
// took 47 us per 1M events; this form is bound by the 40 B/event it moves.
// (1024 threads x 4 events: with 256 x 16 a CU ran four waves, every phase -- ranks, staging, write-out -- at its full
// latency: 19.8 us per 1M events)
__global__ __launch_bounds__(kBsThreads) void k_bin_scatter(EvSets sets, int has_perm,
                                                          const uint16_t* __restrict__ binid, long long n,
                                                          const uint32_t* __restrict__ bin_start,
                                                          uint32_t* __restrict__ cursor, int nbins,
                                                          const DevState* __restrict__ st,
                                                          const uint32_t* __restrict__ armed) {
    if (!*armed) return;
    const int cs = st->hot.cs;
    const EvSetPtrs src = sets.s[cs], dst = sets.s[cs ^ 1];
    const float2* const src_p = st->hot.pp ? src.p2 : src.p;
    float2* const dst_p = st->hot.pp ? dst.p2 : dst.p;
    const bool perm_in = has_perm || st->hot.rebins > 1;
    extern __shared__ uint32_t s_u32[];
    uint32_t* s_cnt = s_u32;                  // [nbins] local histogram, then exclusive local offsets
    uint32_t* s_base = s_u32 + nbins;         // [nbins] global position of the bin's first local event
    uint32_t* s_xy = s_base + nbins;          // staging, sorted by bin
    int32_t* s_t = reinterpret_cast<int32_t*>(s_xy + kBsEvents);
    uint32_t* s_perm = reinterpret_cast<uint32_t*>(s_t + kBsEvents);
    uint16_t* s_bin = reinterpret_cast<uint16_t*>(s_perm + kBsEvents);
    float2* s_p = reinterpret_cast<float2*>(s_bin + kBsEvents);   // (8-byte aligned: all counts above are even)
    __shared__ uint32_t s_wsum[kBsThreads / 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < nbins; i += kBsThreads) s_cnt[i] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * kBsEvents;
    const int live = (int)((n - base) < kBsEvents ? (n - base) : kBsEvents);
    uint32_t rank[kBsPerThread];
    int bin[kBsPerThread];
    // the thread's events are requested together with their bin ids (they are only staged after two barriers and the
    // range reservation: loading them there put a second memory round trip on the work-group's chain)
    uint32_t exy[kBsPerThread], eperm[kBsPerThread];
    int32_t et[kBsPerThread];
    float2 ep[kBsPerThread];
#pragma unroll
    for (int k = 0; k < kBsPerThread; ++k) {
        const int j = k * kBsThreads + tid;
        const long long i = base + (j < live ? j : 0);
        bin[k] = j < live ? (int)binid[i] : -1;
        exy[k] = src.xy[i];
        et[k] = src.t[i];
        ep[k] = src_p[i];
        eperm[k] = perm_in ? src.perm[i] : (uint32_t)i;
    }
#pragma unroll
    for (int k = 0; k < kBsPerThread; ++k)
        if (bin[k] >= 0) rank[k] = atomicAdd(&s_cnt[bin[k]], 1u);
    __syncthreads();
    // reserve the global ranges, then turn the histogram into exclusive local offsets (block scan)
    {
        const int per = (nbins + kBsThreads - 1) / kBsThreads;
        uint32_t local = 0;
        for (int k = 0; k < per; ++k) {
            const int b = tid * per + k;
            if (b < nbins) {
                const uint32_t c = s_cnt[b];
                if (c) s_base[b] = bin_start[b] + atomicAdd(&cursor[b], c);
                local += c;
            }
        }
        // exclusive scan of `local` over the work-group: wave scan (DPP-free shuffles), then wave totals
        uint32_t incl = local;
        const int lane = tid & 63;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) s_wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < (tid >> 6); ++w) wbase += s_wsum[w];
        uint32_t run = wbase + incl - local;
        for (int k = 0; k < per; ++k) {
            const int b = tid * per + k;
            if (b < nbins) {
                const uint32_t c = s_cnt[b];
                s_cnt[b] = run;   // exclusive local offset of bin b
                run += c;
            }
        }
    }
    __syncthreads();
    // stage: event -> LDS slot (local offset of its bin + its rank)
#pragma unroll
    for (int k = 0; k < kBsPerThread; ++k) {
        if (bin[k] >= 0) {
            const uint32_t o = s_cnt[bin[k]] + rank[k];
            s_xy[o] = exy[k];
            s_t[o] = et[k];
            s_p[o] = ep[k];
            s_perm[o] = eperm[k];
            s_bin[o] = (uint16_t)bin[k];
        }
    }
    __syncthreads();
    // write out in sorted order: slot j of bin b goes to s_base[b] + (j - local offset of b)
    for (int j = tid; j < live; j += kBsThreads) {
        const int b = s_bin[j];
        const uint32_t o = s_base[b] + ((uint32_t)j - s_cnt[b]);
        dst.xy[o] = s_xy[j];
        dst.t[o] = s_t[j];
        dst_p[o] = s_p[j];
        dst.perm[o] = s_perm[j];
    }
}

constexpr int kMaxTileRows = 192;   // LR = TSR + 2 D <= 128 + 64

// One event of the tile-binned scatter, from its previous projected position: warp (event.h:100-108,164-168 -- same
// arithmetic as k_warp_scatter), store of the new products, splat centre (accel_lib.h:154-158), LDS accumulate or --
// drifted out of this bin's tile -- the exact overflow path.
struct ScatterGeo {
    int X0, Y0, L, LR;
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
// and splats entries with LDS atomics, so duplicates simply add.  No LDS tile: occupancy is set by registers, all bins of
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
        row = lx;
        return (uint32_t)(__mul24(lx, sg.L) + ly);
    }
    if (take_overflow) {
        overflow_add(hs, a, X, Y, (unsigned long long)((long long)ti - hs.tmin));
        ++n_ovf;
    }
    return kNoEntry;
}
// exclusive scan of the row counts s_row[1 .. LR] (one wave, 64 rows per step): cursors in LDS, row starts in `crow`
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
        overflow_add(hs, a, sg.X0 + row, sg.Y0 + (int)code - __mul24(row, sg.L), dt);
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
    __shared__ uint32_t s_row[FMT == 2 ? 1 + kMaxTileRows : 1];   // lists: entries per tile row, then the rows' cursors
    __shared__ uint32_t s_mcnt[2];
    if (FMT == 2)
        for (int r = threadIdx.x; r <= kMaxTileRows; r += THREADS) s_row[r] = 0;
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
    const ScatterGeo sg = {X0, Y0, L, LR};
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
        if (tid < 64) list_row_scan(s_row, LR, a.chdr + (size_t)b * (size_t)(LR + 1), tid);
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
    __shared__ uint32_t s_row[FMT == 2 ? 1 + kMaxTileRows : 1];   // lists: entries per tile row, then the rows' cursors
    __shared__ uint32_t s_mcnt[2];
    if (FMT == 2)
        for (int r = threadIdx.x; r <= kMaxTileRows; r += THREADS) s_row[r] = 0;
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
    const ScatterGeo sg = {X0, Y0, L, LR};
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
        if (tid < 64) list_row_scan(s_row, LR, a.chdr + (size_t)b * (size_t)(LR + 1), tid);
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
                                                            int cur_prev, bf_trace_rec* trace, DevState* snap, const uint32_t* lost) {
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
}

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

// K3 (binned): merge the slabs covering each pixel, box-sum, normalise, then the shared tail.
// HS = scale / 2 is a template parameter so that the tile geometry is constexpr (index
// arithmetic by multiply-shift), TS is a power of two (shifts), D <= TS / 2 (a pixel is
// covered by at most 2 x 2 bins) and every slab load of a thread is issued up front.
// (K3Pre: everything the kernel's first phase -- the slab loads' addresses -- is computed from, as leading scalar kernel
// arguments that the command processor preloads into SGPRs (14 dwords, the most the hardware takes; see the scatter kernel):
// the state's load and the address arithmetic no longer wait for the argument block's fetch: 12.44 -> 11.95 us per launch
// alone at config 2 (update in the tail), 6.5 -> 6.4 with the chip full.)
struct K3Pre {
    const unsigned long long* slabs;
    const DevState* st;
    int R, C, D, lg, nbc, nbr, LR, L, TSR;
    uint32_t mul_r;
};
template <int HS, int MODE, int NT>   // MODE: what the scatter kernel wrote -- 0 dense slabs, 1 lists, 2 interior + margin plane
__device__ __forceinline__ void stencil_binned_body(const StencilArgs& a, const K3Pre& pre) {
    constexpr bool COMPACT = MODE == 1;
    tl_stamp(a.tl, a.tl_launch, 0);
    // one burst of scalar loads; the state is not CONSUMED (not even for the early exit of a finished loop) before the
    // first vector loads below are out: their latencies overlap instead of adding up.  (Keeping EVERY scalar load -- state
    // and argument block -- behind the slab loads' issue, which the preloaded arguments allow, was built and measured: no
    // faster than this, EXPERIMENTS.md.)
    const HotState hs = sload(&pre.st->hot);
    tl_stamp(a.tl, a.tl_launch, 1);
    constexpr int TR = kTileR, TC = kTileC;
    constexpr int H = HS + 1;
    constexpr int PR = TR + 2 * H, PC = TC + 2 * H;
    constexpr int TH = TR + 2, TW = TC + 2;
    // thread mapping of the two pixel loops: wave wv takes rows wv, wv + NW, ... (uniform), lane l column l; the columns
    // beyond the 64th go to the first lanes of every wave
    static_assert(TC == 64 && NT % 64 == 0, "one lane per tile column");
    constexpr int NW = NT / 64;
    constexpr int KR = (PR + NW - 1) / NW, XC = PC - 64, XE = (KR * XC + 63) / 64;   // halo plane (slab merge)
    constexpr int KT = (TH + NW - 1) / NW, XT = TW - 64;                             // time tile
    static_assert(KT * XT <= 64, "the time tile's extra columns fit one pass");
    __shared__ unsigned long long s_acc[PR * PC];
    __shared__ float s_time[TH * TW];
    __shared__ Sums s_red[NT / 64];
    __shared__ double s_rcp[kRcpTab];   // 1 / i (time_from_sums): requested with the first loads, in LDS before the first barrier
    static_assert(NT >= kRcpTab, "one table entry per thread");
    const int R = pre.R, C = pre.C;
    const int tid = threadIdx.x;
    // (the table's address comes with the code -- see c_rcp --, so this load leaves at once, ahead of everything that waits for
    // the argument block; parked in LDS before the first barrier)
    const double rcp_v = tid < kRcpTab ? c_rcp.v[tid] : 0.0;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * TC;
    BinGrid g = a.g;   // (the fields the slab addresses use come from the preloaded arguments)
    g.D = pre.D; g.lg = pre.lg; g.nbc = pre.nbc; g.nbr = pre.nbr; g.LR = pre.LR; g.L = pre.L; g.TSR = pre.TSR; g.mul_r = pre.mul_r;
#ifdef BF_CENSUS
    if (a.tl && tid == 0 && !(a.check_done && hs.done)) {   // debug: work-groups resident per CU (third block of the timeline buffer: counts, then maxima)
        const uint32_t hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(6164);
        const uint32_t key = (xcc & 7u) * 128u + ((hw >> 8) & 127u);
        unsigned long long* cen = a.tl + 2 * 64 * 2 * 16;
        const unsigned long long n = atomicAdd(&cen[key], 1ull) + 1ull;
        atomicMax(&cen[1024 + key], n);
    }
#endif
    const int bt = hs.bin_tbits;
    const unsigned long long bm = (1ull << bt) - 1ull;
    // (static indices only: a runtime index would push the HotState copy into scratch memory)
    const bool ovf = sload(a.ovf_cur) != 0;   // events of this iteration took the overflow path (uniform)
    const int LLi = g.LR * g.L;

    constexpr bool compact = COMPACT;
    if (compact) {
        // COMPACT lists (the scatter kernel wrote, per bin, only the touched pixels: index + packed accumulator).
        // Every entry of the bins that can reach this tile is read once and SPLATTED: added to the (2 HS + 1)^2 time
        // pixels whose box contains it (accel_lib.h:160-165 literally), with LDS atomics into a plane that then holds
        // the box sums.  Work and traffic follow the events, not the area: at 1280x720 a tile of 1188 pixels sees ~10-100
        // entries, against 4 slab loads + 9 box terms for each of its pixels in the dense form.
        static_assert(TH * TW <= PR * PC, "the box plane fits the point plane's LDS");
        for (int idx = tid; idx < TH * TW; idx += NT) s_acc[idx] = 0ull;
        // bins whose LDS tile can hold a point within HS of the tile's time pixels (rows r0 - 1 .. r0 + TR, columns alike)
        const int br_lo = row_bin(max(r0 - 1 - HS - g.D, 0), g), br_hi = min(row_bin(min(r0 + TR + HS, R - 1) + g.D, g), g.nbr - 1);
        const int bc_lo = max(c0 - 1 - HS - g.D, 0) >> g.lg, bc_hi = min((min(c0 + TC + HS, C - 1) + g.D) >> g.lg, g.nbc - 1);
        // The entries of the reachable bins (<= 3 bin rows; 3 bin columns with 64-wide tiles, up to 7 with 16-wide ones)
        // form ONE list for the work-group: wave 0 fetches the counts, scans them and leaves, per bin, the first
        // flattened entry number, the element offset of its list and the position of its LDS-tile origin in the tile's
        // time-pixel coordinates; then thread t takes entries t, t + 256, ...: all 256 threads share the gather evenly
        // whatever the bins' sizes, and it costs two memory round trips (counts, entries) like any gather.
        constexpr int kMaxBins = 32;
        __shared__ uint32_t s_eoff[kMaxBins + 1];
        __shared__ int2 s_ebin[kMaxBins];   // x: element offset of the bin's list minus its first flattened entry number; y: oy << 16 | ox & 0xffff
        const int ncol = bc_hi - bc_lo + 1;
        const int nbin_ = min((br_hi - br_lo + 1) * ncol, kMaxBins);
        if (tid < 64) {
            uint32_t n = 0;
            int bin = 0, r = 0, cc = 0;
            uint32_t first = 0;
            if (tid < nbin_) {   // the bin's entries in the tile rows that can reach this stencil tile
                r = tid / ncol; cc = tid - r * ncol;
                bin = (br_lo + r) * g.nbc + bc_lo + cc;
                const int top = (br_lo + r) * g.TSR - g.D;   // image row of tile row 0
                const int lo = min(max(r0 - 1 - HS - top, 0), g.LR), hi = min(max(r0 + TR + HS + 1 - top, 0), g.LR);
                const uint32_t* crow = a.chdr + (size_t)bin * (size_t)(g.LR + 1);
                // (a list holds LL entries: the surplus of a fuller bin went down the overflow path)
                first = min(crow[lo], (uint32_t)LLi);
                n = min(crow[hi], (uint32_t)LLi) - first;
            }
            uint32_t incl = n;
#pragma unroll
            for (int o = 1; o < kMaxBins; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o, 64);
                if (tid >= o) incl += v;
            }
            if (tid < nbin_) {
                s_eoff[tid + 1] = incl;
                const int oy_ = (br_lo + r) * g.TSR - g.D - (r0 - 1), ox_ = ((bc_lo + cc) << g.lg) - g.D - (c0 - 1);   // (|.| < 2^15)
                s_ebin[tid] = make_int2(bin * LLi + (int)first - (int)(incl - n), (int)(((uint32_t)oy_ << 16) | ((uint32_t)ox_ & 0xffffu)));
            }
            if (tid == 0) s_eoff[0] = 0;
        }
        if (a.check_done && hs.done) return;   // (uniform; before the first barrier)
        __syncthreads();   // (the box plane is zero, the bin table is in place)
        const uint32_t E = s_eoff[nbin_];
        for (uint32_t e = tid; e < E; e += NT) {
            // which bin's list holds entry e: the number of bins whose first entry number is <= e (a compare and an add per
            // bin; uniform trip count, LDS broadcast reads), then ONE gather of that bin's record
            int jb = 0;
            for (int j = 1; j < nbin_; ++j) jb += e >= s_eoff[j] ? 1 : 0;
            const int2 eb = s_ebin[jb];
            const int base = eb.x, oy = eb.y >> 16, ox = (int)(short)(eb.y & 0xffff);
            const uint32_t idx = a.cidx[(uint32_t)(base + (int)e)];
            const unsigned long long v = pre.slabs[(uint32_t)(base + (int)e)];
            const int lx = (int)__umulhi(idx, g.mul_l);            // idx / L
            const int tr = oy + lx, tc = ox + (int)idx - lx * g.L;  // the point, in time-pixel coordinates
            if (tr >= -HS && tr < TH + HS && tc >= -HS && tc < TW + HS) {
                // (one unsigned compare per box row and per box column, not four signed ones per add)
                bool okr[2 * HS + 1], okc[2 * HS + 1];
#pragma unroll
                for (int d = 0; d <= 2 * HS; ++d) {
                    okr[d] = (unsigned)(tr + d - HS) < (unsigned)TH;
                    okc[d] = (unsigned)(tc + d - HS) < (unsigned)TW;
                }
                unsigned long long* q = &s_acc[(tr - HS) * TW + (tc - HS)];
#pragma unroll
                for (int da = 0; da <= 2 * HS; ++da)
#pragma unroll
                    for (int db = 0; db <= 2 * HS; ++db)
                        if (okr[da] && okc[db]) atomicAdd(&q[da * TW + db], v);
            }
        }
    } else if constexpr (MODE == 2) {
    // Interior + margin format (flush_split): one word of the tiled image per pixel -- bin (br, bc) keeps its TSR x TS pixels
    // at [(br nbc + bc) TSR TS + (gr - br TSR) TS + (gc - bc TS)] -- plus the margin-plane word where another bin's margin
    // can reach the pixel: within D of a boundary of its own bin.  Rows as in the dense form: the tile's rows cross at most
    // one bin boundary, so the row part of the offset is one of two UNIFORM values.  Thread mapping as in the dense form
    // below: a wave takes whole rows of the halo plane, one column per lane.
    static_assert(TR + 2 * H <= 32, "a tile plus halo must fit the smallest bin height (32)");
    const int TSA = g.TS * g.TSR;
    const int b0r = row_bin(max(r0 - H, 0), g), next_r = (b0r + 1) * g.TSR;
    const int rp0 = b0r * (g.nbc * TSA - g.TSR * g.TS), rp1 = rp0 + g.nbc * TSA - g.TSR * g.TS;
    const int cmask = g.TS - 1;
    auto col_of = [&](int pc, bool& in, bool& edge, int& off, int& gc) {
        gc = c0 - H + pc;
        in = gc >= 0 && gc < C;
        const int cc = gc & cmask;
        off = __mul24(gc >> g.lg, TSA) + cc;
        edge = cc < g.D || cc >= g.TS - g.D;
    };
    auto row_of = [&](int pr, bool& in, bool& edge, int& off, int& moff) {
        const int gr = r0 - H + pr;
        in = pr < PR && gr >= 0 && gr < R;
        const bool up = gr >= next_r;
        const int rr = gr - (up ? next_r : next_r - g.TSR);   // row inside the pixel's bin
        off = (up ? rp1 : rp0) + __mul24(gr, g.TS);
        edge = rr < g.D || rr >= g.TSR - g.D;
        moff = __mul24(gr, C);
    };
    unsigned long long w[KR][2], we[XE][2];
    const __amdgpu_buffer_rsrc_t slab_buf = buf_of(pre.slabs), margin_buf = buf_of(a.m_cur);
    bool c_in, c_edge;
    int c_off, c_gc;
    col_of(lane, c_in, c_edge, c_off, c_gc);
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        bool r_in, r_edge;
        int r_off, r_moff;
        row_of(wv + k * NW, r_in, r_edge, r_off, r_moff);   // (uniform: scalar unit)
        w[k][0] = w[k][1] = 0ull;
        if (r_in && c_in) {
            w[k][0] = buf_ld_u64(slab_buf, (uint32_t)c_off * 8u, (uint32_t)r_off * 8u);
            if (r_edge || c_edge) w[k][1] = buf_ld_u64(margin_buf, (uint32_t)c_gc * 8u, (uint32_t)r_moff * 8u);
        }
    }
#pragma unroll
    for (int x = 0; x < XE; ++x) {   // the columns beyond the 64th: lanes 0 .. KR XC - 1 of every wave, for the wave's own rows
        const int e = lane + 64 * x;
        bool r_in, r_edge, x_in, x_edge;
        int r_off, r_moff, x_off, x_gc;
        row_of(wv + (e / XC) * NW, r_in, r_edge, r_off, r_moff);
        col_of(64 + e % XC, x_in, x_edge, x_off, x_gc);
        we[x][0] = we[x][1] = 0ull;
        if (e < KR * XC && r_in && x_in) {
            we[x][0] = pre.slabs[(uint32_t)(r_off + x_off)];
            if (r_edge || x_edge) we[x][1] = a.m_cur[(uint32_t)(r_moff + x_gc)];
        }
    }
    if (a.check_done && hs.done) return;   // (uniform; before the first barrier -- the loads above are already out)
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int pr = wv + k * NW;
        if (pr < PR) s_acc[pr * PC + lane] = w[k][0] + w[k][1];
    }
#pragma unroll
    for (int x = 0; x < XE; ++x) {
        const int e = lane + 64 * x, pr = wv + (e / XC) * NW;
        if (e < KR * XC && pr < PR) s_acc[pr * PC + 64 + e % XC] = we[x][0] + we[x][1];
    }
    } else {
    static_assert(TR + 2 * H <= 32, "a tile plus halo must fit the smallest bin height (32)");
    // Row -> bin without a per-pixel division: the tile's rows (with halo) span TR + 2 H <= 32 <= TSR rows, so both
    // gr - D and gr + D cross at most one bin boundary inside the tile.  The bin rows at the tile's first row,
    // the boundary rows and the slab offsets of those bin rows are UNIFORM (scalar unit).
    // Thread mapping: a wave takes whole ROWS of the halo plane (rows wv, wv + NW, ...), one column per lane.  Everything
    // that depends on the row -- which bin rows cover it, the row part of the slab offsets, whether a second bin row has to
    // be read at all -- is then uniform and lives in the scalar unit; everything that depends on the column is computed once
    // per thread, not once per pixel; a load is "scalar row base + per-thread byte offset" and costs no vector instruction
    // for its address.  (With pixel idx = tid + c NT the row / column / bin arithmetic was ~60 vector instructions per slab
    // pixel, a third of the kernel's: rocprofv3 counted 1038 per wave, and with several contexts on the GPU the vector
    // units are what the loop saturates.)  The PC - 64 columns beyond the 64th go to the first lanes of every wave, for the
    // wave's own rows, pixel by pixel as before.
    const int bl = row_bin(max(r0 - H - g.D, 0), g), bh = row_bin(max(r0 - H + g.D, 0), g);
    const int bl_next = (bl + 1) * g.TSR, bh_next = (bh + 1) * g.TSR;
    // element offset of pixel (gr, gc) in the slab of bin (br, bc): (br nbc + bc) LL + (gr - br TSR + D) L + (gc - bc TS + D)
    //   = [br nbc LL - (br TSR - D) L]  +  gr L  +  [bc LL - bc TS + D + gc]
    const int rb_l0 = bl * g.nbc * LLi - (bl * g.TSR - g.D) * g.L, rb_l1 = rb_l0 + g.nbc * LLi - g.TSR * g.L;
    const int rb_h0 = bh * g.nbc * LLi - (bh * g.TSR - g.D) * g.L, rb_h1 = rb_h0 + g.nbc * LLi - g.TSR * g.L;
    auto col_of = [&](int pc, bool& in, bool& two, int& lo, int& hi) {
        const int gc = c0 - H + pc;
        in = gc >= 0 && gc < C;
        const int bcl = max(gc - g.D, 0) >> g.lg, bch = min((gc + g.D) >> g.lg, g.nbc - 1);
        lo = __mul24(bcl, LLi) - (bcl << g.lg) + g.D + gc;
        hi = __mul24(bch, LLi) - (bch << g.lg) + g.D + gc;
        two = bch > bcl;
    };
    auto row_of = [&](int pr, bool& in, bool& two, int& lo, int& hi) {
        const int gr = r0 - H + pr;
        in = pr < PR && gr >= 0 && gr < R;
        // bin (br, bc) holds rows [br*TSR - D, br*TSR + TSR + D)
        const bool l_up = max(gr - g.D, 0) >= bl_next, h_up = gr + g.D >= bh_next;
        const int brl = bl + (l_up ? 1 : 0), brh = min(bh + (h_up ? 1 : 0), g.nbr - 1);
        const int grL = __mul24(gr, g.L);
        lo = (l_up ? rb_l1 : rb_l0) + grL;
        hi = ((brh > bh) ? rb_h1 : rb_h0) + grL;
        two = brh > brl;
    };
    unsigned long long w[KR][4], we[XE][4];
    const __amdgpu_buffer_rsrc_t slab_buf = buf_of(pre.slabs);
    bool c_in, c_two;
    int c_lo, c_hi;
    col_of(lane, c_in, c_two, c_lo, c_hi);
    // No per-lane masking of the loads: a lane whose column lies outside the image, and the second-bin slot of a lane whose
    // column has no second bin, read an ALWAYS-ZERO cell instead -- column offset 0 of the row, i.e. local column 0 of the
    // slab of bin column 0, image column -D < 0, which no event is ever added to (the scatter kernel's window test) and which
    // the scatter kernel's flush writes as 0 in every launch.  What depends on the row (is it inside the image, does it have
    // a second bin row) is uniform: scalar branches, and the sums below repeat them -- so nothing is zero-filled and no
    // execution mask is touched: per row two to four loads and one to three 64-bit adds (before: four register pairs
    // zeroed, the execution mask saved and restored around each conditional load, three adds).  Integer sums: same bits.
    const uint32_t vo_lo = c_in ? (uint32_t)c_lo * 8u : 0u;
    const uint32_t vo_hi = (c_in && c_two) ? (uint32_t)c_hi * 8u : 0u;
    bool rin[KR], rtwo[KR];   // (uniform)
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        int r_lo, r_hi;
        row_of(wv + k * NW, rin[k], rtwo[k], r_lo, r_hi);   // (uniform: scalar unit)
        if (rin[k]) {
            w[k][0] = buf_ld_u64(slab_buf, vo_lo, (uint32_t)r_lo * 8u);
            w[k][1] = buf_ld_u64(slab_buf, vo_hi, (uint32_t)r_lo * 8u);
            if (rtwo[k]) {
                w[k][2] = buf_ld_u64(slab_buf, vo_lo, (uint32_t)r_hi * 8u);
                w[k][3] = buf_ld_u64(slab_buf, vo_hi, (uint32_t)r_hi * 8u);
            }
        }
    }
#pragma unroll
    for (int x = 0; x < XE; ++x) {   // the columns beyond the 64th: per-lane rows, so per-lane conditions as before
        const int e = lane + 64 * x;
        bool r_in, r_two, x_in, x_two;
        int r_lo, r_hi, x_lo, x_hi;
        row_of(wv + (e / XC) * NW, r_in, r_two, r_lo, r_hi);
        col_of(64 + e % XC, x_in, x_two, x_lo, x_hi);
        we[x][0] = we[x][1] = we[x][2] = we[x][3] = 0ull;
        if (e < KR * XC && r_in && x_in) {
            // 32-bit byte offsets (the slabs are far below 2^32 bytes)
            we[x][0] = buf_ld_u64(slab_buf, (uint32_t)(r_lo + x_lo) * 8u, 0u);
            if (x_two) we[x][1] = buf_ld_u64(slab_buf, (uint32_t)(r_lo + x_hi) * 8u, 0u);
            if (r_two) we[x][2] = buf_ld_u64(slab_buf, (uint32_t)(r_hi + x_lo) * 8u, 0u);
            if (r_two && x_two) we[x][3] = buf_ld_u64(slab_buf, (uint32_t)(r_hi + x_hi) * 8u, 0u);
        }
    }
    if (a.check_done && hs.done) return;   // (uniform; before the first barrier -- the loads above are already out)
    // The slab accumulators stay PACKED (count << tbits | time sum) through the merge and the box sum: k_bin_scan sized
    // the fields for any sum over the events of up to four bins, which covers the <= 2 x 2 slabs at a pixel and the
    // s x s box around it.  One 64-bit add per contribution, one unpack per pixel.  Events that took the overflow path
    // are outside that bound (they come from any bin): their planes are read unpacked below, only when there are any.
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int pr = wv + k * NW;
        if (pr < PR) {
            unsigned long long sum = 0ull;
            if (rin[k]) {
                sum = w[k][0] + w[k][1];
                if (rtwo[k]) sum += w[k][2] + w[k][3];
            }
            s_acc[pr * PC + lane] = sum;
        }
    }
#pragma unroll
    for (int x = 0; x < XE; ++x) {
        const int e = lane + 64 * x, pr = wv + (e / XC) * NW;
        if (e < KR * XC && pr < PR) s_acc[pr * PC + 64 + e % XC] = (we[x][0] + we[x][1]) + (we[x][2] + we[x][3]);
    }
    }   // dense slabs
    // Overflow events of this iteration (uniform, rare): the scatter kernel flagged their pixels in a bitmap; the rows of it
    // that the tile's boxes can reach (four words each: the tile's 64 columns and 32 on either side) are staged here, and a
    // time pixel reads the overflow planes only if a flagged pixel lies in its box (every pixel reading its whole box from
    // memory cost an iteration with overflow events about twice the time of a clean one at 640x480).
    __shared__ uint32_t s_bits[(TH + 2 * HS) * 4];
    if (ovf) {
        for (int i = tid; i < (TH + 2 * HS) * 4; i += NT) {
            const int gr = r0 - 1 - HS + (i >> 2);
            s_bits[i] = (gr >= 0 && gr < R) ? a.ovf_bits[(size_t)gr * (size_t)a.ovf_pitch + (size_t)((c0 >> 5) + (i & 3))] : 0u;
        }
    }
    if (tid < kRcpTab) s_rcp[tid] = rcp_v;
    tl_stamp(a.tl, a.tl_launch, 2);
    __syncthreads();
    tl_stamp(a.tl, a.tl_launch, 3);
    // (same thread mapping: uniform row, one column per lane -- the box's LDS addresses are one per-thread base plus
    // immediates, the row tests are scalar)
    // s x s box sum == the s x s splat of accel_lib.h:160-165 on integer planes (lists: the plane already holds box sums)
    auto box_at = [&](const int tr, const int tc) {
        if (compact) return s_acc[tr * TW + tc];
        unsigned long long pk = 0;
#pragma unroll
        for (int da = 0; da <= 2 * HS; ++da)
#pragma unroll
            for (int db = 0; db <= 2 * HS; ++db) pk += s_acc[(tr + da) * PC + (tc + db)];
        return pk;
    };
    auto time_px = [&](const int tr, const int tc, const unsigned long long pk) {
        const int idx = tr * TW + tc;
        const int gr = r0 - 1 + tr, gc = c0 - 1 + tc;
        float tv = 0.f;
        if (gr >= 0 && gr < R && gc >= 0 && gc < C) {
            unsigned long long acc = pk & bm;
            uint32_t cacc = (uint32_t)(pk >> bt);
            bool box_dirty = false;
            if (ovf) {
                const int ps = tc + 31 - HS;   // first column of the box, counted from the first staged column (c0 - 32)
                uint32_t any = 0;
#pragma unroll
                for (int da = 0; da <= 2 * HS; ++da) {
                    const uint32_t* rw = &s_bits[(tr + da) * 4 + (ps >> 5)];
                    const unsigned long long w2 = ((unsigned long long)rw[1] << 32) | rw[0];
                    any |= (uint32_t)(w2 >> (ps & 31)) & ((1u << (2 * HS + 1)) - 1u);
                }
                box_dirty = any != 0;
            }
            if (box_dirty) {   // rare: the overflow planes (u64 time sums, u32 counts) straight from memory, box by box
#pragma unroll
                for (int da = -HS; da <= HS; ++da)
#pragma unroll
                    for (int db = -HS; db <= HS; ++db) {
                        const int pr_ = gr + da, pc_ = gc + db;
                        if (pr_ >= 0 && pr_ < R && pc_ >= 0 && pc_ < C) {
                            acc += a.plane[(uint32_t)(__mul24(pr_, C) + pc_)];
                            cacc += a.cplane[(uint32_t)(__mul24(pr_, C) + pc_)];
                        }
                    }
            }
            tv = time_from_sums(cacc, (long long)acc, a.tmin, s_rcp);
            // (no time / count image out of this kernel: only bf_run launches it -- the stand-alone operators read the planes
            // with k_stencil, bf_kernels.hip)
        }
        s_time[idx] = tv;
    };
    constexpr int LW = TH % NW;   // the first LW waves take KT rows, the others KT - 1 (LW == 0: all KT)
    if constexpr (COMPACT) {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int tr = wv + k * NW;
            if (tr < TH) time_px(tr, lane, box_at(tr, lane));
        }
    } else {
        // Dense planes: a wave takes CONSECUTIVE rows (which rows a wave takes is free: the time tile is complete before anybody
        // reads it, and the order of the moment sums is the tail's, not this loop's), so that a lane's boxes share their rows:
        // the horizontal sums of the n + 2 HS plane rows under its n pixels once, then each box as 2 HS + 1 of them -- for
        // scale 3 and five rows 21 LDS reads and 24 64-bit adds instead of 45 and 40.
        const int nrows = (LW == 0 || wv < LW) ? KT : KT - 1;                              // (uniform)
        const int rbeg = (LW == 0 || wv < LW) ? wv * KT : LW * KT + (wv - LW) * (KT - 1);
        unsigned long long hsum[KT + 2 * HS];
#pragma unroll
        for (int j = 0; j < KT + 2 * HS; ++j) {
            hsum[j] = 0ull;
            if (j < nrows + 2 * HS) {
#pragma unroll
                for (int db = 0; db <= 2 * HS; ++db) hsum[j] += s_acc[(rbeg + j) * PC + (lane + db)];
            }
        }
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            if (i < nrows) {
                unsigned long long pk = 0;
#pragma unroll
                for (int da = 0; da <= 2 * HS; ++da) pk += hsum[i + da];
                time_px(rbeg + i, lane, pk);
            }
        }
    }
    // the tile's last XT columns: one column each for the waves that had a row less than the others (with 18 rows on four
    // waves: waves 2 and 3), so that every wave makes the same number of passes
    if constexpr (LW != 0 && NW - LW >= XT && TH <= 64) {
        if (wv >= LW && wv < LW + XT && lane < TH) time_px(lane, 64 + (wv - LW), box_at(lane, 64 + (wv - LW)));
    } else {
        if (lane < KT * XT) {
            const int tr = wv + (lane / XT) * NW;
            if (tr < TH) time_px(tr, 64 + lane % XT, box_at(tr, 64 + lane % XT));
        }
    }
    __syncthreads();
    tl_stamp(a.tl, a.tl_launch, 4);
    const bool do_zero = a.zero_plane && sload(a.ovf_prev) != 0;   // the other plane buffer is dirty: clear it for the next iteration
    // (the accumulator plane is free from here on: the wave totals of the moment sums go through it)
    static_assert((size_t)PR * PC >= (size_t)(NT / 64) * 192, "the reduction's scratch fits the accumulator plane");
    stencil_tail<TR, TC, NT, false>(a, s_time, s_red, r0, c0, do_zero, reinterpret_cast<double*>(s_acc));
}

// Two builds of each: as the compiler allocates it (~100 scalar registers: the rows live in the scalar unit), and with
// the scalar registers capped at what lets a CU hold 8 work-groups of 256 threads (<= 80: 7 up to 96, 6 up to 112 --
// MI355X_MICROARCH.md "Residency"; the surplus is spilled into vector-register lanes).  A launch that fills the GPU
// several times over runs the capped build: measured on one box, same inputs, eight config-2 slices side by side 60.8
// -> 56.2 us, 1280x720 (event lists) 57.4 -> 50.3 us per launch.  A launch of a few work-groups per CU is one
// work-group's latency chain long and the spills only lengthen it (346x260: 13.4 -> 13.8 us): the plain build.
#define BF_K3_PRE_PARAMS const unsigned long long* p_slabs, const DevState* p_st, int p_R, int p_C, int p_D, int p_lg, \
                          int p_nbc, int p_nbr, int p_LR, int p_L, int p_TSR, uint32_t p_mul_r
#define BF_K3_PRE_VALUE K3Pre{p_slabs, p_st, p_R, p_C, p_D, p_lg, p_nbc, p_nbr, p_LR, p_L, p_TSR, p_mul_r}
template <int HS, int MODE, int NT>
__global__ __launch_bounds__(NT) void k_stencil_binned(BF_K3_PRE_PARAMS, StencilArgs a) {
    stencil_binned_body<HS, MODE, NT>(a, BF_K3_PRE_VALUE);
}
template <int HS, int MODE, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(kStencilSgprs))) void k_stencil_binned_full(BF_K3_PRE_PARAMS, StencilArgs a) {
    stencil_binned_body<HS, MODE, NT>(a, BF_K3_PRE_VALUE);
}

// Plain launch, or (profiling armed) an extended launch whose events carry the kernel's own timestamps.
template <class K, class... A>
static void launch_timed(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    LaunchTimer& t = launch_timer();
    if (t.start && !t.consumed) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, t.start, t.stop, 0, args...);
        t.consumed = true;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
    }
}

void launch_stencil_binned(const StencilArgs& a, dim3 grid, hipStream_t s, int n_cus) {
    const bool full = n_cus > 0 && (long long)grid.x * grid.y >= 8ll * n_cus;   // more work-groups than the CUs hold at once
#define BF_K3_PRE a.slabs, a.st, a.R, a.C, a.g.D, a.g.lg, a.g.nbc, a.g.nbr, a.g.LR, a.g.L, a.g.TSR, a.g.mul_r
#define BF_K3(HS_)                                                                                                  \
    if (full) {                                                                                                     \
        if (a.compact == 3) launch_timed(k_stencil_binned_full<HS_, 2, kThreads>, grid, dim3(kThreads), 0, s, BF_K3_PRE, a);   \
        else if (a.compact) launch_timed(k_stencil_binned_full<HS_, 1, kThreads>, grid, dim3(kThreads), 0, s, BF_K3_PRE, a);   \
        else launch_timed(k_stencil_binned_full<HS_, 0, kThreads>, grid, dim3(kThreads), 0, s, BF_K3_PRE, a);                  \
    } else if (a.compact == 3) launch_timed(k_stencil_binned<HS_, 2, kThreads>, grid, dim3(kThreads), 0, s, BF_K3_PRE, a);     \
    else if (a.compact) launch_timed(k_stencil_binned<HS_, 1, kThreads>, grid, dim3(kThreads), 0, s, BF_K3_PRE, a);            \
    else launch_timed(k_stencil_binned<HS_, 0, kThreads>, grid, dim3(kThreads), 0, s, BF_K3_PRE, a)
    switch (a.scale / 2) {
        case 0: BF_K3(0); break;
        case 1: BF_K3(1); break;
        case 2: BF_K3(2); break;
        case 3: BF_K3(3); break;
        default: BF_K3(4); break;
    }
#undef BF_K3
#undef BF_K3_PRE
}

// ---------------------------------------------------------------------------------------
void launch_rebin(const EvSets& sets, int has_perm, long long n, DevState* st, const BinGrid& g,
                  uint16_t* binid, uint32_t* hist_cnt, uint32_t* bin_start,
                  uint32_t* cursor, uint32_t* armed, const WarpParams* prewarp, int pack_limit, hipStream_t s,
                  uint32_t* ftab, uint32_t* lost) {
    if (n <= 0) return;
    const unsigned blocks = (unsigned)((n + kBsEvents - 1) / kBsEvents);
    if (prewarp)
        hipLaunchKernelGGL(k_bin_count<true>, dim3(blocks), dim3(kBsThreads), (size_t)g.nbins * 4, s, sets, n,
                           st, g, binid, hist_cnt, armed, *prewarp);
    else
        hipLaunchKernelGGL(k_bin_count<false>, dim3(blocks), dim3(kBsThreads), (size_t)g.nbins * 4, s, sets, n,
                           st, g, binid, hist_cnt, armed, WarpParams{});
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, hist_cnt, g.nbins, bin_start, cursor, st, armed, pack_limit,
                       g.fz ? g.nbr : 0, g.nbc, ftab, lost);
    const size_t lds = ((size_t)g.nbins * 2 + (g.nbins & 1)) * 4 + (size_t)kBsEvents * (4 + 4 + 4 + 2 + 8);
    hipLaunchKernelGGL(k_bin_scatter, dim3(blocks), dim3(kBsThreads), lds, s, sets,
                       has_perm, binid, n, bin_start, cursor, g.nbins, st, armed);
}

// The scatter kernel's instantiations are the ones the host really picks (bf_run), not the full product: the update's home
// fixes the form -- HEAD: every work-group applies the pending update itself (a context that has the GPU to itself), lean: the
// stencil kernel's last work-group did ("co_schedule") -- and form + format fix the work-group sizes:
//     dense slabs / own pixels + margin plane:  head 1024 threads (bins of >= 1536 events) or 512, lean 512
//     event lists:                              256 (thousands of small bins) or 512, either form
// times 1, 2, 4 or 8 events per thread and warp / no warp (the first pass of a cold run): 80 kernels, where the full product
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

// `threads`: bin_scatter_threads()'s answer for this slice; `per_thread`: events a thread keeps in flight (1, 2, 4 or 8).
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
    return wide ? launch_bws<true, 1024, 0>(a, warp, per_thread, s)
                : (head ? launch_bws<true, 512, 0>(a, warp, per_thread, s) : launch_bws<false, 512, 0>(a, warp, per_thread, s));
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

// Start of a run, one launch: the host's state to the device (the struct travels as a kernel argument), and -- tile-binned
// run, or accumulators left dirty -- the overflow counters (slot j % 3 <- iteration j; slot 2 = "iteration -1") and both
// accumulator parities.
__global__ __launch_bounds__(kThreads) void k_run_init(DevState* st, DevState v, uint32_t* ovf, uint32_t prev_dirty, MomentAcc* acc,
                                                       int init_loop) {
    const int tid = threadIdx.x;
    if (tid < kStateWords) reinterpret_cast<unsigned long long*>(st)[tid] = reinterpret_cast<const unsigned long long*>(&v)[tid];
    if (!init_loop) return;
    // (three overflow-counter slots of 17 lines each; the flag of slot 2 -- "iteration -1" -- says whether the other plane buffer is dirty)
    if (tid < 3 * (1 + kOvfLines)) ovf[(tid / (1 + kOvfLines)) * kOvfSlotWords + (tid % (1 + kOvfLines)) * kOvfStride] = (tid == 2 * (1 + kOvfLines)) ? prev_dirty : 0u;
    // (three accumulator buffers and, behind them, the `lost` word of the one-kernel iteration: bf_ctx::d_acc)
    for (int i = tid; i < 3 * kAccGroups * 16 + 16; i += kThreads) (&acc[0].f[0])[i] = 0ull;
}
void launch_run_init(DevState* st, const DevState& v, uint32_t* ovf, uint32_t prev_dirty, MomentAcc* acc, bool init_loop, hipStream_t s) {
    hipLaunchKernelGGL(k_run_init, dim3(1), dim3(kThreads), 0, s, st, v, ovf, prev_dirty, acc, init_loop ? 1 : 0);
}

void launch_finish_update(DevState* st, MomentAcc* acc, const uint32_t* ovf_prev, int j, int cur_prev, bf_trace_rec* trace,
                          DevState* snap, hipStream_t s, const uint32_t* lost) {
    hipLaunchKernelGGL(k_finish_update, dim3(1), dim3(64), 0, s, st, acc, ovf_prev, j, cur_prev, trace, snap, lost);
}

int bin_kernel_setup() {
    // the LDS-staged counting-sort scatter: 88 KB of staging + two words per bin (+ 32 B static)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 64) != hipSuccess)
        return -1;
    return 0;
}

}  // namespace bf
