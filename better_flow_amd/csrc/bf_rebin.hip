// bf_rebin.hip -- the counting sort of the tile-binned loops (events by the image tile of their CURRENT target: k_bin_count /
// k_bin_scan / k_bin_scatter, once per slice and again when the model has drifted by more than the margin) and the
// start-of-run kernel.  Why tiles at all: bf_scatter.hip.
#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <limits.h>
#include <cstdlib>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {

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

int bin_kernel_setup() {
    // the LDS-staged counting-sort scatter: 88 KB of staging + two words per bin (+ 32 B static)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 64) != hipSuccess)
        return -1;
    return 0;
}

}  // namespace bf
