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
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <limits.h>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {


// Row -> bin row: exact division by the (not necessarily power-of-two) tile height.
__device__ __forceinline__ int row_bin(int row, const BinGrid& g) { return (int)__umulhi((uint32_t)row, g.mul_r); }

// Image tile of the current target of one event (clamped into the grid: events whose target
// is outside the image are rejected by the scatter but still need a home bin).
__device__ __forceinline__ int bin_of(uint32_t xy, float2 p, const HotState& hs, const BinGrid& g) {
    const double pr_x = pr_from_p(xy & 0xffffu, p.x);
    const double pr_y = pr_from_p(xy >> 16, p.y);
    int X = trunc_x86(pr_x * (double)hs.scale + (double)hs.x_sh);
    int Y = trunc_x86(pr_y * (double)hs.scale + (double)hs.y_sh);
    X = min(max(X, 0), hs.R - 1);
    Y = min(max(Y, 0), hs.C - 1);
    return row_bin(X, g) * g.nbc + (Y >> g.lg);
}

// The re-bin kernels are enqueued by the host at a fixed cadence and run only when the update
// asked for it (hot.need_rebin): no host round trip sits between "drifted" and "re-sorted".
//
// R1: per-bin event count; remembers each event's bin.  PREWARP: the warm-start warp
// of OptimizerRolling::set_model (optimizer_rolling.h:294-298) is applied on the way (the events must be
// sorted by where that warp puts them), saving a pass over the events.  A launch that has nothing to do
// also disarms the scatter kernel (see k_bin_scatter).
template <bool PREWARP>
__global__ __launch_bounds__(kThreads) void k_bin_count(EvSets sets, long long n,
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
    extern __shared__ uint32_t s_cnt[];
    for (int i = threadIdx.x; i < g.nbins; i += kThreads) s_cnt[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (long long)gridDim.x * kThreads) {
        const uint32_t v = e.xy[i];
        const int32_t ti = e.t[i];
        float2 q = e.p[i];
        if (PREWARP) {
            double nx, ny;
            warp_products(prewarp, pr_from_p(v & 0xffffu, q.x), pr_from_p(v >> 16, q.y), ti, q, nx, ny);
            e.p[i] = q;
        }
        const int b = bin_of(v, q, hs, g);
        binid[i] = (uint16_t)b;
        atomicAdd(&s_cnt[b], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < g.nbins; i += kThreads) {
        if (s_cnt[i]) atomicAdd(&hist_cnt[i], s_cnt[i]);
    }
}

// R2: exclusive scan of the counts -> bin_start.
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* __restrict__ hist_cnt, int nbins,
                                                   uint32_t* __restrict__ bin_start,
                                                   uint32_t* __restrict__ cursor, DevState* st,
                                                   uint32_t* __restrict__ armed, int pack_limit) {
    if (!st->hot.need_rebin || st->hot.done) return;
    __shared__ uint32_t s_sum[1024];
    __shared__ uint32_t s_maxc[1024];
    const int tid = threadIdx.x;
    const int per = (nbins + 1023) / 1024;
    uint32_t local = 0, maxc = 0;
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < nbins) { local += hist_cnt[b]; maxc = max(maxc, hist_cnt[b]); }
    }
    s_sum[tid] = local; s_maxc[tid] = maxc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan (sum) / running maximum
        uint32_t v = (tid >= off) ? s_sum[tid - off] : 0u;
        uint32_t mc = (tid >= off) ? s_maxc[tid - off] : 0u;
        __syncthreads();
        s_sum[tid] += v;
        s_maxc[tid] = max(s_maxc[tid], mc);
        __syncthreads();
    }
    uint32_t run = s_sum[tid] - local;   // exclusive prefix of this thread's first bin
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < nbins) {
            bin_start[b] = run;
            run += hist_cnt[b];
            cursor[b] = 0;
            hist_cnt[b] = 0;   // ready for the next re-bin
        }
    }
    if (tid == 1023) {
        bin_start[nbins] = s_sum[1023];
        // Packing of the per-bin tiles (count << tbits | time sum).  Whatever is summed in packed form downstream -- a
        // tile pixel, the <= 2 x 2 slabs merged at a pixel, the s x s box around it -- is a sum over events of at most
        // four bins, each adding 1 and at most t_span: the fields need bits(4 maxc) and bits(4 maxc t_span), with maxc
        // the fullest bin.  (A slice-wide bound -- bits(N) + bits(sum of all times) -- stops fitting 64 bits just above
        // 1M events x 30 ms.)  If even this does not fit (nearly all events in one bin), bin_ok = 0 sends every event
        // down the exact overflow path (unpacked u64 + u32 planes).
        const unsigned long long m4 = 4ull * (unsigned long long)s_maxc[1023];
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
constexpr int kBsPerThread = 16;
constexpr int kBsEvents = kThreads * kBsPerThread;   // 4096 events, 80 KB of LDS staging
__global__ __launch_bounds__(kThreads) void k_bin_scatter(EvSets sets, int has_perm,
                                                          const uint16_t* __restrict__ binid, long long n,
                                                          const uint32_t* __restrict__ bin_start,
                                                          uint32_t* __restrict__ cursor, int nbins,
                                                          const DevState* __restrict__ st,
                                                          const uint32_t* __restrict__ armed) {
    if (!*armed) return;
    const int cs = st->hot.cs;
    const EvSetPtrs src = sets.s[cs], dst = sets.s[cs ^ 1];
    const bool perm_in = has_perm || st->hot.rebins > 1;
    extern __shared__ uint32_t s_u32[];
    uint32_t* s_cnt = s_u32;                  // [nbins] local histogram, then exclusive local offsets
    uint32_t* s_base = s_u32 + nbins;         // [nbins] global position of the bin's first local event
    uint32_t* s_xy = s_base + nbins;          // staging, sorted by bin
    int32_t* s_t = reinterpret_cast<int32_t*>(s_xy + kBsEvents);
    uint32_t* s_perm = reinterpret_cast<uint32_t*>(s_t + kBsEvents);
    uint16_t* s_bin = reinterpret_cast<uint16_t*>(s_perm + kBsEvents);
    float2* s_p = reinterpret_cast<float2*>(s_bin + kBsEvents);   // (8-byte aligned: all counts above are even)
    __shared__ uint32_t s_wsum[kThreads / 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < nbins; i += kThreads) s_cnt[i] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * kBsEvents;
    const int live = (int)((n - base) < kBsEvents ? (n - base) : kBsEvents);
    uint32_t rank[kBsPerThread];
    int bin[kBsPerThread];
#pragma unroll
    for (int k = 0; k < kBsPerThread; ++k) {
        const int j = k * kThreads + tid;
        bin[k] = -1;
        if (j < live) {
            bin[k] = binid[base + j];
            rank[k] = atomicAdd(&s_cnt[bin[k]], 1u);
        }
    }
    __syncthreads();
    // reserve the global ranges, then turn the histogram into exclusive local offsets (block scan)
    {
        const int per = (nbins + kThreads - 1) / kThreads;
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
        const int j = k * kThreads + tid;
        if (bin[k] >= 0) {
            const long long i = base + j;
            const uint32_t o = s_cnt[bin[k]] + rank[k];
            s_xy[o] = src.xy[i];
            s_t[o] = src.t[i];
            s_p[o] = src.p[i];
            s_perm[o] = perm_in ? src.perm[i] : (uint32_t)i;
            s_bin[o] = (uint16_t)bin[k];
        }
    }
    __syncthreads();
    // write out in sorted order: slot j of bin b goes to s_base[b] + (j - local offset of b)
    for (int j = tid; j < live; j += kThreads) {
        const int b = s_bin[j];
        const uint32_t o = s_base[b] + ((uint32_t)j - s_cnt[b]);
        dst.xy[o] = s_xy[j];
        dst.t[o] = s_t[j];
        dst.p[o] = s_p[j];
        dst.perm[o] = s_perm[j];
    }
}

// K1 (binned): warp + LDS scatter + slab flush, one work-group per bin.
template <bool WARP, int THREADS>
__global__ __launch_bounds__(THREADS) void k_bin_warp_scatter(
    EvSets sets, const uint32_t* __restrict__ bin_start, unsigned long long* __restrict__ slabs,
    unsigned long long* __restrict__ ovf_plane, uint32_t* __restrict__ ovf_cplane, DevState* st,
    BinGrid g, int cur, int check_done, unsigned long long* tl, int tl_launch) {
    extern __shared__ unsigned long long s_tile[];
    const int L = g.L, LR = g.LR, LL = g.LR * g.L;
    const int b = blockIdx.x;
    tl_stamp(tl, tl_launch, 0);
#ifdef BF_TIMELINE
    // per-work-group stamps of launch 20 (third 2048-entry block of the timeline buffer; tl points at the second)
    unsigned long long* tlw = (tl && tl_launch == 20 && b < 512 && threadIdx.x == 0) ? tl + 2048 + b * 4 : nullptr;
    if (tlw) tlw[0] = wall_clock64();
#endif
    // everything the block needs from global memory is requested up front, in one burst
    const uint32_t beg = bin_start[b], end = bin_start[b + 1];
    const HotState hs = st->hot;
    const int X0 = (b / g.nbc) * g.TSR - g.D, Y0 = (b - (b / g.nbc) * g.nbc) * g.TS - g.D;
    {   // zero the LDS tile, 16 bytes per lane (overlaps the scalar loads above)
        ulonglong2* z = reinterpret_cast<ulonglong2*>(s_tile);
        for (int i = threadIdx.x; i < LL / 2; i += THREADS) z[i] = make_ulonglong2(0ull, 0ull);
    }
    if (check_done && hs.done) return;
    tl_stamp(tl, tl_launch, 1);
    const EvSetPtrs ev = sets.s[hs.cs ^ hs.flip];
    const uint32_t* __restrict__ xy = ev.xy;
    const int32_t* __restrict__ t = ev.t;
    float2* __restrict__ p = ev.p;
    const WarpParams& wp = hs.wp;
    const int s = hs.scale, x_sh = hs.x_sh, y_sh = hs.y_sh, hsc = hs.scale / 2;
    const int wsx = hs.wsx, wsy = hs.wsy, C = hs.C, tbits = hs.bin_tbits;
    const bool bin_ok = hs.bin_ok != 0;
    const long long tmin = hs.tmin;
    uint32_t n_ovf = 0;
    __syncthreads();
    // Events in flight per thread: all loads of a pass are issued first.  U * THREADS covers a whole
    // bin of the usual size in ONE pass: a second pass would wait (vmcnt) for the first pass's
    // write-through stores of p before it sees its own loads (~2 us per extra pass, measured).
    constexpr int U = 8192 / THREADS;
    for (uint32_t base = beg; base < end; base += THREADS * U) {
        uint32_t vxy[U];
        int32_t vt[U];
        float2 vp[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            // unconditional loads from a clamped index (no branch per load); dead slots are skipped below
            uint32_t i = base + k * THREADS + threadIdx.x;
            i = i < end ? i : beg;
            vxy[k] = xy[i];
            vt[k] = t[i];
            vp[k] = p[i];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t i = base + k * THREADS + threadIdx.x;
            if (i >= end) continue;
            const uint32_t v = vxy[k];
            const int32_t ti = vt[k];
            float2 q = vp[k];
            const uint32_t fx = v & 0xffffu, fy = v >> 16;
            double pr_x = pr_from_p(fx, q.x);
            double pr_y = pr_from_p(fy, q.y);
            if (WARP) {   // event.h:100-108,164-168 -- same arithmetic as k_warp_scatter
                double nx, ny;
                warp_products(wp, pr_x, pr_y, ti, q, nx, ny);
                // write-through as well (see the slab flush below)
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(&p[i]),
                                   ((unsigned long long)__float_as_uint(q.y) << 32) | (unsigned long long)__float_as_uint(q.x),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pr_x = pr_from_p(fx, q.x);
                pr_y = pr_from_p(fy, q.y);
            }
            const int X = trunc_x86(pr_x * (double)s + (double)x_sh);   // accel_lib.h:154-158
            const int Y = trunc_x86(pr_y * (double)s + (double)y_sh);
            if (!((X >= wsx + hsc) || (X < hsc) || (Y >= wsy + hsc) || (Y < hsc))) {
                const unsigned long long dt = (unsigned long long)((long long)ti - tmin);
                const int lx = X - X0, ly = Y - Y0;
                if (bin_ok && lx >= 0 && lx < LR && ly >= 0 && ly < L) {
                    atomicAdd(&s_tile[__mul24(lx, L) + ly], (1ull << tbits) + dt);
                } else {   // drifted out of this bin's tile: exact, slow path
                    const size_t kk = (size_t)X * (size_t)C + (size_t)Y;
                    atomicAdd(&ovf_plane[kk], dt);
                    atomicAdd(&ovf_cplane[kk], 1u);
                    ++n_ovf;
                }
            }
        }
    }
    if (n_ovf) atomicAdd(&st->hot.ovf_cnt[cur], n_ovf);
    tl_stamp(tl, tl_launch, 2);
    __syncthreads();
#ifdef BF_TIMELINE
    if (tlw) { tlw[1] = wall_clock64(); tlw[3] = end - beg; }
#endif
    tl_stamp(tl, tl_launch, 3);
    {   // flush: the whole tile (nothing to zero, no atomics).  WRITE-THROUGH stores (agent-scope
        // relaxed = global_store ... sc1): with plain stores the ~15 MB of slabs (+ 8 MB of p) sat
        // dirty in the L2s until the end of the kernel, and their write-back stretched the kernel
        // boundary to ~5.6 us (measured; "B / 6 TB/s" in the MI355X notes).
        // (16 bytes per lane: L is even, so LL is, and a slab starts on a 16-byte boundary)
        unsigned long long* dst = slabs + (size_t)b * (size_t)LL;
        const bf_u32x4* src4 = reinterpret_cast<const bf_u32x4*>(s_tile);
        for (int i = threadIdx.x; i < LL / 2; i += THREADS) {
            const bf_u32x4 v = src4[i];
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + 2 * i), "v"(v) : "memory");
        }
    }
    tl_stamp(tl, tl_launch, 4);
#ifdef BF_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tlw) tlw[2] = wall_clock64();
#endif
}

// K3 (binned): merge the slabs covering each pixel, box-sum, normalise, then the shared tail.
// HS = scale / 2 is a template parameter so that the tile geometry is constexpr (index
// arithmetic by multiply-shift), TS is a power of two (shifts), D <= TS / 2 (a pixel is
// covered by at most 2 x 2 bins) and every slab load of a thread is issued up front.
template <int HS>
__device__ __forceinline__ void stencil_binned_body(const StencilArgs& a) {
    tl_stamp(a.tl, a.tl_launch, 0);
    const HotState hs = a.st->hot;   // one burst of scalar loads, then the branch
    if (a.check_done && hs.done) return;
    tl_stamp(a.tl, a.tl_launch, 1);
    constexpr int TR = kTileR, TC = kTileC;
    constexpr int H = HS + 1;
    constexpr int PR = TR + 2 * H, PC = TC + 2 * H;
    constexpr int TH = TR + 2, TW = TC + 2;
    constexpr int NC = (PR * PC + kThreads - 1) / kThreads;
    __shared__ unsigned long long s_acc[PR * PC];
    __shared__ float s_time[TH * TW];
    __shared__ Sums s_red[kThreads / 64];
    const int R = a.R, C = a.C;
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * TC;
    const BinGrid g = a.g;
    const int bt = hs.bin_tbits;
    const unsigned long long bm = (1ull << bt) - 1ull;
    // (static indices only: a runtime index would push the HotState copy into scratch memory)
    const bool ovf = (a.cur ? hs.ovf_cnt[1] : hs.ovf_cnt[0]) != 0;
    const int LLi = g.LR * g.L;

    static_assert(TR + 2 * H <= 32, "a tile plus halo must fit the smallest bin height (32)");
    // Row -> bin without a per-pixel division: the tile's rows (with halo) span TR + 2 H <= 32 <= TSR rows, so both
    // gr - D and gr + D cross at most one bin boundary inside the tile.  The bin rows at the tile's first row,
    // the boundary rows and the slab offsets of those bin rows are UNIFORM (scalar unit); a pixel only compares
    // and adds.  (Measured, scripts/micro/rates.hip: every vector multiply -- 24-bit, 32-bit lo / hi -- and every
    // f64 op or conversion issues at ~4.3 cycles per wave per SIMD, adds / logic at ~2.5; what counts is the
    // instruction count per pixel, which this form halves for the address arithmetic.)
    const int bl = row_bin(max(r0 - H - g.D, 0), g), bh = row_bin(max(r0 - H + g.D, 0), g);
    const int bl_next = (bl + 1) * g.TSR, bh_next = (bh + 1) * g.TSR;
    // element offset of pixel (gr, gc) in the slab of bin (br, bc): (br nbc + bc) LL + (gr - br TSR + D) L + (gc - bc TS + D)
    //   = [br nbc LL - (br TSR - D) L]  +  gr L  +  [bc LL - bc TS + D + gc]
    const int rb_l0 = bl * g.nbc * LLi - (bl * g.TSR - g.D) * g.L, rb_l1 = rb_l0 + g.nbc * LLi - g.TSR * g.L;
    const int rb_h0 = bh * g.nbc * LLi - (bh * g.TSR - g.D) * g.L, rb_h1 = rb_h0 + g.nbc * LLi - g.TSR * g.L;
    unsigned long long w[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int idx = tid + c * kThreads;
        const int pr = idx / PC, pc = idx - pr * PC;
        const int gr = r0 - H + pr, gc = c0 - H + pc;
        const bool in = idx < PR * PC && gr >= 0 && gr < R && gc >= 0 && gc < C;
        // bin (br, bc) holds rows [br*TSR - D, br*TSR + TSR + D)
        const bool l_up = max(gr - g.D, 0) >= bl_next, h_up = gr + g.D >= bh_next;
        const int brl = bl + (l_up ? 1 : 0), brh = min(bh + (h_up ? 1 : 0), g.nbr - 1);
        const int bcl = max(gc - g.D, 0) >> g.lg, bch = min((gc + g.D) >> g.lg, g.nbc - 1);
        const int grL = __mul24(gr, g.L);
        const int rowpart_l = (l_up ? rb_l1 : rb_l0) + grL;
        const int rowpart_h = ((brh > bh) ? rb_h1 : rb_h0) + grL;
        const int colpart_l = __mul24(bcl, LLi) - (bcl << g.lg) + g.D + gc;
        const int colpart_h = __mul24(bch, LLi) - (bch << g.lg) + g.D + gc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool use = in && (!(q & 2) || brh > brl) && (!(q & 1) || bch > bcl);
            // 32-bit element offsets (the slabs and planes are far below 2^32 bytes): base + offset addressing
            w[c][q] = use ? a.slabs[(uint32_t)(((q & 2) ? rowpart_h : rowpart_l) + ((q & 1) ? colpart_h : colpart_l))] : 0ull;
        }
    }
    // The slab accumulators stay PACKED (count << tbits | time sum) through the merge and the box sum: k_bin_scan sized
    // the fields for any sum over the events of up to four bins, which covers the <= 2 x 2 slabs at a pixel and the
    // s x s box around it.  One 64-bit add per contribution, one unpack per pixel.  Events that took the overflow path
    // are outside that bound (they come from any bin): their planes are read unpacked below, only when there are any.
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int idx = tid + c * kThreads;
        if (idx < PR * PC) s_acc[idx] = (w[c][0] + w[c][1]) + (w[c][2] + w[c][3]);
    }
    tl_stamp(a.tl, a.tl_launch, 2);
    __syncthreads();
    tl_stamp(a.tl, a.tl_launch, 3);
    for (int idx = tid; idx < TH * TW; idx += kThreads) {
        const int tr = idx / TW, tc = idx - tr * TW;
        const int gr = r0 - 1 + tr, gc = c0 - 1 + tc;
        float tv = 0.f;
        if (gr >= 0 && gr < R && gc >= 0 && gc < C) {
            // s x s box sum == the s x s splat of accel_lib.h:160-165 on integer planes
            unsigned long long pk = 0;
#pragma unroll
            for (int da = 0; da <= 2 * HS; ++da)
#pragma unroll
                for (int db = 0; db <= 2 * HS; ++db) pk += s_acc[(tr + da) * PC + (tc + db)];
            unsigned long long acc = pk & bm;
            uint32_t cacc = (uint32_t)(pk >> bt);
            if (ovf) {   // rare: the overflow planes (u64 time sums, u32 counts) straight from memory, box by box
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
            tv = time_from_sums(cacc, (long long)acc, a.tmin);
            if (tr >= 1 && tr <= TR && tc >= 1 && tc <= TC) {
                if (a.time_out) a.time_out[(size_t)gr * C + gc] = tv;
                if (a.count_out) a.count_out[(size_t)gr * C + gc] = cacc;
            }
        }
        s_time[idx] = tv;
    }
    __syncthreads();
    tl_stamp(a.tl, a.tl_launch, 4);
    const bool do_zero = a.zero_plane && (a.cur ? hs.ovf_cnt[0] : hs.ovf_cnt[1]) != 0;
    stencil_tail<TR, TC>(a, s_time, s_red, r0, c0, do_zero);
}

// Two builds of the same body.  The default one takes the registers it wants (~115: four waves per SIMD) and is
// the fastest for one slice at a time.  The "co-scheduled" one is held to five waves per SIMD (<= 102 VGPRs, a
// few spills): slower alone (+0.8 us), but two of its waves fit a SIMD next to a resident K1 work-group of another
// slice context (4 x 77 VGPRs), which is worth +8 % when several contexts share the GPU (option "co_schedule").
template <int HS>
__global__ __launch_bounds__(kThreads) void k_stencil_binned(StencilArgs a) {
    stencil_binned_body<HS>(a);
}
template <int HS>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_stencil_binned_co(StencilArgs a) {
    stencil_binned_body<HS>(a);
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

void launch_stencil_binned(const StencilArgs& a, dim3 grid, hipStream_t s) {
#define BF_K3(HS_)                                                                              \
    if (a.co_schedule) launch_timed(k_stencil_binned_co<HS_>, grid, dim3(kThreads), 0, s, a);   \
    else launch_timed(k_stencil_binned<HS_>, grid, dim3(kThreads), 0, s, a)
    switch (a.scale / 2) {
        case 0: BF_K3(0); break;
        case 1: BF_K3(1); break;
        case 2: BF_K3(2); break;
        case 3: BF_K3(3); break;
        default: BF_K3(4); break;
    }
#undef BF_K3
}

// ---------------------------------------------------------------------------------------
void launch_rebin(const EvSets& sets, int has_perm, long long n, DevState* st, const BinGrid& g,
                  uint16_t* binid, uint32_t* hist_cnt, uint32_t* bin_start,
                  uint32_t* cursor, uint32_t* armed, const WarpParams* prewarp, int pack_limit, hipStream_t s) {
    if (n <= 0) return;
    long long blocks = (n + kThreads * 8 - 1) / (kThreads * 8);
    if (blocks > 1024) blocks = 1024;
    if (prewarp)
        hipLaunchKernelGGL(k_bin_count<true>, dim3((unsigned)blocks), dim3(kThreads), (size_t)g.nbins * 4, s, sets, n,
                           st, g, binid, hist_cnt, armed, *prewarp);
    else
        hipLaunchKernelGGL(k_bin_count<false>, dim3((unsigned)blocks), dim3(kThreads), (size_t)g.nbins * 4, s, sets, n,
                           st, g, binid, hist_cnt, armed, WarpParams{});
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, hist_cnt, g.nbins, bin_start, cursor,
                       st, armed, pack_limit);
    const size_t lds = ((size_t)g.nbins * 2 + (g.nbins & 1)) * 4 + (size_t)kBsEvents * (4 + 4 + 4 + 2 + 8);
    hipLaunchKernelGGL(k_bin_scatter, dim3((unsigned)((n + kBsEvents - 1) / kBsEvents)), dim3(kThreads), lds, s, sets,
                       has_perm, binid, n, bin_start, cursor, g.nbins, st, armed);
}

template <int THREADS>
static void launch_bws(const EvSets& sets, const uint32_t* bin_start, unsigned long long* slabs, unsigned long long* ovf_plane, uint32_t* ovf_cplane,
                       DevState* st, const BinGrid& g, int cur, bool warp, int check_done, unsigned long long* tl,
                       int tl_launch, hipStream_t s) {
    const size_t lds = (size_t)g.LR * g.L * sizeof(unsigned long long);
    if (warp)
        launch_timed(k_bin_warp_scatter<true, THREADS>, dim3(g.nbins), dim3(THREADS), lds, s, sets, bin_start, slabs,
                     ovf_plane, ovf_cplane, st, g, cur, check_done, tl, tl_launch);
    else
        launch_timed(k_bin_warp_scatter<false, THREADS>, dim3(g.nbins), dim3(THREADS), lds, s, sets, bin_start, slabs,
                     ovf_plane, ovf_cplane, st, g, cur, check_done, tl, tl_launch);
}

void launch_bin_warp_scatter(const EvSets& sets, const uint32_t* bin_start, unsigned long long* slabs, unsigned long long* ovf_plane,
                             uint32_t* ovf_cplane, DevState* st, const BinGrid& g, int cur, bool warp,
                             int check_done, int threads, unsigned long long* tl, int tl_launch, hipStream_t s) {
    if (threads >= 1024) launch_bws<1024>(sets, bin_start, slabs, ovf_plane, ovf_cplane, st, g, cur, warp, check_done, tl, tl_launch, s);
    else if (threads >= 512) launch_bws<512>(sets, bin_start, slabs, ovf_plane, ovf_cplane, st, g, cur, warp, check_done, tl, tl_launch, s);
    else launch_bws<256>(sets, bin_start, slabs, ovf_plane, ovf_cplane, st, g, cur, warp, check_done, tl, tl_launch, s);
}

template <bool W, int T>
static hipError_t raise_lds() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_warp_scatter<W, T>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

int bin_kernel_setup() {
    // LDS tiles above 64 KiB need the dynamic-LDS attribute raised (160 KiB per CU on gfx950)
    hipError_t e[6] = {raise_lds<true, 256>(),  raise_lds<false, 256>(),  raise_lds<true, 512>(),
                       raise_lds<false, 512>(), raise_lds<true, 1024>(), raise_lds<false, 1024>()};
    for (hipError_t x : e)
        if (x != hipSuccess) return -1;
    // the LDS-staged counting-sort scatter: 88 KB of staging + two words per bin (+ 32 B static)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 64) != hipSuccess)
        return -1;
    return 0;
}

}  // namespace bf
