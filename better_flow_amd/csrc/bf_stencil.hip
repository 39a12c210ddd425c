// bf_stencil.hip -- the stencil kernel of the tile-binned loop (K3): slab merge / list splat, box sum, time image, Scharr,
// moment sums, exact accumulators [, model / loop update in its tail].
#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <limits.h>
#include <cstdlib>

#include "bf_device.h"
#include "bf_device_fns.h"
#include "bf_kernels.h"

namespace bf {

constexpr int kStencilSgprs = 74;

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
        // With column zones (BinGrid::zw; the scatter kernel sorted a bin's entries by (zone, row)) the unit is a SEGMENT = one
        // zone of one bin, rows clipped as before, and only the segments whose columns can reach the tile's boxes are taken:
        // of a neighbouring bin column the one zone that faces the tile.  A candidate per lane (<= 3 bin rows x 3 bin columns
        // x 3 zones), the kept ones compacted by a ballot.
        constexpr int kMaxBins = 32;
        __shared__ uint32_t s_eoff[kMaxBins + 1];
        __shared__ int2 s_ebin[kMaxBins];   // x: element offset of the segment's entries minus its first flattened entry number; y: oy << 16 | ox & 0xffff
        __shared__ int s_nseg;
        const int ncol = bc_hi - bc_lo + 1;
        const int zw = g.zw, nz = zw ? 3 : 1;
        const int ncand = min((br_hi - br_lo + 1) * ncol * nz, 64);
        if (tid < 64) {
            uint32_t n = 0;
            int bin = 0, r = 0, cc = 0;
            uint32_t first = 0;
            bool keep = false;
            if (tid < ncand) {   // the segment's entries in the tile rows that can reach this stencil tile
                const int bi = tid / nz, z = tid - bi * nz;
                r = bi / ncol; cc = bi - r * ncol;
                bin = (br_lo + r) * g.nbc + bc_lo + cc;
                const int top = (br_lo + r) * g.TSR - g.D;   // image row of tile row 0
                const int lo = min(max(r0 - 1 - HS - top, 0), g.LR), hi = min(max(r0 + TR + HS + 1 - top, 0), g.LR);
                // image columns of the zone [z_lo, z_hi) against the columns whose boxes touch the tile's time pixels
                const int left = ((bc_lo + cc) << g.lg) - g.D;
                const int z_lo = left + (z == 0 ? 0 : (z == 1 ? zw : g.L - zw)), z_hi = left + (nz == 1 ? g.L : (z == 0 ? zw : (z == 1 ? g.L - zw : g.L)));
                keep = z_hi > c0 - 1 - HS && z_lo <= c0 + TC + HS;
                if (keep) {
                    const uint32_t* crow = a.chdr + (size_t)bin * (size_t)(nz * g.LR + 1) + z * g.LR;
                    // (a list holds LL entries: the surplus of a fuller bin went down the overflow path)
                    first = min(crow[lo], (uint32_t)LLi);
                    n = min(crow[hi], (uint32_t)LLi) - first;
                }
            }
            uint32_t incl = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o, 64);
                if (tid >= o) incl += v;
            }
            const unsigned long long kept = __ballot(keep);
            const int pos = __popcll(kept & ((1ull << tid) - 1ull));
            if (keep && pos < kMaxBins) {
                s_eoff[pos + 1] = incl;
                const int oy_ = (br_lo + r) * g.TSR - g.D - (r0 - 1), ox_ = ((bc_lo + cc) << g.lg) - g.D - (c0 - 1);   // (|.| < 2^15)
                s_ebin[pos] = make_int2(bin * LLi + (int)first - (int)(incl - n), (int)(((uint32_t)oy_ << 16) | ((uint32_t)ox_ & 0xffffu)));
            }
            if (tid == 0) { s_eoff[0] = 0; s_nseg = min(__popcll(kept), kMaxBins); }
        }
        if (a.check_done && hs.done) return;   // (uniform; before the first barrier)
        __syncthreads();   // (the box plane is zero, the bin table is in place)
        const int nbin_ = __builtin_amdgcn_readfirstlane(s_nseg);
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

}  // namespace bf
