// bf_device.h -- structures shared between the gfx950 kernels and the C-ABI host code.
//
// HBM layout of one slice (all arrays owned by bf_ctx, sized at bf_create):
//   xy      u32[N]   fr_x | fr_y << 16   (sensor row / column, < 65536)
//   t       i32[N]   ns relative to the slice start (accel_lib.h:85 keeps `int t` too)
//   p       f32x2[N] the two f32 products kx*float(t), ky*float(t) of Event::apply_project
//                    (event.h:164-168).  pr is re-derived from them bit-exactly:
//                    pr_x = (double)(float)fr_x - (double)p.x / 10000.0
//   noise   u8[N]    optional (only when the caller passed a mask)
//   nxny    f64x2[N] written by the final warp only (Event::nx, ny)
//   uv      f64x2[N] written by compute_uv only
//   plane   2 x u64[R*C]  point-scatter accumulators, double buffered:
//                    PACKED: count << tbits | sum(t - tmin)
//                    SPLIT : u64 sum(t - tmin) in plane[], u32 count in cplane[]
//                    (in the tile-binned mode they only take the rare overflow events)
//   set[2]  two copies of (xy, t, p, perm): the tile-binned mode keeps events counting-
//                    sorted by the image tile of their current target and ping-pongs on re-bin
//   slabs   u64[nbins * L * L]  one private (TS+2D)^2 tile per bin, rewritten every iteration
//   time    f32[R*C] time image (stand-alone operators only)
//   gx, gy  f32[R*C] Scharr planes (stand-alone operators only)
//   acc     MomentAcc[2][kAccGroups]  moment sums of the stencil kernel: exact integer / fixed-point
//                    accumulators, added to with device atomics (order-free, so bit-reproducible)
//   state   DevState[2] the model, loop control and warp parameters of the fused run (the tile-binned
//                    loop ping-pongs: the update runs at the head of the next warp+scatter launch)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/bf_accel.h"

namespace bf {

constexpr int kThreads = 256;
constexpr int kEvPerThread = 4;   // 16-byte vector loads of xy / t / p
constexpr int kTileR = 16;        // stencil tile: rows
constexpr int kTileC = 64;        // stencil tile: columns
constexpr int kMaxHalfScale = 4;  // scale <= 9
constexpr int kRcpTab = 256;      // entries of the reciprocal table of the stencil kernel's division by the event count (time_from_sums)
// overflow counter of one iteration (tile-binned loop): a flag line and 16 counter lines of 32 words (bf_device_fns.h)
constexpr int kOvfLines = 16, kOvfStride = 32, kOvfSlotWords = (1 + kOvfLines) * kOvfStride;

// Parameters of one warp (Event::project_4param_reinit, event.h:99-110).
struct WarpParams {
    double dnx, dny, cx, cy, div;
    double c, s;   // cos(crl), sin(crl)
};

// Moment sums of one iteration, accumulated ACROSS work-groups in exact integer arithmetic (A.4 + A.6 in one pass;
// ci = i - R/2, cj = j - C/2 are centred integer pixel coordinates):
//   f[0..2]   n, sum ci, sum cj over valid pixels (two's complement)
//   f[3 + 2k], f[4 + 2k]   the k-th floating-point sum (sgx, sgy, sigx, sigy, sjgx, sjgy) as fixed point:
//             every work-group's f64 partial v (summed in a fixed order inside the group) is split into
//             hi = floor(v 2^12) and lo = floor((v 2^12 - hi) 2^52); the his and the los are added separately.
// Integer addition is associative, so the total does not depend on the order in which the work-groups arrive:
// device atomics replace the ordered gather of per-group records (752 records read by one CU took 3 us).  Resolution
// 2^-64 per partial; ranges: |v| < 2^49 (a partial is at most 2^16 pixels x 2^15 x |g| <= 2^7), <= 2^10 partials.
// kAccGroups copies on different cache lines spread the same-address atomics (~12 ns each; 752 work-groups that
// finish together would queue ~50 deep on 16 copies; the reader holds kAccGroups / 4 words per lane in registers).
constexpr int kAccGroups = 32;
constexpr int kAccFields = 15;
struct MomentAcc {
    unsigned long long f[16];   // 128 bytes: one line per group
};

// Geometry of the tile-binned scatter (bf_scatter.hip): image tiles of TSR rows x TS columns of scaled
// pixels, LDS / slab tiles of LR x L = (TSR + 2 D) x (TS + 2 D), nbr x nbc bins.  TS is a power of two
// (column -> bin by a shift); TSR is any multiple of 16 (row -> bin by an exact multiply-high), chosen by the
// host so that the number of bins fills the CUs (one work-group per bin).
struct BinGrid {
    int32_t TS, D, L, nbr, nbc, nbins;
    int32_t lg, TSR;   // TS == 1 << lg; D <= min(TS, TSR) / 2, so a pixel is covered by <= 2 x 2 bins
    int32_t LR;        // TSR + 2 D
    uint32_t mul_r;    // floor(2^32 / TSR) + 1: row / TSR == __umulhi(row, mul_r) for row < 2^20
    uint32_t mul_l;    // floor(2^32 / L) + 1: index / L of a tile-local pixel index (< 2^16)
    int32_t fz;        // one-kernel iteration (k_fused_pass): width E of the edge strips of a tile, 0 = the two-kernel loop.
                       // The counting sort then keys events by (bin, zone): nbins counts KEYS, kFusedZones per image tile.
    uint32_t mul_h;    // floor(2^32 / (L / 2)) + 1: row of a 16-byte PAIR of tile pixels (interior + margin format; L is even)
    int32_t zw;        // event lists: entries sorted by (column zone, row) with zones [0, zw), [zw, L - zw), [L - zw, L) of the tile's
                       // columns; zw = D + scale / 2 + 1 (what a neighbouring stencil tile can reach); 0: sorted by row only
};

// One-kernel iteration (bf_fused.hip, k_fused_pass).  A tile's events are sorted into nine zones by where their target
// lay when the bins were built: the centre and, clockwise from the top-left corner, the eight pieces of the strip of width
// E = H + D along the tile's edge (H = scale / 2 + 1: box sum + Scharr halo; D: the drift a binning tolerates).  A
// work-group reads its own tile's events and the strips of the eight neighbouring tiles that face it -- ten contiguous
// ranges of the sorted arrays; nothing is stored twice.
constexpr int kFusedZones = 9;    // C, TL, T, TR, R, BR, B, BL, L
constexpr int kFusedRanges = 10;  // own tile; N, S, W neighbours (one range each); E neighbour (two); four corners
struct FusedTab {                 // per tile, written by the counting sort's scan (k_bin_scan), read with scalar loads
    uint32_t pre[kFusedRanges];   // running index of the first event of range r (pre[0] == 0)
    uint32_t off[kFusedRanges];   // global index of an event of range r = its running index + off[r]
    uint32_t total;
    uint32_t zone[kFusedZones - 1];   // running index of the first event of zones 1 .. 8 inside range 0 (the tile's own events)
    uint32_t spare[3];
};
constexpr int kFusedTabWords = (int)(sizeof(FusedTab) / 4);
static_assert(kFusedTabWords == 32, "one s_load_dwordx16 pair");


// Fields every kernel of the loop reads.  They are contiguous so that a kernel issues ONE
// burst of scalar loads for them before it branches on `done` (each dependent scalar load
// that misses L2 costs ~1-2 us on the iteration's critical path).
struct HotState {
    int32_t done, it, binned, bin_tbits;
    uint32_t ovf_cnt[2];      // plane buffer [i] is dirty (stand-alone operators and the global-atomic loop; the
                              // tile-binned loop counts its overflow events in bf_ctx::d_ovf, outside the state)
    int32_t need_rebin, rebins;
    int32_t pp, redo;                 // one-kernel iteration: which of a set's two product arrays is current; the next pass repeats
                                      // the scatter of the last one after a re-bin (its sums were incomplete)
    int32_t pend, spare_;             // ... the sums of the last pass await their update (launch numbers do not tell: passes
                                      // that wait for a re-bin or repeat one do not advance the iteration);
                                      // spare_: launches of the persistent loop kernel (k_fused_loop) completed in this run
    int32_t cs, flip, bin_ok, fmt;    // live event set; flip = a re-bin moved the events to set cs^1; fmt = this slice's scatter writes
                                      // COMPACT lists (1) instead of dense slabs (0) (informative: bf_scatter.hip);
                                      // bin_ok = the per-bin packing of this binning fits 64 bits (else: overflow path)
                                    // (committed by the next update)
    // window (host-written at set_cloud)
    int32_t scale, R, C, wsx, wsy, x_sh, y_sh, tbits;
    long long tmin;
    WarpParams wp;
};

struct DevState {
    HotState hot;
    double x_shift, y_shift;
    // --- loop control (optimizer_rolling.h:36,59-63) ---
    float x_div, y_div, rot_div, div_div;
    float old_dx, old_dy, old_rot, old_div;
    int32_t max_iter, hard_cap, rc, trace_cap, nblocks;
    uint32_t last_ovf;                // overflow events of the last iteration applied (tile-binned loop)
    uint32_t ovf_total, n_events;
    // --- drift tracking of the tile-binned scatter: warp parameters at the last re-bin, and
    //     the largest |t| (ns) and lever arm (sensor px) an event of this slice can have ---
    WarpParams ref_wp;
    double t_abs_max, r_max, drift_limit;
    long long t_span;                 // tmax - tmin of the slice (ns): bound of one event's time addend
    int32_t run_tag;                  // what `done` is set to (non-zero; bf_run gives every run its own)
    int32_t last_j;                   // one-kernel iteration: launch number of the pass that wrote this copy of the state
    // --- model ---
    bf_model model;
};

static_assert(offsetof(DevState, run_tag) % 8 == 0 && offsetof(DevState, last_j) == offsetof(DevState, run_tag) + 4,
              "(run_tag, last_j) is one 8-byte word of the host's polled snapshot");

// The two event sets as kernel arguments; the live one is sets[hot.cs ^ hot.flip].
struct EvSetPtrs {
    uint32_t* xy;
    int32_t* t;
    float2* p;
    uint32_t* perm;
    float2* p2;   // one-kernel iteration: the products ping-pong between p and p2 (hot.pp), so that a neighbouring tile's
                  // work-group reads the previous positions while the owner stores the new ones
};
struct EvSets {
    EvSetPtrs s[2];
};

}  // namespace bf
