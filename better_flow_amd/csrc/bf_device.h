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
//   time    f32[R*C] time image (stand-alone operators only)
//   gx, gy  f32[R*C] Scharr planes (stand-alone operators only)
//   partial Partial[blocks]  per-work-group moment sums of the stencil kernel
//   state   DevState the model, loop control and warp parameters of the fused run
#pragma once
#include <stdint.h>

#include "../../include/bf_accel.h"

namespace bf {

constexpr int kThreads = 256;
constexpr int kEvPerThread = 4;   // 16-byte vector loads of xy / t / p
constexpr int kTileR = 16;        // stencil tile: rows
constexpr int kTileC = 64;        // stencil tile: columns
constexpr int kMaxHalfScale = 4;  // scale <= 9

// Parameters of one warp (Event::project_4param_reinit, event.h:99-110).
struct WarpParams {
    double dnx, dny, cx, cy, div;
    double c, s;   // cos(crl), sin(crl)
};

// Per-work-group partial sums of the moment reduction (A.4 + A.6 in one pass).
// ci = i - R/2, cj = j - C/2 are centred integer pixel coordinates.
struct Partial {
    long long n, sci, scj;          // exact integer sums over valid pixels
    double sgx, sgy;                // sum gx, sum gy
    double sigx, sigy, sjgx, sjgy;  // sum ci*gx, ci*gy, cj*gx, cj*gy
    double pad;
};

struct DevState {
    // --- window (host-written at set_cloud) ---
    int32_t scale, R, C, wsx, wsy, x_sh, y_sh, tbits;
    double x_shift, y_shift;
    long long tmin;
    // --- loop control (optimizer_rolling.h:36,59-63) ---
    float x_div, y_div, rot_div, div_div;
    float old_dx, old_dy, old_rot, old_div;
    int32_t it, done, max_iter, hard_cap, rc, trace_cap, nblocks, pad0;
    // --- model + warp parameters ---
    bf_model model;
    WarpParams wp;
};

}  // namespace bf
