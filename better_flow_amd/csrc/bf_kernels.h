// bf_kernels.h -- launch interface between bf_accel.cpp (C-ABI, host logic) and
// bf_kernels.hip (gfx950 kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bf_device.h"

namespace bf {

// min / max / sum statistics of one staged slice (k_prepare).
struct SliceStats {
    int32_t xmin, xmax, ymin, ymax, tmin, tmax;
    long long tsum;
};

struct WarpScatterArgs {
    const uint32_t* xy;
    const int32_t* t;
    float2* p;
    const uint8_t* noise;          // may be NULL
    double2* nxny;                 // written only by the final warp
    unsigned long long* plane;     // point-scatter accumulator (current buffer)
    uint32_t* cplane;              // SPLIT mode count plane (current buffer)
    const DevState* st;
    long long n;
    int check_done;                // 1 inside the fused loop: return at once if st->done
    bool packed;
};

struct StencilArgs {
    const DevState* st;
    int check_done;
    int R, C, scale, tbits;
    long long tmin;
    const unsigned long long* plane;   // SRC 0 / 1
    const uint32_t* cplane;            // SRC 1
    const float* time_in;              // SRC 2
    float* time_out;                   // optional
    uint32_t* count_out;               // optional
    float* gx_out;                     // optional (gy_out must be set with it)
    float* gy_out;
    Partial* partials;                 // optional
    unsigned long long* zero_plane;    // optional: the OTHER plane buffer, zeroed here
    uint32_t* zero_cplane;
};

void launch_set_state(DevState* st, const DevState& v, hipStream_t s);
void launch_init_stats(SliceStats* st, hipStream_t s);
void launch_warp_scatter(const WarpScatterArgs& a, bool warp, bool scatter, bool write_n,
                         hipStream_t s);
void launch_prepare(const int32_t* fr_x, const int32_t* fr_y, const int32_t* t_in, uint32_t* xy,
                    int32_t* t_out, float2* p, long long n, long long n_pad, SliceStats* stats,
                    hipStream_t s);
void stencil_grid(int R, int C, int* gx, int* gy);
void launch_stencil(const StencilArgs& a, int src, hipStream_t s);
void launch_update(DevState* st, const Partial* partials, int nblocks, bf_trace_rec* trace, int mode,
                   hipStream_t s);
void launch_compute_uv(const double2* nxny, double2* uv, long long n, hipStream_t s);
void launch_expand_pr(const uint32_t* xy, const float2* p, double2* pr, long long n, hipStream_t s);
void launch_copy(const void* src, void* dst, long long bytes, hipStream_t s);

}  // namespace bf
