// bf_device_fns.h -- device-side helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>

#include "bf_device.h"
#include "bf_kernels.h"
#include <math.h>
#include <math.h>

namespace bf {

// `int x = <double>` on x86-64 is cvttsd2si: NaN / out-of-range -> INT_MIN (then rejected by
// the bounds test of accel_lib.h:157).  v_cvt_i32_f64 would saturate / give 0 instead.
__device__ __forceinline__ int trunc_x86(double v) {
    return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN;
}

// The same for a scatter position, whose only consumer is the window test of accel_lib.h:157: anything out of int
// range is rejected there whichever end it saturates to (the window is far inside the int range), so the hardware's
// saturating conversion serves, and only NaN -- which it would turn into 0, a pixel INSIDE the window at scale 1 -- needs
// the x86 answer.  Two compares fewer per coordinate than trunc_x86.
__device__ __forceinline__ int trunc_scatter(double v) { return (v == v) ? __double2int_rz(v) : INT_MIN; }

// f32 form of the reference's `p > 0.000001` (float against a double literal): 1e-6f is the
// largest float below 1e-6, so (double)p > 1e-6  <=>  p > 1e-6f.
__device__ __forceinline__ bool valid_px(float p) { return p > 1e-6f; }

// Exact IEEE division by a constant in three FMA-class operations (Markstein):
//   q0 = x * R,  r = fma(-q0, d, x),  q = fma(r, R, q0),  R = RN(1 / d).
// The dividends here are always a float (converted to double for the f64 form), so the
// identity with x / d was PROVEN EXHAUSTIVELY over all 2^32 floats by tests/exhaustive_div.c:
// exactly three inputs differ -- -0 gives +0 instead of -0, +-inf give NaN instead of +-inf -- and neither can change
// a result here: the quotients are only ever SUBTRACTED from a sensor coordinate (x - (+0) == x - (-0) for every x
// of either sign) or multiplied into a product whose zero sign is lost the same way, and an infinite product only
// arises from a diverged model, whose events the x86 truncation rule rejects as NaN or as infinity alike.  (A fix-up
// select for those three inputs cost 5 of the 8 instructions of each of the six divisions per event.)  Replaces
// ~15-instruction division expansions: the warp+scatter kernel was f64-ALU-bound on them.
__device__ __forceinline__ double div_10000(double x) {   // x == (double)(some float)
    constexpr double R = 1.0 / 10000.0;
    const double q0 = x * R;
    const double r = fma(-q0, 10000.0, x);
    return fma(r, R, q0);
}
__device__ __forceinline__ float div_127(float x) {
    constexpr float R = 1.0f / 127.0f;
    const float q0 = x * R;
    const float r = fmaf(-q0, 127.0f, x);
    return fmaf(r, R, q0);
}

// (double)ts / 1e9 for an integer nanosecond sum ts (accel_lib.h:162): same sequence; checked on
// 2^32 pseudo-random integer dividends of all magnitudes plus every integer below 2^26
// (tests/exhaustive_div.c).  The per-pixel IEEE division expansion was ~1 us of the stencil.
__device__ __forceinline__ double div_1e9(double x) {
    constexpr double R = 1.0 / 1000000000.0;
    const double q0 = x * R;
    const double r = fma(-q0, 1000000000.0, x);
    return fma(r, R, q0);   // (x is a finite integer; x == 0 gives 0 either way)
}

// Previous / new projected position from the stored f32 product (event.h:167-168):
//   pr = float(fr) - (kx * float(t)) / 10000.0      (f32 product, f64 divide and subtract)
// (float(fr) is exact for a 16-bit coordinate, so the u32 -> f64 conversion gives the same double in one step)
__device__ __forceinline__ double pr_from_p(uint32_t fr, float prod) {
    return (double)fr - div_10000((double)prod);
}

// Event::project_4param_reinit (event.h:99-110) + apply_project (event.h:164-168) of one event -- the ONE
// place this arithmetic lives.  In: the event's previous projected position (pr_x, pr_y) and slice-local time.
// Out: the direction vector (nx, ny) and the two f32 products q = (kx * float(t), ky * float(t)) from which
// pr is re-derived with pr_from_p.  No operation may be contracted (-ffp-contract=off).
__device__ __forceinline__ void warp_products(const WarpParams& wp, double pr_x, double pr_y, int32_t t, float2& q,
                                              double& nx, double& ny) {
    // event.h:100-108
    const double rx = pr_x - wp.cx, ry = pr_y - wp.cy;
    const double qx = wp.c * rx - wp.s * ry;
    const double qy = wp.s * rx + wp.c * ry;
    nx = ((-qx) * wp.div + (qx - rx)) + wp.dnx;
    ny = ((-qy) * wp.div + (qy - ry)) + wp.dny;
    // event.h:164-165: float kx = float(nx) / nz.  The double division rounded to float equals the
    // correctly rounded f32 division (double has >= 2 * 24 + 2 digits).
    const float kx = div_127((float)nx);
    const float ky = div_127((float)ny);
    const float ft = (float)t;   // round-to-nearest int -> f32, as float(sll t)
    q.x = kx * ft;
    q.y = ky * ft;
}

// A load the compiler must issue on the SCALAR unit: through the constant address space.  For data that no work-group
// of the running kernel writes (kernel arguments passed inside a struct lose their __restrict__, and the compiler
// then falls back to vector loads for uniform addresses: they complete in order with every other vector load of the
// wave, so a 4-byte state word ended up waiting for ~1 us of earlier requests).
template <class T>
__device__ __forceinline__ T sload(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    typedef const __attribute__((address_space(4))) uint32_t* cptr;
    const cptr src = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(p));
    T out;
    uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) dst[i] = src[i];   // (merged into s_load_dwordx2 .. x16)
    return out;
}

__device__ __forceinline__ int wave_min(int v) {
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Row -> bin row of the tile-binned loops: exact division by the (not necessarily power-of-two) tile height.
__device__ __forceinline__ int row_bin(int row, const BinGrid& g) { return (int)__umulhi((uint32_t)row, g.mul_r); }

constexpr int kTicketGroups = 32;   // arrival counters of the fused reduction (64 B apart)

// Debug timeline: 100 MHz wall-clock stamps of kernel phases for work-groups 0 and `mid`, 16
// slots per (launch, group).  Compiled in only with -DBF_TIMELINE (`make tl`, BF_TIMELINE=<file> at
// run time): the stamps cost registers, so the production library does not contain them.
constexpr int kTlLaunches = 64;
__device__ __forceinline__ void tl_stamp(unsigned long long* tl, int launch, int slot) {
#ifdef BF_TIMELINE
    if (!tl || launch >= kTlLaunches || threadIdx.x != 0) return;
    const int nb = gridDim.x * gridDim.y, me = blockIdx.y * gridDim.x + blockIdx.x;
    int g = -1;
    if (me == 0) g = 0;
    else if (me == nb / 2) g = 1;
    if (g < 0) return;
    tl[((size_t)launch * 2 + g) * 16 + slot] = wall_clock64();
#else
    (void)tl; (void)launch; (void)slot;
#endif
}

struct Sums {
    long long n, sci, scj;
    double sgx, sgy, sigx, sigy, sjgx, sjgy;
};

__device__ __forceinline__ void sums_zero(Sums& s) {
    s.n = s.sci = s.scj = 0;
    s.sgx = s.sgy = s.sigx = s.sigy = s.sjgx = s.sjgy = 0.0;
}
__device__ __forceinline__ void sums_add(Sums& a, const Sums& b) {
    a.n += b.n; a.sci += b.sci; a.scj += b.scj;
    a.sgx += b.sgx; a.sgy += b.sgy;
    a.sigx += b.sigx; a.sigy += b.sigy; a.sjgx += b.sjgx; a.sjgy += b.sjgy;
}
__device__ __forceinline__ void sums_wave_reduce(Sums& s) {
    s.n = wave_sum(s.n); s.sci = wave_sum(s.sci); s.scj = wave_sum(s.scj);
    s.sgx = wave_sum(s.sgx); s.sgy = wave_sum(s.sgy);
    s.sigx = wave_sum(s.sigx); s.sigy = wave_sum(s.sigy);
    s.sjgx = wave_sum(s.sjgx); s.sjgy = wave_sum(s.sjgy);
}

// Wave64 sum of a 64-bit pattern on the DPP network (no LDS traffic): inclusive scan by row_shr 1 / 2 / 4 / 8
// inside each row of 16 lanes, then row_bcast:15 (lane 15 of a row into the next row) and row_bcast:31 (lane 31 into
// rows 2 and 3) -- lane 63 ends up with the total, added in a fixed order; no other lane is meaningful.
// Every step runs with all rows enabled and bound_ctrl:0 (a lane without a source reads 0), so no register has to be
// zeroed for the lanes a step does not feed.  The row_bcast steps then also add into rows that a masked step would
// skip (row 2 picks up row 1's total, rows 0 / 1 pick up nothing); those lanes are never read again: lane 63 adds
// lane 47's value as it was BEFORE the step (DPP reads the operand of the other lane, not its result), then lane 31's,
// which is (row 1 + row 0) -- the same four row totals in the same order as with the masks.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)v, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(v >> 32), CTRL, ROW_MASK, 0xf, true);
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo;
}
__device__ __forceinline__ long long wave_total_dpp(long long v) {
    v += (long long)dpp_u64<0x111, 0xf>((unsigned long long)v);   // row_shr:1
    v += (long long)dpp_u64<0x112, 0xf>((unsigned long long)v);   // row_shr:2
    v += (long long)dpp_u64<0x114, 0xf>((unsigned long long)v);   // row_shr:4
    v += (long long)dpp_u64<0x118, 0xf>((unsigned long long)v);   // row_shr:8
    v += (long long)dpp_u64<0x142, 0xf>((unsigned long long)v);   // row_bcast:15: lane 15 of a row -> the next row
    v += (long long)dpp_u64<0x143, 0xf>((unsigned long long)v);   // row_bcast:31: lane 31 -> rows 2, 3
    return v;   // lane 63: sum of all 64 lanes
}
__device__ __forceinline__ double wave_total_dpp(double v) {
#define BF_DPP_STEP(CTRL, MASK) \
    v += __longlong_as_double((long long)dpp_u64<CTRL, MASK>((unsigned long long)__double_as_longlong(v)))
    BF_DPP_STEP(0x111, 0xf);
    BF_DPP_STEP(0x112, 0xf);
    BF_DPP_STEP(0x114, 0xf);
    BF_DPP_STEP(0x118, 0xf);
    BF_DPP_STEP(0x142, 0xf);
    BF_DPP_STEP(0x143, 0xf);
#undef BF_DPP_STEP
    return v;
}

// Work-group reduction of a Sums: DPP wave totals (above), lane 63 of every wave parks its nine
// values in LDS, and after one barrier every thread adds the per-wave results in wave order.  (The
// wave64 __shfl_down tree costs 107 ds_bpermute_b32 per wave, ~1.9 us with every wave at it; an
// LDS-transposed version ~1.8 us; this one stays on the VALU.)  Every thread gets the total.
// `tid` is the thread's index inside its THREADS-wide group (groups are wave aligned).
// Wave64 total of a 32-bit word, same network; the compiler folds each step into one v_add_u32 with a DPP operand.
__device__ __forceinline__ unsigned int wave_total_dpp(unsigned int v) {
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xf, 0xf, true);
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xf, 0xf, true);
    return v;
}

// PACK_INTS: the group contributes at most 1024 pixels, all inside a tile whose first centred row / column is
// (base_i, base_j) and which is at most 64 x 64 (a 16 x 64 stencil tile).  The three integer sums then travel as two
// 32-bit words of tile-local coordinates -- n << 16 | sum(cj - base_j), and sum(ci - base_i), each field with room
// for 1024 terms -- through the wave reduction (one instruction per step and word, against five for a 64-bit
// integer) and as one 64-bit word through the LDS hop.  Exact: sci = sum + n base_i.
constexpr int kSumFields = 9;
// Two halves: every wave publishes its totals (wave reduction, one LDS hop, work-group barrier); then whoever needs the
// work-group's total adds the partials up -- in the stencil kernels that is the first wave only, the others are done.
// The six f64 wave totals through LDS instead of six 6-step DPP scans (18 instructions each, every lane adding for the
// one lane that is read): one DPP step forms the pair sums (lane 2k+1: v[2k] + v[2k+1]), the odd lanes park them in `scr`
// (this wave's 6 x 32 doubles), lane 8 f + c then adds four consecutive pair sums of field f pairwise -- lanes 8c .. 8c+7
// of the wave -- and three DPP steps among the eight lanes of a field combine the chunks.  The SAME binary tree over the
// 64 lanes as wave_total_dpp (adjacent pairs, then pairs of pairs, ...), so the same bits; ~45 vector instructions
// instead of ~120.  Lane 8 f + 7 ends up with the wave's total of field f.
__device__ __forceinline__ double wave_totals6_lds(const double (&v)[6], double* scr /* this wave's 192 doubles */, const int lane) {
    double p[6];
#pragma unroll
    for (int f = 0; f < 6; ++f)
        p[f] = v[f] + __longlong_as_double((long long)dpp_u64<0x111, 0xf>((unsigned long long)__double_as_longlong(v[f])));   // row_shr:1
    if (lane & 1) {
#pragma unroll
        for (int f = 0; f < 6; ++f) scr[f * 32 + (lane >> 1)] = p[f];
    }
    __builtin_amdgcn_wave_barrier();   // (LDS operations of one wave complete in order)
    const int fl = lane < 48 ? lane : 0;   // lanes 48 .. 63 read along, their result is not used
    const double* q = scr + (fl >> 3) * 32 + (fl & 7) * 4;
    double t = (q[0] + q[1]) + (q[2] + q[3]);
    t += __longlong_as_double((long long)dpp_u64<0x111, 0xf>((unsigned long long)__double_as_longlong(t)));   // chunks c-1, c
    t += __longlong_as_double((long long)dpp_u64<0x112, 0xf>((unsigned long long)__double_as_longlong(t)));   // pairs
    t += __longlong_as_double((long long)dpp_u64<0x114, 0xf>((unsigned long long)__double_as_longlong(t)));   // quads: lane 8 f + 7
    return t;
}

template <int THREADS, bool PACK_INTS = false>
__device__ __forceinline__ void block_reduce_publish(const Sums& sm, unsigned long long* s_part /* 9 * THREADS / 64 */,
                                                     const int tid, const int base_i = 0, const int base_j = 0,
                                                     double* s_scr = nullptr /* THREADS / 64 x 192 doubles, or none */) {
    constexpr int W = THREADS / 64;
    Sums r;
    if constexpr (PACK_INTS) {
        const int tn = (int)sm.n;
        const unsigned int wa = wave_total_dpp(((unsigned int)tn << 16) | (unsigned int)((int)sm.scj - tn * base_j));
        const unsigned int wb = wave_total_dpp((unsigned int)((int)sm.sci - tn * base_i));
        if (s_scr) {
            const int lane = tid & 63, w = tid >> 6;
            const double v6[6] = {sm.sgx, sm.sgy, sm.sigx, sm.sigy, sm.sjgx, sm.sjgy};
            const double t = wave_totals6_lds(v6, s_scr + w * 192, lane);
            if (lane < 48 && (lane & 7) == 7) s_part[(3 + (lane >> 3)) * W + w] = (unsigned long long)__double_as_longlong(t);
            if (lane == 63) s_part[0 * W + w] = ((unsigned long long)wa << 32) | (unsigned long long)wb;
            __syncthreads();
            return;
        }
        r.sgx = wave_total_dpp(sm.sgx); r.sgy = wave_total_dpp(sm.sgy);
        r.sigx = wave_total_dpp(sm.sigx); r.sigy = wave_total_dpp(sm.sigy);
        r.sjgx = wave_total_dpp(sm.sjgx); r.sjgy = wave_total_dpp(sm.sjgy);
        if ((tid & 63) == 63) {
            const int w = tid >> 6;
            s_part[0 * W + w] = ((unsigned long long)wa << 32) | (unsigned long long)wb;
            s_part[3 * W + w] = (unsigned long long)__double_as_longlong(r.sgx);
            s_part[4 * W + w] = (unsigned long long)__double_as_longlong(r.sgy);
            s_part[5 * W + w] = (unsigned long long)__double_as_longlong(r.sigx);
            s_part[6 * W + w] = (unsigned long long)__double_as_longlong(r.sigy);
            s_part[7 * W + w] = (unsigned long long)__double_as_longlong(r.sjgx);
            s_part[8 * W + w] = (unsigned long long)__double_as_longlong(r.sjgy);
        }
        __syncthreads();
        return;
    }
    r.n = wave_total_dpp(sm.n); r.sci = wave_total_dpp(sm.sci); r.scj = wave_total_dpp(sm.scj);
    r.sgx = wave_total_dpp(sm.sgx); r.sgy = wave_total_dpp(sm.sgy);
    r.sigx = wave_total_dpp(sm.sigx); r.sigy = wave_total_dpp(sm.sigy);
    r.sjgx = wave_total_dpp(sm.sjgx); r.sjgy = wave_total_dpp(sm.sjgy);
    if ((tid & 63) == 63) {
        const int w = tid >> 6;
        s_part[0 * W + w] = (unsigned long long)r.n;
        s_part[1 * W + w] = (unsigned long long)r.sci;
        s_part[2 * W + w] = (unsigned long long)r.scj;
        s_part[3 * W + w] = (unsigned long long)__double_as_longlong(r.sgx);
        s_part[4 * W + w] = (unsigned long long)__double_as_longlong(r.sgy);
        s_part[5 * W + w] = (unsigned long long)__double_as_longlong(r.sigx);
        s_part[6 * W + w] = (unsigned long long)__double_as_longlong(r.sigy);
        s_part[7 * W + w] = (unsigned long long)__double_as_longlong(r.sjgx);
        s_part[8 * W + w] = (unsigned long long)__double_as_longlong(r.sjgy);
    }
    __syncthreads();
}
template <int THREADS, bool PACK_INTS = false>
__device__ __forceinline__ Sums block_reduce_total(const unsigned long long* s_part, const int base_i = 0, const int base_j = 0) {
    constexpr int W = THREADS / 64;
    Sums t;
    if constexpr (PACK_INTS) {
        unsigned long long pt = 0;
        double d_[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int w = 0; w < W; ++w) {
            pt += s_part[0 * W + w];
#pragma unroll
            for (int f = 3; f < kSumFields; ++f) d_[f - 3] += __longlong_as_double((long long)s_part[f * W + w]);
        }
        const unsigned int ta = (unsigned int)(pt >> 32), tb = (unsigned int)pt;
        t.n = (long long)(ta >> 16);
        t.sci = (long long)tb + t.n * (long long)base_i;
        t.scj = (long long)(ta & 0xffffu) + t.n * (long long)base_j;
        t.sgx = d_[0]; t.sgy = d_[1]; t.sigx = d_[2]; t.sigy = d_[3]; t.sjgx = d_[4]; t.sjgy = d_[5];
        return t;
    }
    long long in_[3] = {0, 0, 0};
    double d_[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int w = 0; w < W; ++w) {
#pragma unroll
        for (int f = 0; f < 3; ++f) in_[f] += (long long)s_part[f * W + w];
#pragma unroll
        for (int f = 3; f < kSumFields; ++f) d_[f - 3] += __longlong_as_double((long long)s_part[f * W + w]);
    }
    t.n = in_[0]; t.sci = in_[1]; t.scj = in_[2];
    t.sgx = d_[0]; t.sgy = d_[1]; t.sigx = d_[2]; t.sigy = d_[3]; t.sjgx = d_[4]; t.sjgy = d_[5];
    return t;
}
template <int THREADS, bool PACK_INTS = false>
__device__ __forceinline__ Sums block_reduce_sums(const Sums& sm, unsigned long long* s_part /* 9 * THREADS / 64 */,
                                                  const int tid, const int base_i = 0, const int base_j = 0) {
    block_reduce_publish<THREADS, PACK_INTS>(sm, s_part, tid, base_i, base_j);
    return block_reduce_total<THREADS, PACK_INTS>(s_part, base_i, base_j);
}

// Mean time of one pixel from its exact integer sums (accel_lib.h:162,172): the f32 sum of
// seconds is the integer-ns sum rounded once, then the f32 divide by the count.
// `rcp` (optional, LDS): rcp[i] = RN64(1.0 / i) for i < kRcpTab.  The f32 division by the pixel's event count -- ~11 vector
// instructions of scale / reciprocal / refinement / fix-up, once per time pixel -- then becomes a conversion, a table read
// and one f64 product, and gives the SAME bits: with q = a / b the real quotient of an f32 a by an integer 1 <= b < 2^8,
// t = RN64(a * RN64(1 / b)) = q (1 + d), |d| <= 2^-52; RN32(t) differs from RN32(q) only if a midpoint m between two
// adjacent f32 values lies between q and t.  a and b m are both multiples of 2^(e - 24) (e the binade of q), so q - m =
// k 2^(e - 24) / b for an integer k, and k = 0 would mean a = b m with m an odd multiple of 2^(e - 24) of 25 significant
// bits: for odd b >= 3 the product b m is odd and wider than a's 24 bits, for even b halve a and b first, for b a power of
// two the division is exact in f32 and in f64.  So |q - m| >= 2^(e - 24) / 255 > 2^(e - 32), while |t - q| < 2^(e - 51).
// (tests/exhaustive_div.c (d) checks the identity for EVERY finite f32 a and every b in [1, 255]: no mismatch with a normal
// quotient; a sum of integer nanoseconds is 0 or >= 1e-9, far above the subnormal range.)
// RN64(1 / i), i < kRcpTab, for time_from_sums' division by the pixel's event count: a compile-time table in constant memory.
// (Its address comes with the code, not through the argument block: a first version passed a pointer in StencilArgs and
// loaded the table at the kernel's entry -- the load's address then waited for the argument block's fetch ahead of EVERY
// slab load, which the preloaded arguments exist to avoid: alone the kernel was still 9 % faster, with four contexts on the
// GPU the bench fell from 201 to 135 Mevents/s.)
struct RcpTab {
    double v[kRcpTab];
    constexpr RcpTab() : v{} {
        for (int i = 1; i < kRcpTab; ++i) v[i] = 1.0 / (double)i;
    }
};
static __constant__ RcpTab c_rcp = RcpTab();

__device__ __forceinline__ float time_from_sums(uint32_t cnt, long long tsum_biased, long long tmin, const double* rcp = nullptr) {
    if (cnt == 0) return 0.f;
    // (tmin is a slice-local time that fits 32 bits: one v_mad_i64_i32 instead of a 64 x 64-bit multiply)
    const long long ts = tsum_biased + (long long)(int)cnt * (long long)(int)tmin;
    const float sum_s = (float)div_1e9((double)ts);
    // (the empty-pixel and outside-the-image branches stay: without them -- rcp[0] = 0 gives the same +0.f -- the dense stencil
    // kernel was 2-3 % slower, round 5: many pixels of a time tile ARE empty early in a run, and whole waves skip)
    if (rcp && cnt < (uint32_t)kRcpTab) return (float)((double)sum_s * rcp[cnt]);
    return sum_s / (float)cnt;
}

// sin / cos of the (small) rotation angle of the warp: Taylor polynomials in x^2 for |x| <= 0.25 (truncation below
// 2^-64; a 30 ms slice rotates by ~1e-3), the library routine otherwise.  ~10 dependent operations instead of ~100 on
// the one lane every iteration waits for.  The reference calls std::cos / std::sin (event.h:102-103; the stand-alone operator
// bf_project_4param_reinit does so too, on the host); how far these are from the host's libm is MEASURED by
// tests/test_gpu_parity.py::test_device_sincos_against_libm (bf_eval_sincos) and stated in DESIGN.md, "Oracle".
// (the library routine out of line: inlined, its two dozen f64 constants are materialised -- and, in a kernel that loops,
// hoisted into registers for the whole loop -- on a path no real slice takes)
__device__ __attribute__((noinline)) static void sincos_large(double x, double* sn, double* cs) { sincos(x, sn, cs); }
__device__ __forceinline__ void sincos_small(double x, double* sn, double* cs) {
    if (!(fabs(x) <= 0.25)) {
        sincos_large(x, sn, cs);
        return;
    }
    const double z = x * x;
    double ps = -1.0 / 1307674368000.0;                    // sin(x) / x
    ps = fma(ps, z, 1.0 / 6227020800.0);
    ps = fma(ps, z, -1.0 / 39916800.0);
    ps = fma(ps, z, 1.0 / 362880.0);
    ps = fma(ps, z, -1.0 / 5040.0);
    ps = fma(ps, z, 1.0 / 120.0);
    ps = fma(ps, z, -1.0 / 6.0);
    *sn = fma(x * z, ps, x);
    double pc = 1.0 / 20922789888000.0;                    // (cos(x) - 1 + z / 2) / z^2
    pc = fma(pc, z, -1.0 / 87178291200.0);
    pc = fma(pc, z, 1.0 / 479001600.0);
    pc = fma(pc, z, -1.0 / 3628800.0);
    pc = fma(pc, z, 1.0 / 40320.0);
    pc = fma(pc, z, -1.0 / 720.0);
    pc = fma(pc, z, 1.0 / 24.0);
    // cos x = 1 - w, w = z (1/2 - z pc) <= 0.031: ONE rounding at the size of the result (w carries ~2^-52 w of its own, a
    // twentieth of the result's ulp at most).  (1 - z/2) + z^2 pc, the form this replaces, rounded twice at that size: up to
    // 1.008 ulp -- measured by tests/test_gpu_parity.py, round 5.)
    *cs = 1.0 - z * fma(-z, pc, 0.5);
}

// The same polynomials with their fourteen coefficients read from a table (LDS, filled by sincos_table_fill): a kernel that
// LOOPS over iterations would otherwise keep all of them in registers for the whole loop (loop-invariant constants are
// hoisted), 28 VGPRs it needs for its events.  Same operations in the same order: the same bits.
__device__ __forceinline__ void sincos_table_fill(double* tab) {
    tab[0] = -1.0 / 1307674368000.0; tab[1] = 1.0 / 6227020800.0; tab[2] = -1.0 / 39916800.0; tab[3] = 1.0 / 362880.0;
    tab[4] = -1.0 / 5040.0; tab[5] = 1.0 / 120.0; tab[6] = -1.0 / 6.0;
    tab[7] = 1.0 / 20922789888000.0; tab[8] = -1.0 / 87178291200.0; tab[9] = 1.0 / 479001600.0; tab[10] = -1.0 / 3628800.0;
    tab[11] = 1.0 / 40320.0; tab[12] = -1.0 / 720.0; tab[13] = 1.0 / 24.0;
}
__device__ __forceinline__ void sincos_small_tab(double x, const double* tab, double* sn, double* cs) {
    if (!(fabs(x) <= 0.25)) {
        sincos_large(x, sn, cs);
        return;
    }
    const double z = x * x;
    double c_[14];   // (all fourteen reads in flight together)
#pragma unroll
    for (int i = 0; i < 14; ++i) c_[i] = tab[i];
    double ps = c_[0];
    ps = fma(ps, z, c_[1]);
    ps = fma(ps, z, c_[2]);
    ps = fma(ps, z, c_[3]);
    ps = fma(ps, z, c_[4]);
    ps = fma(ps, z, c_[5]);
    ps = fma(ps, z, c_[6]);
    *sn = fma(x * z, ps, x);
    double pc = c_[7];
    pc = fma(pc, z, c_[8]);
    pc = fma(pc, z, c_[9]);
    pc = fma(pc, z, c_[10]);
    pc = fma(pc, z, c_[11]);
    pc = fma(pc, z, c_[12]);
    pc = fma(pc, z, c_[13]);
    *cs = 1.0 - z * fma(-z, pc, 0.5);   // (see sincos_small)
}

// A uniform double out of one lane of a wave (two v_readlane_b32).
__device__ __forceinline__ double lane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, lane);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}
__device__ __forceinline__ long long lane_i64(unsigned long long v, int lane) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, lane);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), lane);
    return (long long)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}

// ObjectModel::update from the reduced sums (object_model.cpp:4-39,103-126) and, in mode 1,
// ObjectModel::update_accumulators (object_model.h:48-53), the glue of iteration_step
// (optimizer_rolling.h:328-346) and the loop control of run() (optimizer_rolling.h:61-101), on a copy of the state in
// LDS.  mode 0: model only (AccelLib::fast_model).
//
// This is the one serial stretch every iteration of the loop waits for, so it is written for latency.  It is executed
// by ONE WAVE (all 64 lanes call it): lane l brings `word`, the total of accumulator field l % 16 (MomentAcc layout).
// A single wave issues one vector instruction every ~5 cycles whatever the number of active lanes, so the work is
// spread over lanes wherever the same operation applies to several values: the integer -> f64 conversions and the
// fixed-point joins run once for all fields, and ALL the divisions the update needs -- the ten moment quotients by the
// pixel count and the five reciprocals of the dividers / the scale -- are ONE lane-parallel IEEE division (~12
// instructions instead of ~180).  The quotients are then broadcast through scalar registers and the rest (a few dozen
// operations on uniform values, sin / cos by sincos_small) is done redundantly by every lane; lane 0 stores.
// ~170 instructions, against ~600 for the same update written for one thread.
//   lane : numerator / denominator
//     1  : (sum ci + n R/2) / n = cx      2 : (sum cj + n C/2) / n = cy       (exact integer numerators)
//     4  : sum ci / n = cxc               6 : sum cj / n = cyc                 (centred)
//     3, 5, 7, 9, 11, 13 : sgx, sgy, sigx, sigy, sjgx, sjgy / n
//     8, 10, 12, 14, 15  : 1 / x_div, 1 / y_div, 1 / rot_div, 1 / div_div, 1 / scale
// rot and div are formed from the quotients (object_model.cpp:26-38 with the division distributed over the terms;
// the dividers enter as reciprocals: <= 1 ulp from the divided form, far inside the moments' 1e-9 bar).
__device__ __forceinline__ void model_update_wave(DevState* st, unsigned long long word, int lane, int mode,
                                                  const double* sc_tab = nullptr) {
    const int f = lane & 15;
    const long long n_i = lane_i64(word, 0), sci = lane_i64(word, 1), scj = lane_i64(word, 2);
    const int R = st->hot.R, C = st->hot.C;
    const double dn = (double)n_i;   // cnt == 0 -> 0/0 = NaN, as in the reference (assert off)
    // fixed-point join: this lane's high word with the next lane's low word (valid in the odd lanes 3 .. 13)
    const double dhi = (double)(long long)word, dlo = (double)word;
    const double lo_next = __shfl_down(dlo, 1, 64);
    double num = (dhi + lo_next * (1.0 / 4503599627370496.0)) * (1.0 / 4096.0);
    double den = dn;
    num = (f == 1) ? (double)(sci + n_i * (long long)(R / 2)) : num;
    num = (f == 2) ? (double)(scj + n_i * (long long)(C / 2)) : num;
    num = (f == 4) ? (double)sci : num;
    num = (f == 6) ? (double)scj : num;
    if (mode != 0) {
        const double xd_ = (double)st->x_div, yd_ = (double)st->y_div, rd_ = (double)st->rot_div, dd_ = (double)st->div_div;
        const double sc_ = (double)st->hot.scale;
        den = (f == 8) ? xd_ : den;
        den = (f == 10) ? yd_ : den;
        den = (f == 12) ? rd_ : den;
        den = (f == 14) ? dd_ : den;
        den = (f == 15) ? sc_ : den;
        num = (f == 8 || f == 10 || f == 12 || f == 14 || f == 15) ? 1.0 : num;
    }
    const double quo = num / den;   // the one division
    bf_model m = st->model;
    m.cx = lane_f64(quo, 1);
    m.cy = lane_f64(quo, 2);
    const double cxc = lane_f64(quo, 4), cyc = lane_f64(quo, 6);
    m.dx = lane_f64(quo, 3);
    m.dy = lane_f64(quo, 5);
    const double qsigx = lane_f64(quo, 7), qsigy = lane_f64(quo, 9), qsjgx = lane_f64(quo, 11), qsjgy = lane_f64(quo, 13);
    // object_model.cpp:26-38 with r = (ci - cxc, cj - cyc)
    m.rot = (qsigy - cxc * m.dy) - (qsjgx - cyc * m.dx);
    m.div = (qsigx - cxc * m.dx) + (qsjgy - cyc * m.dy);
    m.cnt = (uint32_t)n_i;
    if (mode == 0) {
        if (lane == 0) st->model = m;
        return;
    }
    const double inv_xd = lane_f64(quo, 8), inv_yd = lane_f64(quo, 10), inv_rd = lane_f64(quo, 12), inv_dd = lane_f64(quo, 14);
    const double inv_scale = lane_f64(quo, 15);
    // object_model.h:48-53 via optimizer_rolling.h:328.  The four quotients are kept: the convergence test
    // below needs the same ratios against dividers that are either unchanged or exactly doubled.
    const double q_rot = m.rot * inv_rd, q_div = m.div * inv_dd;
    const double q_dx = m.dx * inv_xd, q_dy = m.dy * inv_yd;
    m.total_rot += q_rot;
    m.total_div += q_div;
    m.total_dx += q_dx;
    m.total_dy += q_dy;
    // optimizer_rolling.h:330-331,340-346
    const double cxs = (m.cx - st->x_shift) * inv_scale;
    const double cys = (m.cy - st->y_shift) * inv_scale;
    WarpParams wp;
    wp.dnx = -m.total_dx; wp.dny = -m.total_dy;
    wp.cx = cxs; wp.cy = cys;
    wp.div = m.total_div;
    if (sc_tab) sincos_small_tab(-m.total_rot, sc_tab, &wp.s, &wp.c);
    else sincos_small(-m.total_rot, &wp.s, &wp.c);
    m.cx = cxs;
    m.cy = cys;

    // ---- run(), optimizer_rolling.h:73-101, as a state machine after each step ----
    const int it = st->hot.it + 1;
    float xd = st->x_div, yd = st->y_div, rd = st->rot_div, dd = st->div_div;
    // m.dx / xd of the convergence test (:81-84) == q_dx when the divider is unchanged and q_dx / 2 when it was
    // just doubled (a power of two)
    double hx = 1.0, hy = 1.0, hr = 1.0, hd = 1.0;
    int done = 0, rc = 0;
    bool new_dividers = false;
    if (it > 1) {
        if (st->max_iter > 0 && it > st->max_iter) {   // :94-96 (before the sign flips)
            done = 1;
        } else {                                       // :98-101
            if (m.dx * (double)st->old_dx < 0) { xd *= 2; hx = 0.5; }
            if (m.dy * (double)st->old_dy < 0) { yd *= 2; hy = 0.5; }
            if (m.rot * (double)st->old_rot < 0) { rd *= 2; hr = 0.5; }
            if (m.div * (double)st->old_div < 0) { dd *= 2; hd = 0.5; }
            new_dividers = true;
            if (st->hard_cap > 0 && it >= st->hard_cap) { done = 1; rc = BF_ERR_NOCONV; }
        }
    }
    bool keep_old = false;
    if (!done) {
        if (!(xd < 32 * 10 || yd < 32 * 10 || rd < 32 * 1000 || dd < 32 * 1000)) {   // :76-79
            done = 1;
        } else if (fabs(q_dx * hx) < 1e-5 && fabs(q_dy * hy) < 1e-5 &&
                   fabs(q_rot * hr) < 1e-4 && fabs(q_div * hd) < 1e-1) {   // :81-84
            done = 1;
        } else {                                                                         // :86-89
            keep_old = true;
        }
    }
    const int run_tag = st->run_tag;
    __builtin_amdgcn_wave_barrier();   // (every lane has read what it needs from the state)
    if (lane == 0) {
        st->hot.wp = wp;
        st->model = m;
        st->hot.it = it;
        if (new_dividers) { st->x_div = xd; st->y_div = yd; st->rot_div = rd; st->div_div = dd; }
        if (keep_old) {
            st->old_dx = (float)m.dx; st->old_dy = (float)m.dy;
            st->old_rot = (float)m.rot; st->old_div = (float)m.div;
        }
        if (done) {
            st->rc = rc;
            st->hot.done = run_tag ? run_tag : 1;
        }
    }
}

// Overflow counter of one iteration (tile-binned loop).  Thousands of waves may lose an event in the same launch, and
// atomics on one cache line serialise at ~11 ns each (scripts/micro/ctr_atomics.hip: 5520 waves adding once: 65 us on one
// word or on 16 words of one line, 7 us -- the empty kernel -- on 16 words of 16 lines).  So a slot is 17 lines: word 0 of
// line 0 is a FLAG (plain store of 1: what the kernels that only ask "any?" read, with a scalar load), lines 1 .. 16 hold
// the count, a wave adding to the line of its bin; the one wave that runs the update adds the sixteen up.
// (kOvfLines, kOvfStride, kOvfSlotWords: bf_device.h)
__device__ __forceinline__ uint32_t* ovf_counter(uint32_t* slot, int b) { return slot + kOvfStride * (1 + (b & (kOvfLines - 1))); }
__device__ __forceinline__ uint32_t ovf_part(const uint32_t* slot, int lane) {   // lanes 0 .. 15 of one wave
    return lane < kOvfLines ? slot[kOvfStride * (1 + lane)] : 0u;
}
__device__ __forceinline__ uint32_t ovf_total_wave(uint32_t part) {   // every lane of the wave calls; every lane gets the total
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_total_dpp(part), 63);
}
__device__ __forceinline__ void ovf_slot_clear(uint32_t* slot, int lane) {   // lanes 0 .. 16 of one wave
    if (lane <= kOvfLines) slot[kOvfStride * lane] = 0u;
}

// `cur` is the plane buffer of the iteration the sums belonged to; `ovf_now` the number of events that took the
// overflow path in it (tile-binned loop: the count lives outside the state, see k_bin_warp_scatter).
__device__ __forceinline__ void model_update_rest(DevState* st, bf_trace_rec* trace, int cur, uint32_t ovf_now = 0) {
    // plane-buffer bookkeeping: the stencil of this iteration cleared buffer cur^1 (v1 paths
    // always, the tile-binned path when it was dirty); buffer `cur` is dirty iff something was
    // scattered into it.  Many overflow events -> ask the host for a re-bin.
    const WarpParams wp = st->hot.wp;
    if (st->hot.binned) {
        if (st->hot.flip) {   // a re-bin moved the events to the other set: commit
            st->hot.cs ^= 1;
            st->hot.flip = 0;
        }
        const uint32_t oc = ovf_now;
        st->last_ovf = oc;
        st->ovf_total += oc;
        // Predictive re-bin: bound how far the new warp can have moved any event (scaled pixels)
        // since the bins were built, and re-sort BEFORE events start leaving their LDS tiles.
        //   n = dn(translation) + (div, rot) x lever arm;  displacement = scale * n / 127 * t / 1e4
        const WarpParams& r = st->ref_wp;
        const double dn = fabs(wp.dnx - r.dnx) + fabs(wp.dny - r.dny) +
                          2.0 * st->r_max * (fabs(wp.div - r.div) + fabs(wp.s - r.s)) +
                          (fabs(wp.div) + fabs(wp.s)) * (fabs(wp.cx - r.cx) + fabs(wp.cy - r.cy));
        const double drift = (double)st->hot.scale * dn / 127.0 * st->t_abs_max / 10000.0;
        if (st->hot.bin_ok) {   // (bin_ok == 0: every event takes the overflow path by design, nothing to re-sort)
            if (drift > st->drift_limit) st->hot.need_rebin = 1;
            // safety net: whatever the bound missed shows up as overflow events
            if ((unsigned long long)oc * 64ull > (unsigned long long)st->n_events) st->hot.need_rebin = 1;
        }
    } else {
        st->hot.ovf_cnt[cur] = 1;
        st->hot.ovf_cnt[cur ^ 1] = 0;
    }
    const int it = st->hot.it;
    if (trace && it <= st->trace_cap) {
        bf_trace_rec& r = trace[it - 1];
        r.model = st->model;
        r.x_divider = st->x_div; r.y_divider = st->y_div; r.rot_divider = st->rot_div; r.div_divider = st->div_div;
        r.iteration = it;
    }
}

// One pixel of the gated 3x3 Scharr (accel_lib.h:513-615) and its contribution to the centre-of-
// mass and moment sums (object_model.cpp:4-39,103-126).  tp points at the pixel inside an LDS time
// tile whose row pitch is TW (halo 1 all round); (gr, gc) is the pixel in the R x C image.
// The same sums with 32-bit integer fields: what one THREAD accumulates over its handful of pixels (the 64-bit adds of
// `Sums` are two vector instructions each, three of them per valid pixel); widened by sums_widen before any reduction.
struct SumsT {
    int n, sci, scj;
    double sgx, sgy, sigx, sigy, sjgx, sjgy;
};
__device__ __forceinline__ void sums_zero(SumsT& s) {
    s.n = s.sci = s.scj = 0;
    s.sgx = s.sgy = s.sigx = s.sigy = s.sjgx = s.sjgy = 0.0;
}
__device__ __forceinline__ Sums sums_widen(const SumsT& t) {
    Sums s;
    s.n = t.n; s.sci = t.sci; s.scj = t.scj;
    s.sgx = t.sgx; s.sgy = t.sgy; s.sigx = t.sigx; s.sigy = t.sigy; s.sjgx = t.sjgx; s.sjgy = t.sjgy;
    return s;
}

// BRANCH-FREE: every lane reads its nine taps and does the arithmetic, the gates select at the end.  (With `if (valid)` /
// `if (all)` blocks the compiler kept the running sums in different registers on the two paths and paid for it in copies --
// ~20 v_mov per pixel -- while a wave only skips a block when all 64 lanes agree, i.e. almost never.)  Same bits: what the
// gates used to skip is replaced by adding +0.0 -- x + 0.0 == x for every x but -0.0, and a running sum is never -0.0 (it
// starts as +0.0, and in round-to-nearest a sum is -0.0 only if both terms are) -- and by fma(c, +0.0, s) = s + (+-0.0) = s.
template <int TW, class S>
__device__ __forceinline__ void stencil_px(const float* tp, int gr, int gc, int R, int C, int hR, int hC, S& sm,
                                           float& gx, float& gy) {
    const float ctr = tp[0];
    const bool v = valid_px(ctr);
    // accel_lib.h:594-604: k = column offset (outer), l = row offset (inner),
    // idx = 3k + l; sharr_x = {3,0,-3,10,0,-10,3,0,-3},
    // sharr_y = {3,10,3,0,0,0,-3,-10,-3}; any tap <= 1e-6 -> gradient stays 0.
    const float t00 = tp[-TW - 1], t10 = tp[-1], t20 = tp[TW - 1];
    const float t01 = tp[-TW], t21 = tp[TW];
    const float t02 = tp[-TW + 1], t12 = tp[1], t22 = tp[TW + 1];
    // "all eight taps > 1e-6f" as ONE comparison of their minimum -- taken on the bit patterns as signed integers (three
    // v_min3_i32 + one v_min_i32 instead of eight compares): for non-negative floats the integer order is the float order;
    // a negative tap (mean time before the slice start) or -0.f has the sign bit set, is the integer minimum, and fails
    // the test as it must.  No NaN can be here: a time pixel is 0 or a finite quotient.
    const int m8 = min(min(min(__float_as_int(t00), __float_as_int(t10)), min(__float_as_int(t20), __float_as_int(t01))),
                       min(min(__float_as_int(t21), __float_as_int(t02)), min(__float_as_int(t12), __float_as_int(t22))));
    const bool grad = v && valid_px(__int_as_float(m8)) && gr >= 1 && gr < R - 1 && gc >= 1 && gc < C - 1;
    // The reference adds all nine weight * tap products in this order, including the weights that are 0.
    // Those terms are dropped here without changing a bit: every tap passed valid_px, so it is a finite
    // positive number and tap * 0.f == +0.f; the running sum is never -0.f (it starts as 0.f + a positive
    // product, and a sum of non-zero terms can only cancel to +0.f in round-to-nearest), and x + (+0.f) == x
    // for every x other than -0.f.  (Where the gate is closed the taps may be anything finite: the result is discarded.)
    // k = 0 (column c-1): l = 0,1,2 (rows r-1, r, r+1)
    // (the reference's 0.f + t00 * 3.f: with the gate open t00 > 1e-6f, the product is positive and 0.f + it is itself)
    float dx = t00 * 3.f, dy = dx;
    /* t10 * 0.f */        dy = dy + t10 * 10.f;
    dx = dx + t20 * -3.f;  dy = dy + t20 * 3.f;
    // k = 1 (column c): dy's three weights are 0, dx's centre weight is 0
    dx = dx + t01 * 10.f;
    dx = dx + t21 * -10.f;
    // k = 2 (column c+1)
    dx = dx + t02 * 3.f;   dy = dy + t02 * -3.f;
    /* t12 * 0.f */        dy = dy + t12 * -10.f;
    dx = dx + t22 * -3.f;  dy = dy + t22 * -3.f;
    gx = grad ? dx : 0.f;
    gy = grad ? dy : 0.f;
    // object_model.cpp:22-30 and :112-116 in one pass, centred coordinates; an invalid pixel adds zeros
    const int ci = gr - hR, cj = gc - hC;
    sm.n += v ? 1 : 0;
    sm.sci += v ? ci : 0;
    sm.scj += v ? cj : 0;
    const double gxd = (double)gx, gyd = (double)gy;   // (+0.0 wherever the gradient gate is closed)
    sm.sgx += gxd;
    sm.sgy += gyd;
    // (an integer below 2^21 times an f32 value is exact in f64, so the fused multiply-add rounds exactly where the
    // separate multiply and add did: same bits, four instructions fewer per pixel)
    sm.sigx = fma((double)ci, gxd, sm.sigx);
    sm.sigy = fma((double)ci, gyd, sm.sigy);
    sm.sjgx = fma((double)cj, gxd, sm.sjgx);
    sm.sjgy = fma((double)cj, gyd, sm.sjgy);
}

typedef unsigned int bf_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kStateWords = (int)(sizeof(DevState) / 8);

// A structure in LDS -> scalar registers: every lane reads it, v_readfirstlane makes each word uniform (the values a
// kernel branches and addresses with should not occupy vector registers).
template <class T>
__device__ __forceinline__ T lds_uniform(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    T out;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) dst[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)src[i]);
    return out;
}

// What the scatter loop needs from the state, in scalar registers.
struct ScatterHot {
    int32_t done, bin_tbits, bin_ok, fmt, scale, C, wsx, wsy, x_sh, y_sh;
    long long tmin;
    WarpParams wp;
};
__device__ __forceinline__ int lds_sreg(const int32_t* p) { return __builtin_amdgcn_readfirstlane(*p); }
__device__ __forceinline__ ScatterHot scatter_hot(const DevState* s) {
    ScatterHot h;
    h.done = lds_sreg(&s->hot.done); h.bin_tbits = lds_sreg(&s->hot.bin_tbits); h.bin_ok = lds_sreg(&s->hot.bin_ok);
    h.fmt = lds_sreg(&s->hot.fmt);
    h.scale = lds_sreg(&s->hot.scale); h.C = lds_sreg(&s->hot.C); h.wsx = lds_sreg(&s->hot.wsx); h.wsy = lds_sreg(&s->hot.wsy);
    h.x_sh = lds_sreg(&s->hot.x_sh); h.y_sh = lds_sreg(&s->hot.y_sh);
    const long long tm = s->hot.tmin;
    h.tmin = ((long long)__builtin_amdgcn_readfirstlane((int)(tm >> 32)) << 32) |
             (long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)tm);
    h.wp = lds_uniform(&s->hot.wp);
    return h;
}


// ---- moment sums across work-groups: exact fixed-point accumulators (bf_device.h: MomentAcc) ----------------------
__device__ __forceinline__ void fx_split(double v, unsigned long long& hi, unsigned long long& lo) {
    v = fmin(fmax(v, -562949953421312.0), 562949953421312.0);   // |v| <= 2^49 (never reached; NaN -> bound)
    const double s = v * 4096.0;          // exact (power of two)
    const double fl = floor(s);           // exact
    hi = (unsigned long long)(long long)fl;
    lo = (unsigned long long)((s - fl) * 4503599627370496.0);   // (s - fl) in [0, 1) is exact; 2^52: truncation below 2^-64
}
__device__ __forceinline__ double fx_join(unsigned long long hi, unsigned long long lo) {
    return ((double)(long long)hi + (double)lo * (1.0 / 4503599627370496.0)) * (1.0 / 4096.0);
}

// The per-lane word model_update_wave expects, from sums held in registers (every lane holds `t`): field lane % 16 of
// the MomentAcc layout.  Used where the sums never went through the accumulators (bf_tiles.hip).
__device__ __forceinline__ unsigned long long sums_lane_word(const Sums& t, int lane) {
    const int f = lane & 15;
    const double d[6] = {t.sgx, t.sgy, t.sigx, t.sigy, t.sjgx, t.sjgy};
    const int k = (f >= 3) ? ((f - 3) >> 1) : 0;
    double dv = d[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) dv = (k == q) ? d[q] : dv;
    unsigned long long hi, lo;
    fx_split(dv, hi, lo);
    unsigned long long v = ((f - 3) & 1) ? lo : hi;
    v = (f == 0) ? (unsigned long long)t.n : v;
    v = (f == 1) ? (unsigned long long)t.sci : v;
    v = (f == 2) ? (unsigned long long)t.scj : v;
    v = (f == 15) ? 0ull : v;
    return v;
}


// Adds one work-group's sums into accumulator group `grp`: fifteen fire-and-forget device atomics, one per lane of
// the calling wave's first fifteen lanes (every thread holds `t`).  Nothing is read back and nothing is waited for:
// the end of the kernel (or the caller's own s_waitcnt) completes them.
__device__ __forceinline__ void acc_add(MomentAcc* acc, int grp, const Sums& t, int tid) {
    if (tid >= kAccFields) return;
    (void)__hip_atomic_fetch_add(&acc[grp].f[tid], sums_lane_word(t, tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Total of the kAccGroups accumulator groups, by ONE wave (all 64 lanes; no work-group barrier): lane l loads field
// l % 16 of groups l / 16, l / 16 + 4, ... (sixteen lanes read one 128-byte line) -- acc_load_wave, to be issued as
// early as possible -- and acc_reduce_wave adds them and finishes with two cross-lane steps (lanes l, l ^ 16, l ^ 32):
// afterwards EVERY lane holds the total of field l % 16, the form model_update_wave takes.  AGENT: the accumulators
// were written by other work-groups of THIS launch (L1-bypassing loads); otherwise by an earlier kernel (plain loads).
// ZERO: the reader clears them for their next use.
constexpr int kAccPerLane = kAccGroups / 4;
template <bool AGENT, bool ZERO>
__device__ __forceinline__ void acc_load_wave(MomentAcc* acc, int lane, unsigned long long (&v)[kAccPerLane]) {
    static_assert(kAccGroups % 4 == 0, "four groups per pass");
    unsigned long long* src = &acc[lane >> 4].f[lane & 15];
#pragma unroll
    for (int k = 0; k < kAccPerLane; ++k) {
        unsigned long long* q = src + (size_t)k * 4 * 16;
        v[k] = AGENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    }
    if (ZERO) {
#pragma unroll
        for (int k = 0; k < kAccPerLane; ++k) src[(size_t)k * 4 * 16] = 0ull;
    }
}
__device__ __forceinline__ unsigned long long acc_reduce_wave(const unsigned long long (&v)[kAccPerLane]) {
    unsigned long long tot = 0;
#pragma unroll
    for (int k = 0; k < kAccPerLane; ++k) tot += v[k];
    tot += __shfl_xor(tot, 16, 64);
    tot += __shfl_xor(tot, 32, 64);
    return tot;
}

// Second half of the stencil kernels: given the time tile in LDS ((TR+2) x (TC+2), halo 1),
// the gated 3x3 Scharr (accel_lib.h:513-615), the centre-of-mass and moment sums
// (object_model.cpp:4-39,103-126), optional gradient output, clearing of the other plane
// buffer, and the wave64-shuffle + LDS reduction into one Partial per work-group.
// NT: threads of the work-group (256, or 512 on small images: half the pixels per thread, a shorter dependent chain).
// OUT: the caller may ask for the gradient planes (the stand-alone operators); the loop's stencil kernel never does, and the
// compiled-out stores take their address arithmetic and their scalar registers with them.
template <int TR, int TC, int NT, bool OUT = true>
__device__ __forceinline__ void stencil_tail(const StencilArgs& a, const float* s_time, Sums* s_red,
                                             int r0, int c0, bool do_zero, double* s_scr = nullptr /* NT / 64 x 192 doubles of free LDS */) {
    constexpr int TW = TC + 2;
    const int R = a.R, C = a.C;
    const int tid = threadIdx.x;
    SumsT smt;   // (a thread's own pixels: 32-bit integer sums)
    sums_zero(smt);
    const int hR = R / 2, hC = C / 2;
    // (a wave covers one tile row: the row is uniform -- scalar unit --, the column is the lane)
    static_assert(TC == 64 && NT % 64 == 0, "one lane per tile column");
    const int lc = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int k = 0; k < (TR * TC) / NT; ++k) {
        const int lr = wv + k * (NT / 64);
        const int gr = r0 + lr, gc = c0 + lc;
        if (gr < R && gc < C) {
            float gx, gy;
            stencil_px<TW>(&s_time[(lr + 1) * TW + (lc + 1)], gr, gc, R, C, hR, hC, smt, gx, gy);
            if (OUT && a.gx_out) {
                a.gx_out[(size_t)gr * C + gc] = gx;
                a.gy_out[(size_t)gr * C + gc] = gy;
            }
            if (do_zero) {
                if (a.zero_bits && !a.zero_full) {
                    // tile-binned loop: what the previous iteration's overflow events left is flagged pixel by pixel -- clear
                    // those (12 bytes each) instead of the whole buffer (32 MB at 640x480), then the flags.  (A wave covers
                    // one tile row, so a bitmap word is read and cleared by lanes of one wave, in program order.)
                    uint32_t* wp = a.zero_bits + ((size_t)gr * (size_t)a.ovf_pitch + (size_t)(gc >> 5) + 1);
                    const uint32_t wv = *wp;
                    if ((wv >> (gc & 31)) & 1u) {
                        a.zero_plane[(size_t)gr * C + gc] = 0ull;
                        if (a.zero_cplane) a.zero_cplane[(size_t)gr * C + gc] = 0u;
                    }
                    if (wv != 0u && (gc & 31) == 0) *wp = 0u;
                } else {
                    a.zero_plane[(size_t)gr * C + gc] = 0ull;
                    if (a.zero_cplane) a.zero_cplane[(size_t)gr * C + gc] = 0u;
                    if (a.zero_bits && (gc & 31) == 0) a.zero_bits[(size_t)gr * (size_t)a.ovf_pitch + (size_t)(gc >> 5) + 1] = 0u;
                }
            }
        }
    }
    const Sums sm = sums_widen(smt);
    if (a.acc) {
        tl_stamp(a.tl, a.tl_launch, 5);
        // Fused-update form (a.ticket): the state is stable while this kernel runs (only its own last work-group writes
        // it), so every work-group fetches a copy into LDS now, off the critical path: the update then runs on it at
        // once (a dependent global round trip per field cost 3.2 us; copying after the ticket ~1 us).
        __shared__ DevState s_state;
        static_assert(sizeof(DevState) % 8 == 0 && sizeof(DevState) / 8 <= 64, "one u64 per lane of one wave");
        // (read from `st`, the buffer every work-group took its hot fields from; the update goes to `st_rw` -- the same
        // buffer in the global-atomic loop, the OTHER one in the tile-binned loop, whose next scatter launch then finds the
        // new state without anybody copying it: the lean scatter kernel's work-group 0 used to carry it over, a vector
        // load that its scatter loop's header waited for before the first event load)
        if (a.ticket && tid < (int)(sizeof(DevState) / 8))
            reinterpret_cast<unsigned long long*>(&s_state)[tid] = reinterpret_cast<const unsigned long long*>(a.st)[tid];
        __shared__ unsigned long long s_rpart[kSumFields * (NT / 64)];
        constexpr bool kPack = TR * TC <= 1024 && TR <= 64 && TC <= 64;
        block_reduce_publish<NT, kPack>(sm, s_rpart, tid, r0 - hR, c0 - hC, s_scr);
        // Everything below is the FIRST WAVE's: the work-group's total, the accumulator adds, the ticket and -- in the
        // last work-group -- the update.  The other waves are done (a quarter of a tile's instructions used to be every
        // wave adding up the same partials).  No work-group barrier from here on: one wave, in order.
        if (tid >= 64) return;
        const Sums blk = block_reduce_total<NT, kPack>(s_rpart, r0 - hR, c0 - hC);
        tl_stamp(a.tl, a.tl_launch, 6);
        const int nblk = gridDim.x * gridDim.y;
        const int me = blockIdx.y * gridDim.x + blockIdx.x;
        // every work-group adds its sums to the exact accumulators (order-free: see MomentAcc)
        acc_add(a.acc, me % kAccGroups, blk, tid);
        if (!a.ticket) {
            // Tile-binned loop: that is all.  The total is formed and the model / loop update runs at the head of the next
            // warp+scatter launch (k_bin_warp_scatter), by every work-group for itself -- no ticket, no last work-group,
            // no single-CU tail.  The accumulators of the OTHER parity (consumed by this iteration's warp+scatter head)
            // and the overflow counter of the next iteration are cleared here for their next use.
            if (me == 0) {
                if (a.acc_zero)
                    for (int i = tid; i < kAccGroups * 16; i += 64) (&a.acc_zero[0].f[0])[i] = 0ull;
                if (a.ovf_next) ovf_slot_clear(a.ovf_next, tid);
            }
            return;
        }
        // Fused reduction + update: the LAST work-group to arrive reads the accumulators and runs the model / loop
        // update -- no separate kernel.  Hand-off (cdna_hip_programming.md, Guideline 16): the atomics are drained
        // with s_waitcnt vmcnt(0), then a relaxed agent-scope ticket; the reader uses agent-scope (L1-bypassing) loads.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last = 0;
        if (tid == 0) {
            // Two-level ticket: one word serialises at ~11 ns per atomic (833 work-groups would
            // cost ~9 us), so arrivals are spread over kTicketGroups words on different cache
            // lines and only the last arriver of each group takes the top-level ticket.
            const int grp = me % kTicketGroups;
            const int grp_size = nblk / kTicketGroups + (grp < nblk % kTicketGroups ? 1 : 0);
            const int n_groups = nblk < kTicketGroups ? nblk : kTicketGroups;
            const unsigned int t1 = __hip_atomic_fetch_add(&a.ticket[16 * (1 + grp)], 1u, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
            if (t1 == (unsigned int)(grp_size - 1)) {
                a.ticket[16 * (1 + grp)] = 0;   // re-armed for the next launch
                const unsigned int t0 =
                    __hip_atomic_fetch_add(&a.ticket[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = (t0 == (unsigned int)(n_groups - 1)) ? 1 : 0;
            }
        }
        last = __builtin_amdgcn_readfirstlane(last);   // (lane 0's verdict, for the whole wave)
        tl_stamp(a.tl, a.tl_launch, 7);
        if (!last) return;
        // (the overflow count of this iteration -- final since the scatter kernel ended -- is requested together with the
        // accumulators and used after the update; it used to be a dependent load of one lane behind the update)
        const uint32_t ovf_p = a.ovf_cur ? ovf_part(a.ovf_cur, tid) : 0u;
        unsigned long long accv[kAccPerLane];
        acc_load_wave<true, true>(a.acc, tid, accv);
        const unsigned long long word = acc_reduce_wave(accv);
        model_update_wave(&s_state, word, tid, a.update_mode);
        const uint32_t ovf_now = ovf_total_wave(ovf_p);
        if (tid == 0) {
            a.ticket[0] = 0;   // ready for the next launch (the kernel boundary orders it)
            if (a.update_mode != 0) model_update_rest(&s_state, a.trace, a.cur, ovf_now);
        }
        if (a.ovf_next) ovf_slot_clear(a.ovf_next, tid);   // (tile-binned loop: the next iteration's overflow counter)
        __builtin_amdgcn_wave_barrier();   // (LDS operations of one wave complete in order)
        if (tid < (int)(sizeof(DevState) / 8)) {
            const unsigned long long v = reinterpret_cast<const unsigned long long*>(&s_state)[tid];
            reinterpret_cast<unsigned long long*>(a.st_rw)[tid] = v;
            if (a.snap) reinterpret_cast<unsigned long long*>(a.snap)[tid] = v;
        }
    }
}


}  // namespace bf
