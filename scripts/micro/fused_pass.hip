// fused_pass.hip -- how long would ONE kernel per iteration take?  (the candidate that was sized and not
// built.)  A stand-alone measurement, not product code: it reuses the library's device functions (warp, time image,
// Scharr, moment sums, exact accumulators, model update), so a pass does the arithmetic of one iteration of the loop, but
// the binning is done on the HOST here and no drift gating / re-binning exists.
//
//   Today:  K1 (warp + LDS scatter + slab flush)  ->  K3 (slab merge + box sum + time image + Scharr + moments)
//   Here :  every bin owns an image tile of TSR x 64 scaled pixels and holds the events whose target was within
//           H + D pixels of it at binning time (H = scale / 2 + 1: box sum + Scharr halo; D: the drift a bin tolerates
//           before a re-bin) -- events near a tile edge are DUPLICATED into the neighbouring bins.  One work-group per
//           bin: update at the head, warp + scatter of its events into an LDS tile with halo H, then -- in the same
//           launch -- each 256-thread sub-group runs the box sum / time image / Scharr / moment sums of one 16 x 64
//           sub-tile (the unit the stencil kernel of today works on, so the per-sub-tile f64 partials, and with them
//           the exact accumulators, are bit-identical).  No slabs, no second launch.
//
// The self-check: the accumulator totals of a pass are the same for every tiling / D (integer sums of identical
// sub-tile partials) -- the program prints a digest per configuration.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I better_flow_amd/csrc scripts/micro/fused_pass.hip -o scripts/micro/fused_pass
//   scripts/micro/fused_pass [H W [n_events]]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "bf_device.h"
#include "bf_device_fns.h"

using namespace bf;

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct PassArgs {
    const uint32_t* xy;
    const int32_t* t;
    float2* p;
    const uint32_t* bin_start;
    const DevState* st_in;
    DevState* st_out;
    MomentAcc* acc_in;
    MomentAcc* acc_out;
    int nbc, R, C;
};

constexpr int kStateWords = (int)(sizeof(DevState) / 8);

template <int HS, int NSUB, int U>
__global__ __launch_bounds__(256 * NSUB) void k_fused_pass(PassArgs a) {
    constexpr int THREADS = 256 * NSUB;
    constexpr int TR = kTileR, TC = kTileC;
    constexpr int H = HS + 1;
    constexpr int TSR = TR * NSUB;
    constexpr int AR = TSR + 2 * H, AC = TC + 2 * H;
    constexpr int PR = TR + 2 * H, PC = AC;
    constexpr int TH = TR + 2, TW = TC + 2;
    __shared__ unsigned long long s_acc[AR * AC];
    __shared__ float s_time[NSUB][TH * TW];
    __shared__ unsigned long long s_rpart[NSUB][kSumFields * 4];
    __shared__ DevState s_state, s_scratch;
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t beg = sload(a.bin_start + b), end = sload(a.bin_start + b + 1);
    unsigned long long accv[kAccPerLane];
    if (tid < 64) acc_load_wave<false, false>(a.acc_in, tid, accv);
    unsigned long long state_word = 0;
    if (tid < kStateWords) state_word = reinterpret_cast<const unsigned long long*>(a.st_in)[tid];
    const int br = b / a.nbc, bc = b - br * a.nbc;
    const int X0 = br * TSR - H, Y0 = bc * TC - H;
    const uint32_t* __restrict__ xy = a.xy;
    const int32_t* __restrict__ t = a.t;
    float2* __restrict__ p = a.p;
    uint32_t vxy[U];
    int32_t vt[U];
    float2 vp[U];
    uint32_t base = beg;
    auto load_pass = [&]() {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            uint32_t i = base + k * THREADS + tid;
            i = i < end ? i : beg;
            vxy[k] = xy[i];
            vt[k] = t[i];
            vp[k] = p[i];
        }
    };
    load_pass();
    asm volatile("" ::: "memory");
    {
        ulonglong2* z = reinterpret_cast<ulonglong2*>(s_acc);
        for (int i = tid; i < AR * AC / 2; i += THREADS) z[i] = make_ulonglong2(0ull, 0ull);
    }
    if (tid < kStateWords) {
        reinterpret_cast<unsigned long long*>(&s_state)[tid] = state_word;
        reinterpret_cast<unsigned long long*>(&s_scratch)[tid] = state_word;
    }
    double ppx[U], ppy[U];
    auto previous_positions = [&]() {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            ppx[k] = pr_from_p(vxy[k] & 0xffffu, vp[k].x);
            ppy[k] = pr_from_p(vxy[k] >> 16, vp[k].y);
        }
    };
    if (tid < 64) {   // the update's latency is paid (on a scratch copy: the measurement keeps its warp parameters)
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long word = acc_reduce_wave(accv);
        __builtin_amdgcn_wave_barrier();
        model_update_wave(&s_scratch, word, tid, 1);
        __builtin_amdgcn_s_setprio(0);
    } else {
        previous_positions();
    }
    __syncthreads();
    const HotState* sh = &s_state.hot;
    const int scale = __builtin_amdgcn_readfirstlane(sh->scale), wsx = __builtin_amdgcn_readfirstlane(sh->wsx),
              wsy = __builtin_amdgcn_readfirstlane(sh->wsy), x_sh = __builtin_amdgcn_readfirstlane(sh->x_sh),
              y_sh = __builtin_amdgcn_readfirstlane(sh->y_sh), bt = __builtin_amdgcn_readfirstlane(sh->bin_tbits);
    const long long tmin = (long long)__builtin_amdgcn_readfirstlane((int)sh->tmin);
    WarpParams wp;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&sh->wp);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&wp);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(WarpParams) / 4); ++i) dst[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)src[i]);
    }
    // keep the scratch update alive
    if (b == 0 && tid < kStateWords) reinterpret_cast<unsigned long long*>(a.st_out)[tid] = reinterpret_cast<const unsigned long long*>(&s_scratch)[tid];
    if (tid < 64) previous_positions();
    const int hsc = scale / 2;
    for (;;) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t i = base + k * THREADS + tid;
            if (i >= end) continue;
            float2 q;
            double nx, ny;
            warp_products(wp, ppx[k], ppy[k], vt[k], q, nx, ny);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(&p[i]),
                               ((unsigned long long)__float_as_uint(q.y) << 32) | (unsigned long long)__float_as_uint(q.x),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double px = pr_from_p(vxy[k] & 0xffffu, q.x), py = pr_from_p(vxy[k] >> 16, q.y);
            const int X = trunc_scatter(px * (double)scale + (double)x_sh);
            const int Y = trunc_scatter(py * (double)scale + (double)y_sh);
            if ((X >= wsx + hsc) || (X < hsc) || (Y >= wsy + hsc) || (Y < hsc)) continue;
            const int lx = X - X0, ly = Y - Y0;
            if (lx >= 0 && lx < AR && ly >= 0 && ly < AC)
                atomicAdd(&s_acc[lx * AC + ly], (1ull << bt) + (unsigned long long)((long long)vt[k] - tmin));
        }
        base += THREADS * U;
        if (base >= end) break;
        load_pass();
        previous_positions();
    }
    __syncthreads();
    // ---- the stencil of today's K3, one 16 x 64 sub-tile per 256-thread sub-group, on the LDS tile ----
    const int g = tid >> 8, lt = tid & 255;
    const int R = a.R, C = a.C;
    const int r0 = br * TSR + g * TR, c0 = bc * TC;
    const unsigned long long bm = (1ull << bt) - 1ull;
    const unsigned long long* win = s_acc + (g * TR) * AC;   // rows r0 - H .. of this sub-tile
    for (int idx = lt; idx < TH * TW; idx += 256) {
        const int tr = idx / TW, tc = idx - tr * TW;
        const int gr = r0 - 1 + tr, gc = c0 - 1 + tc;
        float tv = 0.f;
        if (gr >= 0 && gr < R && gc >= 0 && gc < C) {
            unsigned long long pk = 0;
#pragma unroll
            for (int da = 0; da <= 2 * HS; ++da)
#pragma unroll
                for (int db = 0; db <= 2 * HS; ++db) pk += win[(tr + da) * PC + (tc + db)];
            tv = time_from_sums((uint32_t)(pk >> bt), (long long)(pk & bm), tmin);
        }
        s_time[g][idx] = tv;
    }
    __syncthreads();
    Sums sm;
    sums_zero(sm);
    const int hR = R / 2, hC = C / 2;
#pragma unroll
    for (int k = 0; k < (TR * TC) / 256; ++k) {
        const int pidx = lt + k * 256;
        const int lr = pidx / TC, lc = pidx - lr * TC;
        const int gr = r0 + lr, gc = c0 + lc;
        if (gr < R && gc < C) {
            float gx, gy;
            stencil_px<TW>(&s_time[g][(lr + 1) * TW + (lc + 1)], gr, gc, R, C, hR, hC, sm, gx, gy);
        }
    }
    block_reduce_publish<256, true>(sm, s_rpart[g], lt, r0 - hR, c0 - hC);
    if (lt >= 64) return;
    if (r0 >= R) return;   // (a sub-tile below the image: nothing to add)
    const Sums blk = block_reduce_total<256, true>(s_rpart[g], r0 - hR, c0 - hC);
    acc_add(a.acc_out, (b * NSUB + g) % kAccGroups, blk, lt);
}

struct HostBins {
    std::vector<uint32_t> xy, start;
    std::vector<int32_t> t;
    int nbr, nbc;
    double dup;
};

static HostBins bin_events(const std::vector<uint32_t>& xy, const std::vector<int32_t>& t, int S, int R, int C, int TSR, int H, int D) {
    HostBins hb;
    hb.nbr = (R + TSR - 1) / TSR;
    hb.nbc = (C + 63) / 64;
    const int nb = hb.nbr * hb.nbc, E = H + D;
    std::vector<uint32_t> cnt(nb + 1, 0);
    auto each = [&](size_t i, auto&& f) {
        const int X = (int)(xy[i] & 0xffff) * S + S / 2, Y = (int)(xy[i] >> 16) * S + S / 2;
        const int r_lo = std::max((X - E) / TSR, 0), r_hi = std::min((X + E) / TSR, hb.nbr - 1);
        const int c_lo = std::max((Y - E) / 64, 0), c_hi = std::min((Y + E) / 64, hb.nbc - 1);
        for (int r = (X - E < 0 ? 0 : r_lo); r <= r_hi; ++r)
            for (int c = (Y - E < 0 ? 0 : c_lo); c <= c_hi; ++c) f(r * hb.nbc + c);
    };
    for (size_t i = 0; i < xy.size(); ++i) each(i, [&](int b) { ++cnt[b + 1]; });
    for (int b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
    hb.start = cnt;
    hb.xy.resize(cnt[nb]);
    hb.t.resize(cnt[nb]);
    std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
    for (size_t i = 0; i < xy.size(); ++i) each(i, [&](int b) { hb.xy[cur[b]] = xy[i]; hb.t[cur[b]] = t[i]; ++cur[b]; });
    hb.dup = (double)cnt[nb] / (double)xy.size();
    return hb;
}

// K independent slices (own copies of everything) on K streams: the co-scheduled regime bench.py's `value` is measured in.
template <int NSUB, int U>
static void run_streams(const char* tag, const std::vector<uint32_t>& xy, const std::vector<int32_t>& t, int S, int Hs, int Ws, int D, int K) {
    const int R = S * Hs, C = S * Ws, HSc = S / 2, H = HSc + 1, TSR = 16 * NSUB;
    HostBins hb = bin_events(xy, t, S, R, C, TSR, H, D);
    const int nb = hb.nbr * hb.nbc;
    const size_t m = hb.xy.size();
    std::vector<PassArgs> args(K);
    std::vector<hipStream_t> st(K);
    DevState hst;
    memset(&hst, 0, sizeof hst);
    hst.hot.scale = S; hst.hot.R = R; hst.hot.C = C; hst.hot.wsx = R - S; hst.hot.wsy = C - S; hst.hot.x_sh = S / 2; hst.hot.y_sh = S / 2;
    hst.hot.bin_tbits = 40;
    hst.hot.wp.dnx = 0.03; hst.hot.wp.dny = -0.02; hst.hot.wp.cx = Hs / 2.0; hst.hot.wp.cy = Ws / 2.0; hst.hot.wp.c = 1;
    hst.x_div = hst.y_div = 10; hst.rot_div = hst.div_div = 1000;
    std::vector<MomentAcc> acc(kAccGroups);
    memset(acc.data(), 0, sizeof(MomentAcc) * kAccGroups);
    acc[0].f[0] = 500000;
    for (int k = 0; k < K; ++k) {
        uint32_t *d_xy, *d_start; int32_t* d_t; float2* d_p; DevState *d_st, *d_st2; MomentAcc *d_in, *d_out;
        CK(hipMalloc(&d_xy, m * 4)); CK(hipMalloc(&d_t, m * 4)); CK(hipMalloc(&d_p, m * 8)); CK(hipMalloc(&d_start, (nb + 1) * 4));
        CK(hipMalloc(&d_st, sizeof(DevState))); CK(hipMalloc(&d_st2, sizeof(DevState)));
        CK(hipMalloc(&d_in, sizeof(MomentAcc) * kAccGroups)); CK(hipMalloc(&d_out, sizeof(MomentAcc) * kAccGroups));
        CK(hipMemcpy(d_xy, hb.xy.data(), m * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_t, hb.t.data(), m * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_p, 0, m * 8));
        CK(hipMemcpy(d_start, hb.start.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_st, &hst, sizeof hst, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_in, acc.data(), sizeof(MomentAcc) * kAccGroups, hipMemcpyHostToDevice));
        CK(hipMemset(d_out, 0, sizeof(MomentAcc) * kAccGroups));
        args[k] = {d_xy, d_t, d_p, d_start, d_st, d_st2, d_in, d_out, hb.nbc, R, C};
        CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    }
    auto launch = [&](int k) {
        if (S / 2 == 1) hipLaunchKernelGGL((k_fused_pass<1, NSUB, U>), dim3(nb), dim3(256 * NSUB), 0, st[k], args[k]);
        else hipLaunchKernelGGL((k_fused_pass<0, NSUB, U>), dim3(nb), dim3(256 * NSUB), 0, st[k], args[k]);
    };
    const int reps = 300;
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i)
            for (int k = 0; k < K; ++k) launch(k);
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        best = std::min(best, us);
    }
    printf("%-26s %dx%d s%d  tile %3dx64 D %2d  %d streams: %6.2f us per pass and slice (aggregate), %6.2f us per round of %d\n", tag, Ws, Hs, S, TSR, D, K,
           best / (reps * K), best / reps, K);
    for (int k = 0; k < K; ++k) {
        hipFree((void*)args[k].xy); hipFree((void*)args[k].t); hipFree(args[k].p); hipFree((void*)args[k].bin_start);
        hipFree((void*)args[k].st_in); hipFree(args[k].st_out); hipFree(args[k].acc_in); hipFree(args[k].acc_out);
        hipStreamDestroy(st[k]);
    }
}

template <int NSUB, int U>
static void run(const char* tag, const std::vector<uint32_t>& xy, const std::vector<int32_t>& t, int S, int Hs, int Ws, int D) {
    const int R = S * Hs, C = S * Ws, HSc = S / 2, H = HSc + 1, TSR = 16 * NSUB;
    HostBins hb = bin_events(xy, t, S, R, C, TSR, H, D);
    const int nb = hb.nbr * hb.nbc;
    const size_t m = hb.xy.size();
    uint32_t *d_xy, *d_start;
    int32_t* d_t;
    float2* d_p;
    DevState *d_st, *d_st2;
    MomentAcc *d_in, *d_out;
    CK(hipMalloc(&d_xy, m * 4)); CK(hipMalloc(&d_t, m * 4)); CK(hipMalloc(&d_p, m * 8)); CK(hipMalloc(&d_start, (nb + 1) * 4));
    CK(hipMalloc(&d_st, sizeof(DevState))); CK(hipMalloc(&d_st2, sizeof(DevState)));
    CK(hipMalloc(&d_in, sizeof(MomentAcc) * kAccGroups)); CK(hipMalloc(&d_out, sizeof(MomentAcc) * kAccGroups));
    CK(hipMemcpy(d_xy, hb.xy.data(), m * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_t, hb.t.data(), m * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_p, 0, m * 8));
    CK(hipMemcpy(d_start, hb.start.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
    DevState st;
    memset(&st, 0, sizeof st);
    st.hot.scale = S; st.hot.R = R; st.hot.C = C; st.hot.wsx = R - S; st.hot.wsy = C - S; st.hot.x_sh = S / 2; st.hot.y_sh = S / 2;
    st.hot.bin_tbits = 40; st.hot.tmin = 0;
    st.hot.wp.dnx = 0.03; st.hot.wp.dny = -0.02; st.hot.wp.cx = Hs / 2.0; st.hot.wp.cy = Ws / 2.0; st.hot.wp.div = 0; st.hot.wp.c = 1; st.hot.wp.s = 0;
    st.x_div = st.y_div = 10; st.rot_div = st.div_div = 1000;
    CK(hipMemcpy(d_st, &st, sizeof st, hipMemcpyHostToDevice));
    std::vector<MomentAcc> acc(kAccGroups);
    memset(acc.data(), 0, sizeof(MomentAcc) * kAccGroups);
    acc[0].f[0] = 500000;
    CK(hipMemcpy(d_in, acc.data(), sizeof(MomentAcc) * kAccGroups, hipMemcpyHostToDevice));
    PassArgs a = {d_xy, d_t, d_p, d_start, d_st, d_st2, d_in, d_out, hb.nbc, R, C};
    auto launch = [&]() {
        if (S / 2 == 1) hipLaunchKernelGGL((k_fused_pass<1, NSUB, U>), dim3(nb), dim3(256 * NSUB), 0, 0, a);
        else hipLaunchKernelGGL((k_fused_pass<0, NSUB, U>), dim3(nb), dim3(256 * NSUB), 0, 0, a);
    };
    for (int i = 0; i < 20; ++i) launch();
    CK(hipMemset(d_out, 0, sizeof(MomentAcc) * kAccGroups));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(acc.data(), d_out, sizeof(MomentAcc) * kAccGroups, hipMemcpyDeviceToHost));
    unsigned long long tot[kAccFields] = {0}, dig = 1469598103934665603ull;
    for (int gI = 0; gI < kAccGroups; ++gI)
        for (int f = 0; f < kAccFields; ++f) tot[f] += acc[gI].f[f];
    for (int f = 0; f < kAccFields; ++f) dig = (dig ^ tot[f]) * 1099511628211ull;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 300;
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        sum += ms;
    }
    printf("%-26s %dx%d s%d  tile %3dx64 D %2d  bins %5d  dup %.2fx  ev/bin %6.0f  pass %6.2f us (mean %6.2f)  n %llu digest %016llx\n", tag, Ws, Hs, S, TSR, D,
           nb, hb.dup, (double)m / nb, best * 1e3 / reps, sum * 1e3 / (5 * reps), tot[0], dig);
    hipFree(d_xy); hipFree(d_t); hipFree(d_p); hipFree(d_start); hipFree(d_st); hipFree(d_st2); hipFree(d_in); hipFree(d_out);
}

int main(int argc, char** argv) {
    const int Hs = argc > 1 ? atoi(argv[1]) : 260, Ws = argc > 2 ? atoi(argv[2]) : 346;
    const size_t n = argc > 3 ? (size_t)atoll(argv[3]) : 1000000;
    const int S = 3;
    std::mt19937_64 rng(1);
    std::vector<uint32_t> xy(n);
    std::vector<int32_t> t(n);
    // events on moving edges would cluster; uniform positions are the average case the per-bin passes are sized for
    for (size_t i = 0; i < n; ++i) {
        const uint32_t fx = (uint32_t)(rng() % (uint64_t)Hs), fy = (uint32_t)(rng() % (uint64_t)Ws);
        xy[i] = fx | (fy << 16);
        t[i] = (int32_t)(rng() % 30000000ull);
    }
    if (argc > 4) {   // a real slice: int32 fr_x[n], fr_y[n], t[n] (python: np.concatenate of synth.make_slice's arrays .tofile)
        FILE* f = fopen(argv[4], "rb");
        std::vector<int32_t> raw(3 * n);
        if (!f || fread(raw.data(), 4, 3 * n, f) != 3 * n) { fprintf(stderr, "cannot read %zu events from %s\n", n, argv[4]); return 1; }
        fclose(f);
        for (size_t i = 0; i < n; ++i) { xy[i] = (uint32_t)raw[i] | ((uint32_t)raw[n + i] << 16); t[i] = raw[2 * n + i]; }
        printf("events of %s\n", argv[4]);
    }
    for (int D : {4, 8, 16}) {
        run<4, 4>("64x64 / 1024 thr / U4", xy, t, S, Hs, Ws, D);
        run<4, 8>("64x64 / 1024 thr / U8", xy, t, S, Hs, Ws, D);
        run<2, 4>("32x64 / 512 thr / U4", xy, t, S, Hs, Ws, D);
        run<2, 8>("32x64 / 512 thr / U8", xy, t, S, Hs, Ws, D);
        run<1, 4>("16x64 / 256 thr / U4", xy, t, S, Hs, Ws, D);
        run<1, 8>("16x64 / 256 thr / U8", xy, t, S, Hs, Ws, D);
    }
    for (int D : {4, 8})
        for (int K : {2, 4, 8}) {
            run_streams<4, 4>("64x64 / 1024 thr / U4", xy, t, S, Hs, Ws, D, K);
            run_streams<2, 4>("32x64 / 512 thr / U4", xy, t, S, Hs, Ws, D, K);
            run_streams<1, 4>("16x64 / 256 thr / U4", xy, t, S, Hs, Ws, D, K);
        }
    return 0;
}
