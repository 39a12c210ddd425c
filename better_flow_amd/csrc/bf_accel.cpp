// bf_accel.cpp -- C-ABI (include/bf_accel.h) over the gfx950 kernels of bf_kernels.hip.
//
// Host-side counterpart of better-flow's AccelLib (accel_lib.h:14-616) plus the fused
// OptimizerRolling::run (optimizer_rolling.h:48-125).  A bf_ctx owns every device buffer,
// is reusable across slices and never allocates per slice.  There is no CPU fallback.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <utility>
#include <string>
#include <thread>
#include <vector>

#include "bf_device.h"
#include "bf_kernels.h"

using namespace bf;

namespace {

struct ProfRec {
    hipEvent_t a, b;
    int cat;
    long long nev;
};

}  // namespace

struct bf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    long long cap_events = 0;   // padded
    long long n = 0, n_pad = 0;
    size_t cap_px = 0;
    int cap_blocks = 0;

    // two event sets: the tile-binned mode ping-pongs between them on every re-bin
    struct EvSet { uint32_t* xy = nullptr; int32_t* t = nullptr; float2* p = nullptr; uint32_t* perm = nullptr; float2* p2 = nullptr; };
    EvSet set[2];
    int cs = 0;                      // set holding the live events
    bool has_perm = false;           // set[cs] is permuted; perm[] gives the upload index
    uint8_t* d_noise = nullptr;
    // tile-binned scatter
    int opt_binned = 1;              // 0 never, 1 when it pays (dense enough), 2 whenever possible
    bool opt_bin_predict = true;
    int opt_bin_tile = 0, opt_bin_margin = 8, opt_bin_threads = 0;   // bin_threads 0: by the events per bin   // bin_tile 0: chosen per slice
    bool opt_co_schedule = false;    // several slice contexts share the GPU: the update runs in the stencil kernel's last work-group
    int opt_bin_pack_limit = 64;     // bits available to the per-bin packing (lower only to test the fallback)
    int opt_bin_tile_rows = 0;       // 0: chosen per slice so that the bins fill the CUs
    int n_cus = 0;
    bool use_binned = false;         // decided per slice in bf_set_cloud
    BinGrid grid;
    // one-kernel iteration (k_fused_pass): the loop of a context that has the GPU to itself
    int opt_fused = 1;               // 0 never, 1 where it is the faster loop (small slices on small images; sparser ones only when
                                     // the context is co-scheduled with others), 2 whenever possible
    int opt_fused_margin = 8;        // D: scaled pixels an event may move before its tile's neighbours must be re-sorted
    int opt_fused_rows = 0;          // rows of an image tile: 0 auto, 32 or 64
    bool fused_ok = false;           // decided per slice in bf_set_cloud
    bool fused_shared = false;       // ... and it is also the loop to take when the context shares the GPU ("co_schedule")
    BinGrid fgrid;                   // its sort grid: keys = (tile, zone)
    uint32_t* d_ftab = nullptr;      // FusedTab per tile
    int ftab_alloc = 0;
    // persistent form of that loop (k_fused_loop, bf_loop.hip): a context ALONE on the GPU keeps the work-groups resident
    bool counted = false;            // in g_live_ctx
    int opt_persist = 1;             // 0 never, 1 for warm-started runs of the one-kernel loop on a context that is not co-scheduled, 2 cold runs too
    unsigned long long *d_xrec = nullptr, *d_xred = nullptr;   // exchange records of the sub-tiles / of the reducers (two parities each)
    int xrec_alloc = 0;              // records per parity d_xrec holds
    float2* d_xscratch[3] = {nullptr, nullptr, nullptr};       // private product arrays of the strips' readers
    uint16_t* d_binid = nullptr;
    uint32_t *d_hist_cnt = nullptr, *d_bin_start = nullptr, *d_cursor = nullptr;
    uint32_t* d_armed = nullptr;
    unsigned long long* d_slabs = nullptr;
    uint16_t* d_cidx = nullptr;      // compact lists: pixel index per entry (same slot count as d_slabs)
    uint32_t* d_chdr = nullptr;      // compact lists: entries per bin
    int stencil_threads = 256;       // work-group size of the stencil kernels for this slice (bf_set_cloud)
    int opt_stencil_threads = 0;     // 0: 256
    int fmt = 0;                     // what this slice's scatter hands to the stencil: 0 dense slabs, 1 merged lists, 2 event lists (bf_set_cloud)
    int opt_bin_ev = 0;              // events per scatter thread in flight (0: from the events per bin)
    int opt_bin_compact = 1;         // 0 never, 1 when the image is sparse (decided per iteration on the device), 2 always
    // interior + margin format of a dense slice (fmt 3, bf_binned.hip: flush_split): 0 never, 1 when it is the faster one, 2 always
    int opt_bin_split = 1;
    unsigned long long* d_mplane[2] = {nullptr, nullptr};   // margin planes (cap_px words each), double buffered like d_plane
    uint32_t* d_mlist = nullptr;     // per bin: the pixels of the margin plane it added to in its last executed launch
    uint32_t* d_mcount = nullptr;    // per bin: entries of that list
    size_t mlist_alloc = 0;
    int mcount_alloc = 0;
    int m_nbins = 0, m_cap = 0;      // geometry the lists were written with
    int m_dirty_plane = -1;          // the margin plane the lists describe (-1: both planes are clean, the lists empty)
    bool m_unknown = false;          // a run did not complete: clear everything before the next use
    uint32_t* d_ovf_bits[2] = {nullptr, nullptr};   // per plane buffer: one bit per image pixel an overflow event touched (tile-binned loop)
    size_t ovf_bits_words = 0;
    int ovf_pitch = 0;               // words per image row: ceil(C / 32) + 3 (one spare word left, two right: the stencil tile's window)
    int bins_alloc = 0;
    size_t slabs_alloc = 0;
    bool bin_setup_done = false;
    // per-tile optimizers (bf_run_tiles)
    uint32_t *d_tile_hist = nullptr, *d_tile_start = nullptr, *d_tile_cursor = nullptr;
    DevState* d_tile_states = nullptr;
    int tiles_alloc = 0;
    int32_t *d_in_x = nullptr, *d_in_y = nullptr, *d_in_t = nullptr;
    // streaming: a second staging slot, a copy stream and one event per slot
    int32_t* d_in2[3] = {nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done[2] = {nullptr, nullptr};
    hipEvent_t staged[2] = {nullptr, nullptr};   // the staging kernels that read a slot have run (compute stream)
    bool staged_valid[2] = {false, false};
    long long pending_n[2] = {0, 0};
    bool pending_ts64[2] = {false, false};       // slot holds absolute 64-bit timestamps (ring hand-off)
    bool pending_addr16[2] = {false, false};     // ... and 16-bit addresses in d_in16 (bf_upload_ring16_async)
    bool pending_noise[2] = {false, false};      // ... and Event::noise flags in d_in_noise
    uint16_t* d_in16[2] = {nullptr, nullptr};    // row[cap_events] then col[cap_events]
    uint8_t* d_in_noise[2] = {nullptr, nullptr};
    unsigned long long pending_t0[2] = {0, 0};
    unsigned long long* d_in_ts[2] = {nullptr, nullptr};
    int pend_head = 0, pend_count = 0;   // FIFO of pending async uploads (slot = index & 1)
    double2 *d_nxny = nullptr, *d_uv = nullptr;
    unsigned long long* d_plane[2] = {nullptr, nullptr};
    uint32_t* d_cplane[2] = {nullptr, nullptr};
    float *d_time = nullptr, *d_gx = nullptr, *d_gy = nullptr, *d_img = nullptr;
    uint32_t* d_count = nullptr;
    MomentAcc* d_acc = nullptr;      // 3 x kAccGroups exact moment accumulators (two-kernel loop: parity of the iteration in the
                                     // first two; one-kernel loop: launch number mod 3) + one line whose first word is `lost`
    bool acc_dirty = false;          // a head-update loop leaves its last iteration's sums behind: whoever uses the
                                     // accumulators next without a loop_init of its own (a ticket-mode stencil) clears them
    uint32_t* d_ovf = nullptr;       // tile-binned loop: overflow events of iteration j in slot j % 3
    unsigned int* d_ticket = nullptr;
    unsigned long long* d_tl = nullptr;   // debug timeline (BF_TIMELINE=<file>, `make tl` build)
    const char* tl_path = nullptr;
    DevState* d_state = nullptr;     // 2 buffers: the tile-binned loop ping-pongs, everything else uses [0]
    SliceStats* d_stats = nullptr;
    bf_trace_rec* d_trace = nullptr;
    int trace_alloc = 0;
    int trace_valid = 0;

    // contrast-score optimiser (bf_local.hip)
    uint32_t* d_lplane[2] = {nullptr, nullptr};   // point planes, double buffered
    unsigned long long* d_lscore = nullptr;       // non-zero sum / count of the blurred image
    uint8_t* d_limg = nullptr;                    // project_img
    void* d_col_planes = nullptr;                 // colour time image: sum cos, sum sin (i64), count (u32) point planes
    uint8_t* d_col_img = nullptr;                    // ... and its B, G, R bytes
    unsigned long long* h_lscore = nullptr;       // pinned
    bf_local_window lwin;
    bool have_lwin = false;
    int lcur = 0;
    SliceStats stats;                // folded k_prepare statistics of the uploaded slice
    bool stats_valid = false;

    DevState* h_state = nullptr;     // pinned, D2H target only: 2 slots (pipelined polling)
    hipEvent_t poll_ev[2] = {nullptr, nullptr};
    bool opt_blocking_poll = true;
    double opt_watchdog_s = 40.0;    // a cold run whose device iteration counter stands still this long is declared hung
    SliceStats* h_stats = nullptr;   // pinned, D2H target only

    DevState hst;                    // authoritative host mirror outside bf_run
    bf_window win;
    bool uploaded = false, have_window = false;
    bool has_noise = false, all_noise = false;
    bool packed = true;
    bool force_split = false;
    bool degenerate = false;         // window with R <= 0 or C <= 0 (empty slice)
    bool pending_warp = false;       // bf_set_model's warp not applied yet
    bool n_valid = false;            // d_nxny holds the n of the last warp
    uint32_t run_counter = 0;
    int warm_iters_hint = 6;         // iterations the previous warm-started run needed
    bool p_clean = false;            // p is all zero (Event::reset state): set by the upload, cleared by any warp
    bool out_sorted = false;         // d_nxny (and d_uv) are in slot order: un-permute with set[cs].perm before reading back
    double2* d_out_tmp = nullptr;    // second buffer for that un-permutation
    bool uv_valid = false;           // d_uv holds compute_uv of that n (fused into bf_run's final warp)
    int cur = 0;                     // plane buffer that is guaranteed all-zero
    bool planes_unknown = true;      // both buffers must be cleared before use
    int last_R = 0, last_C = 0;

    int prof_mode = 0;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    bf_profile prof;

    char err[512];
    std::mutex err_mu;   // fail() may be called from the uploading thread and the solving thread at once (see bf_accel.h: threading)
};

namespace {

// Contexts alive per device in this process.  The persistent loop kernel needs every one of its work-groups resident at
// once; two such kernels from two contexts could each hold part of the CUs and wait for the rest, so a context takes it
// only while it is the one context on its device (other contexts: "co_schedule" or not, they would share the CUs).
std::atomic<int> g_live_ctx[64];

int fail(bf_ctx* c, int code, const char* fmt, ...) {
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        std::lock_guard<std::mutex> g(c->err_mu);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIP_TRY(c, expr)                                                                      \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail((c), BF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

hipEvent_t get_event(bf_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// Brackets one kernel launch with events when profiling is on.
struct ProfScope {
    bf_ctx* c;
    ProfRec r;
    bool on;
    ProfScope(bf_ctx* c_, int cat, long long nev = 0) : c(c_), on(c_->prof_mode == 1) {
        if (!on) return;
        r.a = get_event(c);
        r.b = get_event(c);
        r.cat = cat;
        r.nev = nev;
        (void)hipEventRecord(r.a, c->stream);
        // the loop kernels' launchers pick these up and time the kernel itself (bf_kernels.h: LaunchTimer)
        LaunchTimer& t = launch_timer();
        t.start = r.a; t.stop = r.b; t.consumed = false;
    }
    ~ProfScope() {
        if (!on) return;
        LaunchTimer& t = launch_timer();
        if (!t.consumed) (void)hipEventRecord(r.b, c->stream);
        t.start = t.stop = nullptr;
        t.consumed = false;
        c->prof_pending.push_back(r);
    }
};

int prof_fold(bf_ctx* c) {
    if (c->prof_pending.empty()) return BF_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& r : c->prof_pending) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        switch (r.cat) {
            case 0: c->prof.warp_scatter_ms += ms; c->prof.warp_scatter_launches++;
                    c->prof.warp_scatter_events += (uint64_t)r.nev; break;
            case 1: c->prof.stencil_ms += ms; c->prof.stencil_launches++; break;
            case 2: c->prof.update_ms += ms; c->prof.update_launches++; break;
            default: c->prof.other_ms += ms; c->prof.other_launches++; break;
        }
        c->ev_pool.push_back(r.a);
        c->ev_pool.push_back(r.b);
    }
    c->prof_pending.clear();
    return BF_OK;
}

int bit_length(unsigned long long v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

WarpParams identity_warp() {
    WarpParams w;
    w.dnx = w.dny = w.cx = w.cy = w.div = 0.0;
    w.c = 1.0;
    w.s = 0.0;
    return w;
}

EvSets ev_sets(const bf_ctx* c) {
    EvSets e;
    for (int i = 0; i < 2; ++i) {
        e.s[i].xy = c->set[i].xy; e.s[i].t = c->set[i].t; e.s[i].p = c->set[i].p; e.s[i].perm = c->set[i].perm;
        e.s[i].p2 = c->set[i].p2;
    }
    return e;
}

WarpScatterArgs ws_args(bf_ctx* c, int buf, int check_done) {
    WarpScatterArgs a;
    const bf_ctx::EvSet& e = c->set[c->cs];
    a.xy = e.xy; a.t = e.t; a.p = e.p;
    a.noise = c->has_noise ? c->d_noise : nullptr;
    a.nxny = c->d_nxny;
    a.uv = nullptr;
    a.perm = c->has_perm ? e.perm : nullptr;
    a.plane = c->d_plane[buf];
    a.cplane = c->d_cplane[buf];
    a.st = c->d_state;
    a.n = c->n;
    a.check_done = check_done;
    a.packed = c->packed;
    a.sets = ev_sets(c);
    a.pick_set = 0;
    a.sorted_out = 0;
    return a;
}

StencilArgs st_args(bf_ctx* c, int buf, int check_done) {
    StencilArgs a;
    memset(&a, 0, sizeof(a));
    a.st = c->d_state;
    a.check_done = check_done;
    a.R = c->win.scale_img_x; a.C = c->win.scale_img_y;
    a.scale = c->win.scale;
    a.tbits = c->hst.hot.tbits;
    a.tmin = c->hst.hot.tmin;
    a.plane = c->d_plane[buf];
    a.cplane = c->d_cplane[buf];
    a.zero_plane = c->d_plane[buf ^ 1];
    a.zero_cplane = (c->packed && !c->use_binned) ? nullptr : c->d_cplane[buf ^ 1];
    a.slabs = c->d_slabs;
    a.cidx = c->d_cidx; a.chdr = c->d_chdr;
    a.compact = c->fmt;
    a.m_cur = c->d_mplane[buf];
    a.threads = c->stencil_threads;
    a.g = c->grid;
    a.ovf_cur = a.ovf_prev = c->d_ovf;   // (the tile-binned loop sets the three counters per launch)
    a.cur = buf;
    return a;
}

// which k_stencil instantiation reads the scatter result of the current mode
int stencil_src(const bf_ctx* c, bool binned_pass) { return binned_pass ? 3 : (c->packed ? 0 : 1); }

int ensure_cplanes(bf_ctx* c) {
    if (c->d_cplane[0]) return BF_OK;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(c, hipMalloc(&c->d_cplane[i], c->cap_px * sizeof(uint32_t)));
        HIP_TRY(c, hipMemsetAsync(c->d_cplane[i], 0, c->cap_px * sizeof(uint32_t), c->stream));
    }
    return BF_OK;
}

int clear_planes(bf_ctx* c) {
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(c, hipMemsetAsync(c->d_plane[i], 0, c->cap_px * sizeof(unsigned long long), c->stream));
        if (c->d_cplane[i])
            HIP_TRY(c, hipMemsetAsync(c->d_cplane[i], 0, c->cap_px * sizeof(uint32_t), c->stream));
        if (c->d_ovf_bits[i])
            HIP_TRY(c, hipMemsetAsync(c->d_ovf_bits[i], 0, c->ovf_bits_words * sizeof(uint32_t), c->stream));
    }
    c->planes_unknown = false;
    c->cur = 0;
    c->hst.hot.ovf_cnt[0] = c->hst.hot.ovf_cnt[1] = 0;
    return BF_OK;
}

// Dirty bitmaps of the overflow planes for an R x C image (a change of R or C clears planes and bitmaps: bf_set_cloud).
int ensure_ovf_bits(bf_ctx* c, int R, int C) {
    const int pitch = (C + 31) / 32 + 3;
    const size_t need = (size_t)R * (size_t)pitch;
    if (need > c->ovf_bits_words) {
        for (int i = 0; i < 2; ++i) {
            if (c->d_ovf_bits[i]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(c->d_ovf_bits[i])); }
            c->d_ovf_bits[i] = nullptr;
            HIP_TRY(c, hipMalloc(&c->d_ovf_bits[i], need * sizeof(uint32_t)));
        }
        c->ovf_bits_words = need;
        c->planes_unknown = true;   // (fresh bitmaps: cleared with the planes below)
    }
    if (pitch != c->ovf_pitch) c->planes_unknown = true;   // (bits set under another row pitch mean other pixels)
    c->ovf_pitch = pitch;
    return BF_OK;
}

int ensure_bin_buffers(bf_ctx* c, const BinGrid& g) {
    if (!c->bin_setup_done) {
        if (bin_kernel_setup() != 0) return fail(c, BF_ERR_HIP, "cannot raise the dynamic LDS limit");
        c->bin_setup_done = true;
    }
    if (!c->d_binid) {
        HIP_TRY(c, hipMalloc(&c->d_binid, (size_t)c->cap_events * sizeof(uint16_t)));
        HIP_TRY(c, hipMalloc(&c->d_armed, 64));
        HIP_TRY(c, hipMemsetAsync(c->d_armed, 0, 64, c->stream));
    }
    // (the second event set and the permutations are shared with bf_run_tiles, which may have allocated them -- and may have
    // left the slice's events IN the second set: replacing the buffers here lost them)
    for (int i = 0; i < 2; ++i)
        if (!c->set[i].perm) HIP_TRY(c, hipMalloc(&c->set[i].perm, (size_t)c->cap_events * sizeof(uint32_t)));
    if (!c->set[1].xy) {
        HIP_TRY(c, hipMalloc(&c->set[1].xy, (size_t)c->cap_events * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->set[1].t, (size_t)c->cap_events * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->set[1].p, (size_t)c->cap_events * sizeof(float2)));
    }
    if (g.nbins > c->bins_alloc) {
        void* old[] = {c->d_hist_cnt, c->d_bin_start, c->d_cursor, c->d_chdr};
        for (void* o : old) if (o) HIP_TRY(c, hipFree(o));
        c->d_hist_cnt = nullptr; c->d_bin_start = nullptr; c->d_cursor = nullptr; c->d_chdr = nullptr;
        const size_t nb = (size_t)g.nbins + 1;
        HIP_TRY(c, hipMalloc(&c->d_hist_cnt, kHistCopies * nb * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_bin_start, nb * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_cursor, nb * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_chdr, nb * 200 * sizeof(uint32_t)));   // compact lists: LR + 1 <= 193 row offsets per bin
        HIP_TRY(c, hipMemsetAsync(c->d_hist_cnt, 0, kHistCopies * nb * sizeof(uint32_t), c->stream));
        c->bins_alloc = g.nbins;
    }
    const size_t need = (size_t)g.nbins * (size_t)g.LR * (size_t)g.L;
    if (need > c->slabs_alloc) {
        if (c->d_slabs) HIP_TRY(c, hipFree(c->d_slabs));
        if (c->d_cidx) HIP_TRY(c, hipFree(c->d_cidx));
        c->d_slabs = nullptr; c->d_cidx = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_slabs, need * sizeof(unsigned long long)));
        HIP_TRY(c, hipMalloc(&c->d_cidx, need * sizeof(uint16_t)));
        c->slabs_alloc = need;
    }
    return BF_OK;
}


// Margin planes and per-bin lists of the interior + margin format.  Iteration j of a run adds to margin plane b0 ^ (j & 1) and
// clears, bin by bin, what the lists say the previous executed launch left in the other one (flush_split); the host keeps
// track of which plane the lists describe.  The lists name pixels by their linear index, so what an earlier bin grid left
// behind is cleared with the earlier grid's list layout before the buffers change hands.
int margin_reset(bf_ctx* c) {
    if (c->m_unknown) {
        for (int i = 0; i < 2; ++i)
            if (c->d_mplane[i]) HIP_TRY(c, hipMemsetAsync(c->d_mplane[i], 0, c->cap_px * sizeof(unsigned long long), c->stream));
        if (c->d_mcount) HIP_TRY(c, hipMemsetAsync(c->d_mcount, 0, (size_t)c->mcount_alloc * sizeof(uint32_t), c->stream));
        c->m_unknown = false;
    } else if (c->m_dirty_plane >= 0) {
        launch_margin_clean(c->d_mplane[c->m_dirty_plane], c->d_mlist, c->d_mcount, c->m_nbins, c->m_cap, c->stream);
        HIP_TRY(c, hipGetLastError());
    }
    c->m_dirty_plane = -1;
    return BF_OK;
}
int ensure_margin_buffers(bf_ctx* c, const BinGrid& g) {
    const int mcap = g.LR * g.L - g.TSR * g.TS;
    if (c->m_unknown || g.nbins != c->m_nbins || mcap != c->m_cap) {
        int rc = margin_reset(c);
        if (rc != BF_OK) return rc;
    }
    for (int i = 0; i < 2; ++i)
        if (!c->d_mplane[i]) {
            HIP_TRY(c, hipMalloc(&c->d_mplane[i], c->cap_px * sizeof(unsigned long long)));
            HIP_TRY(c, hipMemsetAsync(c->d_mplane[i], 0, c->cap_px * sizeof(unsigned long long), c->stream));
        }
    if (g.nbins > c->mcount_alloc) {
        if (c->d_mcount) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(c->d_mcount)); }
        c->d_mcount = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_mcount, (size_t)g.nbins * sizeof(uint32_t)));
        HIP_TRY(c, hipMemsetAsync(c->d_mcount, 0, (size_t)g.nbins * sizeof(uint32_t), c->stream));
        c->mcount_alloc = g.nbins;
    }
    const size_t need = (size_t)g.nbins * (size_t)mcap;
    if (need > c->mlist_alloc) {
        if (c->d_mlist) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(c->d_mlist)); }
        c->d_mlist = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_mlist, need * sizeof(uint32_t)));
        c->mlist_alloc = need;
    }
    c->m_nbins = g.nbins;
    c->m_cap = mcap;
    return BF_OK;
}

// Device-conditional counting sort of the live events by the image tile of their current
// target (runs only when hot.need_rebin is set); no host synchronisation.
uint32_t* lost_flag(const bf_ctx* c) { return reinterpret_cast<uint32_t*>(c->d_acc + 3 * kAccGroups); }

int enqueue_rebin(bf_ctx* c, DevState* st, bool has_perm_at_start, const WarpParams* prewarp = nullptr, bool fused = false, int launch_no = 0) {
    ProfScope ps(c, 3);
    launch_rebin(ev_sets(c), has_perm_at_start ? 1 : 0, c->n, st, fused ? c->fgrid : c->grid, c->d_binid, c->d_hist_cnt,
                 c->d_bin_start, c->d_cursor, c->d_armed, prewarp, c->opt_bin_pack_limit, c->stream,
                 fused ? c->d_ftab : nullptr, fused ? lost_flag(c) + (launch_no + 2) % 3 : nullptr);   // (the last pass's word)
    HIP_TRY(c, hipGetLastError());
    return BF_OK;
}

// Apply the warp bf_set_model left pending (optimizer_rolling.h:294-298).
int flush_pending(bf_ctx* c) {
    if (!c->pending_warp) return BF_OK;
    launch_set_state(c->d_state, c->hst, c->stream);
    {
        ProfScope ps(c, 3);
        launch_warp_scatter(ws_args(c, c->cur, 0), true, false, true, c->stream);
    }
    c->pending_warp = false;
    c->p_clean = false;
    c->n_valid = true;
    c->uv_valid = false;
    c->out_sorted = false;
    HIP_TRY(c, hipGetLastError());
    return BF_OK;
}

int d2h_state(bf_ctx* c) {
    HIP_TRY(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(DevState), hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

// Waits for an event without occupying a host core: query, sleep ~20 us (+ the kernel's timer slack), repeat.
// hipEventSynchronize spins here whatever the event's flags say (measured: one full core per waiting thread).
int wait_event_sleeping(bf_ctx* c, hipEvent_t ev) {
    bool waited = false;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) HIP_TRY(c, e);
        waited = true;
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
    }
    if (waited) (void)hipGetLastError();   // "not ready" is recorded as the thread's last error: it is not one
    return BF_OK;
}

// Folds the per-work-group min / max / sum records k_prepare wrote for the uploaded slice (one
// device-to-host copy per slice, cached).
int fold_stats(bf_ctx* c) {
    if (c->stats_valid) return BF_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // (the records are in pinned host memory once k_prepare has completed)
    SliceStats s = c->h_stats[0];
    for (int k = 1; k < kPrepBlocks; ++k) {
        const SliceStats& q = c->h_stats[k];
        if (q.xmin < s.xmin) s.xmin = q.xmin;
        if (q.xmax > s.xmax) s.xmax = q.xmax;
        if (q.ymin < s.ymin) s.ymin = q.ymin;
        if (q.ymax > s.ymax) s.ymax = q.ymax;
        if (q.tmin < s.tmin) s.tmin = q.tmin;
        if (q.tmax > s.tmax) s.tmax = q.tmax;
        s.tsum += q.tsum;
    }
    c->stats = s;
    c->stats_valid = true;
    return BF_OK;
}

int after_upload(bf_ctx* c, long long n) {
    c->p_clean = true;   // k_prepare wrote p = 0
    c->stats_valid = false;
    c->have_lwin = false;
    c->n = n;
    c->uploaded = true;
    c->have_window = false;
    c->all_noise = false;
    c->pending_warp = false;
    c->n_valid = false;
    c->uv_valid = false;
    c->out_sorted = false;
    return BF_OK;
}

}  // namespace

// copy stream, its events and the second staging slot of the asynchronous uploads (created on first use)
static int streaming_setup(bf_ctx* c) {
    if (c->copy_stream) return BF_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->copy_done[i], hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->staged[i], hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) HIP_TRY(c, hipMalloc(&c->d_in2[i], (size_t)c->cap_events * sizeof(int32_t)));
    return BF_OK;
}

// The slice hand-off of DVS_flow::recompute (dvs_flow.h:185-216) for a structure-of-arrays ring in pinned
// memory: up to two contiguous pieces per array, no repacking on the host.  ADDR is int32_t (bf_upload_ring_async) or
// uint16_t (bf_upload_ring16_async: the addresses travel as 16-bit values and are widened by the staging kernel).
template <class ADDR>
static int upload_ring(bf_ctx* c, const ADDR* ring_x, const ADDR* ring_y, const uint64_t* ring_ts, const uint8_t* ring_noise,
                       int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    if (!c) return BF_ERR_ARG;
    if (n <= 0 || cap <= 0 || first < 0 || first >= cap || n > cap || !ring_x || !ring_y || !ring_ts)
        return fail(c, BF_ERR_ARG, "bad ring slice (cap %lld, first %lld, n %lld)", (long long)cap, (long long)first, (long long)n);
    if (n > c->cap_events) return fail(c, BF_ERR_CAPACITY, "n=%lld exceeds capacity %lld", (long long)n, c->cap_events);
    if (c->pend_count >= 2) return fail(c, BF_ERR_STATE, "two uploads are already pending");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rc = streaming_setup(c);
        if (rc != BF_OK) return rc;
    }
    const int slot = (c->pend_head + c->pend_count) & 1;
    if (c->staged_valid[slot]) HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->staged[slot], 0));
    if (!c->d_in_ts[slot]) HIP_TRY(c, hipMalloc(&c->d_in_ts[slot], (size_t)c->cap_events * sizeof(unsigned long long)));
    const bool narrow = sizeof(ADDR) == 2;
    if (narrow && !c->d_in16[slot]) HIP_TRY(c, hipMalloc(&c->d_in16[slot], (size_t)c->cap_events * 2 * sizeof(uint16_t)));
    if (ring_noise && !c->d_in_noise[slot]) HIP_TRY(c, hipMalloc(&c->d_in_noise[slot], (size_t)c->cap_events));
    // destinations of the two address columns: the slot's int32 staging arrays, or (16-bit form) two halves of d_in16
    ADDR* dx = narrow ? reinterpret_cast<ADDR*>(c->d_in16[slot]) : reinterpret_cast<ADDR*>(slot ? c->d_in2[0] : c->d_in_x);
    ADDR* dy = narrow ? reinterpret_cast<ADDR*>(c->d_in16[slot] + c->cap_events) : reinterpret_cast<ADDR*>(slot ? c->d_in2[1] : c->d_in_y);
    const int64_t n0 = (first + n <= cap) ? n : cap - first, n1 = n - n0;   // [first, first + n0) then [0, n1)
    HIP_TRY(c, hipMemcpyAsync(dx, ring_x + first, (size_t)n0 * sizeof(ADDR), hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(dy, ring_y + first, (size_t)n0 * sizeof(ADDR), hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_in_ts[slot], ring_ts + first, (size_t)n0 * 8, hipMemcpyHostToDevice, c->copy_stream));
    if (ring_noise) HIP_TRY(c, hipMemcpyAsync(c->d_in_noise[slot], ring_noise + first, (size_t)n0, hipMemcpyHostToDevice, c->copy_stream));
    if (n1 > 0) {
        HIP_TRY(c, hipMemcpyAsync(dx + n0, ring_x, (size_t)n1 * sizeof(ADDR), hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipMemcpyAsync(dy + n0, ring_y, (size_t)n1 * sizeof(ADDR), hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_in_ts[slot] + n0, ring_ts, (size_t)n1 * 8, hipMemcpyHostToDevice, c->copy_stream));
        if (ring_noise) HIP_TRY(c, hipMemcpyAsync(c->d_in_noise[slot] + n0, ring_noise, (size_t)n1, hipMemcpyHostToDevice, c->copy_stream));
    }
    HIP_TRY(c, hipEventRecord(c->copy_done[slot], c->copy_stream));
    c->pending_n[slot] = n;
    c->pending_ts64[slot] = true;
    c->pending_addr16[slot] = narrow;
    c->pending_noise[slot] = ring_noise != nullptr;
    c->pending_t0[slot] = t0;
    c->pend_count++;
    return BF_OK;
}

extern "C" {

const char* bf_version(void) { return "bf_accel gfx950 r1"; }

int bf_device_count(int32_t* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    return (e == hipSuccess && n > 0) ? BF_OK : BF_ERR_NODEVICE;
}

int bf_abi_struct_sizes(int32_t* out, int32_t n) {
    const int32_t sz[8] = {(int32_t)sizeof(bf_model),     (int32_t)sizeof(bf_window),
                           (int32_t)sizeof(bf_run_opts),  (int32_t)sizeof(bf_run_info),
                           (int32_t)sizeof(bf_trace_rec), (int32_t)sizeof(bf_profile),
                           (int32_t)sizeof(bf_local_window), (int32_t)sizeof(bf_local_state)};
    for (int i = 0; out && i < n && i < 8; ++i) out[i] = sz[i];
    return 8;
}

void bf_run_opts_default(bf_run_opts* o) {
    if (!o) return;
    o->max_iter = -1;       // OptimizerRolling(): max_itercount(-1)
    o->min_events = 1000;   // optimizer_rolling.h:57
    o->res_x = 180;         // common.h:39
    o->res_y = 240;         // common.h:40
    o->hard_iter_cap = 100000;
    o->poll_interval = 8;
    o->trace_cap = 0;
    o->want_uv = 0;
}

int bf_create(int32_t device, int64_t max_events, int32_t max_rows, int32_t max_cols, void* hip_stream,
              bf_ctx** out) {
    if (!out || max_events <= 0 || max_rows <= 0 || max_cols <= 0) return BF_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BF_ERR_NODEVICE;
    if (device < 0 || device >= ndev) return BF_ERR_ARG;
    bf_ctx* c = new (std::nothrow) bf_ctx();
    if (!c) return BF_ERR_HIP;
    c->err[0] = 0;
    memset(&c->hst, 0, sizeof(c->hst));
    memset(&c->win, 0, sizeof(c->win));
    memset(&c->prof, 0, sizeof(c->prof));
    c->device = device;
    int rc = [&]() -> int {
        HIP_TRY(c, hipSetDevice(device));
        (void)hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device);
        if (hip_stream) {
            c->stream = (hipStream_t)hip_stream;
        } else {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->own_stream = true;
        }
        const long long gran = (long long)kThreads * kEvPerThread;
        c->cap_events = ((long long)max_events + gran - 1) / gran * gran;
        c->cap_px = (size_t)max_rows * (size_t)max_cols;
        int gx, gy;
        stencil_grid(max_rows, max_cols, &gx, &gy);
        // a window with the same pixel count but another aspect ratio can need more tiles
        c->cap_blocks = gx * gy * 2 + 64;
        const size_t ne = (size_t)c->cap_events;
        HIP_TRY(c, hipMalloc(&c->set[0].xy, ne * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->set[0].t, ne * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->set[0].p, ne * sizeof(float2)));
        HIP_TRY(c, hipMalloc(&c->d_noise, ne));
        HIP_TRY(c, hipMalloc(&c->d_in_x, ne * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->d_in_y, ne * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->d_in_t, ne * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->d_nxny, ne * sizeof(double2)));
        HIP_TRY(c, hipMalloc(&c->d_uv, ne * sizeof(double2)));
        for (int i = 0; i < 2; ++i)
            HIP_TRY(c, hipMalloc(&c->d_plane[i], c->cap_px * sizeof(unsigned long long)));
        HIP_TRY(c, hipMalloc(&c->d_time, c->cap_px * sizeof(float)));
        HIP_TRY(c, hipMalloc(&c->d_gx, c->cap_px * sizeof(float)));
        HIP_TRY(c, hipMalloc(&c->d_gy, c->cap_px * sizeof(float)));
        HIP_TRY(c, hipMalloc(&c->d_img, c->cap_px * sizeof(float)));
        HIP_TRY(c, hipMalloc(&c->d_count, c->cap_px * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_acc, (3 * kAccGroups + 1) * sizeof(MomentAcc)));
        HIP_TRY(c, hipMemsetAsync(c->d_acc, 0, (3 * kAccGroups + 1) * sizeof(MomentAcc), c->stream));
        HIP_TRY(c, hipMalloc(&c->d_ovf, 3 * kOvfSlotWords * sizeof(uint32_t)));   // (three slots of 17 lines: bf_device_fns.h)
        HIP_TRY(c, hipMemsetAsync(c->d_ovf, 0, 3 * kOvfSlotWords * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMalloc(&c->d_state, 2 * sizeof(DevState)));
        HIP_TRY(c, hipMalloc(&c->d_ticket, 16 * 64 * sizeof(unsigned int)));   // 1 + 32 counters, 64 B apart
        HIP_TRY(c, hipMemsetAsync(c->d_ticket, 0, 16 * 64 * sizeof(unsigned int), c->stream));

        HIP_TRY(c, hipHostMalloc(&c->h_state, 2 * sizeof(DevState), hipHostMallocDefault));
        for (int i = 0; i < 2; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->poll_ev[i], hipEventDisableTiming));
        HIP_TRY(c, hipHostMalloc(&c->h_stats, kPrepBlocks * sizeof(SliceStats), hipHostMallocDefault));
        c->d_stats = c->h_stats;   // k_prepare writes its per-work-group records straight into pinned host memory: no copy command
        HIP_TRY(c, hipMemsetAsync(c->d_state, 0, 2 * sizeof(DevState), c->stream));
        c->tl_path = getenv("BF_TIMELINE");
        if (c->tl_path && *c->tl_path) {
            HIP_TRY(c, hipMalloc(&c->d_tl, 3 * 64 * 2 * 16 * sizeof(unsigned long long)));
            HIP_TRY(c, hipMemsetAsync(c->d_tl, 0, 3 * 64 * 2 * 16 * sizeof(unsigned long long), c->stream));
        }
        int r = clear_planes(c);
        if (r != BF_OK) return r;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return BF_OK;
    }();
    if (rc != BF_OK) {
        fprintf(stderr, "bf_create: %s\n", c->err);
        bf_destroy(c);
        return rc;
    }
    // BF_ACCEL_OPTIONS="key=value,key=value": bf_set_option calls for every context of the process -- for A/B runs through a
    // host that has no flag for an option (the command line).  A bad entry fails the creation loudly.
    if (const char* env = getenv("BF_ACCEL_OPTIONS")) {
        std::string all(env);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(pos, end - pos);
            pos = end + 1;
            if (item.empty()) continue;
            const size_t eq = item.find('=');
            // (strtoll with an end pointer: "fused=abc" or "fused=" must not quietly become 0)
            bool syntax = eq == std::string::npos || eq + 1 >= item.size();
            long long val = 0;
            if (!syntax) {
                char* endp = nullptr;
                errno = 0;
                val = strtoll(item.c_str() + eq + 1, &endp, 10);
                syntax = errno != 0 || endp == item.c_str() + eq + 1 || *endp != '\0';
            }
            const int orc = syntax ? BF_ERR_ARG : bf_set_option(c, item.substr(0, eq).c_str(), val);
            if (orc != BF_OK) {
                fprintf(stderr, "bf_create: BF_ACCEL_OPTIONS entry '%s': %s\n", item.c_str(), syntax ? "expected key=<integer>" : c->err);
                bf_destroy(c);
                return orc;
            }
        }
    }
    g_live_ctx[device & 63].fetch_add(1);
    c->counted = true;
    *out = c;
    return BF_OK;
}

void bf_destroy(bf_ctx* c) {
    if (!c) return;
    if (c->counted) g_live_ctx[c->device & 63].fetch_sub(1);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_tl) {   // debug timeline dump: launch group slot ticks(100 MHz)
        std::vector<unsigned long long> tl(3 * 64 * 2 * 16);   // [kernel][launch][group][slot]; third block: per-work-group stamps of K1 launch 20
        (void)hipMemcpy(tl.data(), c->d_tl, tl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE* f = fopen(c->tl_path, "w")) {
            for (size_t i = 0; i < tl.size(); ++i)
                if (tl[i]) fprintf(f, "%zu %zu %zu %zu %llu\n", i / 2048, (i / 32) % 64, (i / 16) % 2, i % 16, tl[i]);
            fclose(f);
        }
        (void)hipFree(c->d_tl);
    }
    for (auto& r : c->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (c->poll_ev[i]) (void)hipEventDestroy(c->poll_ev[i]);
    for (int i = 0; i < 2; ++i) if (c->copy_done[i]) (void)hipEventDestroy(c->copy_done[i]);
    for (int i = 0; i < 2; ++i) if (c->staged[i]) (void)hipEventDestroy(c->staged[i]);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (int i = 0; i < 3; ++i) if (c->d_in2[i]) (void)hipFree(c->d_in2[i]);
    for (int i = 0; i < 2; ++i) if (c->d_in_ts[i]) (void)hipFree(c->d_in_ts[i]);
    for (int i = 0; i < 2; ++i) if (c->d_in16[i]) (void)hipFree(c->d_in16[i]);
    for (int i = 0; i < 2; ++i) if (c->d_in_noise[i]) (void)hipFree(c->d_in_noise[i]);
    void* bufs[] = {c->d_xrec, c->d_xred, c->d_xscratch[0], c->d_xscratch[1], c->d_xscratch[2], c->set[0].p2, c->set[1].p2, c->d_ftab, c->set[0].xy, c->set[0].t, c->set[0].p, c->set[0].perm, c->set[1].xy, c->set[1].t,
                    c->set[1].p, c->set[1].perm, c->d_binid, c->d_hist_cnt, c->d_bin_start,
                    c->d_cursor, c->d_slabs, c->d_cidx, c->d_chdr, c->d_mplane[0], c->d_mplane[1], c->d_mlist, c->d_mcount, c->d_ovf_bits[0], c->d_ovf_bits[1], c->d_armed, c->d_acc, c->d_ovf, c->d_out_tmp, c->d_lplane[0], c->d_lplane[1], c->d_lscore, c->d_limg, c->d_col_planes, c->d_col_img, c->d_tile_hist, c->d_tile_start, c->d_tile_cursor, c->d_tile_states,
                    c->d_noise, c->d_in_x, c->d_in_y, c->d_in_t, c->d_nxny,
                    c->d_uv, c->d_plane[0], c->d_plane[1], c->d_cplane[0], c->d_cplane[1], c->d_time,
                    c->d_gx, c->d_gy, c->d_img, c->d_count, c->d_ticket, c->d_state,
                    c->d_trace};   // (d_stats is h_stats: freed below)
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->h_lscore) (void)hipHostFree(c->h_lscore);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* bf_last_error(const bf_ctx* c) {
    if (!c) return "null ctx";
    // a copy private to the calling thread: another thread of the context's owner (the uploading one) may fail meanwhile
    static thread_local char text[sizeof(c->err)];
    std::lock_guard<std::mutex> g(const_cast<bf_ctx*>(c)->err_mu);
    memcpy(text, c->err, sizeof(text));
    text[sizeof(text) - 1] = 0;
    return text;
}

int bf_get_stat(bf_ctx* c, const char* key, int64_t* value) {
    if (!c || !key || !value) return BF_ERR_ARG;
    if (!strcmp(key, "scatter_format")) {
        *value = c->use_binned ? c->fmt : -1;
        return BF_OK;
    }
    if (!strcmp(key, "one_kernel")) {
        *value = (c->fused_ok && (!c->opt_co_schedule || c->fused_shared)) ? 1 : 0;
        return BF_OK;
    }
    if (!strcmp(key, "persistent")) {   // would bf_run, called now, take the persistent loop kernel?
        *value = (c->fused_ok && !c->opt_co_schedule && (c->opt_persist == 2 || (c->opt_persist == 1 && c->pending_warp)) &&
                  g_live_ctx[c->device & 63].load() == 1 &&
                  fused_loop_resident(c->win.scale / 2, c->fgrid.TSR, c->n_cus, c->fgrid.nbr * c->fgrid.nbc)) ? 1 : 0;
        return BF_OK;
    }
    return fail(c, BF_ERR_ARG, "unknown statistic '%s'", key);
}

int bf_set_option(bf_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return BF_ERR_ARG;
    if (!strcmp(key, "force_split")) {
        c->force_split = value != 0;
        return BF_OK;
    }
    if (!strcmp(key, "binned")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "binned must be 0, 1 or 2");
        c->opt_binned = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_tile")) {
        if (value != 0 && value != 16 && value != 32 && value != 64 && value != 128)
            return fail(c, BF_ERR_ARG, "bin_tile must be 0 (auto), 16, 32, 64 or 128");
        c->opt_bin_tile = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_pack_limit")) {
        if (value < 1 || value > 64) return fail(c, BF_ERR_ARG, "bin_pack_limit must be in [1, 64]");
        c->opt_bin_pack_limit = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "co_schedule")) {
        c->opt_co_schedule = value != 0;
        return BF_OK;
    }
    if (!strcmp(key, "blocking_poll")) {
        c->opt_blocking_poll = value != 0;
        return BF_OK;
    }
    if (!strcmp(key, "stream_prealloc")) {   // everything the 16-bit ring hand-off allocates on first use, now
        if (!value) return BF_OK;
        HIP_TRY(c, hipSetDevice(c->device));
        const int rc = streaming_setup(c);
        if (rc != BF_OK) return rc;
        for (int slot = 0; slot < 2; ++slot) {
            if (!c->d_in_ts[slot]) HIP_TRY(c, hipMalloc(&c->d_in_ts[slot], (size_t)c->cap_events * sizeof(unsigned long long)));
            if (!c->d_in16[slot]) HIP_TRY(c, hipMalloc(&c->d_in16[slot], (size_t)c->cap_events * 2 * sizeof(uint16_t)));
            if (!c->d_in_noise[slot]) HIP_TRY(c, hipMalloc(&c->d_in_noise[slot], (size_t)c->cap_events));   // (else: first use, possibly in the middle of a solve)
        }
        return BF_OK;
    }
    if (!strcmp(key, "watchdog_ms")) {
        if (value < 1) return fail(c, BF_ERR_ARG, "watchdog_ms must be >= 1");
        c->opt_watchdog_s = (double)value * 1e-3;
        return BF_OK;
    }
    if (!strcmp(key, "bin_tile_rows")) {
        // (>= 32: the stencil kernel relies on a 16-row tile plus its halo crossing at most one bin boundary)
        if (value != 0 && (value < 32 || value > 128 || value % 16))
            return fail(c, BF_ERR_ARG, "bin_tile_rows must be 0 (auto) or a multiple of 16 in [32, 128]");
        c->opt_bin_tile_rows = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "stencil_threads")) {
        if (value != 0 && value != 256 && value != 512) return fail(c, BF_ERR_ARG, "stencil_threads must be 0 (auto), 256 or 512");
        c->opt_stencil_threads = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_ev")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(c, BF_ERR_ARG, "bin_ev must be 0, 1, 2, 4 or 8");
        c->opt_bin_ev = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_compact")) {
        if (value < 0 || value > 3) return fail(c, BF_ERR_ARG, "bin_compact must be 0, 1, 2 or 3");
        c->opt_bin_compact = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_split")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "bin_split must be 0, 1 or 2");
        c->opt_bin_split = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_predict")) {
        c->opt_bin_predict = value != 0;
        return BF_OK;
    }
    if (!strcmp(key, "bin_threads")) {
        if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(c, BF_ERR_ARG, "bin_threads must be 0 (auto), 256, 512 or 1024");
        c->opt_bin_threads = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "fused")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "fused must be 0, 1 (auto) or 2");
        c->opt_fused = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "persist")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "persist must be 0, 1 (auto) or 2");
        c->opt_persist = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "fused_margin")) {
        if (value < 1 || value > 30) return fail(c, BF_ERR_ARG, "fused_margin must be in [1, 30]");
        c->opt_fused_margin = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "fused_rows")) {
        if (value != 0 && value != 32 && value != 64) return fail(c, BF_ERR_ARG, "fused_rows must be 0 (auto), 32 or 64");
        c->opt_fused_rows = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_margin")) {
        if (value < 2 || value > 64 || value % 2) return fail(c, BF_ERR_ARG, "bin_margin must be even, in [2, 64]");
        c->opt_bin_margin = (int)value;
        return BF_OK;
    }
    return fail(c, BF_ERR_ARG, "unknown option '%s'", key);
}

int bf_synchronize(bf_ctx* c) {
    if (!c) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

// ---- slice set-up ---------------------------------------------------------------------

static int stage_common(bf_ctx* c, const int32_t* dx, const int32_t* dy, const int32_t* dt, long long n) {
    const long long gran = (long long)kThreads * kEvPerThread;
    c->n_pad = (n + gran - 1) / gran * gran;
    {
        ProfScope ps(c, 3);
        launch_prepare(dx, dy, dt, c->set[0].xy, c->set[0].t, c->set[0].p, n, c->n_pad, c->d_stats,
                       c->stream);
        c->cs = 0;
        c->has_perm = false;
    }
    HIP_TRY(c, hipGetLastError());
    return after_upload(c, n);
}

int bf_upload_events(bf_ctx* c, const int32_t* fr_x, const int32_t* fr_y, const int32_t* t_ns,
                     const uint8_t* noise, int64_t n) {
    if (!c) return BF_ERR_ARG;
    if (n < 0 || (n > 0 && (!fr_x || !fr_y || !t_ns))) return fail(c, BF_ERR_ARG, "bad event arrays");
    if (n > c->cap_events) return fail(c, BF_ERR_CAPACITY, "n=%lld exceeds capacity %lld", (long long)n, c->cap_events);
    if (c->pend_count > 0)   // (the blocking upload stages through slot 0, which a pending asynchronous upload may own)
        return fail(c, BF_ERR_STATE, "bf_upload_events while %d asynchronous upload(s) are pending", c->pend_count);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->staged_valid[0]) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->staged[0], 0));
    const size_t nb = (size_t)n * sizeof(int32_t);
    if (n > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->d_in_x, fr_x, nb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_in_y, fr_y, nb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_in_t, t_ns, nb, hipMemcpyHostToDevice, c->stream));
    }
    c->has_noise = false;
    if (noise && n > 0) {
        const long long gran = (long long)kThreads * kEvPerThread;
        const long long n_pad = (n + gran - 1) / gran * gran;
        HIP_TRY(c, hipMemsetAsync(c->d_noise, 0, (size_t)n_pad, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_noise, noise, (size_t)n, hipMemcpyHostToDevice, c->stream));
        c->has_noise = true;
    }
    int rc = stage_common(c, c->d_in_x, c->d_in_y, c->d_in_t, n);
    if (rc != BF_OK) return rc;
    // host arrays are only borrowed for the duration of the call
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

int bf_host_alloc(bf_ctx* c, int64_t bytes, void** out) {
    if (!c || !out || bytes <= 0) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable));   // (portable: a slice farm uploads from one ring to several devices)
    return BF_OK;
}

int bf_host_free(bf_ctx* c, void* ptr) {
    if (!c) return BF_ERR_ARG;
    if (ptr) HIP_TRY(c, hipHostFree(ptr));
    return BF_OK;
}

int bf_upload_events_async(bf_ctx* c, const int32_t* fr_x, const int32_t* fr_y, const int32_t* t_ns, int64_t n) {
    if (!c) return BF_ERR_ARG;
    if (n <= 0 || !fr_x || !fr_y || !t_ns) return fail(c, BF_ERR_ARG, "bad event arrays");
    if (n > c->cap_events) return fail(c, BF_ERR_CAPACITY, "n=%lld exceeds capacity %lld", (long long)n, c->cap_events);
    if (c->pend_count >= 2) return fail(c, BF_ERR_STATE, "two uploads are already pending");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rc = streaming_setup(c);
        if (rc != BF_OK) return rc;
    }
    const int slot = (c->pend_head + c->pend_count) & 1;
    // the slot's previous content may still be waiting for its staging kernel on the compute stream
    if (c->staged_valid[slot]) HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->staged[slot], 0));
    int32_t* dx = slot ? c->d_in2[0] : c->d_in_x;
    int32_t* dy = slot ? c->d_in2[1] : c->d_in_y;
    int32_t* dt = slot ? c->d_in2[2] : c->d_in_t;
    const size_t nb = (size_t)n * sizeof(int32_t);
    HIP_TRY(c, hipMemcpyAsync(dx, fr_x, nb, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(dy, fr_y, nb, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(dt, t_ns, nb, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipEventRecord(c->copy_done[slot], c->copy_stream));
    c->pending_n[slot] = n;
    c->pending_ts64[slot] = false;
    c->pend_count++;
    return BF_OK;
}

int bf_upload_ring_async(bf_ctx* c, const int32_t* ring_x, const int32_t* ring_y, const uint64_t* ring_ts, const uint8_t* ring_noise,
                         int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    return upload_ring<int32_t>(c, ring_x, ring_y, ring_ts, ring_noise, cap, first, n, t0);
}

int bf_upload_ring16_async(bf_ctx* c, const uint16_t* ring_row, const uint16_t* ring_col, const uint64_t* ring_ts,
                           const uint8_t* ring_noise, int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    return upload_ring<uint16_t>(c, ring_row, ring_col, ring_ts, ring_noise, cap, first, n, t0);
}

int bf_wait_uploads(bf_ctx* c) {
    if (!c) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->copy_stream) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    return BF_OK;
}

int bf_commit_upload(bf_ctx* c) {
    if (!c) return BF_ERR_ARG;
    if (c->pend_count == 0) return fail(c, BF_ERR_STATE, "no upload is pending");
    HIP_TRY(c, hipSetDevice(c->device));
    const int slot = c->pend_head & 1;
    // the staging kernel (compute stream) waits for the copy; nothing blocks on the host
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_done[slot], 0));
    c->has_noise = false;
    if (c->pending_ts64[slot]) {   // absolute timestamps -> slice-local 32-bit times (Event::set_local_time)
        if (c->pending_addr16[slot])   // ... and 16-bit addresses -> the staging kernel's int32 columns, same pass
            launch_local_time16(c->d_in_ts[slot], c->d_in16[slot], c->d_in16[slot] + c->cap_events, c->pending_t0[slot],
                                slot ? c->d_in2[0] : c->d_in_x, slot ? c->d_in2[1] : c->d_in_y, slot ? c->d_in2[2] : c->d_in_t,
                                c->pending_n[slot], c->stream);
        else
            launch_local_time(c->d_in_ts[slot], c->pending_t0[slot], slot ? c->d_in2[2] : c->d_in_t, c->pending_n[slot], c->stream);
        if (c->pending_noise[slot]) {   // Event::noise of the slice (padding: not noise, like the blocking upload's)
            const long long gran = (long long)kThreads * kEvPerThread;
            const long long n_pad = (c->pending_n[slot] + gran - 1) / gran * gran;
            HIP_TRY(c, hipMemsetAsync(c->d_noise, 0, (size_t)n_pad, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->d_noise, c->d_in_noise[slot], (size_t)c->pending_n[slot], hipMemcpyDeviceToDevice, c->stream));
            c->has_noise = true;
        }
    }
    int rc = stage_common(c, slot ? c->d_in2[0] : c->d_in_x, slot ? c->d_in2[1] : c->d_in_y,
                          slot ? c->d_in2[2] : c->d_in_t, c->pending_n[slot]);
    // the slot may be refilled once the staging kernels above have read it
    HIP_TRY(c, hipEventRecord(c->staged[slot], c->stream));
    c->staged_valid[slot] = true;
    c->pend_head++;
    c->pend_count--;
    return rc;
}

int bf_upload_events_device(bf_ctx* c, const int32_t* d_fr_x, const int32_t* d_fr_y, const int32_t* d_t_ns,
                            int64_t n) {
    if (!c) return BF_ERR_ARG;
    if (n < 0 || (n > 0 && (!d_fr_x || !d_fr_y || !d_t_ns))) return fail(c, BF_ERR_ARG, "bad event arrays");
    if (n > c->cap_events) return fail(c, BF_ERR_CAPACITY, "n=%lld exceeds capacity %lld", (long long)n, c->cap_events);
    HIP_TRY(c, hipSetDevice(c->device));
    c->has_noise = false;
    return stage_common(c, d_fr_x, d_fr_y, d_t_ns, n);
}

int bf_set_cloud(bf_ctx* c, int32_t scale, int32_t res_x, int32_t res_y, bf_window* window_out) {
    if (!c) return BF_ERR_ARG;
    if (!c->uploaded) return fail(c, BF_ERR_STATE, "bf_set_cloud before bf_upload_events");
    if (scale < 1 || scale % 2 == 0 || scale / 2 > kMaxHalfScale)   // optimizer_rolling.h:274
        return fail(c, BF_ERR_ARG, "scale must be odd and <= %d (got %d)", 2 * kMaxHalfScale + 1, scale);
    HIP_TRY(c, hipSetDevice(c->device));
    {
        int rc = fold_stats(c);
        if (rc != BF_OK) return rc;
    }
    const SliceStats s = c->stats;
    bf_window w;
    memset(&w, 0, sizeof(w));
    w.scale = scale;
    // optimizer_rolling.h:252-260: min seeded with RES, max with 0
    w.x_min = res_x; w.y_min = res_y; w.x_max = 0; w.y_max = 0;
    if (c->n > 0) {
        if (s.tmin == INT_MIN)
            return fail(c, BF_ERR_ARG, "a slice-local event time does not fit 32 bits (slice longer than 2.1 s?)");
        if (s.xmin < 0 || s.ymin < 0 || s.xmax > 65535 || s.ymax > 65535)
            return fail(c, BF_ERR_ARG, "event coordinates outside [0, 65535]");
        if (s.xmax > w.x_max) w.x_max = s.xmax;
        if (s.ymax > w.y_max) w.y_max = s.ymax;
        if (s.xmin < w.x_min) w.x_min = s.xmin;
        if (s.ymin < w.y_min) w.y_min = s.ymin;
    }
    w.metric_wsizex = scale * (w.x_max - w.x_min);   // :263
    w.metric_wsizey = scale * (w.y_max - w.y_min);   // :264
    w.scale_img_x = w.metric_wsizex + scale;         // :276
    w.scale_img_y = w.metric_wsizey + scale;         // :277
    // :279-282, both "/ 2" are integer divisions
    w.x_shift = -double((w.x_max - w.x_min) / 2 + w.x_min) * double(scale) +
                double(w.metric_wsizex) / 2.0 + scale / 2;
    w.y_shift = -double((w.y_max - w.y_min) / 2 + w.y_min) * double(scale) +
                double(w.metric_wsizey) / 2.0 + scale / 2;
    if (w.scale_img_x <= 0 || w.scale_img_y <= 0) {
        // e.g. an empty slice: x_min = RES_X > x_max = 0.  The reference carries on and run()
        // returns 1 at the window guard (:49-55); no image operator is usable on it.
        c->win = w;
        c->have_window = true;
        c->degenerate = true;
        memset(&c->hst.model, 0, sizeof(c->hst.model));
        c->pending_warp = false;
        if (window_out) *window_out = w;
        return BF_OK;
    }
    c->degenerate = false;
    if ((size_t)w.scale_img_x * (size_t)w.scale_img_y > c->cap_px)
        return fail(c, BF_ERR_CAPACITY, "window %d x %d exceeds the image capacity", w.scale_img_x, w.scale_img_y);
    if (w.scale_img_x > 65535 || w.scale_img_y > 65535)   // 16-bit pixel coordinates in the packed moment sums
        return fail(c, BF_ERR_CAPACITY, "window %d x %d: at most 65535 rows / columns", w.scale_img_x, w.scale_img_y);
    int gx, gy;
    stencil_grid(w.scale_img_x, w.scale_img_y, &gx, &gy);
    if (gx * gy > c->cap_blocks) return fail(c, BF_ERR_CAPACITY, "window needs %d tiles > %d", gx * gy, c->cap_blocks);

    // Accumulator packing: count << tbits | sum(t - tmin).  Exact iff both fields can hold the
    // whole slice (no pixel can collect more than all events / all time).
    long long tmin = (c->n > 0) ? (long long)s.tmin : 0;
    unsigned long long span_sum = (c->n > 0) ? (unsigned long long)(s.tsum - tmin * c->n) : 0ull;
    int tbits = bit_length(span_sum);
    if (tbits < 1) tbits = 1;
    int cbits = bit_length((unsigned long long)c->n);
    c->packed = !c->force_split && (tbits + cbits <= 64);
    if (!c->packed) {
        int rc = ensure_cplanes(c);
        if (rc != BF_OK) return rc;
        tbits = 64;
    }

    c->win = w;
    c->have_window = true;
    DevState& h = c->hst;
    h.hot.scale = scale;
    h.hot.R = w.scale_img_x; h.hot.C = w.scale_img_y;
    h.hot.wsx = w.metric_wsizex; h.hot.wsy = w.metric_wsizey;
    h.hot.x_sh = (int)w.x_shift;   // double -> int parameter conversion of accel_lib.h:147
    h.hot.y_sh = (int)w.y_shift;
    h.hot.tbits = tbits;
    h.x_shift = w.x_shift; h.y_shift = w.y_shift;
    h.hot.tmin = tmin;
    h.nblocks = gx * gy;
    // Stencil work-group size: 256 threads per 16 x 64 tile.  "stencil_threads" = 512 halves the pixels per thread and
    // shortens a work-group's dependent chain: one context alone on a small image gains 1.5 % (20.65 -> 20.35 us per
    // iteration at 346x260), four contexts sharing the GPU lose 12 % (164 -> 144 Mev/s: twice the waves for the same
    // work), and the f64 partial sums are formed in another order, so its bits differ from the 256-thread build's.
    // Hence an option, not a default.
    c->stencil_threads = c->opt_stencil_threads > 0 ? c->opt_stencil_threads : 256;
    memset(&h.model, 0, sizeof(h.model));   // a fresh OptimizerRolling has a zero ObjectModel
    h.hot.wp = identity_warp();
    h.hot.it = 0; h.hot.done = 0; h.rc = 0;
    c->pending_warp = false;
    c->all_noise = false;
    // Event::reset for every event (set_cloud :260).  bf_upload_events already reset p.
    if (!c->p_clean) HIP_TRY(c, hipMemsetAsync(c->set[c->cs].p, 0, (size_t)c->n_pad * sizeof(float2), c->stream));
    c->p_clean = true;
    c->n_valid = false;
    c->uv_valid = false;
    c->out_sorted = false;
    // Tile-binned scatter: usable when there is no noise mask and the bin grid fits the kernels' LDS.  Its own
    // per-bin packing is decided on the device by the counting sort (k_bin_scan), with the overflow path as fallback.
    {
        BinGrid g;
        memset(&g, 0, sizeof(g));
        // Tile shape: one work-group per bin.  Cost model of one iteration (calibrated on config 2, in us):
        //   waves of work-groups x events per tile x 1.7 ns   (the fullest CU sets the length of the scatter kernel)
        // + slab pixels x 2.3 ps                               (every slab pixel is written and re-read)
        // over widths {16, 32, 64} (a power of two) and heights {32 .. 128}; ties go to the larger tile.  Small dense
        // images get small tiles (enough bins to fill the CUs), large images large ones (less margin overhead).
        // "bin_tile" / "bin_tile_rows" override.
        g.TS = c->opt_bin_tile > 0 ? c->opt_bin_tile : 64;
        g.TSR = c->opt_bin_tile_rows > 0 ? c->opt_bin_tile_rows : (g.TS < 32 ? 32 : g.TS);
        if (c->n_cus > 0 && (c->opt_bin_tile <= 0 || c->opt_bin_tile_rows <= 0)) {
            const double density = (double)c->n / ((double)w.scale_img_x * (double)w.scale_img_y);
            double best = -1.0;
            int best_area = 0;
            for (int cols = 16; cols <= 64; cols *= 2) {
                if (c->opt_bin_tile > 0 && cols != c->opt_bin_tile) continue;
                for (int rows = 32; rows <= 128; rows += 16) {
                    if (c->opt_bin_tile_rows > 0 && rows != c->opt_bin_tile_rows) continue;
                    const int d = c->opt_bin_margin > cols / 2 ? cols / 2 : c->opt_bin_margin;
                    if ((size_t)(rows + 2 * d) * (cols + 2 * d) * 8 > 64 * 1024) continue;
                    const int nb = ((w.scale_img_x + rows - 1) / rows) * ((w.scale_img_y + cols - 1) / cols);
                    if (nb > 8192) continue;
                    const double cost = (double)((nb + c->n_cus - 1) / c->n_cus) * rows * cols * density * 1.7e-3 +
                                        (double)nb * (rows + 2 * d) * (cols + 2 * d) * 2.3e-6;
                    if (best < 0 || cost < best * 0.999 || (cost <= best * 1.001 && rows * cols > best_area)) {
                        best = cost; best_area = rows * cols; g.TS = cols; g.TSR = rows;
                    }
                }
            }
        }
        g.lg = 0;
        while ((1 << g.lg) < g.TS) ++g.lg;
        g.nbc = (w.scale_img_y + g.TS - 1) / g.TS;
        const int tmin_ = g.TS < g.TSR ? g.TS : g.TSR;
        g.D = c->opt_bin_margin > tmin_ / 2 ? tmin_ / 2 : c->opt_bin_margin;   // <= 2 x 2 bins per pixel
        g.L = g.TS + 2 * g.D;
        g.LR = g.TSR + 2 * g.D;
        g.mul_r = (uint32_t)(0x100000000ull / (unsigned)g.TSR) + 1u;
        g.mul_l = (uint32_t)(0x100000000ull / (unsigned)g.L) + 1u;
        g.mul_h = (uint32_t)(0x100000000ull / (unsigned)(g.L / 2 > 0 ? g.L / 2 : 1)) + 1u;
        g.nbr = (w.scale_img_x + g.TSR - 1) / g.TSR;
        g.nbins = g.nbr * g.nbc;
        // Density rule: every iteration writes and re-reads one slab pixel (8 B x (L / TS)^2) per image pixel, a global
        // atomic costs ~48 ns per event; below ~1 event per 12 pixels the plain atomic scatter is the faster one
        // (measured: 300k events on a 3550 x 6350 image, 0.41 vs 0.66 ms per iteration).
        // ... on an image of tens of megapixels: the event-list form of the binned loop follows the events, and up to the
        // 8.3 M pixels of a 1280x720 sensor at scale 3 it beats the atomics for sparse slices too (20k .. 500k events:
        // 640x480 22 .. 28 us per iteration against 31 .. 40, 1280x720 48 .. 64 against 62 .. 86).
        const bool dense = (double)w.scale_img_x * (double)w.scale_img_y < 12.0 * (double)c->n ||
                           (double)w.scale_img_x * (double)w.scale_img_y <= 9.0e6;
        c->use_binned = (c->opt_binned == 2 || (c->opt_binned == 1 && dense)) && !c->force_split && !c->has_noise && c->n > 0 &&
                        g.nbins <= 8192 &&
                        (size_t)g.LR * g.L * 8 <= (size_t)kBinTileLdsMax && w.scale_img_x < (1 << 20);
        if (c->use_binned) {
            int rc = ensure_cplanes(c);
            if (rc == BF_OK) rc = ensure_bin_buffers(c, g);
            if (rc == BF_OK) rc = ensure_ovf_bits(c, w.scale_img_x, w.scale_img_y);
            if (rc != BF_OK) return rc;
            c->grid = g;
        }
        // The one-kernel iteration (k_fused_pass; used by bf_run unless the context is co-scheduled with others): image
        // tiles of 32 x 64 pixels -- 64 x 64 when the nine sort keys per tile would not fit the counting sort -- and a
        // margin D that keeps a tile's edge strips (H + D wide, H = scale / 2 + 1) from overlapping.
        // Where it pays (measured on MI355X, one context, cold runs; us per iteration fused / best two-kernel or atomic loop):
        //   240x180: 50k events 16.1 / 22.0, 200k 17.9 / 19.6, 400k 20.5 / 18.2;   346x260: 20k 16.0 / 17.1, 50k 15.9 / 19.6,
        //   100k 17.5 / 23.1, 200k 17.8 / 20.2, 400k 19.6 / 20.3, 1M 26.8 / 19.7;   640x480: 20k .. 400k 31 .. 38 / 22 .. 34.
        // The events of a tile's edge strips are warped by up to four work-groups (2.1 x the events at D = 8) and a
        // dense slice meets in few LDS words, so "auto" takes it for slices of at most one event per two image pixels on
        // images up to 1.2 M pixels; a launch chain half as long is what it buys there.
        c->fused_ok = false;
        const double Pimg = (double)w.scale_img_x * (double)w.scale_img_y;
        const bool fused_pays = Pimg <= 1.2e6 && 2.0 * (double)c->n <= Pimg;
        if ((c->opt_fused == 2 || (c->opt_fused == 1 && fused_pays)) && c->opt_binned != 0 && !c->force_split && !c->has_noise && c->n > 0 && scale / 2 <= 4 &&
            c->stencil_threads == 256 && w.scale_img_x < (1 << 20) && (long long)c->n < (1ll << 31)) {
            BinGrid f;
            memset(&f, 0, sizeof(f));
            const int Hh = scale / 2 + 1;
            auto tiles = [&](int rows) { return ((w.scale_img_x + rows - 1) / rows) * ((w.scale_img_y + 63) / 64); };
            int rows = c->opt_fused_rows > 0 ? c->opt_fused_rows : (tiles(32) * kFusedZones <= 8192 ? 32 : 64);
            int Dm = c->opt_fused_margin;
            if (Dm > rows / 2 - Hh) Dm = rows / 2 - Hh;
            if (Dm >= 1 && tiles(rows) * kFusedZones <= 8192) {
                f.TS = 64; f.lg = 6; f.TSR = rows; f.D = Dm; f.fz = Hh + Dm;
                f.nbc = (w.scale_img_y + 63) / 64;
                f.nbr = (w.scale_img_x + rows - 1) / rows;
                f.nbins = f.nbr * f.nbc * kFusedZones;   // sort keys
                f.mul_r = (uint32_t)(0x100000000ull / (unsigned)f.TSR) + 1u;
                f.L = f.LR = 0; f.mul_l = 0;
                int rc = ensure_cplanes(c);
                if (rc == BF_OK) rc = ensure_bin_buffers(c, f);
                if (rc != BF_OK) return rc;
                const int nt = f.nbr * f.nbc;
                if (nt > c->ftab_alloc) {
                    if (c->d_ftab) HIP_TRY(c, hipFree(c->d_ftab));
                    c->d_ftab = nullptr;
                    HIP_TRY(c, hipMalloc(&c->d_ftab, (size_t)nt * sizeof(FusedTab)));
                    c->ftab_alloc = nt;
                }
                for (int i = 0; i < 2; ++i)
                    if (!c->set[i].p2) HIP_TRY(c, hipMalloc(&c->set[i].p2, (size_t)c->cap_events * sizeof(float2)));
                c->fgrid = f;
                c->fused_ok = true;
                // Contexts that share the GPU: with dense slices the two loop kernels are bandwidth-bound and the tail-update
                // form keeps the CUs full, so the two-kernel loop stays; sparse slices remain launch-bound even with eight
                // contexts in flight (346x260, 2 / 4 / 8 contexts: 50k events 12.4 / 11.4 / 10.6 us per iteration and slice
                // against 16.7 / 14.6 / 12.8; 200k events 11.8 / 9.3 / 9.4 against 13.3 / 9.2 / 9.1).
                c->fused_shared = c->opt_fused == 2 || 8.0 * (double)c->n <= Pimg;
            }
        }
        h.hot.binned = (c->use_binned || c->fused_ok) ? 1 : 0;
        h.hot.pp = 0; h.hot.redo = 0; h.hot.pend = 0; h.last_j = -1;
        h.n_events = (uint32_t)c->n;
        h.hot.bin_tbits = tbits > 62 ? 62 : tbits; h.hot.bin_ok = 1; h.hot.need_rebin = 0; h.hot.rebins = 0; h.ovf_total = 0;
        h.hot.flip = 0;
        // Dense slabs, merged lists or event lists.  A dense slice (one event per four pixels or more) merges its events in
        // the bin's LDS tile and writes the tile.  A sparse one writes lists, work and traffic following the events: one
        // entry per EVENT and no LDS tile where events rarely meet at a pixel (at most two events per sensor pixel of the
        // window: a 1280x720 sensor with 1M events -- the tile of such a bin would fill the CU's LDS and leave one
        // work-group per CU), one entry per touched PIXEL, merged in the LDS tile, where they do (a small sensor at a large
        // scale: a third of the entries, and the stencil kernel splats every entry into s x s pixels).  "auto" decides once
        // per slice: the kernels are compiled per format.  Measured per iteration (dense / merged / events): 1280x720
        // scale 3: 90 / 81 / 68 us; 346x260 scale 7: 96 / 61 / 103; 640x480 scale 3: 44 / 53 / 52.
        {
            const double P = (double)w.scale_img_x * (double)w.scale_img_y;
            const double sensor_px = P / ((double)scale * (double)scale);
            const size_t LLg = (size_t)g.LR * (size_t)g.L;
            const bool lists_ok = c->use_binned && LLg <= 65536;                         // 16-bit tile-local pixel indices
            const bool merged_ok = lists_ok && LLg * 10 + 16 <= (size_t)kBinTileLdsMax;   // tile + index list in LDS
            const int mode = lists_ok ? c->opt_bin_compact : 0;
            c->fmt = 0;
            if (mode == 2) c->fmt = 2;
            else if (mode == 3) c->fmt = merged_ok ? 1 : 2;
            else if (mode == 1 && 4.0 * (double)c->n < P) c->fmt = ((double)c->n <= 2.0 * sensor_px || !merged_ok) ? 2 : 1;
            // Dense slices: the bin's own pixels + a margin plane instead of whole-tile slabs (flush_split).  It moves 0.6 x the
            // slab bytes and a quarter of the stencil kernel's loads; "auto" takes it where that is what the iteration
            // waits for -- a context that has the GPU to itself (update at the scatter head) on an image of >= 1.5 M
            // pixels: 640x480 scale 3, 1M events: K1 14.7 -> 11.3 us, iteration 37.2 -> 32.9 us.  At 346x260 the loop is a
            // latency chain and nothing moves (18.8 us either way); with the update in the stencil tail ("co_schedule") the
            // lean scatter kernel LOSES 1.7 us per launch (8.0 -> 9.7 us at 346x260, value 196 -> 178 Mevents/s).
            const bool split_pays = !c->opt_co_schedule && P >= 1.5e6;
            if (c->fmt == 0 && c->use_binned && (c->opt_bin_split == 2 || (c->opt_bin_split == 1 && split_pays)) && g.D >= 2 &&
                (g.D & (g.D - 1)) == 0 && g.TS >= 4) {   // (D a power of two)
                int rc = ensure_margin_buffers(c, g);
                if (rc != BF_OK) return rc;
                c->fmt = 3;
            }
            h.hot.fmt = c->fmt;
        }
        h.t_span = (c->n > 0) ? (long long)s.tmax - (long long)s.tmin : 0;
        h.t_abs_max = (c->n > 0) ? std::fmax(std::fabs((double)s.tmin), std::fabs((double)s.tmax)) : 0.0;
        h.r_max = std::hypot((double)(w.x_max - w.x_min), (double)(w.y_max - w.y_min)) + 64.0;
        h.drift_limit = c->opt_bin_predict ? 0.6 * (double)c->grid.D : 1e300;
    }
    if (c->planes_unknown || w.scale_img_x != c->last_R || w.scale_img_y != c->last_C) {
        int rc = clear_planes(c);
        if (rc != BF_OK) return rc;
    }
    c->last_R = w.scale_img_x;
    c->last_C = w.scale_img_y;
    // (the device copy of the state is written by whoever uses it next -- bf_run, the AccelLib operators,
    // flush_pending all upload c->hst first; a launch here would only add ~5 us to every slice)
    if (window_out) *window_out = w;
    return BF_OK;
}

// ---- AccelLib operators ----------------------------------------------------------------

int bf_project_4param_reinit(bf_ctx* c, double dnx_, double dny_, double cx, double cy, double div,
                             double crl) {
    if (!c) return BF_ERR_ARG;
    if (!c->have_window) return fail(c, BF_ERR_STATE, "bf_project_4param_reinit before bf_set_cloud");
    if (c->degenerate) return fail(c, BF_ERR_STATE, "bf_project_4param_reinit on a degenerate (empty) window");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != BF_OK) return rc;
    WarpParams& w = c->hst.hot.wp;
    w.dnx = dnx_; w.dny = dny_; w.cx = cx; w.cy = cy; w.div = div;
    w.c = std::cos(crl);   // event.h:102-103 evaluates std::cos / std::sin on the host
    w.s = std::sin(crl);
    launch_set_state(c->d_state, c->hst, c->stream);
    {
        ProfScope ps(c, 0, c->n);
        launch_warp_scatter(ws_args(c, c->cur, 0), true, false, true, c->stream);
    }
    c->p_clean = false;
    c->n_valid = true;
    c->uv_valid = false;
    c->out_sorted = false;
    HIP_TRY(c, hipGetLastError());
    return BF_OK;
}

int bf_get_time_img(bf_ctx* c, float* time_out, uint32_t* count_out) {
    if (!c) return BF_ERR_ARG;
    if (!c->have_window) return fail(c, BF_ERR_STATE, "bf_get_time_img before bf_set_cloud");
    if (c->degenerate) return fail(c, BF_ERR_STATE, "bf_get_time_img on a degenerate (empty) window");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != BF_OK) return rc;
    launch_set_state(c->d_state, c->hst, c->stream);
    const int buf = c->cur;
    if (!c->all_noise) {
        ProfScope ps(c, 0, c->n);
        launch_warp_scatter(ws_args(c, buf, 0), false, true, false, c->stream);
    }
    StencilArgs a = st_args(c, buf, 0);
    a.time_out = c->d_time;
    a.count_out = c->d_count;
    a.zero_cplane = c->d_cplane[buf ^ 1];   // may be NULL (never allocated): nothing to clear
    {
        ProfScope ps(c, 1);
        launch_stencil(a, stencil_src(c, false), c->stream);
    }
    HIP_TRY(c, hipGetLastError());
    c->cur = buf ^ 1;   // the stencil zeroed the other buffer; `buf` is cleared by the next pass
    c->hst.hot.ovf_cnt[buf] = 1;
    c->hst.hot.ovf_cnt[buf ^ 1] = 0;
    const size_t P = (size_t)c->win.scale_img_x * (size_t)c->win.scale_img_y;
    if (time_out)
        HIP_TRY(c, hipMemcpyAsync(time_out, c->d_time, P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (count_out)
        HIP_TRY(c, hipMemcpyAsync(count_out, c->d_count, P * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

static int image_pass(bf_ctx* c, const float* d_src, int rows, int cols, bool grads, bool moments) {
    StencilArgs a;
    memset(&a, 0, sizeof(a));
    a.st = c->d_state;
    a.R = rows; a.C = cols; a.scale = 1;
    a.time_in = d_src;
    if (grads) { a.gx_out = c->d_gx; a.gy_out = c->d_gy; }
    if (moments) {   // sums -> exact accumulators; the last work-group forms the model (mode 0)
        if (c->acc_dirty) {
            if (hipMemsetAsync(c->d_acc, 0, 2 * kAccGroups * sizeof(MomentAcc), c->stream) != hipSuccess) return BF_ERR_HIP;
            c->acc_dirty = false;
        }
        a.acc = c->d_acc;
        a.ticket = c->d_ticket;
        a.st_rw = c->d_state;
        a.update_mode = 0;
    }
    ProfScope ps(c, 1);
    launch_stencil(a, 2, c->stream);
    return BF_OK;
}

int bf_sobel(bf_ctx* c, const float* img, int32_t rows, int32_t cols, float* grad_x, float* grad_y) {
    if (!c) return BF_ERR_ARG;
    if (!img || !grad_x || !grad_y || rows <= 0 || cols <= 0) return fail(c, BF_ERR_ARG, "bad image");
    const size_t P = (size_t)rows * (size_t)cols;
    if (P > c->cap_px) return fail(c, BF_ERR_CAPACITY, "image %d x %d exceeds capacity", rows, cols);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_img, img, P * sizeof(float), hipMemcpyHostToDevice, c->stream));
    image_pass(c, c->d_img, rows, cols, true, false);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(grad_x, c->d_gx, P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(grad_y, c->d_gy, P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

int bf_fast_model(bf_ctx* c, const float* img, int32_t rows, int32_t cols, bf_model* model) {
    if (!c) return BF_ERR_ARG;
    if (!model) return fail(c, BF_ERR_ARG, "model is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    const float* src = c->d_time;
    if (img) {
        if (rows <= 0 || cols <= 0) return fail(c, BF_ERR_ARG, "bad image");
        const size_t P = (size_t)rows * (size_t)cols;
        if (P > c->cap_px) return fail(c, BF_ERR_CAPACITY, "image %d x %d exceeds capacity", rows, cols);
        HIP_TRY(c, hipMemcpyAsync(c->d_img, img, P * sizeof(float), hipMemcpyHostToDevice, c->stream));
        src = c->d_img;
    } else {
        if (!c->have_window) return fail(c, BF_ERR_STATE, "no resident time image");
        rows = c->win.scale_img_x;
        cols = c->win.scale_img_y;
    }
    int gx, gy;
    stencil_grid(rows, cols, &gx, &gy);
    if (gx * gy > c->cap_blocks) return fail(c, BF_ERR_CAPACITY, "image needs %d tiles", gx * gy);
    DevState tmp = c->hst;
    tmp.hot.R = rows; tmp.hot.C = cols;
    launch_set_state(c->d_state, tmp, c->stream);
    image_pass(c, src, rows, cols, false, true);
    HIP_TRY(c, hipGetLastError());
    int rc = d2h_state(c);
    if (rc != BF_OK) return rc;
    const bf_model& m = c->h_state->model;
    model->cx = m.cx; model->cy = m.cy;
    model->dx = m.dx; model->dy = m.dy;
    model->rot = m.rot; model->div = m.div;
    model->cnt = m.cnt;
    return BF_OK;
}

// The final warp of bf_run writes its per-event outputs in slot (tile-sorted) order -- coalesced stores instead of
// 16-byte stores scattered through perm[] (31 -> 10 us per 1M events) -- and they are put back into upload order only
// when somebody reads them.
static int materialize_outputs(bf_ctx* c) {
    if (!c->out_sorted) return BF_OK;
    c->out_sorted = false;
    if (!c->has_perm || c->n == 0) return BF_OK;
    if (!c->d_out_tmp) HIP_TRY(c, hipMalloc(&c->d_out_tmp, (size_t)c->cap_events * sizeof(double2)));
    const uint32_t* perm = c->set[c->cs].perm;
    launch_unpermute(c->d_nxny, perm, c->d_out_tmp, c->n, c->stream);
    std::swap(c->d_nxny, c->d_out_tmp);
    if (c->uv_valid) {
        launch_unpermute(c->d_uv, perm, c->d_out_tmp, c->n, c->stream);
        std::swap(c->d_uv, c->d_out_tmp);
    }
    HIP_TRY(c, hipGetLastError());
    return BF_OK;
}

static int copy_pairs(bf_ctx* c, const double2* d_src, double* a, double* b) {
    std::vector<double2> tmp((size_t)c->n);
    HIP_TRY(c, hipMemcpyAsync(tmp.data(), d_src, (size_t)c->n * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (long long i = 0; i < c->n; ++i) {
        if (a) a[i] = tmp[(size_t)i].x;
        if (b) b[i] = tmp[(size_t)i].y;
    }
    return BF_OK;
}

int bf_writeout_events(bf_ctx* c, double* pr_x, double* pr_y, double* nx, double* ny) {
    if (!c) return BF_ERR_ARG;
    if (!c->have_window) return fail(c, BF_ERR_STATE, "bf_writeout_events before bf_set_cloud");
    if (c->degenerate) return fail(c, BF_ERR_STATE, "bf_writeout_events on a degenerate (empty) window");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != BF_OK) return rc;
    if (c->n == 0) return BF_OK;
    rc = materialize_outputs(c);
    if (rc != BF_OK) return rc;
    if (pr_x || pr_y) {
        c->uv_valid = false;   // d_uv is the staging buffer of the expanded positions below
        {
            ProfScope ps(c, 3);
            launch_expand_pr(c->set[c->cs].xy, c->set[c->cs].p, c->has_perm ? c->set[c->cs].perm : nullptr,
                             c->d_uv, c->n, c->stream);
        }
        HIP_TRY(c, hipGetLastError());
        rc = copy_pairs(c, c->d_uv, pr_x, pr_y);
        if (rc != BF_OK) return rc;
    }
    if (nx || ny) {
        if (!c->n_valid) {   // Event::reset leaves nx = ny = 0 (event.h:57)
            for (long long i = 0; i < c->n; ++i) {
                if (nx) nx[i] = 0.0;
                if (ny) ny[i] = 0.0;
            }
        } else {
            rc = copy_pairs(c, c->d_nxny, nx, ny);
            if (rc != BF_OK) return rc;
        }
    }
    return BF_OK;
}

int bf_compute_uv(bf_ctx* c, double* u, double* v) {
    if (!c) return BF_ERR_ARG;
    if (!c->have_window) return fail(c, BF_ERR_STATE, "bf_compute_uv before bf_set_cloud");
    if (c->degenerate) return fail(c, BF_ERR_STATE, "bf_compute_uv on a degenerate (empty) window");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != BF_OK) return rc;
    if (c->n == 0) return BF_OK;
    if (!c->n_valid) {
        for (long long i = 0; i < c->n; ++i) {
            if (u) u[i] = 0.0;
            if (v) v[i] = 0.0;
        }
        return BF_OK;
    }
    rc = materialize_outputs(c);
    if (rc != BF_OK) return rc;
    if (!c->uv_valid) {   // (bf_run with want_uv already produced it in its final warp)
        ProfScope ps(c, 3);
        launch_compute_uv(c->d_nxny, c->d_uv, c->n, c->stream);
        c->uv_valid = true;
    }
    HIP_TRY(c, hipGetLastError());
    return copy_pairs(c, c->d_uv, u, v);
}

int bf_compute_uv_ring(bf_ctx* c, double* uv_ring, int64_t cap, int64_t first) {
    if (!c) return BF_ERR_ARG;
    if (!c->have_window) return fail(c, BF_ERR_STATE, "bf_compute_uv_ring before bf_set_cloud");
    if (c->degenerate) return fail(c, BF_ERR_STATE, "bf_compute_uv_ring on a degenerate (empty) window");
    if (!uv_ring || cap <= 0 || first < 0 || first >= cap || c->n > cap)
        return fail(c, BF_ERR_ARG, "bad flow ring (cap %lld, first %lld, n %lld)", (long long)cap, (long long)first, c->n);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != BF_OK) return rc;
    if (c->n == 0) return BF_OK;
    const int64_t n0 = (first + c->n <= cap) ? c->n : cap - first, n1 = c->n - n0;
    if (!c->n_valid) {   // Event::reset state: no flow yet
        memset(uv_ring + 2 * first, 0, (size_t)n0 * 16);
        memset(uv_ring, 0, (size_t)n1 * 16);
        return BF_OK;
    }
    rc = materialize_outputs(c);
    if (rc != BF_OK) return rc;
    if (!c->uv_valid) {
        ProfScope ps(c, 3);
        launch_compute_uv(c->d_nxny, c->d_uv, c->n, c->stream);
        c->uv_valid = true;
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(uv_ring + 2 * first, c->d_uv, (size_t)n0 * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    if (n1 > 0) HIP_TRY(c, hipMemcpyAsync(uv_ring, c->d_uv + n0, (size_t)n1 * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

// ---- fused optimizer ---------------------------------------------------------------------

int bf_set_model(bf_ctx* c, const bf_model* model) {
    if (!c || !model) return BF_ERR_ARG;
    if (!c->have_window) return fail(c, BF_ERR_STATE, "bf_set_model before bf_set_cloud");
    if (c->degenerate) {   // nothing to warp; get_model() still returns what was set
        c->hst.model = *model;
        return BF_OK;
    }
    // optimizer_rolling.h:289-299: model <- m; warp(-total_dx, -total_dy, cx, cy, total_div, -total_rot)
    c->hst.model = *model;
    WarpParams& w = c->hst.hot.wp;
    w.dnx = -model->total_dx; w.dny = -model->total_dy;
    w.cx = model->cx; w.cy = model->cy;
    w.div = model->total_div;
    w.c = std::cos(-model->total_rot);
    w.s = std::sin(-model->total_rot);
    c->pending_warp = true;
    return BF_OK;
}

int bf_run(bf_ctx* c, const bf_run_opts* opts_in, bf_model* model_out, bf_run_info* info) {
    if (!c) return BF_ERR_ARG;
    if (!c->have_window) return fail(c, BF_ERR_STATE, "bf_run before bf_set_cloud");
    bf_run_opts o;
    if (opts_in) o = *opts_in; else bf_run_opts_default(&o);
    if (o.poll_interval < 1) o.poll_interval = 1;
    bf_run_info inf;
    memset(&inf, 0, sizeof(inf));
    inf.x_divider = inf.y_divider = 1.0f;
    inf.rot_divider = inf.div_divider = 10000.0f;
    HIP_TRY(c, hipSetDevice(c->device));
    const bf_window& w = c->win;

    // optimizer_rolling.h:49-55 (integer arithmetic) and :57-58
    if ((w.scale_img_x < w.scale * o.res_x / 15) && (w.scale_img_y < w.scale * o.res_y / 15)) {
        c->all_noise = true;   // "for (auto &e : *events) e.noise = true;"
        inf.rc = BF_SKIPPED;
    } else if (c->n < (long long)o.min_events) {
        inf.rc = BF_SKIPPED;
    }
    if (inf.rc == BF_SKIPPED) {
        if (model_out) *model_out = c->hst.model;
        if (info) *info = inf;
        return BF_SKIPPED;
    }

    if (o.trace_cap > c->trace_alloc) {
        if (c->d_trace) HIP_TRY(c, hipFree(c->d_trace));
        c->d_trace = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_trace, (size_t)o.trace_cap * sizeof(bf_trace_rec)));
        c->trace_alloc = o.trace_cap;
    }
    c->p_clean = false;   // the loop warps the events
    // One slice context alone on the GPU: the one-kernel iteration when the slice qualifies (bf_set_cloud), else the
    // two-kernel tile-binned loop when the slice is dense enough for it, else global atomics.
    const bool fused = c->fused_ok && (!c->opt_co_schedule || c->fused_shared);
    const bool binned = c->use_binned || fused;
    // The persistent form of the one-kernel loop (bf_loop.hip): the work-groups stay resident over many iterations and
    // exchange their moment sums through memory -- for a context that has the GPU to itself (two such kernels from two
    // contexts could each hold half of the CUs and wait for the other half), when all tiles can be resident at once.
    // A cold run re-bins a dozen times in its first iterations, and every re-bin ends a launch of the persistent kernel with
    // a host round trip (measured on 50 000 events, 240x180: 25 us per iteration against 17); a warm-started slice of a stream
    // -- the reference's own mode, ~115 iterations and one or two re-bins -- is where it pays (11.1 against 12.2 us per
    // iteration all in): "auto" takes it for warm starts.
    const bool persist = fused && !c->opt_co_schedule && (c->opt_persist == 2 || (c->opt_persist == 1 && c->pending_warp)) &&
                         g_live_ctx[c->device & 63].load() == 1 &&
                         fused_loop_resident(c->win.scale / 2, c->fgrid.TSR, c->n_cus, c->fgrid.nbr * c->fgrid.nbc);   // (else: one launch per iteration)
    DevState& h = c->hst;
    // Tile-binned mode sorts the events by the tile of their CURRENT target, so a warm-start
    // warp (bf_set_model) is applied before the sort rather than inside the first iteration.
    bool first_warp = c->pending_warp;
    const bool warm_start = c->pending_warp;
    WarpParams prewarp_wp = h.hot.wp;
    const bool prewarp = binned && c->pending_warp;   // fused into the first counting sort (k_bin_count<true>)
    if (prewarp) first_warp = false;
    c->pending_warp = false;
    h.x_div = h.y_div = 1.0f;            // :61
    h.rot_div = h.div_div = 10000.0f;    // :62-63
    h.old_dx = h.old_dy = h.old_rot = h.old_div = 0.f;
    h.hot.it = 0; h.hot.done = 0; h.rc = 0;
    h.run_tag = (int32_t)((++c->run_counter & 0x3fffffff) | 0x40000000);   // `done` is set to this (non-zero) tag
    h.max_iter = o.max_iter;
    h.hard_cap = o.hard_iter_cap;
    h.trace_cap = o.trace_cap;
    h.hot.binned = binned ? 1 : 0;
    h.hot.need_rebin = binned ? 1 : 0;   // the first enqueued re-bin builds the bins
    h.hot.rebins = 0; h.ovf_total = 0;
    h.hot.cs = c->cs; h.hot.flip = 0;
    h.hot.pp = 0; h.hot.redo = 0; h.hot.pend = 0; h.last_j = -1;
    // (the persistent loop re-bins AT the request -- it returns for it --, the other loops one or two batches of launches
    // after it: the same effective threshold)
    if (binned) h.drift_limit = c->opt_bin_predict ? (persist ? 0.85 : 0.6) * (double)(fused ? c->fgrid.D : c->grid.D) : 1e300;
    if (!first_warp) h.hot.wp = identity_warp();
    h.ref_wp = h.hot.wp;
    const bool perm_at_start = c->has_perm;

    const int b0 = c->cur;
    int buf = b0;
    bool first = true;
    bf_trace_rec* trace = o.trace_cap > 0 ? c->d_trace : nullptr;
    int launched_iters = 0;
    DevState fin;
    // Tile-binned loop: the update of iteration j runs at the head of warp+scatter launch j + 1, so the state
    // ping-pongs between two buffers (launch j reads [j & 1], writes [(j + 1) & 1]), the moment accumulators alternate
    // with the iteration's parity, and the overflow events of iteration j are counted in slot j % 3 (slot 2 stands
    // for "iteration -1": is plane buffer b0 ^ 1 still dirty from an earlier operator?).
    auto state_of = [&](int j) { return c->d_state + (j & 1); };
    auto acc_of = [&](int j) { return c->d_acc + (size_t)(fused ? ((j % 3) + 3) % 3 : (j & 1)) * kAccGroups; };
    auto ovf_of = [&](int j) { return c->d_ovf + (((j % 3) + 3) % 3) * kOvfSlotWords; };
    // (one launch: the state, and the loop's counters / accumulators)
    launch_run_init(c->d_state, h, c->d_ovf, h.hot.ovf_cnt[b0 ^ 1] ? 1u : 0u, c->d_acc, binned || c->acc_dirty, c->stream);
    c->acc_dirty = false;
    // Interior + margin format: iteration j adds to margin plane b0 ^ (j & 1) and clears, bin by bin, what the lists say the
    // previous executed launch left in the other one.  That works across runs as long as the plane the lists describe is not
    // the one the first iteration adds to; otherwise (or after a run that did not complete) it is cleared up front.
    const bool split = c->use_binned && !fused && c->fmt == 3;
    if (split) {
        if (c->m_unknown || c->m_dirty_plane == b0) {
            int rcm = margin_reset(c);
            if (rcm != BF_OK) return rcm;
        }
        c->m_unknown = true;   // (until this run has completed)
    }
    // Where the model / loop update runs.  One slice context alone: at the head of the next warp+scatter launch (every
    // work-group for itself; shortest iteration).  Several contexts sharing the GPU ("co_schedule"): in the last
    // work-group of the stencil kernel -- a serial tail on ONE CU that the other contexts' kernels fill, instead of
    // ~1.5 us on all 256 CUs.
    const bool head_update = fused || (binned && !c->opt_co_schedule);   // (the one-kernel loop has no other form)
    if (head_update) c->acc_dirty = true;   // (the sums of the last iteration are consumed, not cleared)
    // events a scatter thread keeps in flight: one pass should cover a bin of 1.5 x the average size
    // (and its work-group size: 1024 threads for bins of thousands of events, 512 where a bin holds a few hundred --
    // large images --, so that twice as many bins are in flight per CU: 84 instead of 91 us per iteration at 1280x720)
    int ev_per_thread = 8;
    const double ev_per_bin = binned ? (double)c->n / (double)(c->grid.nbins > 0 ? c->grid.nbins : 1) : 0.0;
    // (contexts sharing the GPU, "co_schedule": 512-thread work-groups even for full bins -- a 1024-thread work-group with its
    // 51 KB tile needs half a CU's wave slots free at once and waits for them while the other contexts' kernels hold a few
    // each: its launches take 16.7 us instead of 8.0 under four contexts; with 512 threads 170 -> 190 Mevents/s.  A context
    // alone is faster with 1024: 8.0 against 8.9 us)
    const int bin_threads = c->opt_bin_threads > 0 ? c->opt_bin_threads : ((ev_per_bin >= 1536.0 && !c->opt_co_schedule) ? 1024 : 512);
    if (binned && c->opt_bin_ev > 0) ev_per_thread = c->opt_bin_ev;
    else if (binned) {
        // (event lists: registers, not LDS, set the occupancy there -- two events per thread keep four work-groups on a
        // CU, and a bin above the pass size takes a second pass; measured at 1280x720: 512 x 2 69.8 us, 512 x 4 73.5)
        // (dense tiles: a pass should cover the AVERAGE bin, fuller bins take a second pass -- sizing it for 1.5 x the
        // average left half of every thread's slots empty at 640x480: 512 x 8 19.9 us, 512 x 4 15.3 us)
        const double per_bin = (c->fmt == 2 ? 1.0 : 1.1) * ev_per_bin / (double)bin_threads;
        // (... and between two and four, two up to 2.83 -- the geometric middle: bins of ~2100 events on 1024 threads ran
        // 15.3 us with four events per thread, half of every thread's slots empty, against 12.1 us with two and a second pass
        // for the fuller bins; measured at 1M events on 440 / 520 / 560 x 480 sensors)
        ev_per_thread = per_bin <= 1 ? 1 : (per_bin <= 2.83 ? 2 : (per_bin <= 4 ? 4 : 8));
    }
    // Pipelined polling: batch b+1 is enqueued BEFORE the host waits for the state snapshot
    // taken after batch b, so the GPU never idles on the host (a blocking poll costs ~25 us of
    // idle GPU).  Kernels launched after `done` was set return at once (~1 us each).
    // The persistent form of the one-kernel loop (bf_loop.hip): the work-groups stay resident over many iterations and
    // exchange their moment sums through memory -- for a context that has the GPU to itself (two such kernels from two
    // contexts could each hold half of the CUs and wait for the other half).
    if (persist) {
        const int nsub = c->fgrid.TSR / 16, nrec = c->fgrid.nbr * c->fgrid.nbc * nsub;
        if (nrec > c->xrec_alloc) {
            if (c->d_xrec) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(c->d_xrec)); }
            c->d_xrec = nullptr;
            HIP_TRY(c, hipMalloc(&c->d_xrec, (size_t)2 * (size_t)nrec * 32 * sizeof(unsigned long long)));
            HIP_TRY(c, hipMemsetAsync(c->d_xrec, 0, (size_t)2 * (size_t)nrec * 32 * sizeof(unsigned long long), c->stream));
            c->xrec_alloc = nrec;
        }
        if (!c->d_xred) {
            HIP_TRY(c, hipMalloc(&c->d_xred, (size_t)2 * 16 * 32 * sizeof(unsigned long long)));
            HIP_TRY(c, hipMemsetAsync(c->d_xred, 0, (size_t)2 * 16 * 32 * sizeof(unsigned long long), c->stream));
            for (int i = 0; i < 3; ++i) HIP_TRY(c, hipMalloc(&c->d_xscratch[i], (size_t)c->cap_events * sizeof(float2)));
        }
    }
    bool want_rebin = false;
    int last_rebin_at = 0;
    // A warm start that is expected to converge in a handful of iterations (the previous one did) is polled batch by batch,
    // the final warp riding along: "quick".  One that is expected to run long -- the reference's own ring: ~115 iterations per
    // warm-started slice -- is fed and polled like a cold run: two-iteration batches with a blocking poll each cost it a
    // host round trip every other iteration (22 instead of 14 us per iteration on a 50 000-event slice).
    const bool quick_warm = warm_start && c->warm_iters_hint < 3 * o.poll_interval;
    const bool snap_polled = binned && !quick_warm && !persist;   // progress is read from the pinned snapshot (below)
    if (snap_polled) {
        *reinterpret_cast<volatile unsigned long long*>(&c->h_state[0]) = 0ull;
        *reinterpret_cast<volatile unsigned long long*>(&c->h_state[0].run_tag) = 0ull;
    }
    int stall_allowance = 0;   // launches that may have been spent waiting for a re-bin (one-kernel iteration)
    bool final_done = false;   // the gated final warp of a warm start's first batch already ran
    int skip_rebin_checks = 0;
    static const bool host_timing = getenv("BF_HOST_TIMING") != nullptr;   // debug: where the host thread's time goes
    double ht_launch = 0, ht_wait = 0;
    auto ht_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double ht_mark = host_timing ? ht_now() : 0;
    for (int batch = 0; persist; ++batch) {
        // One round: the (device-gated) re-bin, the loop kernel -- which returns when the loop is over, when a re-bin is due
        // or after max_passes iterations --, the final warp gated on `done`, and the state for the host.  A round ends with
        // a host round trip (~20 us of idle GPU); a cold run takes about one per re-bin.
        {
            int rc = enqueue_rebin(c, c->d_state, perm_at_start, (prewarp && batch == 0) ? &prewarp_wp : nullptr, true, 0);
            if (rc != BF_OK) return rc;
        }
        FusedLoopArgs la;
        la.sets = ev_sets(c);
        la.ftab = c->d_ftab;
        la.st = c->d_state; la.st_other = c->d_state + 1;
        la.snap = nullptr;
        la.rec = c->d_xrec; la.red = c->d_xred;
        for (int i = 0; i < 3; ++i) la.scratch[i] = c->d_xscratch[i];
        la.trace = trace;
        la.nbr = c->fgrid.nbr; la.nbc = c->fgrid.nbc;
        la.R = c->win.scale_img_x; la.C = c->win.scale_img_y;
        la.max_passes = 4096;
        la.first_warp = first_warp ? 1 : 0;
        la.tl = c->d_tl;
        {
            ProfScope ps(c, 0, c->n);
            HIP_TRY(c, launch_fused_loop(la, c->win.scale / 2, c->fgrid.TSR, c->n_cus, c->stream));
        }
        {
            ProfScope ps(c, 3);
            WarpScatterArgs fa = ws_args(c, buf, 2);
            fa.st = c->d_state;
            fa.pick_set = 1;
            fa.sorted_out = 1;
            if (o.want_uv) fa.uv = c->d_uv;
            launch_final_warp(fa, c->stream);
        }
        inf.launches += 5;
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(&c->h_state[batch & 1], c->d_state, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipEventRecord(c->poll_ev[batch & 1], c->stream));
        if (warm_start || !c->opt_blocking_poll) {
            HIP_TRY(c, hipEventSynchronize(c->poll_ev[batch & 1]));
        } else {
            int rcw = wait_event_sleeping(c, c->poll_ev[batch & 1]);
            if (rcw != BF_OK) return rcw;
        }
        inf.polls++;
        const DevState& ws = c->h_state[batch & 1];
        launched_iters = ws.last_j + 1;
        if (host_timing && batch < 40)
            fprintf(stderr, "persist round %d: it %d done %d need_rebin %d redo %d last_j %d rebins %d rc %d launches %d ovf_total %u\n", batch, ws.hot.it,
                    ws.hot.done, ws.hot.need_rebin, ws.hot.redo, ws.last_j, ws.hot.rebins, ws.rc, ws.hot.spare_, ws.ovf_total);
        if (ws.hot.done) {
            fin = ws;
            final_done = true;
            break;
        }
        if (batch > 100000) return fail(c, BF_ERR_NOCONV, "device loop did not terminate");
    }
    const bool persist_ran = final_done;
    for (int batch = 0; !persist_ran; ++batch) {
        // The re-bin kernels are device-gated (they run only if hot.need_rebin is set), but even a
        // no-op launch costs ~4.5 us here, so they are enqueued only before the first iteration and
        // when a polled snapshot shows the update asking for one.  The request is predictive
        // (0.6 x margin of drift), which covers the one-to-two batches of polling lag; anything
        // that still escapes takes the exact overflow path.
        if (binned && (batch == 0 || want_rebin)) {
            int rc = enqueue_rebin(c, state_of(launched_iters), perm_at_start, (prewarp && batch == 0) ? &prewarp_wp : nullptr, fused, launched_iters);
            if (rc != BF_OK) return rc;
            inf.launches += 3;
            want_rebin = false;
            skip_rebin_checks = 1;   // the next snapshot predates this re-bin
            last_rebin_at = launched_iters;
            if (fused && batch > 0) stall_allowance += 3 * o.poll_interval;
        }
        // A warm start (bf_set_model) converges in a handful of iterations: its first batch is short and is
        // polled at once, so that ~20 no-op launches and a second poll are not queued behind it.
        int batch_len = o.poll_interval;
        if (quick_warm) {   // one more iteration than the previous warm start needed, then two at a time
            batch_len = batch == 0 ? c->warm_iters_hint + 1 : 2;
            if (batch_len < 2) batch_len = 2;
            if (batch_len > o.poll_interval) batch_len = o.poll_interval;
        }
        for (int k = 0; k < batch_len; ++k) {
            const bool warp = first ? first_warp : true;
            const int j = launched_iters;
            if (fused) {   // warp + scatter + stencil + moments in one launch; the update at the head of the next
                FusedArgs fa;
                fa.sets = ev_sets(c);
                fa.ftab = c->d_ftab;
                fa.st_in = state_of(j); fa.st_out = state_of(j + 1);
                fa.snap = quick_warm ? nullptr : &c->h_state[0];
                fa.acc_in = acc_of(j - 1); fa.acc_out = acc_of(j); fa.acc_zero = acc_of(j + 1);
                fa.lost = lost_flag(c);
                fa.trace = trace;
                fa.nbr = c->fgrid.nbr; fa.nbc = c->fgrid.nbc;
                fa.R = c->win.scale_img_x; fa.C = c->win.scale_img_y;
                fa.j = j;
                fa.warp = warp ? 1 : 0;
                fa.tl = c->d_tl;
                ProfScope ps(c, 0, c->n);
                HIP_TRY(c, launch_fused_pass(fa, c->win.scale / 2, c->fgrid.TSR, c->stream));
                first = false;
                buf ^= 1;
                ++launched_iters;
                inf.launches += 1;
                continue;
            }
            if (binned) {
                BinScatterArgs ba;
                ba.sets = ev_sets(c);
                ba.bin_start = c->d_bin_start;
                ba.slabs = c->d_slabs;
                ba.cidx = c->d_cidx; ba.chdr = c->d_chdr;
                ba.compact = c->fmt;
                ba.ovf_plane = c->d_plane[buf]; ba.ovf_cplane = c->d_cplane[buf];
                ba.ovf_bits = c->d_ovf_bits[buf]; ba.ovf_pitch = c->ovf_pitch;
                ba.st_in = state_of(j); ba.st_out = state_of(j + 1);
                ba.acc = head_update ? acc_of(j - 1) : nullptr;
                ba.ovf_cur = ovf_of(j); ba.ovf_prev = ovf_of(j - 1);
                ba.snap = quick_warm ? nullptr : &c->h_state[0];
                ba.trace = trace;
                ba.g = c->grid;
                ba.cur = buf; ba.j = j;
                ba.tl = c->d_tl ? c->d_tl + 64 * 2 * 16 : nullptr;
                ba.m_cur = c->d_mplane[buf]; ba.m_prev = c->d_mplane[buf ^ 1];
                ba.mlist = c->d_mlist; ba.mcount = c->d_mcount; ba.mcap = c->m_cap;
                ProfScope ps(c, 0, c->n);
                HIP_TRY(c, launch_bin_warp_scatter(ba, warp, bin_threads, ev_per_thread, c->stream));
            } else {
                ProfScope ps(c, 0, c->n);
                launch_warp_scatter(ws_args(c, buf, 1), warp, true, false, c->stream);
            }
            {   // stencil + moments; its last work-group reduces and runs the model / loop update
                StencilArgs a = st_args(c, buf, 1);
                if (binned) {
                    a.ovf_bits = c->d_ovf_bits[buf]; a.zero_bits = c->d_ovf_bits[buf ^ 1]; a.ovf_pitch = c->ovf_pitch;
                    a.zero_full = j == 0 ? 1 : 0;   // (what an earlier operator left in the other buffer is not in the bitmap)
                    a.st = state_of(j + 1);
                    a.ovf_cur = ovf_of(j); a.ovf_prev = ovf_of(j - 1); a.ovf_next = ovf_of(j + 1);
                }
                if (head_update) {   // accumulate only: the update runs at the head of the next warp+scatter launch
                    a.acc = acc_of(j); a.acc_zero = acc_of(j + 1);
                } else if (binned) {   // "co_schedule": the last work-group of the stencil kernel updates
                    a.acc = c->d_acc;
                    a.ticket = c->d_ticket;
                    // it reads the state the (lean) scatter kernel read and writes the new one where the next scatter
                    // launch looks for it -- and to the pinned snapshot the host polls; nobody copies the state in between
                    a.st = state_of(j);
                    a.st_rw = state_of(j + 1);
                    a.snap = quick_warm ? nullptr : &c->h_state[0];
                } else {        // the last work-group reduces and updates
                    a.acc = c->d_acc;
                    a.ticket = c->d_ticket;
                    a.st_rw = c->d_state;
                }
                a.trace = trace;
                a.update_mode = 1;
                a.tl = c->d_tl;
                a.tl_launch = launched_iters;
                ProfScope ps(c, 1);
                launch_stencil(a, stencil_src(c, binned), c->stream);
            }
            first = false;
            buf ^= 1;
            ++launched_iters;
            inf.launches += 2;
        }
        if (quick_warm) {
            // A warm start is polled batch by batch (no pipelining: it rarely needs a second batch), and the
            // final warp rides along with every batch, gated on `done` (check_done 2) and picking the event
            // set on the device: when the batch was enough -- the usual case -- nothing is left to launch
            // after the poll (a blocking poll + launch costs ~20 us of idle GPU).
            if (head_update) {   // `done` of the batch's last iteration: apply its update now (normally the next launch would)
                launch_finish_update(state_of(launched_iters), acc_of(launched_iters - 1), ovf_of(launched_iters - 1),
                                     launched_iters, buf ^ 1, trace, &c->h_state[batch & 1], c->stream, fused ? lost_flag(c) + (launched_iters + 2) % 3 : nullptr);
                inf.launches++;
            }
            ProfScope ps(c, 3);
            WarpScatterArgs fa = ws_args(c, buf, 2);
            fa.st = state_of(binned ? launched_iters : 0);
            fa.pick_set = binned ? 1 : 0;
            fa.sorted_out = 1;
            if (o.want_uv) fa.uv = c->d_uv;
            launch_final_warp(fa, c->stream);
            inf.launches++;
        }
        HIP_TRY(c, hipGetLastError());
        if (snap_polled) {
            // Tile-binned cold run: no copy command, no event.  Whoever computes the new state -- work-group 0 of the warp+scatter
            // launch (update at its head) or the stencil kernel's last work-group (update in its tail) -- writes it to pinned
            // host memory as well; its first 8-byte word -- (done, it), one lane's store
            // -- tells the host how far the device is and whether the loop is over (`done` carries this run's tag: a
            // straggler launch of an earlier run on this context cannot be mistaken for it).  The host enqueues the next batch
            // when less than one batch is left in the queue and sleeps in between (the queue hides its wake-up latency).
            const volatile unsigned long long* w0p = reinterpret_cast<const volatile unsigned long long*>(&c->h_state[0]);
            const volatile int32_t* rebin_p = &c->h_state[0].hot.need_rebin;
            const volatile unsigned long long* lastj_p = reinterpret_cast<const volatile unsigned long long*>(&c->h_state[0].run_tag);
            bool done_seen = false;
            int gpu_it = 0;
            if (host_timing) { const double t = ht_now(); ht_launch += t - ht_mark; ht_mark = t; }
            // Invariant of this mode: the snapshot's `done` word is 0 while the loop runs and takes this run's tag --
            // nothing else -- when it ends (a straggler launch of an earlier run can only leave an older tag, which is
            // read as "not started yet": `it` 0).  The watchdog is a wall-clock deadline since the last PROGRESS of
            // the device's iteration counter, not a count of looks: a spinning poll (blocking_poll = 0) takes a few
            // nanoseconds per look, and one batch can legitimately take long (large poll_interval, 1280x720
            // iterations, several contexts sharing the GPU, a first launch loading code objects).
            auto wd_clock = [] { return std::chrono::steady_clock::now(); };
            auto wd_mark = wd_clock();
            int wd_it = -1;
            for (unsigned spins = 0;; ++spins) {
                const unsigned long long w0 = *w0p;
                const int32_t sdone = (int32_t)(uint32_t)(w0 & 0xffffffffull), sit = (int32_t)(uint32_t)(w0 >> 32);
                if (sdone == h.run_tag) { done_seen = true; break; }
                gpu_it = (sdone == 0) ? sit : 0;
                // (one-kernel iteration: progress is counted in LAUNCHES -- passes that wait for a re-bin, or repeat one,
                // do not advance the iteration counter)
                if (fused) {
                    const unsigned long long wj = *lastj_p;   // (run_tag, last_j): one 8-byte store of the device
                    gpu_it = ((int32_t)(uint32_t)(wj & 0xffffffffull) == h.run_tag) ? (int32_t)(uint32_t)(wj >> 32) + 1 : 0;
                }
                if (launched_iters - gpu_it <= o.poll_interval) break;   // less than a batch left in the queue: feed it
                if (c->opt_blocking_poll) {
                    struct timespec ts = {0, 20000};
                    nanosleep(&ts, nullptr);
                }
                if (gpu_it != wd_it) { wd_it = gpu_it; wd_mark = wd_clock(); }
                else if ((spins & 1023u) == 0 &&
                         std::chrono::duration<double>(wd_clock() - wd_mark).count() > c->opt_watchdog_s) {
                    const hipError_t e = hipStreamQuery(c->stream);
                    if (e != hipSuccess && e != hipErrorNotReady) HIP_TRY(c, e);
                    return fail(c, BF_ERR_HIP, "device loop makes no progress");
                }
            }
            if (host_timing) { const double t = ht_now(); ht_wait += t - ht_mark; ht_mark = t; }
            inf.polls++;
            if (done_seen) break;
            // (a snapshot older than the last re-bin does not count.  With the update at the scatter head the snapshot of
            // iteration count L is written by launch L itself, behind a re-bin enqueued at L; with the update in the stencil
            // tail -- and in the one-kernel loop -- it is written by launch L - 1, ahead of that re-bin)
            if (((fused || !head_update) ? gpu_it > last_rebin_at : gpu_it >= last_rebin_at) && gpu_it > 0 && *rebin_p) want_rebin = true;
            if (launched_iters - stall_allowance > (o.hard_iter_cap > 0 ? o.hard_iter_cap : INT_MAX - 64) + 3 * o.poll_interval)
                return fail(c, BF_ERR_NOCONV, "device loop did not terminate");
            continue;
        }
        if (!(quick_warm && head_update))   // (there k_finish_update has written the state to the pinned copy itself)
            HIP_TRY(c, hipMemcpyAsync(&c->h_state[batch & 1], state_of(binned ? launched_iters : 0), sizeof(DevState),
                                      hipMemcpyDeviceToHost, c->stream));
        // A cold run is polled one batch behind the launches, so its wait can sleep (the wake-up latency hides
        // behind the batch already queued) instead of burning a host core per slice context; a warm start waits
        // for the batch it has just launched and spins.
        hipEvent_t* pev = c->poll_ev;
        HIP_TRY(c, hipEventRecord(pev[batch & 1], c->stream));
        if (batch == 0 && !quick_warm) continue;
        if (quick_warm) {   // look at this batch straight away
            HIP_TRY(c, hipEventSynchronize(pev[batch & 1]));
            inf.polls++;
            const DevState& ws = c->h_state[batch & 1];
            if (ws.hot.done) {
                fin = ws;
                final_done = true;
                break;
            }
            // (a warm start's follow-up batches are two iterations long: a re-bin -- three kernels, ~30 us on a large image --
            // pays only where the overflow path would cost more, i.e. when a good part of the events took it)
            // (the one-kernel loop has no overflow path: it WAITS for the re-bin it asks for)
            if (binned && ws.hot.need_rebin && (fused || (unsigned long long)ws.last_ovf * 8ull > (unsigned long long)ws.n_events)) want_rebin = true;
            if (launched_iters - stall_allowance > (o.hard_iter_cap > 0 ? o.hard_iter_cap : INT_MAX - 64) + 3 * o.poll_interval)
                return fail(c, BF_ERR_NOCONV, "device loop did not terminate");
            continue;
        }
        if (host_timing) { const double t = ht_now(); ht_launch += t - ht_mark; ht_mark = t; }
        if (c->opt_blocking_poll) {
            int rcw = wait_event_sleeping(c, pev[(batch - 1) & 1]);
            if (rcw != BF_OK) return rcw;
        } else {
            HIP_TRY(c, hipEventSynchronize(pev[(batch - 1) & 1]));
        }
        if (host_timing) { const double t = ht_now(); ht_wait += t - ht_mark; ht_mark = t; }
        inf.polls++;
        const DevState& snap = c->h_state[(batch - 1) & 1];
        if (snap.hot.done) {
            fin = snap;
            break;
        }
        if (skip_rebin_checks > 0) --skip_rebin_checks;
        else if (binned && snap.hot.need_rebin) want_rebin = true;
        if (launched_iters - stall_allowance > (o.hard_iter_cap > 0 ? o.hard_iter_cap : INT_MAX - 64) + 3 * o.poll_interval)
            return fail(c, BF_ERR_NOCONV, "device loop did not terminate");
    }
    if (host_timing)
        fprintf(stderr, "bf_run host time: launching %.3f ms, waiting %.3f ms, %d launches\n", 1e3 * ht_launch,
                1e3 * ht_wait, (int)inf.launches);
    // final warp: the last project_4param_reinit of iteration_step (:340-344), kept so that
    // pr / nx / ny describe the converged model; n is written for compute_uv / writeout.
    if (!final_done) {
        ProfScope ps(c, 3);
        WarpScatterArgs fa = ws_args(c, buf, 0);
        fa.st = state_of(binned ? launched_iters : 0);   // (after `done` every launch keeps both buffers identical)
        fa.pick_set = binned ? 1 : 0;                    // the device knows which set holds the (tile-sorted) events
        fa.sorted_out = 1;
        if (o.want_uv) fa.uv = c->d_uv;   // Event::compute_uv (event.h:135-142) in the same pass
        launch_final_warp(fa, c->stream);
        inf.launches++;
    }
    if (snap_polled) {   // the final state, consistently: behind everything that is queued
        HIP_TRY(c, hipMemcpyAsync(&c->h_state[1], state_of(launched_iters), sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipEventRecord(c->poll_ev[1], c->stream));
        if (c->opt_blocking_poll) {
            int rcw = wait_event_sleeping(c, c->poll_ev[1]);
            if (rcw != BF_OK) return rcw;
        } else {
            HIP_TRY(c, hipEventSynchronize(c->poll_ev[1]));
        }
        fin = c->h_state[1];
    }
    if (binned) {   // the device chose which set holds the (tile-sorted) events
        c->cs = fin.hot.cs;
        c->has_perm = true;
    }
    if (warm_start) c->warm_iters_hint = fin.hot.it;
    c->n_valid = true;
    c->uv_valid = o.want_uv != 0;
    c->out_sorted = true;
    HIP_TRY(c, hipGetLastError());
    if (o.want_uv) HIP_TRY(c, hipStreamSynchronize(c->stream));

    const DevState d = fin;
    h = d;   // model, dividers, warp parameters, plane-buffer dirtiness
    h.hot.pp = 0; h.hot.redo = 0; h.hot.pend = 0;   // (the final warp left the products in the set's first array)
    if (split) {   // the last executed iteration added to margin plane b0 ^ ((it - 1) & 1), and the lists name those pixels
        if (d.hot.it > 0) c->m_dirty_plane = b0 ^ ((d.hot.it - 1) & 1);
        c->m_unknown = d.rc < 0;   // (a run stopped at the iteration cap may have one executed launch more than `it` counts)
        if (getenv("BF_DEBUG_MARGIN")) {   // entries of the bins' margin lists after the last executed launch
            std::vector<uint32_t> mc((size_t)c->m_nbins);
            HIP_TRY(c, hipMemcpy(mc.data(), c->d_mcount, mc.size() * 4, hipMemcpyDeviceToHost));
            unsigned long long tot = 0; uint32_t mx = 0;
            for (uint32_t v : mc) { tot += v; mx = v > mx ? v : mx; }
            fprintf(stderr, "margin entries after %d iterations: %llu in %d bins (max %u of %d)\n", d.hot.it, tot, c->m_nbins, mx, c->m_cap);
        }
    }
    if (binned && !fused) {   // the last iteration scattered its overflow events into buffer b0 ^ ((it - 1) & 1); the other one is clean
        h.hot.ovf_cnt[b0 ^ (d.hot.it & 1)] = 0;
        h.hot.ovf_cnt[b0 ^ (d.hot.it & 1) ^ 1] = d.last_ovf ? 1u : 0u;
    }

    // iterations executed alternate buffers starting at b0; the next scatter goes to the
    // buffer the last stencil left clean.  (The one-kernel loop touches neither plane buffer: what was dirty stays dirty,
    // hot.ovf_cnt came back from the device as it went.)
    c->cur = fused ? b0 : (b0 ^ (d.hot.it & 1));
    c->trace_valid = d.hot.it < o.trace_cap ? d.hot.it : o.trace_cap;
    inf.rc = d.rc;
    inf.iterations = d.hot.it;
    inf.x_divider = d.x_div; inf.y_divider = d.y_div;
    inf.rot_divider = d.rot_div; inf.div_divider = d.div_div;
    inf.rebins = d.hot.rebins;
    inf.overflow_events = (int32_t)(d.ovf_total > 0x7fffffffu ? 0x7fffffffu : d.ovf_total);
    if (model_out) *model_out = d.model;
    if (info) *info = inf;
    if (d.rc < 0) return fail(c, d.rc, "iteration cap (%d) reached without convergence", o.hard_iter_cap);
    return d.rc;
}

int bf_run_many(bf_ctx* const* ctxs, int32_t n, const bf_run_opts* opts, bf_model* models_out, bf_run_info* infos_out) {
    if (!ctxs || n < 0) return BF_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return BF_ERR_ARG;
        for (int k = 0; k < i; ++k)
            if (ctxs[k] == ctxs[i]) return fail(ctxs[i], BF_ERR_ARG, "bf_run_many: context %d is also context %d (a context holds one slice)", i, k);
    }
    std::vector<int> rc((size_t)n, BF_OK);
    auto one = [&](int i) {
        bf_model m;
        bf_run_info inf;
        rc[(size_t)i] = bf_run(ctxs[i], opts, &m, &inf);
        if (models_out) models_out[i] = m;
        if (infos_out) { infos_out[i] = inf; infos_out[i].rc = rc[(size_t)i]; }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n; ++i) th.emplace_back(one, i);
    if (n > 0) one(0);
    for (auto& t : th) t.join();
    for (int i = 0; i < n; ++i)
        if (rc[(size_t)i] < 0) return rc[(size_t)i];
    return BF_OK;
}

int bf_run_tiles(bf_ctx* c, const bf_tile_opts* o, bf_model* models_out, bf_run_info* infos_out) {
    if (!c || !o) return BF_ERR_ARG;
    if (!c->uploaded) return fail(c, BF_ERR_STATE, "bf_run_tiles before bf_upload_events");
    if (o->grid_rows < 1 || o->grid_cols < 1 || (long long)o->grid_rows * o->grid_cols > 16384)
        return fail(c, BF_ERR_ARG, "bad tile grid %d x %d", o->grid_rows, o->grid_cols);
    if (o->scale < 1 || o->scale % 2 == 0 || o->scale / 2 > kMaxHalfScale) return fail(c, BF_ERR_ARG, "scale must be odd");
    if (o->sensor_res_x < 1 || o->sensor_res_y < 1) return fail(c, BF_ERR_ARG, "bad sensor size");
    if (c->has_noise) return fail(c, BF_ERR_ARG, "bf_run_tiles does not take a noise mask");
    HIP_TRY(c, hipSetDevice(c->device));
    const int nt = o->grid_rows * o->grid_cols;
    // second event set + permutation (shared with the tile-binned scatter)
    if (!c->set[1].xy) {
        HIP_TRY(c, hipMalloc(&c->set[1].xy, (size_t)c->cap_events * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->set[1].t, (size_t)c->cap_events * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->set[1].p, (size_t)c->cap_events * sizeof(float2)));
    }
    for (int i = 0; i < 2; ++i)
        if (!c->set[i].perm) HIP_TRY(c, hipMalloc(&c->set[i].perm, (size_t)c->cap_events * sizeof(uint32_t)));
    if (nt > c->tiles_alloc) {
        void* old[] = {c->d_tile_hist, c->d_tile_start, c->d_tile_cursor, c->d_tile_states};
        for (void* p : old) if (p) HIP_TRY(c, hipFree(p));
        c->d_tile_hist = c->d_tile_start = c->d_tile_cursor = nullptr;
        c->d_tile_states = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_tile_hist, (size_t)(nt + 1) * 4));
        HIP_TRY(c, hipMalloc(&c->d_tile_start, (size_t)(nt + 1) * 4));
        HIP_TRY(c, hipMalloc(&c->d_tile_cursor, (size_t)(nt + 1) * 4));
        HIP_TRY(c, hipMalloc(&c->d_tile_states, (size_t)nt * sizeof(DevState)));
        HIP_TRY(c, hipMemsetAsync(c->d_tile_hist, 0, (size_t)(nt + 1) * 4, c->stream));
        c->tiles_alloc = nt;
    }
    // LDS image capacity: the largest window a tile can have
    const int tr = (o->sensor_res_x + o->grid_rows - 1) / o->grid_rows + 1;
    const int tc = (o->sensor_res_y + o->grid_cols - 1) / o->grid_cols + 1;
    const long long max_px = (long long)(o->scale * tr + o->scale) * (o->scale * tc + o->scale);
    if (max_px * 16 > 156 * 1024)
        return fail(c, BF_ERR_CAPACITY, "a tile window of up to %lld pixels does not fit the LDS", max_px);

    DevState tmpl;
    memset(&tmpl, 0, sizeof(tmpl));
    tmpl.x_div = tmpl.y_div = 1.0f;           // optimizer_rolling.h:61-63
    tmpl.rot_div = tmpl.div_div = 10000.0f;
    tmpl.max_iter = o->max_iter;
    tmpl.hard_cap = o->hard_iter_cap;
    tmpl.hot.wp = identity_warp();
    launch_fill_states(c->d_tile_states, tmpl, nt, c->stream);

    TileGrid g;
    g.rows = o->grid_rows; g.cols = o->grid_cols; g.res_x = o->sensor_res_x; g.res_y = o->sensor_res_y;
    const bf_ctx::EvSet& src = c->set[c->cs];
    const bf_ctx::EvSet& dst = c->set[c->cs ^ 1];
    {
        ProfScope ps(c, 3);
        launch_tile_sort(src.xy, src.t, c->has_perm ? src.perm : nullptr, c->n, g, c->d_tile_hist, c->d_tile_start,
                         c->d_tile_cursor, dst.xy, dst.t, dst.p, dst.perm, c->stream);
    }
    c->cs ^= 1;
    c->has_perm = true;
    TileArgs a;
    a.xy = dst.xy; a.t = dst.t; a.p = dst.p; a.perm = dst.perm;
    a.nxny = c->d_nxny;
    a.tile_start = c->d_tile_start;
    a.states = c->d_tile_states;
    a.scale = o->scale;
    a.seed_res_x = o->sensor_res_x; a.seed_res_y = o->sensor_res_y;
    a.guard_res_x = o->guard_res_x; a.guard_res_y = o->guard_res_y;
    a.min_events = o->min_events;
    a.max_px = (int32_t)max_px;
    {
        ProfScope ps(c, 0, c->n);
        if (launch_tile_optimizer(a, nt, c->stream) != 0) return fail(c, BF_ERR_HIP, "cannot configure the tile kernel");
    }
    HIP_TRY(c, hipGetLastError());
    std::vector<DevState> st((size_t)nt);
    HIP_TRY(c, hipMemcpyAsync(st.data(), c->d_tile_states, (size_t)nt * sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < nt; ++i) {
        if (models_out) models_out[i] = st[(size_t)i].model;
        if (infos_out) {
            bf_run_info inf;
            memset(&inf, 0, sizeof(inf));
            inf.rc = st[(size_t)i].rc;
            inf.iterations = st[(size_t)i].hot.it;
            inf.x_divider = st[(size_t)i].x_div; inf.y_divider = st[(size_t)i].y_div;
            inf.rot_divider = st[(size_t)i].rot_div; inf.div_divider = st[(size_t)i].div_div;
            infos_out[i] = inf;
        }
    }
    c->p_clean = false;
    c->n_valid = true;
    c->uv_valid = false;
    c->out_sorted = false;
    c->pending_warp = false;
    c->have_window = true;    // per-event read-back (bf_compute_uv / bf_writeout_events) is valid now
    c->degenerate = false;
    c->use_binned = false;    // the events are now sorted by sensor tile, not by image tile
    c->fused_ok = false;
    return BF_OK;
}

// ---- OptimizerLocal: the contrast-score optimiser (optimizer_sampler.h / .cpp) ------------------

namespace {

// Event::project -> apply_project (event.h:65-70,164-168) of one event on the host (the centre
// event of the window); this file is compiled with -ffp-contract=off like the kernels.
void project_one(int32_t fr_x, int32_t fr_y, int64_t t, float kx, float ky, double* pr_x, double* pr_y) {
    const float ft = (float)t;
    const float px = kx * ft, py = ky * ft;
    *pr_x = (double)(float)fr_x - (double)px / 10000.0;
    *pr_y = (double)(float)fr_y - (double)py / 10000.0;
}

int local_step(bf_ctx* c, double nx, double ny, double* score, bool want_img) {
    const bf_local_window& w = c->lwin;
    LocalGeom g;
    memset(&g, 0, sizeof(g));
    g.scale = w.scale; g.wsx = w.metric_wsizex; g.wsy = w.metric_wsizey;
    g.R = w.scale_img_x; g.C = w.scale_img_y;
    g.kx = (float)((double)(float)nx / 127.0);   // event.h:164-165, nz is the double 127
    g.ky = (float)((double)(float)ny / 127.0);
    double cpx, cpy;
    project_one(w.c_fr_x, w.c_fr_y, w.c_t, g.kx, g.ky, &cpx, &cpy);           // optimizer_sampler.cpp:122
    g.x_shift = -cpx * (double)w.scale + (double)w.metric_wsizex / 2.0;        // :126
    g.y_shift = -cpy * (double)w.scale + (double)w.metric_wsizey / 2.0;        // :127
    const bf_ctx::EvSet& e = c->set[c->cs];
    HIP_TRY(c, hipMemsetAsync(c->d_lscore, 0, 2 * sizeof(unsigned long long), c->stream));
    launch_local_project_count(e.xy, e.t, c->n, g, c->d_lplane[c->lcur], c->stream);
    if (launch_local_blur_score(c->d_lplane[c->lcur], c->d_lplane[c->lcur ^ 1], g, c->d_lscore,
                                want_img ? c->d_limg : nullptr, c->stream) != 0)
        return fail(c, BF_ERR_ARG, "the 8-bit Gaussian is defined for scale <= 7 (got %d)", w.scale);
    c->lcur ^= 1;
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_lscore, c->d_lscore, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // get_event_score, optimizer_sampler.cpp:192-205 (integer sums are exact in a double far beyond any image)
    *score = c->h_lscore[1] == 0 ? 0.0 : (double)c->h_lscore[0] / (double)c->h_lscore[1];
    return BF_OK;
}

}  // namespace

int bf_local_set_window(bf_ctx* c, int32_t scale, int32_t wsz, int32_t c_fr_x, int32_t c_fr_y, int64_t c_t,
                        bf_local_window* window_out) {
    if (!c) return BF_ERR_ARG;
    if (!c->uploaded) return fail(c, BF_ERR_STATE, "bf_local_set_window before bf_upload_events");
    if (scale < 1 || scale % 2 == 0 || scale > 7)   // optimizer_sampler.cpp:206 (odd); the Gaussian is stated to 7
        return fail(c, BF_ERR_ARG, "scale must be odd and <= 7 (got %d)", scale);
    HIP_TRY(c, hipSetDevice(c->device));
    bf_local_window w;
    memset(&w, 0, sizeof(w));
    w.scale = scale;
    if (wsz <= 0) {   // OptimizerLocal(events, scale), optimizer_sampler.h:35-48
        if (c->n <= 0) return fail(c, BF_ERR_STATE, "the bounding box of an empty cloud is undefined");
        int rc = fold_stats(c);
        if (rc != BF_OK) return rc;
        const SliceStats& s = c->stats;
        w.metric_wsizex = scale * (s.xmax - s.xmin);
        w.metric_wsizey = scale * (s.ymax - s.ymin);
        w.c_fr_x = (s.xmax - s.xmin) / 2 + s.xmin;
        w.c_fr_y = (s.ymax - s.ymin) / 2 + s.ymin;
        w.c_t = 0;
    } else {          // OptimizerLocal(events, e, scale, wsz), :29-33
        w.metric_wsizex = scale * wsz;
        w.metric_wsizey = scale * wsz;
        w.c_fr_x = c_fr_x; w.c_fr_y = c_fr_y; w.c_t = c_t;
    }
    w.scale_img_x = w.metric_wsizex + scale;   // optimizer_sampler.cpp:208-209
    w.scale_img_y = w.metric_wsizey + scale;
    if ((size_t)w.scale_img_x * (size_t)w.scale_img_y > c->cap_px)
        return fail(c, BF_ERR_CAPACITY, "window %d x %d exceeds the image capacity", w.scale_img_x, w.scale_img_y);
    if (!c->d_lplane[0]) {
        for (int i = 0; i < 2; ++i) HIP_TRY(c, hipMalloc(&c->d_lplane[i], c->cap_px * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_lscore, 2 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMalloc(&c->d_limg, c->cap_px));
        HIP_TRY(c, hipHostMalloc(&c->h_lscore, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    }
    // a new window lays the planes out afresh
    for (int i = 0; i < 2; ++i) HIP_TRY(c, hipMemsetAsync(c->d_lplane[i], 0, c->cap_px * sizeof(uint32_t), c->stream));
    c->lcur = 0;
    c->lwin = w;
    c->have_lwin = true;
    if (window_out) *window_out = w;
    return BF_OK;
}

int bf_local_iteration_step(bf_ctx* c, double nx, double ny, double* score, uint8_t* img_out) {
    if (!c || !score) return BF_ERR_ARG;
    if (!c->have_lwin) return fail(c, BF_ERR_STATE, "bf_local_iteration_step before bf_local_set_window");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = local_step(c, nx, ny, score, img_out != nullptr);
    if (rc != BF_OK) return rc;
    if (img_out)
        HIP_TRY(c, hipMemcpy(img_out, c->d_limg, (size_t)c->lwin.scale_img_x * (size_t)c->lwin.scale_img_y,
                             hipMemcpyDeviceToHost));
    return BF_OK;
}

int bf_local_run(bf_ctx* c, int32_t res_x, int32_t res_y, int64_t max_evaluations, bf_local_state* out) {
    if (!c || !out) return BF_ERR_ARG;
    if (!c->have_lwin) return fail(c, BF_ERR_STATE, "bf_local_run before bf_local_set_window");
    HIP_TRY(c, hipSetDevice(c->device));
    const bf_local_window& w = c->lwin;
    bf_local_state st;
    memset(&st, 0, sizeof(st));
    st.dnx = 0.01; st.dny = 0.01;   // optimizer_sampler.cpp:7
    // (NZ * T_DIVIDER * 1000.0) / (10 * scale * FROM_MS(MAX_TIME_MS)), common.h:36,49,60,64
    st.dn_th = (127 * 1 * 1000.0) / (double)(10ull * (unsigned long long)w.scale * 100000000ull);
    *out = st;
    if ((w.scale_img_x < w.scale * res_x / 15) && (w.scale_img_y < w.scale * res_y / 15)) return BF_SKIPPED;   // :9-13
    int rc = local_step(c, st.nx, st.ny, &st.last_score, false);   // :16
    if (rc != BF_OK) return rc;
    st.evaluations = 1;
    while (std::hypot(st.dnx, st.dny) > st.dn_th) {   // :20
        {   // compute_new_nx, :90-102
            const double nx_new = st.nx + st.dnx;
            double new_score;
            if ((rc = local_step(c, nx_new, st.ny, &new_score, false)) != BF_OK) return rc;
            const double dscore = new_score - st.last_score;
            st.last_score = new_score;
            if (dscore <= 0) st.dnx = -st.dnx / 2.0;
            st.nx = nx_new;
        }
        {   // compute_new_ny, :105-117
            const double ny_new = st.ny + st.dny;
            double new_score;
            if ((rc = local_step(c, st.nx, ny_new, &new_score, false)) != BF_OK) return rc;
            const double dscore = new_score - st.last_score;
            st.last_score = new_score;
            if (dscore <= 0) st.dny = -st.dny / 2.0;
            st.ny = ny_new;
        }
        st.evaluations += 2;
        if (max_evaluations > 0 && st.evaluations >= max_evaluations) {
            *out = st;
            return fail(c, BF_ERR_NOCONV, "evaluation cap (%lld) reached", (long long)max_evaluations);
        }
    }
    *out = st;
    return BF_OK;
}

int bf_projection_img(bf_ctx* c, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final, uint8_t* img_out) {
    if (!c || !img_out) return BF_ERR_ARG;
    if (!c->uploaded) return fail(c, BF_ERR_STATE, "bf_projection_img before bf_upload_events");
    if (scale < 1 || scale % 2 == 0 || scale > 7) return fail(c, BF_ERR_ARG, "scale must be odd and <= 7 (got %d)", scale);
    if (res_x < 2 || res_y < 2) return fail(c, BF_ERR_ARG, "bad sensor size");
    const size_t px = (size_t)res_x * scale * (size_t)res_y * scale;
    if (px > c->cap_px) return fail(c, BF_ERR_CAPACITY, "image %d x %d exceeds the image capacity", res_x * scale, res_y * scale);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = flush_pending(c);   // a pending bf_set_model warp moves the events first
    if (rc != BF_OK) return rc;
    if (!c->d_lplane[0]) {
        for (int i = 0; i < 2; ++i) HIP_TRY(c, hipMalloc(&c->d_lplane[i], c->cap_px * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_lscore, 2 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMalloc(&c->d_limg, c->cap_px));
        HIP_TRY(c, hipHostMalloc(&c->h_lscore, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    }
    // the point planes are shared with the contrast-score path: lay them out afresh for this geometry
    for (int i = 0; i < 2; ++i) HIP_TRY(c, hipMemsetAsync(c->d_lplane[i], 0, c->cap_px * sizeof(uint32_t), c->stream));
    c->have_lwin = false;
    c->lcur = 0;
    HIP_TRY(c, hipMemsetAsync(c->d_lscore, 0, 2 * sizeof(unsigned long long), c->stream));
    const bf_ctx::EvSet& e = c->set[c->cs];
    launch_proj_count(e.xy, e.p, c->has_noise ? c->d_noise : nullptr, c->n, scale, res_x, res_y, show_final ? 1 : 0,
                      c->d_lplane[0], c->stream);
    LocalGeom g;
    memset(&g, 0, sizeof(g));
    g.scale = scale; g.R = res_x * scale; g.C = res_y * scale;
    if (launch_local_blur_score(c->d_lplane[0], c->d_lplane[1], g, c->d_lscore, c->d_limg, c->stream) != 0)
        return fail(c, BF_ERR_ARG, "unsupported scale %d", scale);
    launch_proj_scale(c->d_limg, (long long)px, c->d_lscore, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(img_out, c->d_limg, px, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

int bf_color_time_img(bf_ctx* c, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final, uint8_t* bgr_out) {
    if (!c || !bgr_out) return BF_ERR_ARG;
    if (!c->uploaded) return fail(c, BF_ERR_STATE, "bf_color_time_img before bf_upload_events");
    if (scale == 0) scale = 11;   // event_file.h:650
    if (scale < 1 || scale > 15) return fail(c, BF_ERR_ARG, "scale must be in 1..15 (got %d)", scale);
    if (res_x < 1 || res_y < 1) return fail(c, BF_ERR_ARG, "bad sensor size");
    ColorGeom g;
    memset(&g, 0, sizeof(g));
    g.scale = scale; g.show_final = show_final ? 1 : 0;
    g.mx = scale * res_x; g.my = scale * res_y;
    g.R = g.mx + scale; g.C = g.my + scale;
    const size_t px = (size_t)g.R * (size_t)g.C;
    if (px > c->cap_px) return fail(c, BF_ERR_CAPACITY, "image %d x %d exceeds the image capacity", g.R, g.C);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = flush_pending(c);   // a pending bf_set_model warp moves the events first
    if (rc != BF_OK) return rc;
    if (c->n > 0) {
        rc = fold_stats(c);
        if (rc != BF_OK) return rc;
        g.t_min = c->stats.tmin;                                             // :659-662: t_max starts at 0
        g.t_range = std::max<long long>(c->stats.tmax, 0) - g.t_min;
    }
    g.x_shift = -double(res_x / 2) * double(scale) + double(g.mx) / 2.0;     // :677-678 with x_min = 0, x_max = RES_X
    g.y_shift = -double(res_y / 2) * double(scale) + double(g.my) / 2.0;
    if (!c->d_col_planes) {
        HIP_TRY(c, hipMalloc(&c->d_col_planes, c->cap_px * 20));   // 2 x i64 sums + u32 count per pixel
        HIP_TRY(c, hipMalloc(&c->d_col_img, c->cap_px * 3));
    }
    HIP_TRY(c, hipMemsetAsync(c->d_col_planes, 0, px * 20, c->stream));
    const bf_ctx::EvSet& e = c->set[c->cs];
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(c->d_col_planes);
    launch_color_time(e.xy, e.t, e.p, c->has_noise ? c->d_noise : nullptr, c->n, g,
                      reinterpret_cast<uint32_t*>(sums + 2 * px), sums, sums + px, c->d_col_img, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(bgr_out, c->d_col_img, px * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

int bf_get_trace(bf_ctx* c, bf_trace_rec* out, int32_t cap, int32_t* written) {
    if (!c || !out || cap < 0) return BF_ERR_ARG;
    int n = c->trace_valid < cap ? c->trace_valid : cap;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n > 0) {
        HIP_TRY(c, hipMemcpyAsync(out, c->d_trace, (size_t)n * sizeof(bf_trace_rec), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (written) *written = n;
    return BF_OK;
}

// ---- raw device buffers -----------------------------------------------------------------------

int bf_device_malloc(bf_ctx* c, int64_t bytes, void** out) {
    if (!c || !out || bytes <= 0) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMalloc(out, (size_t)bytes));
    return BF_OK;
}

int bf_device_free(bf_ctx* c, void* ptr) {
    if (!c) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (ptr) HIP_TRY(c, hipFree(ptr));
    return BF_OK;
}

int bf_memcpy_h2d(bf_ctx* c, void* dst, const void* src, int64_t bytes) {
    if (!c || !dst || !src || bytes < 0) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return BF_OK;
}

// ---- measurement ---------------------------------------------------------------------------

int bf_profile_enable(bf_ctx* c, int32_t mode) {
    if (!c || mode < 0 || mode > 1) return BF_ERR_ARG;
    int rc = prof_fold(c);
    c->prof_mode = mode;
    return rc;
}

int bf_profile_reset(bf_ctx* c) {
    if (!c) return BF_ERR_ARG;
    int rc = prof_fold(c);
    memset(&c->prof, 0, sizeof(c->prof));
    return rc;
}

int bf_profile_get(bf_ctx* c, bf_profile* out) {
    if (!c || !out) return BF_ERR_ARG;
    int rc = prof_fold(c);
    *out = c->prof;
    return rc;
}

int bf_copy_bandwidth(bf_ctx* c, int64_t bytes, int32_t reps, double* gbps_out) {
    if (!c || !gbps_out || bytes < 4096 || reps < 1) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    bytes &= ~(int64_t)15;
    void *a = nullptr, *b = nullptr;
    HIP_TRY(c, hipMalloc(&a, (size_t)bytes));
    HIP_TRY(c, hipMalloc(&b, (size_t)bytes));
    HIP_TRY(c, hipMemsetAsync(a, 1, (size_t)bytes, c->stream));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    // the ceiling is the best of a few launch shapes (work-groups per CU, plain / non-temporal accesses), each warmed up
    double best = 0.0;
    for (int nt = 0; nt < 2; ++nt)
        for (int blocks : {1024, 2048, 4096, 8192}) {
            launch_copy(a, b, bytes, blocks, nt != 0, c->stream);   // warm-up
            for (int r = 0; r < reps; ++r) {
                HIP_TRY(c, hipEventRecord(e0, c->stream));
                launch_copy(a, b, bytes, blocks, nt != 0, c->stream);
                HIP_TRY(c, hipEventRecord(e1, c->stream));
                HIP_TRY(c, hipEventSynchronize(e1));
                float ms = 0.f;
                HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
                const double g = 2.0 * (double)bytes / ((double)ms * 1e-3) / 1e9;
                if (g > best) best = g;
            }
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    *gbps_out = best;
    return BF_OK;
}

}  // extern "C"
