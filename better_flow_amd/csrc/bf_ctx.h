// bf_ctx.h -- private to the C-ABI implementation files (bf_context.cpp, bf_upload.cpp, bf_operators.cpp, bf_run.cpp,
// bf_extras.cpp): the context structure behind `bf_ctx`, and the small helpers they share (error text, profiling brackets,
// kernel-argument builders, buffer management).  Nothing here is part of the ABI (include/bf_accel.h).
#pragma once
#pragma clang diagnostic ignored "-Wunused-function"   // (every file uses its own subset of the helpers below)
#include <functional>
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

// Contexts alive per device in this process (defined in bf_context.cpp).  The persistent loop kernel needs every one of its
// work-groups resident at once; two such kernels from two contexts could each hold part of the CUs and wait for the rest,
// so a context takes it only while it is the one context on its device.
extern std::atomic<int> g_live_ctx[64];

namespace {

struct ProfRec {
    hipEvent_t a, b;
    int cat;
    long long nev;
};

}  // namespace

// Margins of the tile-binned loops, in scaled pixels: how far an event may drift from where the counting sort found it before
// its bin must be re-sorted.  Swept flat over 4 .. 8 in rounds 3 and 4 (EXPERIMENTS: 195.9 / 195.1 / 193.1 Mevents/s at 8 / 6 / 4),
// so no longer an option.  The tile shape, the scatter work-groups' size and their events per thread are chosen per slice
// (bf_set_cloud, bf_run): the options that overrode them went with it.
constexpr int kBinMargin = 8;      // two-kernel loop: D of BinGrid (capped at half the smaller tile side)
constexpr int kFusedMargin = 8;    // one-kernel loops: D on top of the stencil halo H = scale / 2 + 1

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
    bool opt_co_schedule = false;    // several slice contexts share the GPU: the update runs in the stencil kernel's last work-group
    int opt_bin_pack_limit = 64;     // bits available to the per-bin packing (lower only to test the fallback)
    int n_cus = 0;
    bool use_binned = false;         // decided per slice in bf_set_cloud
    BinGrid grid;
    // one-kernel iteration (k_fused_pass): the loop of a context that has the GPU to itself
    int opt_fused = 1;               // 0 never, 1 where it is the faster loop (small slices on small images; sparser ones only when
                                     // the context is co-scheduled with others), 2 whenever possible
    bool fused_ok = false;           // decided per slice in bf_set_cloud
    bool fused_shared = false;       // ... and it is also the loop to take when the context shares the GPU ("co_schedule")
    BinGrid fgrid;                   // its sort grid: keys = (tile, zone)
    uint32_t* d_ftab = nullptr;      // FusedTab per tile
    int ftab_alloc = 0;
    // persistent form of that loop (k_fused_loop, bf_loop.hip): a context ALONE on the GPU keeps the work-groups resident
    bool counted = false;            // in g_live_ctx
    int opt_persist = 1;             // 0 never, 1 for warm-started runs of the one-kernel loop on a context that is not co-scheduled, 2 cold runs too
    unsigned long long *d_xrec = nullptr, *d_xred = nullptr;   // exchange records of the sub-tiles / of the reducers (two parities each)
    unsigned long long* d_verdict = nullptr;   // (launch id << 2) | COMMIT / ABORT of the persistent kernel's current launch (bf_loop.hip)
    int* h_broken = nullptr;         // pinned: set by the device if a committed launch could not be read back (never observed)
    int xrec_alloc = 0;              // records per parity d_xrec holds
    float2* d_xscratch[4] = {nullptr, nullptr, nullptr, nullptr};       // private product arrays of the strips' readers
    // a launch that gave up (something else holds part of the GPU) costs 0.2 s: the context then leaves the persistent kernel
    // alone for persist_backoff runs (1, 2, 4, ... 64; a launch that completes resets it)
    int persist_skip = 0, persist_backoff = 0;
    long long persist_giveups = 0;   // bf_get_stat "persist_giveups"
    // test hooks, read from the environment ONCE at bf_create (BF_DEBUG_PERSIST_ABORT / _MUTE / _SPLIT=<pass>[,late])
    int dbg_persist_abort = -1, dbg_persist_mute = -1, dbg_persist_split = -1, dbg_persist_split_late = 0;
    int dbg_margin = 0;              // BF_DEBUG_MARGIN=<n>: both loops' margin (tests: margins of 1 .. 4 pixels make events outrun their bins)
    uint16_t* d_binid = nullptr;
    uint32_t *d_hist_cnt = nullptr, *d_bin_start = nullptr, *d_cursor = nullptr;
    uint32_t* d_armed = nullptr;
    unsigned long long* d_slabs = nullptr;
    uint16_t* d_cidx = nullptr;      // compact lists: pixel index per entry (same slot count as d_slabs)
    uint32_t* d_chdr = nullptr;      // compact lists: entries per bin
    int fmt = 0;                     // what this slice's scatter hands to the stencil: 0 dense slabs, 2 event lists, 3 own pixels + margin plane (bf_set_cloud)
    int opt_bin_compact = 1;         // 0 never, 1 when the image is sparse (decided per iteration on the device), 2 always
    // interior + margin format of a dense slice (fmt 3, bf_scatter.hip: flush_split): 0 never, 1 when it is the faster one, 2 always
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
    void* d_ltile = nullptr;         // bf_local_run_tiles: the windows' states, then their return codes
    int ltile_alloc = 0;
    void* d_many_args = nullptr;     // bf_run_tiles_many (lead context): the slices' launch arguments + the claim counter
    void* h_many_args = nullptr;     // ... and their pinned staging copy
    int many_alloc = 0;
    int32_t *d_in_x = nullptr, *d_in_y = nullptr, *d_in_t = nullptr;
    // streaming: a second staging slot, a copy stream and one event per slot
    int32_t* d_in2[3] = {nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done[2] = {nullptr, nullptr};
    hipEvent_t staged[2] = {nullptr, nullptr};   // the staging kernels that read a slot have run (compute stream)
    bool staged_valid[2] = {false, false};
    long long pending_n[2] = {0, 0};
    bool pending_ts64[2] = {false, false};       // slot holds absolute 64-bit timestamps (ring hand-off)
    bool pending_ts32[2] = {false, false};       // ... of which only the low 32 bits were sent (bf_upload_ring16t32_async)
    bool pending_addr16[2] = {false, false};     // ... and 16-bit addresses in d_in16 (bf_upload_ring16_async)
    bool pending_noise[2] = {false, false};      // ... and Event::noise flags in d_in_noise
    uint16_t* d_in16[2] = {nullptr, nullptr};    // row[cap_events] then col[cap_events]
    uint8_t* d_in_noise[2] = {nullptr, nullptr};
    unsigned long long pending_t0[2] = {0, 0};
    unsigned long long* d_in_ts[2] = {nullptr, nullptr};
    int pend_head = 0, pend_count = 0;   // FIFO of pending async uploads (slot = index & 1)
    // Early staging (round 6): an asynchronous upload WITHOUT a noise ring runs its staging kernels (widening, k_prepare)
    // on the COPY stream, right behind its copies, into the slot's own event arrays and its own pinned statistics record --
    // under the previous slice's solve.  bf_commit_upload then only swaps those arrays with set[0]'s (pointers) and makes the
    // compute stream wait for the slot's `prepared` event: no staging kernel and no statistics round trip are left on a
    // warm-started chain's critical path (~35 us of a ~230 us slice at 346x260).
    EvSet inc[2];                                 // xy, t, p of the slice staged in slot i
    SliceStats* h_stats_slot[2] = {nullptr, nullptr};   // pinned: k_prepare's per-work-group records of slot i
    hipEvent_t prepared[2] = {nullptr, nullptr};  // slot i's staging kernels have run (copy stream)
    hipEvent_t inc_free[2] = {nullptr, nullptr};  // the arrays swapped INTO inc[i] at a commit are free (compute stream is past that commit)
    bool inc_free_valid[2] = {false, false};
    bool pending_early[2] = {false, false};
    // "defer_uploads": an asynchronous upload only takes its slot and remembers what to copy; its HIP calls (three copies, the
    // staging kernels, the events: ~25 us of host time) are issued by the next bf_run once that run's first batch of kernels is in
    // the queue -- or by whoever needs the slot sooner (bf_commit_upload, bf_wait_uploads).  For a single-threaded caller
    // driving one warm-started chain those 25 us otherwise sit between two runs, with the GPU idle.
    int opt_sep_update = 1;          // co-scheduled contexts: the update as a kernel of its own -- 0 never, 1 for event lists, 2 always (bf_run.cpp)
    bool opt_defer_uploads = false;
    std::function<int()> deferred[2];
    std::mutex stats_mu;                          // fold_stats: bf_set_cloud's thread and an uploading thread (stage_early's guard) may both fold
    const SliceStats* stats_src = nullptr;        // where fold_stats reads (h_stats, or the committed slot's record)
    hipEvent_t stats_event = nullptr;             // ... once this event has completed (null: the compute stream)
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
    unsigned long long* h_seq = nullptr;   // pinned: "snapshot complete" sequence number of a warm start's batch (k_finish_update)
    unsigned long long seq_counter = 0;
    hipEvent_t poll_ev[2] = {nullptr, nullptr};
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
        HIP_TRY(c, hipMalloc(&c->d_chdr, nb * 580 * sizeof(uint32_t)));   // event lists: 3 LR + 1 <= 577 key offsets per bin ((column zone, row) keys)
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
    // (one folder at a time: with early staging the thread that uploads the NEXT slice into this record's slot folds the current
    // slice's statistics first if nobody has -- bf_upload.cpp: stage_early -- while bf_set_cloud may be doing the same)
    std::lock_guard<std::mutex> lk(c->stats_mu);
    if (c->stats_valid) return BF_OK;
    // (the records are in pinned host memory once k_prepare has completed: on the compute stream, or -- early staging -- on the
    // copy stream, usually long ago)
    if (c->stats_event) HIP_TRY(c, hipEventSynchronize(c->stats_event));
    else HIP_TRY(c, hipStreamSynchronize(c->stream));
    const SliceStats* src = c->stats_src ? c->stats_src : c->h_stats;
    SliceStats s = src[0];
    for (int k = 1; k < kPrepBlocks; ++k) {
        const SliceStats& q = src[k];
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

// the HIP side of uploads that were only recorded ("defer_uploads"), oldest first
int issue_deferred_uploads(bf_ctx* c) {
    for (int k = 0; k < c->pend_count; ++k) {
        const int slot = (c->pend_head + k) & 1;
        if (c->deferred[slot]) {
            std::function<int()> f;
            f.swap(c->deferred[slot]);
            const int rc = f();
            if (rc != BF_OK) return rc;
        }
    }
    return BF_OK;
}

// copy stream, its events and the second staging slot of the asynchronous uploads (created on first use)
int streaming_setup(bf_ctx* c) {
    if (c->copy_stream) return BF_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->copy_done[i], hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->staged[i], hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) HIP_TRY(c, hipMalloc(&c->d_in2[i], (size_t)c->cap_events * sizeof(int32_t)));
    for (int i = 0; i < 2; ++i) {   // early staging: the slots' own event arrays, statistics records and events
        HIP_TRY(c, hipEventCreateWithFlags(&c->prepared[i], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->inc_free[i], hipEventDisableTiming));
        HIP_TRY(c, hipMalloc(&c->inc[i].xy, (size_t)c->cap_events * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->inc[i].t, (size_t)c->cap_events * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->inc[i].p, (size_t)c->cap_events * sizeof(float2)));
        HIP_TRY(c, hipHostMalloc(&c->h_stats_slot[i], kPrepBlocks * sizeof(SliceStats), hipHostMallocDefault));
    }
    return BF_OK;
}


}  // namespace
