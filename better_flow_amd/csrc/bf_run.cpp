// bf_run.cpp -- C-ABI: the fused OptimizerRolling::set_model / run (optimizer_rolling.h:48-125,289-347): the host side of the device loops
// (two-kernel tile-binned loop, one-kernel iteration, persistent loop kernel, global-atomic fallback) and bf_run_many.
#include "bf_ctx.h"

extern "C" {

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
    bool persist = fused && !c->opt_co_schedule && (c->opt_persist == 2 || (c->opt_persist == 1 && c->pending_warp)) &&
                   g_live_ctx[c->device & 63].load() == 1 &&
                   fused_loop_resident(c->win.scale / 2, c->fgrid.TSR, c->n_cus, c->fgrid.nbr * c->fgrid.nbc);   // (else: one launch per iteration)
    // (g_live_ctx only knows this process: another process's kernels -- or anything else that keeps work-groups from becoming
    // resident -- shows as a launch that gives up after 0.2 s.  The context then stays away from the kernel for a while.)
    if (persist && c->persist_skip > 0) { --c->persist_skip; persist = false; }
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
    h.hot.spare_ = 0;   // launches of the persistent loop kernel completed in THIS run (with run_tag: the launch's id)
    // (the persistent loop re-bins AT the request -- it returns for it --, the other loops one or two batches of launches
    // after it: the same effective threshold)
    if (binned) h.drift_limit = c->opt_bin_predict ? (persist ? 0.85 : 0.6) * (double)(fused ? c->fgrid.D : c->grid.D) : 1e300;
    // A warm start whose warp is fused into the first counting sort: the bins are built for the positions THAT warp gives,
    // so it is the reference the drift bound measures from (k_bin_scan: ref_wp <- hot.wp; the first pass does not warp, and the
    // first update overwrites hot.wp).  With the identity there, the first update -- whose warp is the previous model's plus
    // one small step -- looked like a jump of the whole flow and asked for a re-bin right after the one just made: one more
    // counting sort per warm slice, and in the persistent loop one more launch with its host round trip.
    if (prewarp) h.hot.wp = prewarp_wp;
    else if (!first_warp) h.hot.wp = identity_warp();
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
    bool sep_update = false;   // (decided below, with the update's home; then ONE state buffer)
    auto state_of = [&](int j) { return c->d_state + (sep_update ? 0 : (j & 1)); };
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
    // A third home ("sep_update", round 6): contexts that share the GPU run the lean scatter kernel and a stencil kernel that
    // only ACCUMULATES (no drain of its atomics, no ticket, no serial tail), and the update is a kernel of its own
    // (k_finish_update: one wave) ahead of every scatter launch, on ONE state buffer (nobody reads the state while that kernel
    // writes it).  Bookkeeping -- accumulator parities, overflow slots, when a snapshot is behind a re-bin -- is the head
    // form's.  A stencil work-group that has to see its fifteen atomics acknowledged and then wait for its ticket holds its LDS and a
    // wave slot ~1 us longer -- 10 % of its life: with thousands of work-groups per launch (event lists: 8100 tiles at 1280x720)
    // the third launch per iteration is the cheaper way (stencil kernel 42.7 -> 37.9 us under co_schedule, config 5's batch +2.7 %);
    // with a few hundred (config 2: 752) the launch costs more than the tickets (bench 206.7 -> 200.9): "auto" takes it for event
    // lists only.  Same bits either way.
    sep_update = binned && !fused && !head_update && (c->opt_sep_update == 2 || (c->opt_sep_update == 1 && c->fmt == 2));
    const bool head_like = head_update || sep_update;
    if (head_like) c->acc_dirty = true;   // (the sums of the last iteration are consumed, not cleared)
    // events a scatter thread keeps in flight: one pass should cover a bin of 1.5 x the average size
    // (and its work-group size: 1024 threads for bins of thousands of events, 512 where a bin holds a few hundred --
    // large images --, so that twice as many bins are in flight per CU: 84 instead of 91 us per iteration at 1280x720)
    int ev_per_thread = 8;
    const double ev_per_bin = binned ? (double)c->n / (double)(c->grid.nbins > 0 ? c->grid.nbins : 1) : 0.0;
    // Work-group size of the scatter kernel (bin_scatter_threads, bf_scatter.hip).  Dense tiles: 1024 threads for a context that
    // has the GPU to itself and bins of thousands of events (8.0 against 8.9 us per launch at config 2; at 640x480, bins of
    // ~1500 events, 512 threads: 11.7 against 17.4 us), 512 for contexts sharing the GPU ("co_schedule": a
    // 1024-thread work-group with its 51 KB tile needs half a CU's wave slots free at once and waits for them while the other
    // contexts' kernels hold a few each -- 16.7 instead of 8.0 us under four contexts; with 512 threads 170 -> 190 Mevents/s).
    // Event lists over thousands of small bins -- 1280x720 at scale 3: 1620 bins of ~600 events, six per CU -- run 256-thread
    // work-groups (16.6 against 18.6 us per scatter launch there at 1 M events, 8.2 against 13.3 at 100 k); with a couple of
    // bins per CU -- 640x480, 540 bins -- 512 threads stay ahead (6.3 against 8.2).
    const bool many_small_bins = c->fmt == 2 && c->n_cus > 0 && c->grid.nbins >= 4 * c->n_cus && ev_per_bin < 1024.0;
    const int bin_threads = bin_scatter_threads(c->fmt, head_update, many_small_bins, ev_per_bin);
    if (binned) {
        // events a scatter thread keeps in flight:
        // (event lists: registers, not LDS, set the occupancy there -- two events per thread keep four work-groups on a
        // CU, and a bin above the pass size takes a second pass; measured at 1280x720: 512 x 2 69.8 us, 512 x 4 73.5)
        // (dense tiles: a pass should cover the AVERAGE bin, fuller bins take a second pass -- sizing it for 1.5 x the
        // average left half of every thread's slots empty at 640x480: 512 x 8 19.9 us, 512 x 4 15.3 us)
        const double per_bin = (c->fmt == 2 ? 1.0 : 1.1) * ev_per_bin / (double)bin_threads;
        // (... and between two and four, two up to 2.83 -- the geometric middle: bins of ~2100 events on 1024 threads ran
        // 15.3 us with four events per thread, half of every thread's slots empty, against 12.1 us with two and a second pass
        // for the fuller bins; measured at 1M events on 440 / 520 / 560 x 480 sensors)
        ev_per_thread = per_bin <= 1 ? 1 : (per_bin <= 2.83 ? 2 : (per_bin <= 4 ? 4 : 8));
        // (dense slabs on 512-thread work-groups with bins of thousands of events -- config 2 under "co_schedule": 272 bins, 3673
        // events on average, 4912 in the fullest -- : the pass covers the FULLEST bin, ~1.35 x the average; with 8 per thread two
        // thirds of the bins took a second pass: 8.45 -> 8.13 us per launch alone, 8.2 -> 7.8 under four contexts)
        if (c->fmt == 0 && bin_threads == 512 && per_bin > 6.5) ev_per_thread = per_bin <= 8.2 ? 10 : 12;
        // (event lists, update in the stencil tail, thousands of small bins on 256 threads -- 1280x720: 1620 bins, 608 events on average,
        // 877 in the fullest: four per thread cover every bin in one pass, 14.3-14.6 -> 13.5 us per launch; the head form, whose
        // registers also hold the update, loses with four: 14.8 -> 16.5)
        if (c->fmt == 2 && !head_update && bin_threads == 256 && per_bin > 2.0) ev_per_thread = 4;
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
        // (each buffer under its own check: an allocation that fails half-way must not leave the others looking ready)
        if (!c->d_xred) {
            HIP_TRY(c, hipMalloc(&c->d_xred, (size_t)2 * 16 * 32 * sizeof(unsigned long long)));
            HIP_TRY(c, hipMemsetAsync(c->d_xred, 0, (size_t)2 * 16 * 32 * sizeof(unsigned long long), c->stream));
        }
        if (!c->d_verdict) {
            HIP_TRY(c, hipMalloc(&c->d_verdict, 64));
            HIP_TRY(c, hipMemsetAsync(c->d_verdict, 0, 64, c->stream));
        }
        if (!c->h_broken) {
            HIP_TRY(c, hipHostMalloc(&c->h_broken, 64, hipHostMallocDefault));
            *c->h_broken = 0;
        }
        for (int i = 0; i < 4; ++i)
            if (!c->d_xscratch[i]) HIP_TRY(c, hipMalloc(&c->d_xscratch[i], (size_t)c->cap_events * sizeof(float2)));
    }
    bool want_rebin = false;
    int last_rebin_at = 0;
    // A warm start that is expected to converge in a handful of iterations (the previous one did) is polled batch by batch,
    // the final warp riding along: "quick".  One that is expected to run long -- the reference's own ring: ~115 iterations per
    // warm-started slice -- is fed and polled like a cold run: two-iteration batches with a blocking poll each cost it a
    // host round trip every other iteration (22 instead of 14 us per iteration on a 50 000-event slice).
    const bool quick_warm = warm_start && c->warm_iters_hint < 3 * o.poll_interval;
    int stall_allowance = 0;   // launches that may have been spent waiting for a re-bin (one-kernel iteration)
    bool final_done = false;   // the gated final warp of a warm start's first batch already ran
    int skip_rebin_checks = 0;
#ifdef BF_DEBUG_HOOKS
    static const bool host_timing = getenv("BF_HOST_TIMING") != nullptr;   // debug build: where the host thread's time goes
#else
    constexpr bool host_timing = false;
#endif
    double ht_launch = 0, ht_wait = 0;
    auto ht_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double ht_mark = host_timing ? ht_now() : 0;
    bool prewarp_done = false, persist_gave_up = false;
    unsigned long long seq_wait = 0;   // sequence number the current quick-warm batch's k_finish_update will publish (0: none)
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
        for (int i = 0; i < 4; ++i) la.scratch[i] = c->d_xscratch[i];
        la.trace = trace;
        la.nbr = c->fgrid.nbr; la.nbc = c->fgrid.nbc;
        la.R = c->win.scale_img_x; la.C = c->win.scale_img_y;
        la.max_passes = 4096;
        la.first_warp = first_warp ? 1 : 0;
        la.tl = c->d_tl;
        la.verdict = c->d_verdict; la.broken = c->h_broken;
        la.debug_abort = c->dbg_persist_abort; la.debug_mute = c->dbg_persist_mute;   // (test hooks: bf_create read the environment)
        la.debug_split = -1; la.debug_split_late = c->dbg_persist_split_late;
        if (c->dbg_persist_split >= 0) {   // that pass of THIS launch, made its last one
            la.debug_split = launched_iters + c->dbg_persist_split;
            la.max_passes = c->dbg_persist_split + 1;
        }
        prewarp_done = true;
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
        if (batch == 0) {   // ("defer_uploads": the next slice's copies and staging go out now, under this round)
            const int rcd = issue_deferred_uploads(c);
            if (rcd != BF_OK) return rcd;
        }
        if (warm_start) {
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
        if (*reinterpret_cast<volatile int*>(c->h_broken)) {
            *c->h_broken = 0;
            return fail(c, BF_ERR_HIP, "persistent loop kernel: a committed launch could not be read back");
        }
        if (ws.hot.spare_ < 0) {
            // The launch gave up (a work-group waited 0.2 s for others that were not resident: something else holds part of
            // the GPU) and undid itself: events and state are as it found them.  The rest of the run takes one launch per
            // iteration.
            persist_gave_up = true;
            c->persist_giveups++;
            c->persist_backoff = c->persist_backoff ? (c->persist_backoff < 64 ? 2 * c->persist_backoff : 64) : 1;
            c->persist_skip = c->persist_backoff;
            if (host_timing) fprintf(stderr, "persistent loop kernel gave up in round %d (last_j %d): falling back to one launch per iteration\n", batch, ws.last_j);
            first = ws.last_j < 0;
            h.hot.spare_ = 0;
            HIP_TRY(c, hipMemcpyAsync(&c->d_state[0].hot.spare_, &h.hot.spare_, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(&c->d_state[1].hot.spare_, &h.hot.spare_, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            break;
        }
        if (ws.hot.done) {
            fin = ws;
            final_done = true;
            break;
        }
        if (batch > 100000) return fail(c, BF_ERR_NOCONV, "device loop did not terminate");
    }
    if (persist && !persist_gave_up) c->persist_backoff = 0;   // (a run that went through: the next give-up starts at one run again)
    const bool persist_ran = final_done;
    const bool snap_polled = binned && !quick_warm && !persist_ran;   // progress is read from the pinned snapshot (below)
    if (snap_polled) {
        *reinterpret_cast<volatile unsigned long long*>(&c->h_state[0]) = 0ull;
        *reinterpret_cast<volatile unsigned long long*>(&c->h_state[0].run_tag) = 0ull;
    }
    for (int batch = 0; !persist_ran; ++batch) {
        // The re-bin kernels are device-gated (they run only if hot.need_rebin is set), but even a
        // no-op launch costs ~4.5 us here, so they are enqueued only before the first iteration and
        // when a polled snapshot shows the update asking for one.  The request is predictive
        // (0.6 x margin of drift), which covers the one-to-two batches of polling lag; anything
        // that still escapes takes the exact overflow path.
        if (binned && (batch == 0 || want_rebin)) {
            int rc = enqueue_rebin(c, state_of(launched_iters), perm_at_start, (prewarp && batch == 0 && !prewarp_done) ? &prewarp_wp : nullptr, fused, launched_iters);
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
        if (quick_warm) {
            // One more iteration than the previous warm start needed, then two at a time -- and, where the previous one needed
            // seven or more (a 640x480 stream: 5 .. 14 per slice), two more and then four at a time: a launch that finds the loop
            // over costs ~2 us, a second look at the batch ~28 us (blocking poll, follow-up launches, another gated final warp).
            // The first batch may be two polling intervals long (it was capped at one -- 8 -- which sent every slice of 9+
            // iterations through extra polls: config 3 averaged 2.2 looks per slice).
            const bool longish = c->warm_iters_hint >= 7;
            batch_len = batch == 0 ? c->warm_iters_hint + (longish ? 2 : 1) : (longish ? 4 : 2);
            if (batch_len < 2) batch_len = 2;
            if (batch_len > 2 * o.poll_interval) batch_len = 2 * o.poll_interval;
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
            if (sep_update) {   // the pending update of iteration j - 1 (none at j = 0: the state's own counter says so)
                launch_finish_update(state_of(j), acc_of(j - 1), ovf_of(j - 1), j, buf ^ 1, trace, quick_warm ? nullptr : &c->h_state[0], c->stream);
                inf.launches++;
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
                if (head_like) {   // accumulate only: the update runs at the head of the next warp+scatter launch (or in its own kernel)
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
                // (contexts of this process sharing the GPU: the other contexts' kernels hold CU slots too.  For event lists -- large
                // sparse images -- count the GPU as full whatever this grid's size, i.e. take the stencil kernel's build that fits 8
                // work-groups per CU: config 5 with four contexts 3.27 against 2.90 Mevents/s.  Dense tiles on a grid that does not fill
                // the GPU by itself keep the plain build: config 2 with four contexts 208.3 against 206.5, round 5 -- since its
                // instruction diet the kernel gains less from two more work-groups per CU than it loses to the spills.)
                const bool shared_lists = c->opt_co_schedule && c->fmt == 2 && g_live_ctx[c->device & 63].load() > 1;
                launch_stencil(a, stencil_src(c, binned), c->stream, shared_lists ? 1 : c->n_cus);
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
            if (head_like) {   // `done` of the batch's last iteration: apply its update now (normally the next launch would)
                seq_wait = ++c->seq_counter;
                launch_finish_update(state_of(launched_iters), acc_of(launched_iters - 1), ovf_of(launched_iters - 1),
                                     launched_iters, buf ^ 1, trace, &c->h_state[batch & 1], c->stream, fused ? lost_flag(c) + (launched_iters + 2) % 3 : nullptr,
                                     c->h_seq, seq_wait);
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
        if (batch == 0) {   // ("defer_uploads": the next slice's copies and staging go out now, under this batch)
            const int rcd = issue_deferred_uploads(c);
            if (rcd != BF_OK) return rcd;
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
            // the device's iteration counter, not a count of looks: a look takes a few
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
                struct timespec ts = {0, 20000};
                nanosleep(&ts, nullptr);
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
            if (((fused || !head_like) ? gpu_it > last_rebin_at : gpu_it >= last_rebin_at) && gpu_it > 0 && *rebin_p) want_rebin = true;
            if (launched_iters - stall_allowance > (o.hard_iter_cap > 0 ? o.hard_iter_cap : INT_MAX - 64) + 3 * o.poll_interval)
                return fail(c, BF_ERR_NOCONV, "device loop did not terminate");
            continue;
        }
        if (!(quick_warm && head_like))   // (there k_finish_update has written the state to the pinned copy itself)
            HIP_TRY(c, hipMemcpyAsync(&c->h_state[batch & 1], state_of(binned ? launched_iters : 0), sizeof(DevState),
                                      hipMemcpyDeviceToHost, c->stream));
        // A cold run is polled one batch behind the launches, so its wait can sleep (the wake-up latency hides
        // behind the batch already queued) instead of burning a host core per slice context; a warm start waits
        // for the batch it has just launched and spins.
        hipEvent_t* pev = c->poll_ev;
        HIP_TRY(c, hipEventRecord(pev[batch & 1], c->stream));
        if (batch == 0 && !quick_warm) continue;
        if (quick_warm) {   // look at this batch straight away
            // Update at the scatter head: k_finish_update has stored the state in the pinned snapshot itself and, behind a
            // system-scope fence, this batch's sequence number -- the host spins on that word and has the model while the final
            // warp (whose results stay on the device) is still running.  A bounded spin: past 2 ms the event decides.
            bool seen = false;
            if (seq_wait) {
                const volatile unsigned long long* sp = c->h_seq;
                const auto t_spin = std::chrono::steady_clock::now();
                for (unsigned spins = 0; !seen; ++spins) {
                    if (*sp == seq_wait) { seen = true; break; }
#if !defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("pause" ::: "memory");
#endif
                    if ((spins & 4095u) == 4095u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spin).count() > 2e-3) break;
                }
                std::atomic_thread_fence(std::memory_order_acquire);
            }
            if (!seen) HIP_TRY(c, hipEventSynchronize(pev[batch & 1]));
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
        {
            int rcw = wait_event_sleeping(c, pev[(batch - 1) & 1]);
            if (rcw != BF_OK) return rcw;
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
        {
            int rcw = wait_event_sleeping(c, c->poll_ev[1]);
            if (rcw != BF_OK) return rcw;
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
    // (want_uv: the final warp writes the per-event flow into d_uv; whoever reads it -- bf_compute_uv, bf_compute_uv_ring, the
    // writers -- does so on this stream, behind it: the run does not wait for it.  It used to drain the stream here, which kept a
    // warm-started slice's ~15 MB of flow output on the chain's critical path.)

    const DevState d = fin;
    h = d;   // model, dividers, warp parameters, plane-buffer dirtiness
    h.hot.pp = 0; h.hot.redo = 0; h.hot.pend = 0;   // (the final warp left the products in the set's first array)
    if (split) {   // the last executed iteration added to margin plane b0 ^ ((it - 1) & 1), and the lists name those pixels
        if (d.hot.it > 0) c->m_dirty_plane = b0 ^ ((d.hot.it - 1) & 1);
        c->m_unknown = d.rc < 0;   // (a run stopped at the iteration cap may have one executed launch more than `it` counts)
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
        memset(&m, 0, sizeof(m));   // (a run refused before it starts writes neither)
        memset(&inf, 0, sizeof(inf));
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

}  // extern "C"
