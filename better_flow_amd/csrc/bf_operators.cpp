// bf_operators.cpp -- C-ABI: OptimizerRolling::set_cloud / set_scale (optimizer_rolling.h:248-283) with the per-slice choice of loop and
// scatter format, and the one-to-one AccelLib operators (accel_lib.h:147-178,263-267,310-341,513-615) with the per-event read-backs.
#include "bf_ctx.h"

extern "C" {

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
        // (the margin: kBinMargin, even; BF_DEBUG_MARGIN overrides it for tests)
        const int bin_margin = c->dbg_margin > 0 ? ((c->dbg_margin + 1) & ~1) : kBinMargin;
        g.TS = 64;
        g.TSR = 64;
        if (c->n_cus > 0) {
            const double density = (double)c->n / ((double)w.scale_img_x * (double)w.scale_img_y);
            double best = -1.0;
            int best_area = 0;
            for (int cols = 16; cols <= 64; cols *= 2) {
                for (int rows = 32; rows <= 128; rows += 16) {
                    const int d = bin_margin > cols / 2 ? cols / 2 : bin_margin;
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
        g.D = bin_margin > tmin_ / 2 ? tmin_ / 2 : bin_margin;   // <= 2 x 2 bins per pixel
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
                        g.nbins <= 8192 && c->cap_events < (1ll << 29) &&   // (32-bit byte offsets into the event arrays: ld_idx)
                        (size_t)g.LR * g.L * 8 <= (size_t)kBinTileLdsMax && w.scale_img_x < (1 << 20);
        // (<= 8192 bins x <= 156 KB: slabs, tiled image and margin plane stay below 2^31 bytes -- the stencil kernel's buffer loads
        // carry 32-bit byte offsets, buf_ld_u64)
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
            w.scale_img_x < (1 << 20) && (long long)c->n < (1ll << 31)) {
            BinGrid f;
            memset(&f, 0, sizeof(f));
            const int Hh = scale / 2 + 1;
            auto tiles = [&](int rows) { return ((w.scale_img_x + rows - 1) / rows) * ((w.scale_img_y + 63) / 64); };
            const int rows = tiles(32) * kFusedZones <= 8192 ? 32 : 64;
            int Dm = c->dbg_margin > 0 ? c->dbg_margin : kFusedMargin;
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
        // Dense slabs or event lists.  A dense slice (one event per four pixels or more) merges its events in the bin's LDS
        // tile and writes the tile.  A sparse one writes lists, work and traffic following the events: one entry per EVENT
        // and no LDS tile (a 1280x720 sensor with 1M events -- the tile of such a bin would fill the CU's LDS and leave one
        // work-group per CU).  "auto" decides once per slice: the kernels are compiled per format.  Measured per iteration
        // (dense / events): 1280x720 scale 3: 90 / 68 us; 640x480 scale 3: 44 / 52.  (A third form -- lists merged per pixel
        // in the LDS tile, for small sensors at large scales: 346x260 scale 7 61 against 96 / 103 us -- was removed in round 5:
        // no BASELINE configuration took it, and every form multiplies the bit-identity matrix.)
        {
            const double P = (double)w.scale_img_x * (double)w.scale_img_y;
            const size_t LLg = (size_t)g.LR * (size_t)g.L;
            const bool lists_ok = c->use_binned && LLg <= 65536;                         // 16-bit tile-local pixel indices
            const int mode = lists_ok ? c->opt_bin_compact : 0;
            c->fmt = 0;
            if (mode == 2 || (mode == 1 && 4.0 * (double)c->n < P)) c->fmt = 2;
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
            // Event lists on 64-column bins (every sensor of BASELINE.json's configurations): entries sorted by (column zone,
            // row), so that a stencil tile gathers from the bins beside its own only the zone that faces it (bf_scatter.hip,
            // "event lists").  zw: the columns of a bin's tile a box sum of the neighbouring stencil tile can reach.
            c->grid.zw = 0;
            if (c->fmt == 2 && g.TS == 64 && g.L >= 2 * (g.D + scale / 2 + 1)) c->grid.zw = g.D + scale / 2 + 1;
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

static int materialize_outputs(bf_ctx* c);

int bf_project_4param(bf_ctx* c, double dnx_, double dny_, double cx, double cy, double div, double crl) {
    if (!c) return BF_ERR_ARG;
    if (!c->have_window) return fail(c, BF_ERR_STATE, "bf_project_4param before bf_set_cloud");
    if (c->degenerate) return fail(c, BF_ERR_STATE, "bf_project_4param on a degenerate (empty) window");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != BF_OK) return rc;
    rc = materialize_outputs(c);   // (the events' (nx, ny) in upload order, whatever left them)
    if (rc != BF_OK) return rc;
    WarpParams& w = c->hst.hot.wp;
    w.dnx = dnx_; w.dny = dny_; w.cx = cx; w.cy = cy; w.div = div;
    w.c = std::cos(crl);   // event.h:91-92 evaluates std::cos / std::sin on the host
    w.s = std::sin(crl);
    launch_set_state(c->d_state, c->hst, c->stream);
    const bf_ctx::EvSet& e = c->set[c->cs];
    {
        ProfScope ps(c, 0, c->n);
        launch_project_dn(e.xy, e.t, e.p, c->d_nxny, c->has_perm ? e.perm : nullptr, c->n_valid ? 1 : 0, c->d_state, c->n, c->stream);
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
        launch_stencil(a, stencil_src(c, false), c->stream, c->n_cus);
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

}  // extern "C"
