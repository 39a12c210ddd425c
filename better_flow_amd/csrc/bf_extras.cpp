// bf_extras.cpp -- C-ABI: the grid of per-tile optimizers (BASELINE config 4), the contrast-score optimiser OptimizerLocal
// (optimizer_sampler.h / .cpp), the frame renderers (event_file.h:460-515,649-747) and bf_get_trace.
#include "bf_ctx.h"

extern "C" {

namespace {

// Everything of bf_run_tiles ahead of the optimizer launch, on stream `s` (the context's own, or the batch's): buffers, the
// tiles' zero-model states, the counting sort of the slice's events by sensor tile.  Fills the launch's arguments.
int tiles_prepare(bf_ctx* c, const bf_tile_opts* o, hipStream_t s, TileArgs& a, bool rolling = true) {
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
        HIP_TRY(c, hipMemsetAsync(c->d_tile_hist, 0, (size_t)(nt + 1) * 4, s));
        c->tiles_alloc = nt;
    }
    // LDS image capacity: the largest window a tile can have
    const int tr = (o->sensor_res_x + o->grid_rows - 1) / o->grid_rows + 1;
    const int tc = (o->sensor_res_y + o->grid_cols - 1) / o->grid_cols + 1;
    const long long max_px = (long long)(o->scale * tr + o->scale) * (o->scale * tc + o->scale);
    if (rolling && max_px * 16 > 156 * 1024)   // (the rolling optimizer's planes; bf_local_run_tiles sizes its own window)
        return fail(c, BF_ERR_CAPACITY, "a tile window of up to %lld pixels does not fit the LDS", max_px);

    DevState tmpl;
    memset(&tmpl, 0, sizeof(tmpl));
    tmpl.x_div = tmpl.y_div = 1.0f;           // optimizer_rolling.h:61-63
    tmpl.rot_div = tmpl.div_div = 10000.0f;
    tmpl.max_iter = o->max_iter;
    tmpl.hard_cap = o->hard_iter_cap;
    tmpl.hot.wp = identity_warp();
    launch_fill_states(c->d_tile_states, tmpl, nt, s);

    TileGrid g;
    g.rows = o->grid_rows; g.cols = o->grid_cols; g.res_x = o->sensor_res_x; g.res_y = o->sensor_res_y;
    const bf_ctx::EvSet& src = c->set[c->cs];
    const bf_ctx::EvSet& dst = c->set[c->cs ^ 1];
    {
        ProfScope ps(c, 3);
        launch_tile_sort(src.xy, src.t, c->has_perm ? src.perm : nullptr, c->n, g, c->d_tile_hist, c->d_tile_start,
                         c->d_tile_cursor, dst.xy, dst.t, dst.p, dst.perm, s);
    }
    c->cs ^= 1;
    c->has_perm = true;
    a.xy = dst.xy; a.t = dst.t; a.p = dst.p; a.perm = dst.perm;
    a.nxny = c->d_nxny;
    a.tile_start = c->d_tile_start;
    a.states = c->d_tile_states;
    a.scale = o->scale;
    a.seed_res_x = o->sensor_res_x; a.seed_res_y = o->sensor_res_y;
    a.guard_res_x = o->guard_res_x; a.guard_res_y = o->guard_res_y;
    a.min_events = o->min_events;
    a.max_px = (int32_t)max_px;
    return BF_OK;
}

// The tiles' final states (already on the host) into the caller's arrays, and the context's flags after a tile run.
void tiles_collect(bf_ctx* c, const std::vector<DevState>& st, int nt, bf_model* models_out, bf_run_info* infos_out) {
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
}

}  // namespace

int bf_run_tiles(bf_ctx* c, const bf_tile_opts* o, bf_model* models_out, bf_run_info* infos_out) {
    if (!c || !o) return BF_ERR_ARG;
    TileArgs a;
    int rc = tiles_prepare(c, o, c->stream, a);
    if (rc != BF_OK) return rc;
    const int nt = o->grid_rows * o->grid_cols;
    {
        ProfScope ps(c, 0, c->n);
        if (launch_tile_optimizer(a, nt, c->stream) != 0) return fail(c, BF_ERR_HIP, "cannot configure the tile kernel");
    }
    HIP_TRY(c, hipGetLastError());
    std::vector<DevState> st((size_t)nt);
    HIP_TRY(c, hipMemcpyAsync(st.data(), c->d_tile_states, (size_t)nt * sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    tiles_collect(c, st, nt, models_out, infos_out);
    return BF_OK;
}

// The tile grids of n slices -- one per context, each uploaded -- in ONE launch on the first context's stream
// (k_tile_optimizer_many): work-groups claim (slice, tile) pairs from a device counter, so the straggler tiles of the first
// slices run under the bulk of the later ones without a stream (and a hardware queue) per slice.
int bf_run_tiles_many(bf_ctx* const* ctxs, int32_t n, const bf_tile_opts* o, bf_model* models_out, bf_run_info* infos_out) {
    if (!ctxs || n < 1 || !o || !ctxs[0]) return BF_ERR_ARG;
    bf_ctx* lead = ctxs[0];
    if (n > 4096) return fail(lead, BF_ERR_ARG, "at most 4096 slices per call (got %d)", n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(lead, BF_ERR_ARG, "context %d is NULL", i);
        if (ctxs[i]->device != lead->device) return fail(lead, BF_ERR_ARG, "context %d lives on another device", i);
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(lead, BF_ERR_ARG, "context %d appears twice (one slice per context)", i);
    }
    HIP_TRY(lead, hipSetDevice(lead->device));
    const int nt = o->grid_rows * o->grid_cols;
    if (n > lead->many_alloc) {
        if (lead->d_many_args) HIP_TRY(lead, hipFree(lead->d_many_args));
        if (lead->h_many_args) HIP_TRY(lead, hipHostFree(lead->h_many_args));
        lead->d_many_args = nullptr; lead->h_many_args = nullptr; lead->many_alloc = 0;
        HIP_TRY(lead, hipMalloc(&lead->d_many_args, (size_t)n * sizeof(TileArgs) + 64));
        HIP_TRY(lead, hipHostMalloc(&lead->h_many_args, (size_t)n * sizeof(TileArgs), hipHostMallocDefault));
        lead->many_alloc = n;
    }
    TileArgs* host_args = static_cast<TileArgs*>(lead->h_many_args);
    TileArgs* dev_args = static_cast<TileArgs*>(lead->d_many_args);
    uint32_t* counter = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lead->d_many_args) + (size_t)lead->many_alloc * sizeof(TileArgs));
    // every slice's preparation on the lead's stream, behind whatever its own context's stream still holds for it
    for (int i = 0; i < n; ++i) {
        bf_ctx* c = ctxs[i];
        if (c != lead) {
            HIP_TRY(c, hipEventRecord(c->poll_ev[0], c->stream));
            HIP_TRY(lead, hipStreamWaitEvent(lead->stream, c->poll_ev[0], 0));
        }
        TileArgs a;
        const int rc = tiles_prepare(c, o, lead->stream, a);
        if (rc != BF_OK) {
            if (c != lead) fail(lead, rc, "slice %d: %s", i, c->err);
            (void)hipStreamSynchronize(lead->stream);
            return rc;
        }
        host_args[i] = a;
    }
    HIP_TRY(lead, hipMemcpyAsync(dev_args, host_args, (size_t)n * sizeof(TileArgs), hipMemcpyHostToDevice, lead->stream));
    HIP_TRY(lead, hipMemsetAsync(counter, 0, sizeof(uint32_t), lead->stream));
    {
        long long ev = 0;
        for (int i = 0; i < n; ++i) ev += ctxs[i]->n;
        ProfScope ps(lead, 0, ev);
        if (launch_tile_optimizer_many(dev_args, n, nt, o->scale, host_args[0].max_px, counter, lead->n_cus, lead->stream) != 0)
            return fail(lead, BF_ERR_HIP, "cannot configure the tile kernel");
    }
    HIP_TRY(lead, hipGetLastError());
    std::vector<DevState> st((size_t)n * (size_t)nt);
    for (int i = 0; i < n; ++i)
        HIP_TRY(lead, hipMemcpyAsync(st.data() + (size_t)i * nt, ctxs[i]->d_tile_states, (size_t)nt * sizeof(DevState),
                                     hipMemcpyDeviceToHost, lead->stream));
    HIP_TRY(lead, hipStreamSynchronize(lead->stream));   // (everything of every slice is complete: the other contexts' streams need no event)
    for (int i = 0; i < n; ++i) {
        std::vector<DevState> one(st.begin() + (size_t)i * nt, st.begin() + (size_t)(i + 1) * nt);
        tiles_collect(ctxs[i], one, nt, models_out ? models_out + (size_t)i * nt : nullptr, infos_out ? infos_out + (size_t)i * nt : nullptr);
    }
    return BF_OK;
}

// A grid of OptimizerLocal windows, one per sensor tile, each on the tile's own events (k_local_tile_optimizer, bf_local.hip).
int bf_local_run_tiles(bf_ctx* c, const bf_local_tile_opts* o, bf_local_state* states_out, int32_t* rc_out) {
    if (!c || !o) return BF_ERR_ARG;
    if (o->wsz < 1) return fail(c, BF_ERR_ARG, "wsz must be >= 1");
    if (o->scale < 1 || o->scale % 2 == 0 || o->scale > 7)   // optimizer_sampler.cpp:206 (odd); the Gaussian is stated to 7
        return fail(c, BF_ERR_ARG, "scale must be odd and <= 7 (got %d)", o->scale);
    bf_tile_opts to;
    memset(&to, 0, sizeof(to));
    to.grid_rows = o->grid_rows; to.grid_cols = o->grid_cols; to.scale = o->scale;
    to.sensor_res_x = o->sensor_res_x; to.sensor_res_y = o->sensor_res_y;
    to.guard_res_x = o->guard_res_x; to.guard_res_y = o->guard_res_y;
    TileArgs ta;
    int rc = tiles_prepare(c, &to, c->stream, ta, false);   // (the counting sort by sensor tile; the rolling optimizers' states are not used)
    if (rc != BF_OK) return rc;
    const int nt = o->grid_rows * o->grid_cols;
    if (nt > c->ltile_alloc) {
        if (c->d_ltile) HIP_TRY(c, hipFree(c->d_ltile));
        c->d_ltile = nullptr; c->ltile_alloc = 0;
        HIP_TRY(c, hipMalloc(&c->d_ltile, (size_t)nt * (sizeof(bf_local_state) + sizeof(int32_t))));
        c->ltile_alloc = nt;
    }
    bf_local_state* d_states = static_cast<bf_local_state*>(c->d_ltile);
    int32_t* d_rcs = reinterpret_cast<int32_t*>(d_states + c->ltile_alloc);
    TileGrid g;
    g.rows = o->grid_rows; g.cols = o->grid_cols; g.res_x = o->sensor_res_x; g.res_y = o->sensor_res_y;
    {
        ProfScope ps(c, 0, c->n);
        const int lr = launch_local_tile_optimizer(ta.xy, ta.t, ta.tile_start, d_states, d_rcs, g, o->scale, o->wsz, o->guard_res_x,
                                                   o->guard_res_y, (long long)o->max_evaluations, c->stream);
        if (lr == -3) return fail(c, BF_ERR_CAPACITY, "a window of %d x %d pixels does not fit the LDS", o->scale * o->wsz + o->scale, o->scale * o->wsz + o->scale);
        if (lr != 0) return fail(c, BF_ERR_HIP, "cannot configure the window kernel");
    }
    HIP_TRY(c, hipGetLastError());
    std::vector<bf_local_state> st((size_t)nt);
    std::vector<int32_t> rcs((size_t)nt);
    HIP_TRY(c, hipMemcpyAsync(st.data(), d_states, (size_t)nt * sizeof(bf_local_state), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(rcs.data(), d_rcs, (size_t)nt * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < nt; ++i) {
        if (states_out) states_out[i] = st[(size_t)i];
        if (rc_out) rc_out[i] = rcs[(size_t)i];
    }
    // the events are now sorted by sensor tile (reset products, permutation kept); no per-event result was written
    c->p_clean = true;
    c->n_valid = false;
    c->uv_valid = false;
    c->out_sorted = false;
    c->pending_warp = false;
    c->use_binned = false;
    c->fused_ok = false;
    c->have_lwin = false;
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

}  // extern "C"
