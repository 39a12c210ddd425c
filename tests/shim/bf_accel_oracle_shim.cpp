// TEST INFRASTRUCTURE ONLY -- never built or linked by the product.
//
// Implements the subset of the C-ABI (include/bf_accel.h) that the host front end
// (better_flow_amd/host/) calls, on top of the CPU oracle (oracle/bf_oracle.c).  It exists so
// that the slice manager / CLI plumbing (ring buffer, triggers, STM chain, accumulation,
// de-duplication, text formats -- BASELINE config 1) can be exercised on a machine without a
// GPU, and so that the GPU CLI's output has a CPU-side expectation to be compared with.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <utility>
#include <vector>

#include "../../include/bf_accel.h"
#include "../../oracle/bf_oracle.h"

struct bf_ctx {
    std::vector<int32_t> fx, fy;
    std::vector<int64_t> t;
    std::vector<uint8_t> noise;
    std::vector<double> pr_x, pr_y, nx, ny;
    bfo_cloud cloud;
    bfo_window win;
    bfo_model model;
    bool have_window = false;
    bool reversed = false;   // events of a ring slice are held newest -> oldest, the order the reference iterates in
    struct PendingUpload {   // bf_upload_*_async ... bf_commit_upload: a FIFO, like the two staging slots of the product
        std::vector<int32_t> x, y;
        std::vector<int64_t> t;
        std::vector<uint8_t> noise;
        bool linear = false;
    };
    std::deque<PendingUpload> pend;
    int64_t cap_events = 0;   // bf_create's capacity: an upload above it fails like the product's (BF_ERR_CAPACITY)
    bfo_local_window lwin;
    std::vector<float> time_img;
    char err[128] = "";
    void bind() {
        cloud.n = (int64_t)fx.size();
        cloud.fr_x = fx.data(); cloud.fr_y = fy.data(); cloud.t = t.data(); cloud.noise = noise.data();
        cloud.pr_x = pr_x.data(); cloud.pr_y = pr_y.data(); cloud.nx = nx.data(); cloud.ny = ny.data();
    }
};

static void to_abi(const bfo_model &m, bf_model *o) {
    o->cx = m.cx; o->cy = m.cy; o->dx = m.dx; o->dy = m.dy; o->rot = m.rot; o->div = m.div;
    o->cnt = m.cnt; o->_pad = 0;
    o->total_dx = m.total_dx; o->total_dy = m.total_dy; o->total_rot = m.total_rot; o->total_div = m.total_div;
}
static void from_abi(const bf_model &m, bfo_model *o) {
    o->cx = m.cx; o->cy = m.cy; o->dx = m.dx; o->dy = m.dy; o->rot = m.rot; o->div = m.div;
    o->cnt = m.cnt;
    o->total_dx = m.total_dx; o->total_dy = m.total_dy; o->total_rot = m.total_rot; o->total_div = m.total_div;
}

// BF_SHIM_EVENT_ORDER=reversed hands every slice to the oracle in the opposite event order (outputs are mapped back).
// The reference accumulates its time image in f32 in container order (accel_lib.h:162), so this is the smallest
// perturbation the reference itself is subject to; tests use it to measure the oracle's OWN spread over an STM chain
// and derive the admissible GPU-vs-oracle difference from it.
static bool flip_order() {
    static const bool f = [] { const char *e = std::getenv("BF_SHIM_EVENT_ORDER"); return e && !strcmp(e, "reversed"); }();
    return f;
}
template <class T> static void reverse_vec(std::vector<T> &v) {
    for (size_t i = 0, n = v.size(); i < n / 2; ++i) std::swap(v[i], v[n - 1 - i]);
}

template <class ADDR>
static int ring_slice(bf_ctx *c, const ADDR *rx, const ADDR *ry, const uint64_t *rts, const uint8_t *rnoise, int64_t cap,
                      int64_t first, int64_t n, uint64_t t0) {
    if (n > c->cap_events) { std::snprintf(c->err, sizeof(c->err), "n=%lld exceeds capacity %lld", (long long)n, (long long)c->cap_events); return BF_ERR_CAPACITY; }
    c->pend.emplace_back();
    bf_ctx::PendingUpload &u = c->pend.back();
    u.x.resize(n); u.y.resize(n); u.t.resize(n); u.noise.assign(n, 0);
    for (int64_t i = 0; i < n; ++i) {   // newest -> oldest, like `for (auto &e : ev_buffer)` (dvs_flow.h:195-197)
        const int64_t k = (first + (n - 1 - i)) % cap;
        u.x[i] = (int32_t)rx[k]; u.y[i] = (int32_t)ry[k];
        u.t[i] = rts[k] > t0 ? (int64_t)(rts[k] - t0) : -(int64_t)(t0 - rts[k]);   // event.h:61-63
        if (rnoise) u.noise[i] = rnoise[k];
    }
    return BF_OK;
}

extern "C" {

const char *bf_version(void) { return "bf_accel ORACLE TEST SHIM (CPU) -- not the product"; }
const char *bf_last_error(const bf_ctx *c) { return c ? c->err : "null ctx"; }

void bf_run_opts_default(bf_run_opts *o) {
    o->max_iter = -1; o->min_events = 1000; o->res_x = 180; o->res_y = 240;
    o->hard_iter_cap = 100000; o->poll_interval = 8; o->trace_cap = 0; o->want_uv = 0;
}

int bf_create(int32_t, int64_t max_events, int32_t, int32_t, void *, bf_ctx **out) {
    *out = new bf_ctx();
    (*out)->cap_events = max_events;
    return BF_OK;
}
void bf_destroy(bf_ctx *c) { delete c; }

int bf_upload_events(bf_ctx *c, const int32_t *fr_x, const int32_t *fr_y, const int32_t *t_ns,
                     const uint8_t *noise, int64_t n) {
    c->fx.assign(fr_x, fr_x + n);
    c->fy.assign(fr_y, fr_y + n);
    c->t.resize(n);
    for (int64_t i = 0; i < n; ++i) c->t[i] = t_ns[i];
    c->noise.assign(n, 0);
    if (noise) c->noise.assign(noise, noise + n);
    c->pr_x.assign(n, 0); c->pr_y.assign(n, 0); c->nx.assign(n, 0); c->ny.assign(n, 0);
    c->reversed = false;
    if (flip_order()) {
        reverse_vec(c->fx); reverse_vec(c->fy); reverse_vec(c->t); reverse_vec(c->noise);
        c->reversed = true;
    }
    c->bind();
    c->have_window = false;
    return BF_OK;
}

int bf_set_cloud(bf_ctx *c, int32_t scale, int32_t res_x, int32_t res_y, bf_window *w) {
    c->bind();
    bfo_set_cloud(&c->cloud, scale, res_x, res_y, &c->win);
    bfo_model_init(&c->model);
    c->have_window = true;
    if (w) {
        memset(w, 0, sizeof(*w));
        w->scale = scale;
        w->x_min = c->win.x_min; w->y_min = c->win.y_min; w->x_max = c->win.x_max; w->y_max = c->win.y_max;
        w->metric_wsizex = c->win.metric_wsizex; w->metric_wsizey = c->win.metric_wsizey;
        w->scale_img_x = c->win.scale_img_x; w->scale_img_y = c->win.scale_img_y;
        w->x_shift = c->win.x_shift; w->y_shift = c->win.y_shift;
    }
    return BF_OK;
}

int bf_project_4param_reinit(bf_ctx *c, double a, double b, double cx, double cy, double div, double crl) {
    bfo_project_4param_reinit(&c->cloud, a, b, cx, cy, div, crl);
    return BF_OK;
}

int bf_get_time_img(bf_ctx *c, float *time_out, uint32_t *count_out) {
    const size_t P = (size_t)c->win.scale_img_x * c->win.scale_img_y;
    c->time_img.resize(P);
    std::vector<float> cnt(P);
    bfo_get_time_img(&c->cloud, c->win.metric_wsizex, c->win.metric_wsizey, c->win.scale, (int)c->win.x_shift,
                     (int)c->win.y_shift, c->time_img.data(), cnt.data());
    if (time_out) memcpy(time_out, c->time_img.data(), P * sizeof(float));
    if (count_out)
        for (size_t i = 0; i < P; ++i) count_out[i] = (uint32_t)cnt[i];
    return BF_OK;
}

int bf_sobel(bf_ctx *, const float *img, int32_t rows, int32_t cols, float *gx, float *gy) {
    bfo_sobel(img, rows, cols, gx, gy);
    return BF_OK;
}

int bf_fast_model(bf_ctx *c, const float *img, int32_t rows, int32_t cols, bf_model *model) {
    bfo_model m;
    from_abi(*model, &m);
    if (img) bfo_fast_model(img, rows, cols, &m);
    else bfo_fast_model(c->time_img.data(), c->win.scale_img_x, c->win.scale_img_y, &m);
    model->cx = m.cx; model->cy = m.cy; model->dx = m.dx; model->dy = m.dy;
    model->rot = m.rot; model->div = m.div; model->cnt = m.cnt;
    return BF_OK;
}

int bf_writeout_events(bf_ctx *c, double *pr_x, double *pr_y, double *nx, double *ny) {
    const size_t n = c->fx.size();
    const std::vector<double> *src[4] = {&c->pr_x, &c->pr_y, &c->nx, &c->ny};
    double *dst[4] = {pr_x, pr_y, nx, ny};
    for (int k = 0; k < 4; ++k) {
        if (!dst[k]) continue;
        for (size_t i = 0; i < n; ++i) dst[k][i] = (*src[k])[c->reversed ? n - 1 - i : i];
    }
    return BF_OK;
}

int bf_compute_uv(bf_ctx *c, double *u, double *v) {
    const size_t n = c->fx.size();
    std::vector<double> uu(n), vv(n);
    bfo_compute_uv(c->nx.data(), c->ny.data(), (int64_t)n, uu.data(), vv.data());
    if (c->reversed) {   // a ring slice is held newest -> oldest; its outputs go back oldest -> newest
        for (size_t i = 0; i < n / 2; ++i) { std::swap(uu[i], uu[n - 1 - i]); std::swap(vv[i], vv[n - 1 - i]); }
    }
    if (u) memcpy(u, uu.data(), n * 8);
    if (v) memcpy(v, vv.data(), n * 8);
    return BF_OK;
}

int bf_set_model(bf_ctx *c, const bf_model *model) {
    bfo_model last;
    from_abi(*model, &last);
    bfo_set_model(&c->cloud, &c->model, &last);
    return BF_OK;
}

int bf_run(bf_ctx *c, const bf_run_opts *opts, bf_model *model_out, bf_run_info *info) {
    bf_run_opts o;
    if (opts) o = *opts; else bf_run_opts_default(&o);
    bfo_loop lp;
    int rc = bfo_run(&c->cloud, &c->win, &c->model, o.max_iter, o.res_x, o.res_y, o.min_events, o.hard_iter_cap,
                     &lp, NULL, 0);
    if (model_out) to_abi(c->model, model_out);
    if (info) {
        memset(info, 0, sizeof(*info));
        info->rc = rc < 0 ? BF_ERR_NOCONV : rc;
        info->iterations = (int32_t)lp.itercount;
        info->x_divider = lp.x_divider; info->y_divider = lp.y_divider;
        info->rot_divider = lp.rot_divider; info->div_divider = lp.div_divider;
    }
    return rc < 0 ? BF_ERR_NOCONV : rc;
}


// ---- OptimizerLocal (optimizer_sampler.cpp) on the oracle ----
int bf_local_set_window(bf_ctx *c, int32_t scale, int32_t wsz, int32_t c_fr_x, int32_t c_fr_y, int64_t c_t,
                        bf_local_window *w) {
    if (wsz <= 0) bfo_local_window_cloud(&c->cloud, scale, &c->lwin);
    else bfo_local_window_at(scale, wsz, c_fr_x, c_fr_y, c_t, &c->lwin);
    if (w) {
        w->scale = c->lwin.scale; w->metric_wsizex = c->lwin.metric_wsizex; w->metric_wsizey = c->lwin.metric_wsizey;
        w->scale_img_x = c->lwin.scale_img_x; w->scale_img_y = c->lwin.scale_img_y;
        w->c_fr_x = c->lwin.c_fr_x; w->c_fr_y = c->lwin.c_fr_y; w->pad_ = 0; w->c_t = c->lwin.c_t;
    }
    return BF_OK;
}

int bf_local_iteration_step(bf_ctx *c, double nx, double ny, double *score, uint8_t *img_out) {
    const size_t px = (size_t)c->lwin.scale_img_x * (size_t)c->lwin.scale_img_y;
    std::vector<uint8_t> img(px), scratch(px);
    *score = bfo_local_iteration_step(&c->cloud, &c->lwin, nx, ny, img.data(), scratch.data());
    if (img_out) memcpy(img_out, img.data(), px);
    return BF_OK;
}

int bf_local_run(bf_ctx *c, int32_t res_x, int32_t res_y, int64_t max_evaluations, bf_local_state *out) {
    const size_t px = (size_t)c->lwin.scale_img_x * (size_t)c->lwin.scale_img_y;
    std::vector<uint8_t> img(px + 1), scratch(px + 1);
    bfo_local_state st;
    int rc = bfo_local_run(&c->cloud, &c->lwin, res_x, res_y, max_evaluations, &st, img.data(), scratch.data());
    out->nx = st.nx; out->ny = st.ny; out->last_score = st.last_score;
    out->dnx = st.dnx; out->dny = st.dny; out->dn_th = st.dn_th; out->evaluations = st.evaluations;
    return rc == -2 ? BF_ERR_NOCONV : rc;
}


// ---- structure-of-arrays ring hand-off (better_flow/stream_flow.h) ----
int bf_host_alloc(bf_ctx *, int64_t bytes, void **out) { *out = std::malloc((size_t)bytes); return *out ? BF_OK : BF_ERR_HIP; }
int bf_host_free(bf_ctx *, void *ptr) { std::free(ptr); return BF_OK; }
int bf_synchronize(bf_ctx *) { return BF_OK; }
// (no device, no NUMA node of a device: the farm's workers stay where they are)
int bf_device_numa_node(int32_t, int32_t *node_out) { if (node_out) *node_out = -1; return BF_OK; }
int bf_bind_thread_to_numa_node(int32_t, int32_t *cpus_out) { if (cpus_out) *cpus_out = 0; return BF_OK; }
int bf_bind_thread_to_device_numa(int32_t, int32_t *node_out) { if (node_out) *node_out = -1; return BF_OK; }
int bf_wait_uploads(bf_ctx *) { return BF_OK; }

int bf_upload_ring_async(bf_ctx *c, const int32_t *rx, const int32_t *ry, const uint64_t *rts, const uint8_t *rnoise,
                         int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    return ring_slice(c, rx, ry, rts, rnoise, cap, first, n, t0);
}
int bf_upload_ring16_async(bf_ctx *c, const uint16_t *rrow, const uint16_t *rcol, const uint64_t *rts, const uint8_t *rnoise,
                           int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    return ring_slice(c, rrow, rcol, rts, rnoise, cap, first, n, t0);
}
// low 32 bits of the timestamps: the difference modulo 2^32, read as signed (see bf_accel.h)
int bf_upload_ring16t32_async(bf_ctx *c, const uint16_t *rrow, const uint16_t *rcol, const uint32_t *rt32, const uint8_t *rnoise,
                              int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    std::vector<uint64_t> full((size_t)cap);
    for (int64_t k = 0; k < cap; ++k) full[(size_t)k] = (uint64_t)((int64_t)t0 + (int64_t)(int32_t)(rt32[k] - (uint32_t)t0));
    return ring_slice(c, rrow, rcol, full.data(), rnoise, cap, first, n, t0);
}

// slice event i (oldest -> newest) -> uv_ring[2 * ((first + i) % cap)]
int bf_compute_uv_ring(bf_ctx *c, double *uv_ring, int64_t cap, int64_t first) {
    const size_t n = c->fx.size();
    std::vector<double> u(n), v(n);
    bf_compute_uv(c, u.data(), v.data());
    for (size_t i = 0; i < n; ++i) {
        const size_t k = (size_t)((first + (int64_t)i) % cap);
        uv_ring[2 * k] = u[i]; uv_ring[2 * k + 1] = v[i];
    }
    return BF_OK;
}

int bf_set_option(bf_ctx *, const char *, int64_t) { return BF_OK; }   // device tuning knobs: nothing to tune here
int bf_get_stat(bf_ctx *, const char *, int64_t *value) { if (value) *value = -1; return BF_OK; }   // (no device loops here)

// linear int32 arrays with slice-local times (the slice farm's second input form); held in upload order
int bf_upload_events_async(bf_ctx *c, const int32_t *fr_x, const int32_t *fr_y, const int32_t *t_ns, int64_t n) {
    if (n > c->cap_events) { std::snprintf(c->err, sizeof(c->err), "n=%lld exceeds capacity %lld", (long long)n, (long long)c->cap_events); return BF_ERR_CAPACITY; }
    c->pend.emplace_back();
    bf_ctx::PendingUpload &u = c->pend.back();
    u.x.assign(fr_x, fr_x + n); u.y.assign(fr_y, fr_y + n);
    u.t.assign(t_ns, t_ns + n);
    u.noise.assign(n, 0);
    u.linear = true;
    return BF_OK;
}

int bf_upload_events16_async(bf_ctx *c, const uint16_t *fr_x, const uint16_t *fr_y, const int32_t *t_ns, int64_t n) {
    std::vector<int32_t> x(fr_x, fr_x + n), y(fr_y, fr_y + n);
    return bf_upload_events_async(c, x.data(), y.data(), t_ns, n);
}

int bf_commit_upload(bf_ctx *c) {
    if (c->pend.empty()) return BF_ERR_STATE;
    bf_ctx::PendingUpload u = std::move(c->pend.front());
    c->pend.pop_front();
    const int64_t n = (int64_t)u.x.size();
    c->fx = std::move(u.x); c->fy = std::move(u.y); c->t = std::move(u.t); c->noise = std::move(u.noise);
    c->pr_x.assign(n, 0); c->pr_y.assign(n, 0); c->nx.assign(n, 0); c->ny.assign(n, 0);
    c->reversed = !u.linear;   // ring slices arrive newest -> oldest, linear arrays in upload order
    if (flip_order()) {
        reverse_vec(c->fx); reverse_vec(c->fy); reverse_vec(c->t); reverse_vec(c->noise);
        c->reversed = !c->reversed;
    }
    c->bind();
    c->have_window = false;
    return BF_OK;
}


int bf_projection_img(bf_ctx *c, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final, uint8_t *img_out) {
    std::vector<uint8_t> scratch((size_t)res_x * scale * (size_t)res_y * scale);
    bfo_projection_img(&c->cloud, scale, res_x, res_y, show_final, img_out, scratch.data());
    return BF_OK;
}

int bf_color_time_img(bf_ctx *c, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final, uint8_t *bgr_out) {
    const int32_t sc = scale ? scale : 11;
    std::vector<float> scratch((size_t)(res_x * sc + sc) * (size_t)(res_y * sc + sc) * 3);
    bfo_color_time_img(&c->cloud, scale, res_x, res_y, show_final, bgr_out, scratch.data());
    return BF_OK;
}

}  // extern "C"
