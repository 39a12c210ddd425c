// accel_lib.h -- the operator API of the motion-compensation path (mirror of the reference's
// `class AccelLib`, better_flow/accel_lib.h:14-616), implemented over the C-ABI of
// libbf_accel.so (include/bf_accel.h): every operator runs on the MI355X.  There is no CPU
// branch: `gpu_enabled` is always true and a missing device is a hard error.
//
// Differences that are forced by the device path (all documented in INTEGRATION.md):
//   * images are bf::Image2D<float> (row-major, rows x cols) instead of cv::Mat;
//   * init_gpu() only remembers the cloud; the slice is staged by sync() once the local
//     time has been set (the reference re-uploads `t` on every projection, accel_lib.h:284-289);
//   * one device context is shared by all AccelLib objects of a thread and reused across
//     slices (the reference allocates per slice, accel_lib.h:86-115).
#ifndef BF_HOST_ACCEL_LIB_H
#define BF_HOST_ACCEL_LIB_H

#include <better_flow/event.h>
#include <better_flow/object_model.h>
#include <bf_accel.h>

#include <stdexcept>
#include <string>

namespace bf {

// A failed call of the C-ABI (include/bf_accel.h) as a C++ exception: `code` is the bf_* return code (BF_ERR_*), what()
// the context's last error text.  Nothing in the host front end terminates the process; the command line catches this
// at top level, a library user wherever it likes.
struct AccelError : std::runtime_error {
    int code;
    AccelError(int code_, const std::string &msg) : std::runtime_error(msg), code(code_) {}
};

template <class T> struct Image2D {
    int rows = 0, cols = 0;
    std::vector<T> data;
    Image2D() {}
    Image2D(int r, int c) : rows(r), cols(c), data((size_t)r * (size_t)c) {}
    T &at(int r, int c) { return data[(size_t)r * cols + c]; }
    const T &at(int r, int c) const { return data[(size_t)r * cols + c]; }
    T *ptr(int r) { return &data[(size_t)r * cols]; }
};

// Process-wide device context, created on first use and grown on demand.
class DeviceContext {
public:
    static int &device() {
        static int d = 0;
        return d;
    }
    static bf_ctx *get(long long events, int rows, int cols) {
        State &s = state();
        if (s.ctx && events <= s.events && (long long)rows * cols <= (long long)s.rows * s.cols) return s.ctx;
        if (s.ctx) bf_destroy(s.ctx);
        s.ctx = nullptr;
        s.events = events > s.events ? events : s.events;
        if ((long long)rows * cols > (long long)s.rows * s.cols) { s.rows = rows; s.cols = cols; }
        int rc = bf_create(device(), s.events, s.rows, s.cols, nullptr, &s.ctx);
        if (rc != BF_OK) {
            s.ctx = nullptr;
            throw AccelError(rc, "bf_create failed (" + std::to_string(rc) + "): the motion-compensation path needs a HIP "
                                 "device (there is no CPU fallback)");
        }
        return s.ctx;
    }
    static void release() {
        State &s = state();
        if (s.ctx) bf_destroy(s.ctx);
        s.ctx = nullptr;
    }

private:
    struct State {
        bf_ctx *ctx = nullptr;
        long long events = 0;
        int rows = 0, cols = 0;
        ~State() { if (ctx) bf_destroy(ctx); }
    };
    static State &state() {
        static thread_local State s;
        return s;
    }
};

}  // namespace bf

class AccelLib {
private:
    bf_ctx *ctx;
    size_t size;   // number of events staged
    bool staged;

    void check(int rc, const char *what) const {
        if (rc < 0)
            throw bf::AccelError(rc, std::string("AccelLib::") + what + " failed (" + std::to_string(rc) + "): " +
                                         (ctx ? bf_last_error(ctx) : "no ctx"));
    }

public:
    bool gpu_enabled;   // accel_lib.h:41 -- here: always true

    AccelLib() : ctx(nullptr), size(0), staged(false), gpu_enabled(true) {}

    bool is_staged() const { return staged; }
    bf_ctx *context() const { return ctx; }

    // AccelLib::init_gpu, accel_lib.h:71-145.  Stages the slice: AoS -> SoA (int fr_x, fr_y, t as
    // at :83-85,:101-103) and one upload.  Event::t must already hold the slice-local time.
    template <class T> inline void init_gpu(T *events, int nRows, int nCols) {
        this->size = events->size();
        this->ctx = bf::DeviceContext::get((long long)(size > 1024 ? size : 1024), nRows > 64 ? nRows : 64,
                                           nCols > 64 ? nCols : 64);
        std::vector<int32_t> fx(size), fy(size), ft(size);
        std::vector<uint8_t> noise(size);
        bool any_noise = false;
        size_t i = 0;
        for (auto &e : *events) {
            fx[i] = (int32_t)e.fr_x;
            fy[i] = (int32_t)e.fr_y;
            if (e.t > (sll)INT32_MAX || e.t <= (sll)INT32_MIN)
                throw bf::AccelError(BF_ERR_ARG, "AccelLib::init_gpu: slice-local time " + std::to_string((long long)e.t) +
                                                     " ns does not fit 32 bits (slices span at most 2.1 s)");
            ft[i] = (int32_t)e.t;
            noise[i] = e.noise ? 1 : 0;
            any_noise |= e.noise;
            ++i;
        }
        check(bf_upload_events(ctx, fx.data(), fy.data(), ft.data(), any_noise ? noise.data() : nullptr,
                               (int64_t)size), "init_gpu");
        this->staged = true;
    }

    // OptimizerRolling::set_cloud's device half: bounding box, window, Event::reset on the device.
    inline bf_window set_window(int scale) {
        bf_window w;
        check(bf_set_cloud(ctx, scale, RES_X, RES_Y, &w), "set_window");
        return w;
    }

    // accel_lib.h:211-217 -> :147-178.  w, h, scale, x_sh, y_sh are implied by the window that
    // set_window() computed from the same events (they are asserted to agree).
    template <class T>
    inline bf::Image2D<float> get_time_img(T * /*events*/, int w, int h, int scale, int /*x_sh*/, int /*y_sh*/) {
        bf::Image2D<float> img(w + scale, h + scale);
        check(bf_get_time_img(ctx, img.data.data(), nullptr), "get_time_img");
        return img;
    }

    inline bf::Image2D<uint32_t> get_count_img(int w, int h, int scale) {
        bf::Image2D<uint32_t> img(w + scale, h + scale);
        check(bf_get_time_img(ctx, nullptr, img.data.data()), "get_count_img");
        return img;
    }

    // accel_lib.h:275-281: the incremental form (its only call in the reference is commented out, optimizer_rolling.h:333-339)
    template <class T>
    inline void project_4param(T * /*events*/, double dnx_, double dny_, double cx, double cy, double div, double crl) {
        check(bf_project_4param(ctx, dnx_, dny_, cx, cy, div, crl), "project_4param");
    }

    // accel_lib.h:263-267
    template <class T>
    inline void project_4param_reinit(T * /*events*/, double dnx_, double dny_, double cx, double cy, double div,
                                      double crl) {
        check(bf_project_4param_reinit(ctx, dnx_, dny_, cx, cy, div, crl), "project_4param_reinit");
    }

    // accel_lib.h:310-329: copy pr / n back into the events.
    template <class T> inline void writeout_events(T *events) {
        if (!staged || size == 0) return;
        std::vector<double> px(size), py(size), nx(size), ny(size);
        check(bf_writeout_events(ctx, px.data(), py.data(), nx.data(), ny.data()), "writeout_events");
        size_t i = 0;
        for (auto &e : *events) {
            e.pr_x = px[i]; e.pr_y = py[i];
            e.nx = nx[i]; e.ny = ny[i];
            ++i;
        }
    }

    // Event::compute_uv (event.h:135-142) for every event, evaluated on the device.
    template <class T> inline void compute_uv(T *events) {
        if (!staged || size == 0) return;
        std::vector<double> u(size), v(size);
        check(bf_compute_uv(ctx, u.data(), v.data()), "compute_uv");
        size_t i = 0;
        for (auto &e : *events) {
            e.set_uv(u[i], v[i]);
            ++i;
        }
    }

    // accel_lib.h:337-341
    void fast_model(ObjectModel &model, bf::Image2D<float> &time_img) {
        bf_model m = model.to_abi();
        check(bf_fast_model(ctx, time_img.data.data(), time_img.rows, time_img.cols, &m), "fast_model");
        model.cx = m.cx; model.cy = m.cy; model.dx = m.dx; model.dy = m.dy;
        model.rot = m.rot; model.div = m.div; model.cnt = m.cnt;
    }
    ObjectModel fast_model(bf::Image2D<float> &time_img) {
        ObjectModel model;
        this->fast_model(model, time_img);
        return model;
    }

    // accel_lib.h:400-432 / :513-543
    void Sobel(bf::Image2D<float> &img, bf::Image2D<float> &grad_x, bf::Image2D<float> &grad_y) {
        if (!ctx) ctx = bf::DeviceContext::get(1024, img.rows, img.cols);
        grad_x = bf::Image2D<float>(img.rows, img.cols);
        grad_y = bf::Image2D<float>(img.rows, img.cols);
        check(bf_sobel(ctx, img.data.data(), img.rows, img.cols, grad_x.data.data(), grad_y.data.data()), "Sobel");
    }

    // OptimizerRolling::set_model's warp (optimizer_rolling.h:294-298)
    void set_model(const ObjectModel &m) {
        bf_model a = m.to_abi();
        check(bf_set_model(ctx, &a), "set_model");
    }

    // EventFile::projection_img (event_file.h:460-515) of the staged slice: the 8-bit event image at the current
    // projected positions (motion compensated once run() has converged) or, show_final, at the sensor positions.
    bf::Image2D<uint8_t> projection_img(int scale, bool show_final) {
        bf::Image2D<uint8_t> img(RES_X * scale, RES_Y * scale);
        check(bf_projection_img(ctx, scale, RES_X, RES_Y, show_final ? 1 : 0, img.ptr(0)), "projection_img");
        return img;
    }

    // EventFile::color_time_img (event_file.h:649-747) of the staged slice: B, G, R bytes, (RES_X * scale + scale) x
    // (RES_Y * scale + scale) pixels; hue = phase of the events' time in the slice
    std::vector<uint8_t> color_time_img(int scale, bool show_final, int *rows, int *cols) {
        const int sc = scale ? scale : 11;
        *rows = RES_X * sc + sc; *cols = RES_Y * sc + sc;
        std::vector<uint8_t> img((size_t)*rows * (size_t)*cols * 3);
        check(bf_color_time_img(ctx, scale, RES_X, RES_Y, show_final ? 1 : 0, img.data()), "color_time_img");
        return img;
    }

    // OptimizerLocal on the device (optimizer_sampler.h:29-48; optimizer_sampler.cpp:4-38,120-153)
    void local_set_window(int scale, int wsz, int c_fr_x, int c_fr_y, long long c_t, bf_local_window *w) {
        check(bf_local_set_window(ctx, scale, wsz, c_fr_x, c_fr_y, c_t, w), "local_set_window");
    }
    double local_iteration_step(double nx, double ny, uint8_t *img_out) {
        double score = 0;
        check(bf_local_iteration_step(ctx, nx, ny, &score, img_out), "local_iteration_step");
        return score;
    }
    int local_run(bf_local_state *st) {
        int rc = bf_local_run(ctx, RES_X, RES_Y, 0, st);
        check(rc, "local_run");
        return rc;
    }

    // The fused OptimizerRolling::run (optimizer_rolling.h:48-125) on the device.
    int run(int max_itercount, ObjectModel &model, bf_run_info *info) {
        bf_run_opts o;
        bf_run_opts_default(&o);
        o.max_iter = max_itercount;
        o.res_x = RES_X;
        o.res_y = RES_Y;
        bf_model m;
        int rc = bf_run(ctx, &o, &m, info);
        check(rc, "run");
        model = ObjectModel(m);
        return rc;
    }
};

#endif  // BF_HOST_ACCEL_LIB_H
