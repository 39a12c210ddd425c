// optimizer_rolling.h -- the 4-parameter (dx, dy, div, rot) gradient-descent optimizer (mirror
// of the reference's better_flow/optimizer_rolling.h:17-347).  Public surface and call order
// are the reference's (set_cloud, set_time, set_maxiter, set_model, run, get_model); the loop
// body (iteration_step, :305-347) and the loop control (run, :61-101) execute on the GPU inside
// AccelLib::run.  The interactive mode (manual(), :128-233) needs an OpenCV GUI and is not part
// of this path.
#ifndef BF_HOST_OPTIMIZER_ROLLING_H
#define BF_HOST_OPTIMIZER_ROLLING_H

#include <better_flow/accel_lib.h>
#include <better_flow/common.h>
#include <better_flow/event.h>
#include <better_flow/object_model.h>

template <class T> class OptimizerRolling {
protected:
    AccelLib accel;
    T *events;

    int scale;
    int metric_wsizex, metric_wsizey;
    int max_itercount;
    int scale_img_x, scale_img_y;
    double x_shift, y_shift;
    int x_min, y_min, x_max, y_max;
    ull current_time;

    ObjectModel model;
    float x_divider, y_divider, rot_divider, div_divider;
    bool have_warm_model;
    ObjectModel warm_model;
    bf_run_info last_info;

    // Stage the slice on the device once the slice-local times are known, then let the device
    // compute the window (set_cloud's bounding box + set_scale, :252-283).
    void stage() {
        if (accel.is_staged()) return;
        accel.init_gpu(this->events, this->scale * RES_X + this->scale, this->scale * RES_Y + this->scale);
        bf_window w = accel.set_window(this->scale);
        this->x_min = w.x_min; this->y_min = w.y_min; this->x_max = w.x_max; this->y_max = w.y_max;
        this->metric_wsizex = w.metric_wsizex; this->metric_wsizey = w.metric_wsizey;
        this->scale_img_x = w.scale_img_x; this->scale_img_y = w.scale_img_y;
        this->x_shift = w.x_shift; this->y_shift = w.y_shift;
        if (this->have_warm_model) accel.set_model(this->warm_model);   // :294-298
    }

public:
    OptimizerRolling()
        : events(NULL), scale(0), metric_wsizex(0), metric_wsizey(0), max_itercount(-1), scale_img_x(0),
          scale_img_y(0), x_shift(0), y_shift(0), x_min(0), y_min(0), x_max(0), y_max(0), current_time(0),
          x_divider(1), y_divider(1), rot_divider(10000), div_divider(10000), have_warm_model(false) {
        std::memset(&last_info, 0, sizeof(last_info));
    }

    // optimizer_rolling.h:48-125.  Returns 0 (optimised) or 1 (skipped by a guard).
    int run() {
        this->stage();
        int rc = accel.run(this->max_itercount, this->model, &this->last_info);
        this->x_divider = last_info.x_divider; this->y_divider = last_info.y_divider;
        this->rot_divider = last_info.rot_divider; this->div_divider = last_info.div_divider;
        if (rc == BF_SKIPPED) {
            // :49-55: a too-small window marks every event as noise
            if ((this->scale_img_x < this->scale * RES_X / 15) && (this->scale_img_y < this->scale * RES_Y / 15))
                for (auto &e : *this->events) e.noise = true;
            return 1;
        }
        if (VERBOSE)
            std::cout << "\tMinimization: iterations: " << last_info.iterations << " (device launches "
                      << last_info.launches << ", re-bins " << last_info.rebins << ")\n";
        return 0;
    }

    int manual() {   // :128-233 is an OpenCV trackbar GUI
        std::cerr << "interactive mode is not available in the MI355X build; running the optimizer\n";
        return this->run();
    }

    void set_maxiter(int val) { this->max_itercount = val; }   // :236-238

    inline void set_time(ull t_) {   // :241-245
        this->current_time = t_;
        for (auto &e : *this->events) e.set_local_time(this->current_time);
    }

    void set_cloud(T *events_, int sc_) {   // :248-270 (the bounding box is reduced on the device)
        this->events = events_;
        this->scale = sc_;
        assert(this->scale % 2 != 0);   // :274
        for (auto &e : *this->events) e.reset();
    }

    ObjectModel get_model() { return this->model; }   // :285-287

    void set_model(ObjectModel m) {   // :289-299 (applied on the device when the slice is staged)
        this->model = m;
        this->warm_model = m;
        this->have_warm_model = true;
        if (accel.is_staged()) accel.set_model(m);
    }

    // per-event results, pulled from the device on demand
    void fetch_uv() { this->stage(); accel.compute_uv(this->events); }
    void writeout_events() { this->stage(); accel.writeout_events(this->events); }

    bf::Image2D<float> get_time_img() {
        this->stage();
        return accel.get_time_img(this->events, metric_wsizex, metric_wsizey, scale, (int)x_shift, (int)y_shift);
    }
    bf::Image2D<uint32_t> get_count_img() {
        this->stage();
        return accel.get_count_img(metric_wsizex, metric_wsizey, scale);
    }

    // EventFile::projection_img of this optimizer's events (event_file.h:460-515)
    bf::Image2D<uint8_t> get_projection_img(int sc, bool show_final) {
        this->stage();
        return accel.projection_img(sc, show_final);
    }

    // EventFile::color_time_img of this optimizer's events (event_file.h:649-747)
    std::vector<uint8_t> get_color_time_img(int sc, bool show_final, int *rows, int *cols) {
        this->stage();
        return accel.color_time_img(sc, show_final, rows, cols);
    }

    const bf_run_info &get_run_info() const { return last_info; }
    int get_scale_img_x() { this->stage(); return scale_img_x; }
    int get_scale_img_y() { this->stage(); return scale_img_y; }
};

#endif  // BF_HOST_OPTIMIZER_ROLLING_H
