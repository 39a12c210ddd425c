// optimizer_sampler.h -- the contrast-score optimiser (mirror of the reference's
// better_flow/optimizer_sampler.h:12-68 and optimizer_sampler.cpp).  Same constructors and public
// methods (run, get_nx, get_ny); the score evaluations (iteration_step: Event::project of every
// event, saturating 8-bit count image, Gaussian blur, mean of the non-zero pixels, :120-153,192-205)
// execute on the GPU behind bf_local_iteration_step, the coordinate descent of run() (:4-38) around
// them.  manual() (:41-87) is an OpenCV trackbar GUI and is not part of this path.
//
// The blur is this build's own stated 8-bit Gaussian (include/bf_accel.h): the reference calls
// cv::GaussianBlur of an un-versioned OpenCV there, so that one stage has no pinned parity.
#ifndef BF_HOST_OPTIMIZER_SAMPLER_H
#define BF_HOST_OPTIMIZER_SAMPLER_H

#include <better_flow/accel_lib.h>
#include <better_flow/common.h>
#include <better_flow/datastructures.h>
#include <better_flow/event.h>

class OptimizerLocal {
protected:
    LinearEventCloud *events;
    Event event_c;
    AccelLib accel;
    bf::Image2D<uint8_t> project_img;

    int scale;
    int metric_wsizex, metric_wsizey;
    int scale_img_x, scale_img_y;
    int wsz;   // > 0: window of wsz sensor pixels around event_c; 0: bounding box of the cloud

    // Gradient descent variables (optimizer_sampler.h:25-28)
    double nx, ny;
    double last_score, dscore;
    double dnx, dny, dn_th;
    long long evaluations;

    void stage() {
        if (accel.is_staged()) return;
        accel.init_gpu(this->events, this->scale_img_x > 64 ? this->scale_img_x : 64,
                       this->scale_img_y > 64 ? this->scale_img_y : 64);
        bf_local_window w;
        accel.local_set_window(this->scale, this->wsz, (int)event_c.fr_x, (int)event_c.fr_y, (long long)event_c.t, &w);
        // the device computed the same window from the same events
        assert(w.metric_wsizex == metric_wsizex && w.metric_wsizey == metric_wsizey);
        assert(wsz > 0 || (w.c_fr_x == (int)event_c.fr_x && w.c_fr_y == (int)event_c.fr_y));
    }

    void update_fields() {   // optimizer_sampler.cpp:205-212
        assert(this->scale % 2 != 0);
        this->scale_img_x = this->metric_wsizex + this->scale;
        this->scale_img_y = this->metric_wsizey + this->scale;
    }

public:
    // optimizer_sampler.h:29-33
    OptimizerLocal(LinearEventCloud *events_, Event &e_, int sc_, int wsz_)
        : events(events_), event_c(e_), scale(sc_), metric_wsizex(sc_ * wsz_), metric_wsizey(sc_ * wsz_), wsz(wsz_),
          nx(0), ny(0), last_score(0), dscore(0), dnx(0), dny(0), dn_th(0), evaluations(0) {
        this->update_fields();
    }

    // optimizer_sampler.h:35-48: window = bounding box of the cloud, centre event in its middle (t = 0)
    OptimizerLocal(LinearEventCloud *events_, int sc_)
        : events(events_), scale(sc_), wsz(0), nx(0), ny(0), last_score(0), dscore(0), dnx(0), dny(0), dn_th(0),
          evaluations(0) {
        int x_min = this->events->x_min, y_min = this->events->y_min;
        int x_max = this->events->x_max, y_max = this->events->y_max;
        this->metric_wsizex = sc_ * (x_max - x_min);
        this->metric_wsizey = sc_ * (y_max - y_min);
        this->event_c = Event((x_max - x_min) / 2 + x_min, (y_max - y_min) / 2 + y_min, 0);
        this->update_fields();
    }

    // optimizer_sampler.cpp:4-38.  Returns 0 (optimised) or 1 (window too small, :9-13).
    int run() {
        this->stage();
        bf_local_state st;
        int rc = accel.local_run(&st);
        this->nx = st.nx; this->ny = st.ny;
        this->last_score = st.last_score;
        this->dnx = st.dnx; this->dny = st.dny; this->dn_th = st.dn_th;
        this->evaluations = st.evaluations;
        if (rc == BF_SKIPPED) {
            if (VERBOSE)
                std::cout << "Window size is too small; (" << this->metric_wsizex << ", " << this->metric_wsizey
                          << "). Skipping...\n";
            return 1;
        }
        if (VERBOSE) std::cout << "\tMinimization: " << st.evaluations << " score evaluations\n";
        return 0;
    }

    int manual() {   // optimizer_sampler.cpp:41-87 is an OpenCV trackbar GUI
        std::cerr << "interactive mode is not available in the MI355X build; running the optimizer\n";
        return this->run();
    }

    inline double get_nx() { return this->nx; }   // optimizer_sampler.h:53-54
    inline double get_ny() { return this->ny; }

    // One score evaluation (the private iteration_step, optimizer_sampler.cpp:120-153); project_img is
    // refreshed when want_img is set.
    double iteration_step(double nx_, double ny_, bool want_img = false) {
        this->stage();
        if (want_img) this->project_img = bf::Image2D<uint8_t>(scale_img_x, scale_img_y);
        return accel.local_iteration_step(nx_, ny_, want_img ? this->project_img.ptr(0) : nullptr);
    }
    const bf::Image2D<uint8_t> &get_project_img() const { return project_img; }
    double get_last_score() const { return last_score; }
    long long get_evaluations() const { return evaluations; }
};

#endif  // BF_HOST_OPTIMIZER_SAMPLER_H
