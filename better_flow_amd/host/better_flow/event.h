// event.h -- one DVS event and the metadata the pipeline keeps with it (mirror of the
// reference's better_flow/event.h:7-59,123-142).
//
// The per-event warp arithmetic of the reference (project_4param_reinit / apply_project,
// event.h:99-110,164-168) is NOT here: it runs on the GPU (k_warp_scatter / k_bin_warp_scatter)
// and its results come back through AccelLib::writeout_events / compute_uv.
#ifndef BF_HOST_EVENT_H
#define BF_HOST_EVENT_H

#include <better_flow/common.h>

class Event {
public:
    uint fr_x, fr_y;   // row, column
    sll t;             // ns relative to the slice start
    ull timestamp;     // ns
    bool noise;
    bool valid;

    double pr_x, pr_y;
    double nx, ny, nz;
    double u, v;

    double best_u, best_v, max_score;
    double best_pr_x, best_pr_y;

    Event()
        : fr_x(UINT_MAX), fr_y(UINT_MAX), t(ULLONG_MAX), timestamp(LLONG_MAX), noise(true), valid(false),
          pr_x(NAN), pr_y(NAN), nx(0), ny(0), nz(NZ), u(0), v(0), best_u(0), best_v(0), max_score(0),
          best_pr_x(NAN), best_pr_y(NAN) {}

    Event(uint x_, uint y_, ull t_)
        : fr_x(x_), fr_y(y_), t(t_), timestamp(t_), noise(false), valid(false), pr_x(x_), pr_y(y_), nx(0),
          ny(0), nz(NZ), u(0), v(0), best_u(0), best_v(0), max_score(0), best_pr_x(x_), best_pr_y(y_) {}

    inline sll operator-(const Event &rhs) { return sll(this->timestamp) - sll(rhs.timestamp); }

    // event.h:39-45, including its timestamp / t mix-up in the second branch
    inline bool operator==(const Event &rhs) {
        bool coord_eq = (this->fr_x == rhs.fr_x) && (this->fr_y == rhs.fr_y);
        ull dt = (this->timestamp >= rhs.timestamp) ? this->timestamp - rhs.timestamp
                                                    : rhs.timestamp - this->t;
        bool time_eq = dt < 100000;   // dt < 0.1 ms
        return coord_eq && time_eq;
    }
    inline bool operator!=(const Event &rhs) { return !(*this == rhs); }

    inline uint get_x() const { return this->fr_x; }
    inline uint get_y() const { return this->fr_y; }

    inline void reset() {   // event.h:54-59
        this->pr_x = this->fr_x;
        this->pr_y = this->fr_y;
        this->nx = this->ny = 0;
        this->u = this->v = 0;
    }

    inline void set_local_time(ull t_) {   // event.h:61-63
        this->t = (this->timestamp > t_) ? this->timestamp - t_ : -sll(t_ - this->timestamp);
    }

    inline void assume_score(double score) {   // event.h:123-129
        this->max_score = score;
        this->best_u = this->u;
        this->best_v = this->v;
        this->best_pr_x = this->pr_x;
        this->best_pr_y = this->pr_y;
    }

    // Event::compute_uv (event.h:135-142) with the device-computed u, v
    inline void set_uv(double u_, double v_) {
        this->u = u_;
        this->v = v_;
        this->best_u = u_;
        this->best_v = v_;
    }
};

#endif  // BF_HOST_EVENT_H
