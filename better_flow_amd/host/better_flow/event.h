// event.h -- the per-event record the host side of the pipeline carries around.
//
// Public data members and method names are those callers of the reference's Event use
// (better_flow/event.h:7-63,123-142: `ev.fr_x`, `ev.best_u`, `ev.set_local_time(...)`); the bodies
// are this build's.  The warp arithmetic of the reference record (project_4param_reinit /
// apply_project, event.h:99-110,164-168) is deliberately absent: it runs on the GPU
// (k_warp_scatter / k_bin_warp_scatter) and only its results come back, through
// AccelLib::writeout_events / compute_uv.
#ifndef BF_HOST_EVENT_H
#define BF_HOST_EVENT_H

#include <better_flow/common.h>

class Event {
public:
    // sensor address: fr_x is the ROW, fr_y the COLUMN (the file reader swaps, bf_mc.cpp:192,200)
    uint fr_x, fr_y;
    sll t;            // ns, signed, relative to the start of the slice being solved
    ull timestamp;    // ns, absolute
    bool noise, valid;

    double pr_x, pr_y;        // warped position
    double nx, ny, nz;        // per-event flow in n-units (nz is the constant NZ)
    double u, v;              // px/s
    double best_u, best_v, max_score, best_pr_x, best_pr_y;

    // A default-constructed record is a "no event": all-ones address, flagged as noise.
    Event() { blank(); fr_x = fr_y = UINT_MAX; t = sll(ULLONG_MAX); timestamp = ull(LLONG_MAX);
              noise = true; pr_x = pr_y = best_pr_x = best_pr_y = NAN; }

    Event(uint row, uint col, ull ns) { blank(); fr_x = row; fr_y = col; t = sll(ns); timestamp = ns;
                                        pr_x = best_pr_x = row; pr_y = best_pr_y = col; }

    uint get_x() const { return fr_x; }
    uint get_y() const { return fr_y; }

    // signed distance in time between two events, ns
    sll operator-(const Event &o) const { return sll(timestamp) - sll(o.timestamp); }

    // "Same event" for the overlap de-duplication of DVS_flow::get_accumulated: same pixel, less
    // than 0.1 ms apart.  When `o` is the later event the reference measures the gap from this
    // event's slice-relative `t`, not from its `timestamp` (event.h:42); callers observe that, so
    // it is reproduced: gap = o.timestamp - ull(t).
    bool operator==(const Event &o) const {
        if (fr_x != o.fr_x || fr_y != o.fr_y) return false;
        const ull gap = timestamp < o.timestamp ? o.timestamp - ull(t) : timestamp - o.timestamp;
        return gap < 100000ull;
    }
    bool operator!=(const Event &o) const { return !(*this == o); }

    // undo every warp: position back to the sensor address, no flow (event.h:54-59)
    void reset() { pr_x = double(fr_x); pr_y = double(fr_y); nx = ny = u = v = 0.0; }

    // slice-relative time; events older than the slice start get negative times (event.h:61-63)
    void set_local_time(ull slice_start) {
        t = timestamp > slice_start ? sll(timestamp - slice_start) : -sll(slice_start - timestamp);
    }

    // keep the current flow / position as the best seen so far (event.h:123-129)
    void assume_score(double score) {
        max_score = score;
        best_u = u; best_v = v;
        best_pr_x = pr_x; best_pr_y = pr_y;
    }

    // Event::compute_uv + the copy to best_* at its call site (event.h:135-142, dvs_flow.h:233-235),
    // with u, v computed on the device (k_compute_uv)
    void set_uv(double u_, double v_) { u = best_u = u_; v = best_v = v_; }

private:
    void blank() {
        noise = valid = false;
        nx = ny = u = v = best_u = best_v = max_score = 0.0;
        nz = NZ;
    }
};

#endif  // BF_HOST_EVENT_H
