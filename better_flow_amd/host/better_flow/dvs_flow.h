// dvs_flow.h -- the slice manager (mirror of the reference's better_flow/dvs_flow.h:21-389):
// a time-span ring of events, triggers on new-event count / elapsed time, one optimizer per
// slice warm-started from the previous model ("STM"), accumulation of processed events.
// Same public interface; the optimizer it drives runs on the MI355X.
//
// --img / --video render the reference's per-slice frame (dvs_flow.h:256-335): the 2 x 2 mosaic of the projection
// images and colour-coded time images, raw on top, motion compensated below -- the four tiles are computed on the
// device, composition and containers are in better_flow/frame_writer.h (PPM / uncompressed AVI, statistics in a
// side-car text file instead of cv::putText).  Not reproduced: the unbounded `motion_memory` copies of every slice
// (:239-242): only the per-slice summary the reference prints (:245-252) is kept.
#ifndef BF_HOST_DVS_FLOW_H
#define BF_HOST_DVS_FLOW_H

#include <better_flow/common.h>
#include <better_flow/event.h>
#include <better_flow/event_file.h>
#include <better_flow/frame_writer.h>
#include <better_flow/optimizer_rolling.h>

#include <queue>

inline std::string f2str(double v) {   // dvs_flow.h:14-19, as written ("1.05" prints as "1.5")
    int base = int(v * 100);
    std::string ret = std::to_string(base / 100) + ".";
    ret += std::to_string(std::abs(base) % 100);
    return ret;
}

template <size_t MAX_SZ, sll SPAN> class DVS_flow {
public:
    // Buffer for incoming events (aka 'slice')
    CircularArray<Event, MAX_SZ, SPAN> ev_buffer;

protected:
    ull on_ev_change, on_time_change;          // triggers
    sll time_diff, event_diff;                 // time passed / new event count
    ull last_slice_time, current_slice_time;
    ObjectModel last_model;                    // starting point for the next minimizer
    bool accumulate;
    std::vector<LinearEventCloudTemplate<Event>> accumulated;

    struct SliceSummary {                      // what dvs_flow.h:245-252 prints per past slice
        ObjectModel model;
        size_t size;
        ull first_ts, last_ts;
    };
    std::vector<SliceSummary> motion_memory;

    bool manual_mode;
    int max_iter;
    int scale;
    bool generate_video;
    int video_fps;
    std::string video_name;
    bf::AviWriter outputvideo;
    bool generate_pictures;
    std::string img_prefix;
    bool stm_disable;
    bool quiet;
    ull slices_done, slices_skipped, iterations_total;
    ull frame_count = 0;

public:
    DVS_flow(ull on_ev_change_, ull on_time_change_, ull start_time = 0)
        : on_ev_change(on_ev_change_), on_time_change(on_time_change_), time_diff(0), event_diff(0),
          last_slice_time(start_time), current_slice_time(start_time), accumulate(false), manual_mode(false),
          max_iter(-1), scale(3), generate_video(false), video_fps(30), generate_pictures(false),
          stm_disable(false), quiet(false), slices_done(0), slices_skipped(0), iterations_total(0) {}

    bool add_event(Event &ev);
    void recompute();

    void set_accumulate(bool val = true) { this->accumulate = val; }
    LinearEventCloudTemplate<Event> get_accumulated();
    void set_manual_mode(bool val = true) { this->manual_mode = val; }
    void set_max_iter(int val = -1) { this->max_iter = val; }
    void set_scale(int val = 3) { this->scale = val; }
    void set_generate_video(bool val = true, std::string name = "out.avi", int framerate = 30) {   // :115-129
        this->video_fps = framerate;
        this->generate_video = val;
        this->video_name = name;   // opened with the first frame, at the size of the mosaic
    }
    void set_generate_pictures(bool val = true, std::string img_prefix_ = "./") {
        this->generate_pictures = val;
        this->img_prefix = img_prefix_;
    }
    void set_stm_disable(bool val = true) { this->stm_disable = val; }
    void set_quiet(bool val = true) { this->quiet = val; }   // the reference parses --quiet but ignores it

    sll get_buf_size() { return this->ev_buffer.size(); }
    sll get_time_diff() { return this->time_diff; }
    sll get_buf_time_diff() {   // dvs_flow.h:150-159
        ull slice_start_time = 0;
        if (this->ev_buffer.size() == MAX_SZ) {
            slice_start_time = this->ev_buffer[MAX_SZ - 1].timestamp;
        } else {
            slice_start_time = (this->current_slice_time > (ull)SPAN) ? this->current_slice_time - SPAN : 0;
        }
        return this->current_slice_time - slice_start_time;
    }
    ObjectModel get_last_model() { return this->last_model; }
    ull get_slices_done() const { return slices_done; }
    ull get_slices_skipped() const { return slices_skipped; }
    ull get_iterations_total() const { return iterations_total; }
};

template <size_t MAX_SZ, sll SPAN> bool DVS_flow<MAX_SZ, SPAN>::add_event(Event &ev) {   // :164-181
    this->ev_buffer.push_back(ev);
    this->event_diff++;
    this->current_slice_time = ev.timestamp;
    this->time_diff = this->current_slice_time - this->last_slice_time;   // time only increases
    if ((this->event_diff < (sll)this->on_ev_change) && (this->time_diff < (sll)this->on_time_change)) {
        return false;
    }
    this->recompute();
    return true;
}

template <size_t MAX_SZ, sll SPAN> void DVS_flow<MAX_SZ, SPAN>::recompute() {   // :185-347
    ull slice_start_time = 0;
    if (this->ev_buffer.size() == MAX_SZ) {
        slice_start_time = this->ev_buffer[MAX_SZ - 1].timestamp;
    } else {
        slice_start_time = (this->current_slice_time > (ull)SPAN) ? this->current_slice_time - SPAN : 0;
    }

    if (this->generate_video || this->generate_pictures)   // the frames are rendered at scale 3 whatever `scale` is
        (void)bf::DeviceContext::get((long long)MAX_SZ, 3 * RES_X + 3, 3 * RES_Y + 3);

    LinearEventPtrs e_ptrs;
    e_ptrs.reserve(this->ev_buffer.size());
    for (auto &e : this->ev_buffer) e_ptrs.push_back(&e);

    // The queue of 'objects' to process; an object is a pair of events and an object model
    std::queue<std::pair<LinearEventPtrs, ObjectModel>> task_queue;
    task_queue.push(std::make_pair(e_ptrs, this->last_model));

    while (!task_queue.empty()) {
        OptimizerRolling<LinearEventPtrs> optimizer;
        optimizer.set_cloud(&task_queue.front().first, this->scale);
        optimizer.set_time(slice_start_time);
        optimizer.set_maxiter(this->max_iter);
        if (!this->stm_disable) optimizer.set_model(task_queue.front().second);   // :218-219
        int rc = this->manual_mode ? optimizer.manual() : optimizer.run();
        this->last_model = optimizer.get_model();
        // :233-235 "compute the actual u and v after minimizations are done" (on the device)
        optimizer.fetch_uv();
        if (this->generate_video || this->generate_pictures) {   // :256-335
            const int fr = RES_X * 3, fc = RES_Y * 3;
            bf::Image2D<uint8_t> pr_f = optimizer.get_projection_img(3, false), pr_t = optimizer.get_projection_img(3, true);
            int cr = 0, cc = 0;
            bf::FrameBGR col_f(0, 0), col_t(0, 0);
            col_f.px = optimizer.get_color_time_img(3, false, &cr, &cc); col_f.rows = cr; col_f.cols = cc;
            col_t.px = optimizer.get_color_time_img(3, true, &cr, &cc); col_t.rows = cr; col_t.cols = cc;
            const bf::FrameBGR frame = bf::mosaic_2x2(
                bf::resize_bilinear(bf::gray_to_bgr(pr_t.ptr(0), pr_t.rows, pr_t.cols), fr, fc), bf::resize_bilinear(col_t, fr, fc),
                bf::resize_bilinear(bf::gray_to_bgr(pr_f.ptr(0), pr_f.rows, pr_f.cols), fr, fc), bf::resize_bilinear(col_f, fr, fc));
            if (this->generate_pictures) {
                const std::string base = this->img_prefix + "/frame_" + std::to_string(this->frame_count);
                if (!bf::write_ppm(base + ".ppm", frame) && !this->quiet) std::cerr << "cannot write " << base << ".ppm\n";
                if (FILE *f = std::fopen((base + ".txt").c_str(), "w")) {   // the cv::putText lines, :277-315
                    const double slice_time_width = double(this->time_diff) / 1000000000.0;
                    const double speedup = double(this->on_time_change) / double(this->time_diff);
                    const ObjectModel &m = this->last_model;
                    std::fprintf(f, "timestamp: %s\n%%realtime: %s\nTime diff (new): %s\nEvents: %zu\nNew events: %lld\n",
                                 f2str(double(this->current_slice_time) / 1000000000.0).c_str(), f2str(speedup).c_str(),
                                 f2str(slice_time_width).c_str(), (size_t)this->ev_buffer.size(), (long long)this->event_diff);
                    std::fprintf(f, "Model:\nC: (%s, %s)\nShift: (%s, %s); total: (%s, %s)\nRot: %s total: %s\nDiv: %s total: %s\n",
                                 f2str(m.cx).c_str(), f2str(m.cy).c_str(), f2str(m.dx).c_str(), f2str(m.dy).c_str(),
                                 f2str(m.total_dx).c_str(), f2str(m.total_dy).c_str(), f2str(m.rot).c_str(),
                                 f2str(m.total_rot).c_str(), f2str(m.div).c_str(), f2str(m.total_div).c_str());
                    std::fclose(f);
                }
                this->frame_count++;
            }
            if (this->generate_video) {
                if (!outputvideo.is_open() && !outputvideo.open(this->video_name, frame.rows, frame.cols, this->video_fps))
                    std::cout << "Could not open the output video for write" << std::endl;   // :329-331
                if (outputvideo.is_open()) outputvideo.write(frame);
            }
        }
        slices_done++;
        if (rc != 0) slices_skipped++;
        iterations_total += optimizer.get_run_info().iterations;
        task_queue.pop();
    }

    // per-slice summary in the reference's format (:239-252)
    SliceSummary sm;
    sm.model = this->last_model;
    sm.size = this->ev_buffer.size();
    if (sm.size > 0) {
        size_t n_iter = 0;
        for (auto &e : this->ev_buffer) {
            if (n_iter == 0) sm.first_ts = e.timestamp;
            sm.last_ts = e.timestamp;
            n_iter++;
        }
        sm.size = n_iter;
    } else {
        sm.first_ts = sm.last_ts = 0;
    }
    this->motion_memory.push_back(sm);
    if (!this->quiet) {
        std::cout << "\n\n------------------------\n";
        for (auto &slice : this->motion_memory) {
            std::cout << slice.model << "\n";
            std::cout << slice.size << "\t" << slice.first_ts << "\t" << slice.last_ts << "\n";
        }
    }

    this->event_diff = 0;
    this->last_slice_time = this->current_slice_time;

    if (this->accumulate) {   // :341-346, oldest -> newest
        LinearEventCloudTemplate<Event> cur_buf;
        for (long int i = (long int)this->ev_buffer.size() - 1; i >= 0; i--) cur_buf.push_back(this->ev_buffer[i]);
        this->accumulated.push_back(cur_buf);
    }
}

template <size_t MAX_SZ, sll SPAN>
LinearEventCloudTemplate<Event> DVS_flow<MAX_SZ, SPAN>::get_accumulated() {   // :351-389
    LinearEventCloudTemplate<Event> ret;
    if (!quiet) std::cout << "Aggregating events into one cloud...\n";
    for (ull i = 0; i < this->accumulated.size(); ++i) {
        if (!quiet) std::cout << "\tBuffer: " << i << "\n";
        auto &buf = this->accumulated[i];
        for (auto &e : buf) {
            if (e.t == -1) continue;
            Event ev = e;
            float avg_cnt = 1;
            for (ull j = i + 1; j < this->accumulated.size(); ++j) {
                auto &buf_next = this->accumulated[j];
                for (auto &e_ : buf_next) {
                    if (e_ - e > 0) break;
                    if (e_.t == -1) continue;
                    if (e != e_) continue;
                    e_.t = -1;
                }
            }
            ev.best_u /= avg_cnt;
            ev.best_v /= avg_cnt;
            ret.push_back(ev);
        }
    }
    if (!quiet) std::cout << "FInal buffer contains " << ret.size() << " events." << std::endl;
    return ret;
}

#endif  // BF_HOST_DVS_FLOW_H
