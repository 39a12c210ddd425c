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

#include <chrono>
#include <unordered_map>

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
    FILE *slice_log = nullptr;

    ull slice_origin();
    void render_frame(OptimizerRolling<LinearEventPtrs> &optimizer);

public:
    DVS_flow(ull on_ev_change_, ull on_time_change_, ull start_time = 0)
        : on_ev_change(on_ev_change_), on_time_change(on_time_change_), time_diff(0), event_diff(0),
          last_slice_time(start_time), current_slice_time(start_time), accumulate(false), manual_mode(false),
          max_iter(-1), scale(3), generate_video(false), video_fps(30), generate_pictures(false),
          stm_disable(false), quiet(false), slices_done(0), slices_skipped(0), iterations_total(0) {}

    bool add_event(Event &ev);
    void recompute();
    // one CSV record per slice (slice,events,new_events,rc,iterations,ms,mevents_per_s)
    bool open_slice_log(const std::string &path) {
        slice_log = std::fopen(path.c_str(), "w");
        if (slice_log) std::fprintf(slice_log, "slice,events,new_events,rc,iterations,ms,mevents_per_s\n");
        return slice_log != nullptr;
    }
    ~DVS_flow() { if (slice_log) std::fclose(slice_log); }

    void set_accumulate(bool on = true) { accumulate = on; }
    LinearEventCloudTemplate<Event> get_accumulated();
    void set_manual_mode(bool on = true) { manual_mode = on; }
    void set_max_iter(int cap = -1) { max_iter = cap; }
    void set_scale(int s = 3) { scale = s; }
    void set_generate_video(bool val = true, std::string name = "out.avi", int framerate = 30) {   // :115-129
        generate_video = val; video_name = name; video_fps = framerate;   // (opened with the first frame, at the mosaic's size)
    }
    void set_generate_pictures(bool val = true, std::string img_prefix_ = "./") {
        generate_pictures = val; img_prefix = img_prefix_;
    }
    void set_stm_disable(bool off = true) { stm_disable = off; }
    void set_quiet(bool val = true) { this->quiet = val; }   // the reference parses --quiet but ignores it

    sll get_buf_size() { return this->ev_buffer.size(); }
    sll get_time_diff() { return this->time_diff; }
    sll get_buf_time_diff() { return (sll)(current_slice_time - slice_origin()); }   // dvs_flow.h:150-159
    ObjectModel get_last_model() { return this->last_model; }
    ull get_slices_done() const { return slices_done; }
    ull get_slices_skipped() const { return slices_skipped; }
    ull get_iterations_total() const { return iterations_total; }
};

// A new event enters the ring; a slice is solved as soon as enough events OR enough time have accumulated since the
// previous slice (either trigger; dvs_flow.h:164-181).  Returns whether this event closed a slice.
template <size_t MAX_SZ, sll SPAN> bool DVS_flow<MAX_SZ, SPAN>::add_event(Event &ev) {
    ev_buffer.push_back(ev);
    ++event_diff;
    current_slice_time = ev.timestamp;                        // (timestamps only grow)
    time_diff = (sll)(current_slice_time - last_slice_time);
    const bool due = event_diff >= (sll)on_ev_change || time_diff >= (sll)on_time_change;
    if (due) recompute();
    return due;
}

// Time origin of the slice in the ring: the oldest event of a full ring, else `SPAN` before the newest event (clamped
// at 0) -- dvs_flow.h:186-193.
template <size_t MAX_SZ, sll SPAN> ull DVS_flow<MAX_SZ, SPAN>::slice_origin() {
    if (ev_buffer.size() == MAX_SZ) return ev_buffer[MAX_SZ - 1].timestamp;
    return current_slice_time > (ull)SPAN ? current_slice_time - (ull)SPAN : 0;
}

// The reference's per-slice frame (dvs_flow.h:256-335): a 2 x 2 mosaic -- events as recorded on top, motion compensated
// below; projection image left, colour-coded time image right --, the four tiles computed on the device at scale 3.
template <size_t MAX_SZ, sll SPAN>
void DVS_flow<MAX_SZ, SPAN>::render_frame(OptimizerRolling<LinearEventPtrs> &optimizer) {
    const int rows = RES_X * 3, cols = RES_Y * 3;
    auto gray_tile = [&](bool compensated) {
        bf::Image2D<uint8_t> g = optimizer.get_projection_img(3, compensated);
        return bf::resize_bilinear(bf::gray_to_bgr(g.ptr(0), g.rows, g.cols), rows, cols);
    };
    auto colour_tile = [&](bool compensated) {
        int r = 0, c = 0;
        bf::FrameBGR f(0, 0);
        f.px = optimizer.get_color_time_img(3, compensated, &r, &c);
        f.rows = r; f.cols = c;
        return bf::resize_bilinear(f, rows, cols);
    };
    const bf::FrameBGR frame = bf::mosaic_2x2(gray_tile(true), colour_tile(true), gray_tile(false), colour_tile(false));
    if (generate_pictures) {
        const std::string base = img_prefix + "/frame_" + std::to_string(frame_count++);
        if (!bf::write_ppm(base + ".ppm", frame) && !quiet) std::cerr << "cannot write " << base << ".ppm\n";
        if (FILE *f = std::fopen((base + ".txt").c_str(), "w")) {   // what the reference draws with cv::putText, :277-315
            const ObjectModel &m = last_model;
            std::fprintf(f, "timestamp: %s\n%%realtime: %s\nTime diff (new): %s\nEvents: %zu\nNew events: %lld\n",
                         f2str(double(current_slice_time) * 1e-9).c_str(), f2str(double(on_time_change) / double(time_diff)).c_str(),
                         f2str(double(time_diff) * 1e-9).c_str(), (size_t)ev_buffer.size(), (long long)event_diff);
            std::fprintf(f, "Model:\nC: (%s, %s)\nShift: (%s, %s); total: (%s, %s)\nRot: %s total: %s\nDiv: %s total: %s\n",
                         f2str(m.cx).c_str(), f2str(m.cy).c_str(), f2str(m.dx).c_str(), f2str(m.dy).c_str(), f2str(m.total_dx).c_str(),
                         f2str(m.total_dy).c_str(), f2str(m.rot).c_str(), f2str(m.total_rot).c_str(), f2str(m.div).c_str(),
                         f2str(m.total_div).c_str());
            std::fclose(f);
        }
    }
    if (generate_video) {
        if (!outputvideo.is_open() && !outputvideo.open(video_name, frame.rows, frame.cols, video_fps))
            std::cout << "Could not open the output video for write" << std::endl;   // :329-331
        if (outputvideo.is_open()) outputvideo.write(frame);
    }
}

// One slice: the events now in the ring, their times made relative to the slice origin, one OptimizerRolling run --
// warm-started from the previous slice's model unless --stm-disable (the "short-term memory", dvs_flow.h:218-224) --,
// per-event flow fetched from the device, optional frame, bookkeeping (dvs_flow.h:185-347).
template <size_t MAX_SZ, sll SPAN> void DVS_flow<MAX_SZ, SPAN>::recompute() {
    const auto t_begin = std::chrono::steady_clock::now();
    const ull origin = slice_origin();
    const bool frames = generate_video || generate_pictures;
    if (frames)   // the frames are rendered at scale 3 whatever `scale` is: make room once
        (void)bf::DeviceContext::get((long long)MAX_SZ, 3 * RES_X + 3, 3 * RES_Y + 3);

    LinearEventPtrs slice;                     // newest -> oldest, the ring's iteration order
    slice.reserve(ev_buffer.size());
    for (auto &e : ev_buffer) slice.push_back(&e);

    int rc;
    bf_run_info info;
    {
        OptimizerRolling<LinearEventPtrs> optimizer;
        optimizer.set_cloud(&slice, scale);
        optimizer.set_time(origin);
        optimizer.set_maxiter(max_iter);
        if (!stm_disable) optimizer.set_model(last_model);
        rc = manual_mode ? optimizer.manual() : optimizer.run();
        last_model = optimizer.get_model();
        optimizer.fetch_uv();                  // "compute the actual u and v after minimizations are done", :233-235
        if (frames) render_frame(optimizer);
        info = optimizer.get_run_info();
    }
    ++slices_done;
    if (rc != 0) ++slices_skipped;
    iterations_total += (ull)info.iterations;

    // what the reference keeps of every past slice, and prints after each new one (:239-252)
    SliceSummary sm;
    sm.model = last_model;
    sm.size = 0;
    sm.first_ts = sm.last_ts = 0;
    for (auto &e : ev_buffer) {
        if (sm.size == 0) sm.first_ts = e.timestamp;
        sm.last_ts = e.timestamp;
        ++sm.size;
    }
    motion_memory.push_back(sm);
    if (!quiet) {
        std::cout << "\n\n------------------------\n";
        for (const SliceSummary &past : motion_memory)
            std::cout << past.model << "\n" << past.size << "\t" << past.first_ts << "\t" << past.last_ts << "\n";
    }
    if (slice_log) {   // one structured record per slice
        const double ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        std::fprintf(slice_log, "%llu,%zu,%lld,%d,%d,%.3f,%.3f\n", (unsigned long long)(slices_done - 1), sm.size,
                     (long long)event_diff, rc, (int)info.iterations, ms, ms > 0 ? sm.size / ms * 1e-3 : 0.0);
        std::fflush(slice_log);
    }

    if (accumulate) {   // a copy of the slice for get_accumulated(), oldest -> newest (:341-346)
        accumulated.emplace_back();
        LinearEventCloudTemplate<Event> &copy = accumulated.back();
        for (size_t k = ev_buffer.size(); k-- > 0;) copy.push_back(ev_buffer[k]);
    }
    event_diff = 0;
    last_slice_time = current_slice_time;
}

// Consecutive slices overlap (the ring spans more than the trigger interval), so an event is in several accumulated
// slices; the output keeps its FIRST copy (dvs_flow.h:351-389).  The reference finds the later copies with a linear
// scan of every later slice per event -- O(n^2) --, calling Event::operator== (event.h:39-45): same pixel, and, for
// a later-slice event e' at or before e in time (the scan stops at the first e' after e), e.timestamp - e'.timestamp
// < 0.1 ms.  Here every slice gets an index pixel -> positions (ascending in time) once, and an event only visits the
// events of its own pixel in the later slices: the same marks in the same order, O(n x slices).
template <size_t MAX_SZ, sll SPAN>
LinearEventCloudTemplate<Event> DVS_flow<MAX_SZ, SPAN>::get_accumulated() {
    typedef std::unordered_map<unsigned long long, std::vector<uint32_t>> PixelIndex;
    auto pixel_key = [](const Event &e) { return ((unsigned long long)e.fr_x << 32) | (unsigned long long)e.fr_y; };
    if (!quiet) std::cout << "Aggregating events into one cloud...\n";
    const size_t nslices = accumulated.size();
    std::vector<PixelIndex> index(nslices);
    for (size_t j = 1; j < nslices; ++j) {   // (slice 0 is never searched)
        uint32_t pos = 0;
        for (auto &e : accumulated[j]) index[j][pixel_key(e)].push_back(pos++);
    }
    LinearEventCloudTemplate<Event> unique;
    for (size_t i = 0; i < nslices; ++i) {
        if (!quiet) std::cout << "\tBuffer: " << i << "\n";
        for (auto &e : accumulated[i]) {
            if (e.t == -1) continue;          // a later copy of an event already written
            for (size_t j = i + 1; j < nslices; ++j) {
                const auto hit = index[j].find(pixel_key(e));
                if (hit == index[j].end()) continue;
                for (uint32_t pos : hit->second) {
                    Event &later = accumulated[j][pos];
                    if (later - e > 0) break;                 // positions ascend in time: nothing further can match
                    if (later.t != -1 && e == later) later.t = -1;
                }
            }
            unique.push_back(e);
        }
    }
    if (!quiet) std::cout << "Final buffer contains " << unique.size() << " events." << std::endl;
    return unique;
}

#endif  // BF_HOST_DVS_FLOW_H
