// stream_flow.h -- the slice former for high event rates: the reference's event ring and triggers
// (CircularArray, datastructures.h:6-115; DVS_flow::add_event / recompute, dvs_flow.h:164-347) on a
// structure-of-arrays ring in PINNED memory with a zero-copy hand-off to the device.
//
// Why a second front end next to DVS_flow: DVS_flow keeps the reference's public `ev_buffer` of 152-byte
// Event records, so every slice is repacked AoS -> SoA on the host (accel_lib.h:91-99 in the reference,
// AccelLib::init_gpu here): ~10 ms per 1M-event slice, an order of magnitude more than the GPU needs for a
// warm-started slice.  StreamFlow keeps row / column / timestamp (and the per-event flow that comes back) as
// parallel arrays; a slice is one or two contiguous pieces of the ring, copied by DMA
// (bf_upload_ring_async), and Event::set_local_time runs on the device.
//
// Same semantics as DVS_flow, element for element:
//   * ring of at most MAX_SZ events spanning at most SPAN ns (push_back :31-44, fix_span :46-59);
//   * trigger: events since the last slice >= on_ev_change OR time since it >= on_time_change (:164-181);
//   * slice = the ring content, EXCEPT that a full ring leaves its oldest element out (end() :71-76);
//   * slice start time = oldest timestamp if the ring is full, else max(now - SPAN, 0) (:186-191);
//   * warm start from the previous slice's model unless stm_disable (:218-219).
// Order inside a slice is oldest -> newest here (the reference iterates newest -> oldest); the device
// accumulates integers, so the order does not change any result.
// One deviation: when run() stops at the small-window guard, the reference marks the ring's events as noise
// (optimizer_rolling.h:52-53) and later overlapping slices leave them out of the time image; DVS_flow carries that flag
// in Event::noise, this ring has no noise array (bf_upload_ring_async takes none).  The two front ends therefore agree
// as long as no slice of the stream is skipped by THAT guard (slices skipped for having fewer than 1000 events do not
// set the flag) -- the case the equivalence tests cover.
#ifndef BF_HOST_STREAM_FLOW_H
#define BF_HOST_STREAM_FLOW_H

#include <better_flow/accel_lib.h>
#include <better_flow/common.h>
#include <better_flow/object_model.h>

#include <vector>

namespace bf {

template <size_t MAX_SZ, sll SPAN> class StreamFlow {
    bf_ctx *ctx;
    // the ring (pinned): slot k holds the k-th arrival modulo MAX_SZ
    int32_t *fr_x, *fr_y;
    uint64_t *ts;
    double *best_u, *best_v;       // per-event flow of the last slice an event took part in (pageable)
    std::vector<double> u_tmp, v_tmp;
    size_t head_id, current_size;  // CircularArray's fields (datastructures.h:17-19)
    bool span_checked;

    ull on_ev_change, on_time_change;
    sll time_diff, event_diff;
    ull last_slice_time, current_slice_time;
    ObjectModel last_model;
    int max_iter, scale;
    bool stm_disable, want_flow;
    ull slices_done, slices_skipped, iterations_total;
    bf_run_info last_info;

    void check(int rc, const char *what) const {
        if (rc < 0)
            throw bf::AccelError(rc, std::string("StreamFlow::") + what + " failed (" + std::to_string(rc) + "): " +
                                         (ctx ? bf_last_error(ctx) : "no ctx"));
    }

    void fix_span() {   // datastructures.h:46-59
        if (span_checked) return;
        span_checked = true;
        size_t tail_id = ((1 - int(current_size - head_id)) + MAX_SZ) % MAX_SZ;
        size_t removed = 0;
        while ((sll)(ts[head_id] - ts[tail_id]) > SPAN) {
            removed++;
            tail_id++;
            if (tail_id >= MAX_SZ) tail_id = 0;
        }
        current_size -= removed;
    }

public:
    struct Slice {
        size_t first, n;          // ring index of the oldest event of the slice, number of events
        ull start_time;
    };

    StreamFlow(ull on_ev_change_, ull on_time_change_, ull start_time = 0)
        : ctx(nullptr), fr_x(nullptr), fr_y(nullptr), ts(nullptr), best_u(nullptr), best_v(nullptr), head_id(0),
          current_size(0), span_checked(true), on_ev_change(on_ev_change_), on_time_change(on_time_change_),
          time_diff(0), event_diff(0), last_slice_time(start_time), current_slice_time(start_time), max_iter(-1),
          scale(3), stm_disable(false), want_flow(true), slices_done(0), slices_skipped(0), iterations_total(0) {
        std::memset(&last_info, 0, sizeof(last_info));
        ctx = bf::DeviceContext::get((long long)MAX_SZ, scale * RES_X + scale, scale * RES_Y + scale);
        void *p = nullptr;   // (pinned host memory is not tied to the ctx object: it survives a re-sized context)
        check(bf_host_alloc(ctx, (int64_t)MAX_SZ * 4, &p), "alloc"); fr_x = (int32_t *)p;
        check(bf_host_alloc(ctx, (int64_t)MAX_SZ * 4, &p), "alloc"); fr_y = (int32_t *)p;
        check(bf_host_alloc(ctx, (int64_t)MAX_SZ * 8, &p), "alloc"); ts = (uint64_t *)p;
        best_u = new double[MAX_SZ]();
        best_v = new double[MAX_SZ]();
    }
    ~StreamFlow() {
        if (ctx) {
            (void)bf_synchronize(ctx);
            (void)bf_host_free(ctx, fr_x); (void)bf_host_free(ctx, fr_y); (void)bf_host_free(ctx, ts);
        }
        delete[] best_u;
        delete[] best_v;
    }
    StreamFlow(const StreamFlow &) = delete;
    StreamFlow &operator=(const StreamFlow &) = delete;

    void set_max_iter(int v = -1) { max_iter = v; }
    void set_scale(int v = 3) { scale = v; }
    void set_stm_disable(bool v = true) { stm_disable = v; }
    void set_want_flow(bool v = true) { want_flow = v; }   // fetch per-event (u, v) after every slice

    // DVS_flow::add_event (dvs_flow.h:164-181); row / column as Event::fr_x / fr_y
    bool add_event(uint32_t row, uint32_t col, ull timestamp) {
        span_checked = false;                                  // CircularArray::push_back, :31-44
        current_size += (current_size >= MAX_SZ) ? 0 : 1;
        head_id++;
        if (head_id >= MAX_SZ) head_id = 0;
        fr_x[head_id] = (int32_t)row; fr_y[head_id] = (int32_t)col; ts[head_id] = timestamp;
        best_u[head_id] = best_v[head_id] = 0.0;
        event_diff++;
        current_slice_time = timestamp;
        time_diff = current_slice_time - last_slice_time;
        if ((event_diff < (sll)on_ev_change) && (time_diff < (sll)on_time_change)) return false;
        recompute();
        return true;
    }

    size_t size() { fix_span(); return current_size; }

    // The slice recompute() would hand to the optimizer now.
    Slice current_slice() {
        fix_span();
        Slice s;
        const size_t oldest = ((1 - int(current_size - head_id)) + MAX_SZ) % MAX_SZ;   // tail of the ring
        if (current_size == MAX_SZ) {            // full ring: iteration stops one short (:71-76), start = oldest ts
            s.start_time = ts[oldest];
            s.first = (oldest + 1) % MAX_SZ;
            s.n = current_size - 1;
        } else {
            s.start_time = (current_slice_time > (ull)SPAN) ? current_slice_time - SPAN : 0;
            s.first = oldest;
            s.n = current_size;
        }
        return s;
    }

    // DVS_flow::recompute (dvs_flow.h:185-347) without the rendering / accumulation branches
    void recompute() {
        const Slice s = current_slice();
        if (s.n > 0) {
            ctx = bf::DeviceContext::get((long long)MAX_SZ, scale * RES_X + scale, scale * RES_Y + scale);
            check(bf_upload_ring_async(ctx, fr_x, fr_y, ts, (int64_t)MAX_SZ, (int64_t)s.first, (int64_t)s.n, s.start_time),
                  "upload_ring");
            check(bf_commit_upload(ctx), "commit_upload");
            bf_window w;
            check(bf_set_cloud(ctx, scale, RES_X, RES_Y, &w), "set_cloud");   // (also waits for the DMA)
            if (!stm_disable) {
                bf_model m = last_model.to_abi();
                check(bf_set_model(ctx, &m), "set_model");                      // :218-219
            }
            bf_run_opts o;
            bf_run_opts_default(&o);
            o.max_iter = max_iter; o.res_x = RES_X; o.res_y = RES_Y; o.want_uv = want_flow ? 1 : 0;
            bf_model out;
            int rc = bf_run(ctx, &o, &out, &last_info);
            check(rc, "run");
            last_model = ObjectModel(out);
            if (want_flow) {   // :233-235; slot i of the slice is ring slot (first + i) mod MAX_SZ
                u_tmp.resize(s.n); v_tmp.resize(s.n);
                check(bf_compute_uv(ctx, u_tmp.data(), v_tmp.data()), "compute_uv");
                const size_t n0 = (s.first + s.n <= MAX_SZ) ? s.n : MAX_SZ - s.first;
                std::memcpy(best_u + s.first, u_tmp.data(), n0 * sizeof(double));
                std::memcpy(best_v + s.first, v_tmp.data(), n0 * sizeof(double));
                std::memcpy(best_u, u_tmp.data() + n0, (s.n - n0) * sizeof(double));
                std::memcpy(best_v, v_tmp.data() + n0, (s.n - n0) * sizeof(double));
            }
            slices_done++;
            if (rc != 0) slices_skipped++;
            iterations_total += last_info.iterations;
        } else {   // the reference runs its optimizer on the empty cloud; the window guard skips it (:49-55)
            slices_done++;
            slices_skipped++;
        }
        event_diff = 0;
        last_slice_time = current_slice_time;
    }

    // element access, idx 0 = newest (CircularArray::operator[], :61-64)
    size_t slot(size_t idx) const { return ((int(head_id) - int(idx)) + MAX_SZ) % MAX_SZ; }
    uint32_t row(size_t idx) const { return (uint32_t)fr_x[slot(idx)]; }
    uint32_t col(size_t idx) const { return (uint32_t)fr_y[slot(idx)]; }
    ull timestamp(size_t idx) const { return ts[slot(idx)]; }
    double u(size_t idx) const { return best_u[slot(idx)]; }
    double v(size_t idx) const { return best_v[slot(idx)]; }

    ObjectModel get_last_model() { return last_model; }
    const bf_run_info &get_run_info() const { return last_info; }
    ull get_slices_done() const { return slices_done; }
    ull get_slices_skipped() const { return slices_skipped; }
    ull get_iterations_total() const { return iterations_total; }
};

}  // namespace bf

#endif  // BF_HOST_STREAM_FLOW_H
