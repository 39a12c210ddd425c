// stream_flow.h -- the slice former for high event rates: the reference's event ring and triggers
// (CircularArray, datastructures.h:6-115; DVS_flow::add_event / recompute, dvs_flow.h:164-347) on a
// structure-of-arrays ring in PINNED memory, handed to the device without a copy on the host, with the slices solved
// by a slice farm (better_flow/slice_farm.h).
//
// Why a second front end next to DVS_flow.  DVS_flow keeps the reference's public `ev_buffer` of 152-byte Event
// records: one add_event() call and one 152-byte store per event, and an AoS -> SoA repack per slice
// (accel_lib.h:91-99; AccelLib::init_gpu here) -- 23 Mevents/s end to end, against a device path that solves
// warm-started 1M-event slices at several Gevents/s.  Here
//   * the ring is three parallel arrays (u64 timestamp, u16 row, u16 column: 12 bytes per event, the column layout
//     of the binary event file) plus a lazily used Event::noise ring; a slice is one or two contiguous pieces of it,
//     copied by DMA (bf_upload_ring16_async), and Event::set_local_time runs on the device;
//   * events arrive in BULK: add_events(rows, cols, timestamps, n) copies a block into the ring, and
//     reserve() / commit() let a producer (a file reader, a socket) write into the ring itself -- then the only host
//     copy is the producer's own.  Trigger points inside a block are found by a binary search on the timestamps and
//     a subtraction on the counts, not by visiting every event;
//   * slices are solved by SliceFarm workers: with set_pipelined(true) the caller goes on filling the ring while
//     slice k is being solved and slice k + 1 uploaded (an STM chain stays sequential on one worker); with
//     --stm-disable style independent slices, several workers / GPUs take them in parallel.  Results are applied in
//     slice order either way;
//   * per-event flow comes back only if asked for (set_want_flow / set_accumulate), straight into a pinned (u, v) ring.
//
// Semantics: those of DVS_flow, element for element --
//   * ring of at most MAX_SZ events spanning at most SPAN ns (push_back :31-44, fix_span :46-59);
//   * trigger: events since the last slice >= on_ev_change OR time since it >= on_time_change (:164-181);
//   * slice = the ring content, EXCEPT that a full ring leaves its oldest element out (end() :71-76);
//   * slice start time = oldest timestamp if the ring is full, else max(now - SPAN, 0) (:186-191);
//   * warm start from the previous slice's model unless stm_disable (:218-219);
//   * a slice stopped by the small-window guard flags its events as noise, and flagged events stay out of the time
//     images of later, overlapping slices (optimizer_rolling.h:49-55, accel_lib.h:152);
//   * set_accumulate + get_accumulated(): every event once, with the flow of the first slice it was solved in --
//     DVS_flow::get_accumulated (dvs_flow.h:351-389) with its exact marking rule.
// Order inside a slice is oldest -> newest here (the reference iterates newest -> oldest); the device accumulates
// integers, so the order does not change any result.  The bulk paths assume what the reference assumes of its input
// ("timestamps only grow"): non-decreasing timestamps; set_assume_sorted(false) makes add_events() visit every event.
// tests/cpp/test_stream.cpp holds this class to DVS_flow slice by slice, on the oracle shim and on the GPU.
#ifndef BF_HOST_STREAM_FLOW_H
#define BF_HOST_STREAM_FLOW_H

#include <better_flow/accel_lib.h>
#include <better_flow/common.h>
#include <better_flow/object_model.h>
#include <better_flow/slice_farm.h>

#include <memory>

namespace bf {

// One solved slice, in slice order (the slice callback, the slice log).
struct SliceRecord {
    uint64_t index = 0;           // 0, 1, ... in trigger order
    uint64_t first_event = 0;     // arrival number of the slice's oldest event
    uint64_t events = 0;          // events in the slice
    uint64_t ring_size = 0;       // CircularArray::size() at the trigger (events + 1 for a full ring)
    uint64_t new_events = 0;      // events since the previous slice
    ull start_time = 0, trigger_time = 0;   // slice origin / newest timestamp, ns
    ull oldest_time = 0;          // timestamp of the slice's oldest event (0 for an empty slice)
    uint64_t events_seen = 0;     // events that had arrived when the slice was triggered
    sll time_diff = 0;            // time since the previous slice
    int rc = 0;
    bool window_guard = false;
    bf_run_info info;
    ObjectModel model;
    double ms = 0;
    int device = 0;
};

// What get_accumulated() returns: the -o table, one row per event.
struct FlowTable {
    std::vector<uint64_t> timestamp;   // ns
    std::vector<uint16_t> row, col;
    std::vector<double> u, v;          // best_u, best_v
    size_t size() const { return timestamp.size(); }
};

class StreamEngine {
public:
    struct Span {                 // a writable piece of the ring (reserve)
        uint64_t *timestamp;
        uint16_t *row, *col;
        size_t n;
    };
    typedef std::function<void(const SliceRecord &)> SliceFn;

    StreamEngine(size_t max_sz_, sll span_, ull on_ev_change_, ull on_time_change_, ull start_time = 0)
        : max_sz(max_sz_), span(span_), on_ev_change(on_ev_change_), on_time_change(on_time_change_), cap(0), ts(nullptr),
          row_(nullptr), col_(nullptr), noise(nullptr), uv(nullptr), head(0), ring_size(0), stale(false), time_diff(0),
          event_diff(0), last_slice_time(start_time), current_slice_time(start_time), time_base(0), max_iter(-1), scale(3),
          stm_disable(false), want_flow(true), accumulate(false), pipelined(false), assume_sorted(true),
          contexts_per_device(1), lookahead(0), slices_submitted(0), last_trigger_plus1(0), noise_live(0),
          protected_from(UINT64_MAX), slices_done(0), slices_skipped(0), iterations_total(0), flow_through_plus1(0),
          failed(false), fail_code(0) {
        if (max_sz < 1) throw AccelError(BF_ERR_ARG, "StreamEngine: ring capacity must be >= 1");
        std::memset(&last_info, 0, sizeof(last_info));
        devices.push_back(DeviceContext::device());
    }
    virtual ~StreamEngine() {
        if (farm) {
            try { farm->drain(); } catch (...) {}
            bf_ctx *c = farm->context(0);
            if (ts) (void)bf_host_free(c, ts);
            if (row_) (void)bf_host_free(c, row_);
            if (col_) (void)bf_host_free(c, col_);
            if (noise) (void)bf_host_free(c, noise);
            if (uv) (void)bf_host_free(c, uv);
        }
        farm.reset();
    }
    StreamEngine(const StreamEngine &) = delete;
    StreamEngine &operator=(const StreamEngine &) = delete;

    // ---- settings (before the first event) ----
    void set_max_iter(int v = -1) { max_iter = v; }
    void set_scale(int v = 3) { scale = v; }
    void set_stm_disable(bool v = true) { stm_disable = v; }
    void set_want_flow(bool v = true) { want_flow = v; }     // fetch per-event (u, v) after every slice
    void set_accumulate(bool v = true) { accumulate = v; }   // keep every slice's events + flow for get_accumulated()
    void set_pipelined(bool v = true) { pipelined = v; }     // add_event(s) return at the trigger; drain() waits
    void set_assume_sorted(bool v = true) { assume_sorted = v; }
    void set_time_base(ull t) { time_base = t; }             // ring timestamps are absolute; logical time = timestamp - base
    void set_lookahead(size_t n) { lookahead = n; }          // ring slots beyond MAX_SZ (default: 2 MAX_SZ, at least 65536)
    void set_devices(const std::vector<int> &d, int contexts = 1) { devices = d; contexts_per_device = contexts; }
    void on_slice(SliceFn fn) { slice_fn = std::move(fn); }

    // Create the workers, their device contexts and the pinned ring now (otherwise: at the first event).
    void warm_up() { ensure_ring(); }

    // ---- input ----
    // DVS_flow::add_event (dvs_flow.h:164-181); row / column as Event::fr_x / fr_y.  Returns whether this event closed a slice.
    bool add_event(uint32_t row, uint32_t col, ull timestamp) {
        Span s[2];
        (void)reserve(1, s);
        s[0].timestamp[0] = timestamp + time_base;
        s[0].row[0] = narrow(row); s[0].col[0] = narrow(col);
        return commit(1) > 0;
    }

    // n add_event calls in one: the block is copied into the ring piecewise; returns the number of slices it closed.
    size_t add_events(const uint32_t *rows, const uint32_t *cols, const ull *timestamps, size_t n) {
        size_t slices = 0, done = 0;
        while (done < n) {
            Span s[2];
            const size_t got = reserve(n - done, s);
            size_t k = done;
            for (int p = 0; p < 2; ++p)
                for (size_t i = 0; i < s[p].n; ++i, ++k) {
                    s[p].timestamp[i] = timestamps[k] + time_base;
                    s[p].row[i] = narrow(rows[k]); s[p].col[i] = narrow(cols[k]);
                }
            slices += commit(got);
            done += got;
        }
        return slices;
    }
    // The same from arrays already in the ring's layout (absolute timestamps = logical + time base).
    size_t add_events(const uint16_t *rows, const uint16_t *cols, const uint64_t *timestamps, size_t n) {
        size_t slices = 0, done = 0;
        while (done < n) {
            Span s[2];
            const size_t got = reserve(n - done, s);
            size_t k = done;
            for (int p = 0; p < 2; ++p) {
                std::memcpy(s[p].timestamp, timestamps + k, s[p].n * 8);
                std::memcpy(s[p].row, rows + k, s[p].n * 2);
                std::memcpy(s[p].col, cols + k, s[p].n * 2);
                k += s[p].n;
            }
            slices += commit(got);
            done += got;
        }
        return slices;
    }

    // Producer interface: up to `want` free ring slots as one or two pieces, to be filled with ABSOLUTE timestamps
    // (logical time + time base), rows, columns -- then commit(n) for the first n of them.  Blocks (pipelined mode) until
    // the slots are no longer needed by a slice in flight.  Returns the number of slots granted (>= 1).
    size_t reserve(size_t want, Span out[2]) {
        ensure_ring();
        rethrow_failure();
        if (want < 1) want = 1;
        const size_t room = cap - max_sz;
        if (want > room) want = room;
        uint64_t prot = protected_from.load(std::memory_order_acquire);
        if (prot != UINT64_MAX && head + 1 > prot + cap) {   // not even one slot: wait for the oldest slice in flight
            const auto t_wait = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { prot = protected_from.load(std::memory_order_acquire); return failed.load() || prot == UINT64_MAX || head + 1 <= prot + cap; });
            g.unlock();
            blocked_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count();
            rethrow_failure();
        }
        if (prot != UINT64_MAX && head + want > prot + cap) want = (size_t)(prot + cap - head);
        const size_t slot = (size_t)(head % cap);
        const size_t n0 = want < cap - slot ? want : cap - slot;
        out[0] = Span{ts + slot, row_ + slot, col_ + slot, n0};
        out[1] = Span{ts, row_, col_, want - n0};
        return want;
    }

    // The first n reserved slots now hold events: run the triggers over them.  Returns the number of slices closed.
    size_t commit(size_t n) {
        size_t slices = 0;
        uint64_t g = head;
        const uint64_t end = head + n;
        if (n == 0) return 0;
        if (noise_live.load(std::memory_order_relaxed) > 0) {   // new events are not noise (Event(x, y, t): noise(false)); before any
                                                                // flag was ever set the whole ring is still zero
            const size_t slot = (size_t)(g % cap), n0 = n < cap - slot ? n : cap - slot;
            std::memset(noise + slot, 0, n0);
            std::memset(noise, 0, n - n0);
        }
        if (accumulate) archive(g, end);
        while (g < end) {
            // first event of [g, end) at which a trigger fires: by count ...
            const uint64_t need = (event_diff + 1 >= (sll)on_ev_change) ? 0 : (uint64_t)((sll)on_ev_change - event_diff - 1);
            uint64_t k = g + need;
            // ... or by time: (sll)(t - last_slice_time) >= on_time_change
            if (assume_sorted) {
                uint64_t lo = g, hi = (k < end ? k : end);   // a time trigger only matters before the count trigger
                if (lo < hi && time_due(logical(hi - 1))) {
                    while (lo < hi) {
                        const uint64_t mid = lo + (hi - lo) / 2;
                        if (time_due(logical(mid))) hi = mid; else lo = mid + 1;
                    }
                    k = lo;
                }
            } else {
                for (uint64_t q = g; q < end && q < k; ++q)
                    if (time_due(logical(q))) { k = q; break; }
            }
            if (k >= end) {   // no trigger in the rest of the block
                advance(end - g, end);
                break;
            }
            advance(k - g + 1, k + 1);
            recompute();
            ++slices;
            g = k + 1;
        }
        return slices;
    }

    // DVS_flow::recompute (dvs_flow.h:185-347) without the rendering branches: solve the ring's content now.
    void recompute() {
        ensure_ring();
        rethrow_failure();
        trim();
        Pending p;
        p.index = slices_submitted++;
        p.ring_size = ring_size;
        const uint64_t oldest = head - ring_size;
        p.full = ring_size == max_sz;
        if (p.full) {            // full ring: iteration stops one short (:71-76), start = oldest timestamp
            p.start_time = logical(oldest);
            p.first = oldest + 1;
            p.n = ring_size - 1;
            p.zero_excluded = want_flow_any() && oldest + 1 > last_trigger_plus1;   // that event has never been in a slice
        } else {
            p.start_time = (current_slice_time > (ull)span) ? current_slice_time - (ull)span : 0;
            p.first = oldest;
            p.n = ring_size;
            p.zero_excluded = false;
        }
        p.trigger_time = current_slice_time;
        p.oldest_time = p.n > 0 ? logical(p.first) : 0;
        p.new_events = (uint64_t)event_diff;
        p.time_diff = time_diff;
        p.trigger_plus1 = head;
        // (the ring slots a slice in flight still needs: its events and, for a full ring, the oldest element it leaves out --
        // deliver() zeroes that element's flow and must still find it there)
        p.protect = oldest;
        SliceFarm::Task t;
        t.ring_row = row_; t.ring_col = col_; t.ring_ts = ts; t.ring_noise = noise; t.noise_live = &noise_live;
        t.first_global = p.first;
        t.cap = (int64_t)cap; t.first = (int64_t)(p.first % cap); t.n = (int64_t)p.n;
        t.t0 = p.start_time + time_base;
        t.scale = scale; t.res_x = RES_X; t.res_y = RES_Y; t.max_iter = max_iter;
        t.warm = stm_disable ? SliceFarm::Warm::Cold : SliceFarm::Warm::FromPrevious;
        if ((accumulate || (want_flow && farm->workers() > 1)) && p.n > 0) {
            // A private block per slice, copied into the ring by deliver() -- which runs in SLICE order.  Needed for
            // get_accumulated(), and whenever several workers solve overlapping slices at once: written straight into
            // the shared ring, an event's flow would be that of whichever slice FINISHED last (and two workers would
            // write overlapping host memory concurrently), not that of the latest slice as in DVS_flow.
            // (uninitialised: bf_compute_uv_ring writes every pair.  A value-initialised vector cost the producer 16 MB of
            // page faults and zeroes per 1M-event slice -- 4 ms, more than the slice's whole solve)
            p.block = std::shared_ptr<double>(new double[2 * (size_t)p.n], std::default_delete<double[]>());
            t.uv_ring = p.block.get(); t.uv_cap = (int64_t)p.n; t.uv_first = 0;
        } else if (want_flow && p.n > 0) {   // one worker: slices complete in order, straight into the pinned ring
            t.uv_ring = uv; t.uv_cap = (int64_t)cap; t.uv_first = t.first;
        }
        t.user = p.index;
        if (farm->workers() > 1 && p.n > 0 && window_guard_on_host(p)) {   // see slice_farm.h: uploads run ahead
            // (earlier slices overlap this one in the ring and their workers may not have read it yet: in the reference's
            // order they do not see this slice's flags -- wait for them first.  The guard stops a slice once in a long while.)
            farm->drain();
            flag_noise(p);
        }
        {
            std::lock_guard<std::mutex> g(mu);
            pending.push_back(p);
            protected_from.store(pending.front().protect, std::memory_order_release);
        }
        last_trigger_plus1 = head;
        event_diff = 0;
        last_slice_time = current_slice_time;
        farm->submit(t);
        if (!pipelined) drain();
    }

    // Wait for every slice triggered so far (pipelined mode); rethrows a failure of a slice as bf::AccelError.
    void drain() {
        if (farm) farm->drain();
        rethrow_failure();
    }

    // ---- the ring, DVS_flow style (idx 0 = newest; CircularArray::operator[], :61-64).  Call drain() first in pipelined mode. ----
    size_t size() { trim(); return ring_size; }
    uint32_t row(size_t idx) const { return row_[slot_of(idx)]; }
    uint32_t col(size_t idx) const { return col_[slot_of(idx)]; }
    ull timestamp(size_t idx) const { return ts[slot_of(idx)] - time_base; }
    double u(size_t idx) const { return flow_of(idx, 0); }
    double v(size_t idx) const { return flow_of(idx, 1); }

    ObjectModel get_last_model() { std::lock_guard<std::mutex> g(mu); return last_model; }
    bf_run_info get_run_info() { std::lock_guard<std::mutex> g(mu); return last_info; }
    ull get_slices_done() { std::lock_guard<std::mutex> g(mu); return slices_done; }
    ull get_slices_skipped() { std::lock_guard<std::mutex> g(mu); return slices_skipped; }
    ull get_iterations_total() { std::lock_guard<std::mutex> g(mu); return iterations_total; }
    sll get_buf_size() { return (sll)size(); }
    sll get_time_diff() const { return time_diff; }
    ull events_seen() const { return head; }
    double seconds_blocked() const { return blocked_s; }   // the producer waited this long in reserve() for slices in flight

    // DVS_flow::get_accumulated (dvs_flow.h:351-389): the events of all slices, each once, with the flow of the first
    // slice that solved it.  The marking rule is the reference's: walking the slices in order and, inside a slice, the
    // events oldest -> newest, an unmarked event e marks, in every LATER slice, the events of e's pixel that are not
    // after e in time and less than 0.1 ms before it (Event::operator==, event.h:39-45) -- its own later copies, and on
    // rare occasions another event; marked events are left out.  An event whose slice-local time is exactly -1 counts
    // as marked from the start (the reference uses t == -1 as the mark).
    FlowTable get_accumulated();

protected:
    struct Pending {
        uint64_t index = 0, first = 0, n = 0, ring_size = 0, new_events = 0, trigger_plus1 = 0, protect = 0;
        ull start_time = 0, trigger_time = 0, oldest_time = 0;
        sll time_diff = 0;
        bool full = false, zero_excluded = false;
        std::shared_ptr<double> block;   // (u, v) pairs of the slice's events
    };
    struct Kept {                 // accumulate: one solved slice
        uint64_t first, n;
        ull start_time;
        std::shared_ptr<double> block;   // (u, v) pairs of the slice's events
        bool lead;                // the ring was full and its oldest element (event first - 1, which the slice leaves out,
                                  // datastructures.h:71-76) had never been in a slice: the reference's copy of the ring
                                  // (dvs_flow.h:340-345) holds it too, with the zero flow of a fresh Event
    };

    // configuration
    size_t max_sz;
    sll span;
    ull on_ev_change, on_time_change;
    // the ring (pinned): event number g lives in slot g % cap, cap = MAX_SZ + lookahead
    size_t cap;
    uint64_t *ts;
    uint16_t *row_, *col_;
    uint8_t *noise;
    double *uv;
    uint64_t head;                // events committed so far
    double blocked_s = 0;
    size_t ring_size;             // CircularArray::current_size
    bool stale;                   // !span_checked
    sll time_diff, event_diff;
    ull last_slice_time, current_slice_time, time_base;
    int max_iter, scale;
    bool stm_disable, want_flow, accumulate, pipelined, assume_sorted;
    std::vector<int> devices;
    int contexts_per_device;
    size_t lookahead;
    SliceFn slice_fn;
    std::unique_ptr<SliceFarm> farm;
    uint64_t slices_submitted, last_trigger_plus1;
    std::atomic<uint64_t> noise_live;        // 1 + arrival number of the newest event flagged as noise (0: none)
    std::atomic<uint64_t> protected_from;    // oldest event a slice in flight still needs (UINT64_MAX: none)
    // results (worker thread -> caller), under `mu`
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Pending> pending;
    ObjectModel last_model;
    bf_run_info last_info;
    ull slices_done, slices_skipped, iterations_total;
    uint64_t flow_through_plus1;             // events below this arrival number have been in a delivered slice
    std::atomic<bool> failed;   // (read by the producer without the lock)
    int fail_code;
    std::string fail_text;
    // accumulate
    std::vector<uint64_t> hist_ts;           // logical timestamps of every event seen
    std::vector<uint16_t> hist_row, hist_col;
    std::vector<Kept> kept;

    static uint16_t narrow(uint32_t v) {
        if (v > 65535u) throw AccelError(BF_ERR_ARG, "StreamEngine: event address " + std::to_string(v) + " does not fit 16 bits");
        return (uint16_t)v;
    }
    bool want_flow_any() const { return want_flow || accumulate; }
    ull logical(uint64_t g) const { return ts[g % cap] - time_base; }
    bool time_due(ull t) const { return !((sll)(t - last_slice_time) < (sll)on_time_change); }
    size_t slot_of(size_t idx) const { return (size_t)((head - 1 - idx) % cap); }
    double flow_of(size_t idx, int which) const {
        const uint64_t g = head - 1 - idx;
        if (!uv || g + 1 > flow_through_plus1) return 0.0;   // newer than the last solved slice: Event(): best_u = best_v = 0
        return uv[2 * (g % cap) + which];
    }

    void rethrow_failure() {
        if (!failed) return;
        std::lock_guard<std::mutex> g(mu);
        throw AccelError(fail_code, fail_text);
    }

    // m more events are in the ring; `upto` = events committed after them (CircularArray::push_back x m + the bookkeeping
    // of DVS_flow::add_event for the last of them)
    void advance(uint64_t m, uint64_t upto) {
        ring_size = (ring_size + m < max_sz) ? (size_t)(ring_size + m) : max_sz;
        stale = true;
        head = upto;
        event_diff += (sll)m;
        current_slice_time = logical(upto - 1);
        time_diff = (sll)(current_slice_time - last_slice_time);
    }

    void trim() {   // CircularArray::fix_span, datastructures.h:46-59
        if (!stale) return;
        stale = false;
        if (ring_size == 0) return;
        const ull newest = logical(head - 1);
        uint64_t oldest = head - ring_size;
        if (assume_sorted) {   // the events too old form a prefix: find its end instead of walking it
            uint64_t lo = oldest, hi = head - 1;   // (the newest event is never too old)
            while (lo < hi) {
                const uint64_t mid = lo + (hi - lo) / 2;
                if ((sll)(newest - logical(mid)) > span) lo = mid + 1; else hi = mid;
            }
            ring_size -= (size_t)(lo - oldest);
            return;
        }
        while ((sll)(newest - logical(oldest)) > span) { ++oldest; --ring_size; }
    }

    void ensure_ring() {
        if (farm) return;
        const bool chained = !stm_disable;
        if (chained && devices.size() * (size_t)contexts_per_device != 1)
            throw AccelError(BF_ERR_ARG, "StreamEngine: several devices / contexts need independent slices (set_stm_disable): a warm-start "
                                         "chain is sequential");
        farm.reset(new SliceFarm(devices, contexts_per_device, (long long)max_sz, scale * RES_X + scale, scale * RES_Y + scale,
                                 [this](const SliceFarm::Result &r) { deliver(r); }, chained));
        size_t extra = lookahead ? lookahead : (2 * max_sz > 65536 ? 2 * max_sz : 65536);   // the producer may run two slices ahead
        cap = max_sz + extra;
        for (size_t w = 0; w < farm->workers(); ++w) (void)bf_set_option(farm->context(w), "stream_prealloc", 1);   // staging slots, copy stream: now, not at the first slice
        bf_ctx *c = farm->context(0);
        void *p = nullptr;   // (pinned host memory is not tied to the ctx object)
        auto alloc = [&](size_t bytes) {
            const int rc = bf_host_alloc(c, (int64_t)bytes, &p);
            if (rc < 0) throw AccelError(rc, std::string("StreamEngine: pinned allocation failed: ") + bf_last_error(c));
            return p;
        };
        ts = (uint64_t *)alloc(cap * 8);
        row_ = (uint16_t *)alloc(cap * 2);
        col_ = (uint16_t *)alloc(cap * 2);
        noise = (uint8_t *)alloc(cap);
        std::memset(noise, 0, cap);
        if (want_flow_any()) { uv = (double *)alloc(cap * 16); std::memset(uv, 0, cap * 16); }
    }

    void archive(uint64_t g, uint64_t end) {   // (two contiguous pieces of the ring, appended in bulk)
        const size_t n = (size_t)(end - g), at = hist_ts.size();
        hist_ts.resize(at + n); hist_row.resize(at + n); hist_col.resize(at + n);
        const size_t slot = (size_t)(g % cap), n0 = n < cap - slot ? n : cap - slot;
        const ull base = time_base;
        for (size_t i = 0; i < n0; ++i) hist_ts[at + i] = ts[slot + i] - base;
        for (size_t i = n0; i < n; ++i) hist_ts[at + i] = ts[i - n0] - base;
        std::memcpy(hist_row.data() + at, row_ + slot, n0 * 2); std::memcpy(hist_row.data() + at + n0, row_, (n - n0) * 2);
        std::memcpy(hist_col.data() + at, col_ + slot, n0 * 2); std::memcpy(hist_col.data() + at + n0, col_, (n - n0) * 2);
    }

    // optimizer_rolling.h:49-55 evaluated on the host: the bounding box of the slice (set_cloud, :248-283) against RES / 15
    bool window_guard_on_host(const Pending &p) const {
        int x_min = RES_X, y_min = RES_Y, x_max = 0, y_max = 0;
        for (uint64_t g = p.first; g < p.first + p.n; ++g) {
            const size_t s = (size_t)(g % cap);
            const int x = row_[s], y = col_[s];
            x_min = x < x_min ? x : x_min; x_max = x > x_max ? x : x_max;
            y_min = y < y_min ? y : y_min; y_max = y > y_max ? y : y_max;
        }
        const int img_x = scale * (x_max - x_min) + scale, img_y = scale * (y_max - y_min) + scale;
        return (img_x < scale * RES_X / 15) && (img_y < scale * RES_Y / 15);
    }

    void flag_noise(const Pending &p) {   // `for (auto &e : *events) e.noise = true`, optimizer_rolling.h:52-53
        if (p.n == 0) return;
        const size_t slot = (size_t)(p.first % cap), n0 = p.n < cap - slot ? (size_t)p.n : cap - slot;
        std::memset(noise + slot, 1, n0);
        std::memset(noise, 1, (size_t)p.n - n0);
        uint64_t cur = noise_live.load(std::memory_order_relaxed);
        while (cur < p.first + p.n && !noise_live.compare_exchange_weak(cur, p.first + p.n, std::memory_order_release)) {}
    }

    // a slice's result, in slice order, on a farm worker's thread
    void deliver(const SliceFarm::Result &r) {
        Pending p;
        {
            std::lock_guard<std::mutex> g(mu);
            p = pending.front();
        }
        if (r.rc < 0) {
            std::lock_guard<std::mutex> g(mu);
            if (!failed) { fail_code = r.rc; fail_text = "StreamEngine: slice " + std::to_string(p.index) + ": " + r.error; failed = true; }
        } else {
            // (several workers: recompute() has already flagged the slice from its own evaluation of the guard -- other
            // workers may be reading the noise ring right now)
            if (r.window_guard && farm->workers() == 1) flag_noise(p);
            if (p.block && uv) {   // accumulate: the slice's own copy of the flow -> the ring's
                const size_t slot = (size_t)(p.first % cap), n0 = p.n < cap - slot ? (size_t)p.n : cap - slot;
                std::memcpy(uv + 2 * slot, p.block.get(), n0 * 16);
                std::memcpy(uv, p.block.get() + 2 * n0, ((size_t)p.n - n0) * 16);
            }
            if (p.zero_excluded && uv) { const size_t s = (size_t)((p.first - 1) % cap); uv[2 * s] = uv[2 * s + 1] = 0.0; }
        }
        SliceRecord rec;
        rec.index = p.index; rec.first_event = p.first; rec.events = p.n; rec.ring_size = p.ring_size; rec.new_events = p.new_events;
        rec.start_time = p.start_time; rec.trigger_time = p.trigger_time; rec.time_diff = p.time_diff;
        rec.oldest_time = p.oldest_time; rec.events_seen = p.trigger_plus1;
        rec.rc = r.rc; rec.window_guard = r.window_guard; rec.info = r.info; rec.model = ObjectModel(r.model); rec.ms = r.ms; rec.device = r.device;
        {
            std::lock_guard<std::mutex> g(mu);
            if (r.rc >= 0) {
                last_model = rec.model;
                last_info = r.info;
                ++slices_done;
                if (r.rc != 0) ++slices_skipped;
                iterations_total += (ull)r.info.iterations;
                if (accumulate) kept.push_back(Kept{p.first, p.n, p.start_time, p.block, p.zero_excluded});
                if (p.trigger_plus1 > flow_through_plus1) flow_through_plus1 = p.trigger_plus1;
            }
        }
        if (slice_fn && r.rc >= 0) slice_fn(rec);
        {
            std::lock_guard<std::mutex> g(mu);
            pending.pop_front();
            protected_from.store(pending.empty() ? UINT64_MAX : pending.front().protect, std::memory_order_release);
        }
        cv.notify_all();
    }
};

inline FlowTable StreamEngine::get_accumulated() {
    drain();
    FlowTable out;
    const size_t K = kept.size();
    const uint64_t N = hist_ts.size();
    if (N >= 0xffffffffull) throw AccelError(BF_ERR_CAPACITY, "StreamEngine::get_accumulated: more than 2^32 - 2 events");
    // chains of events at the same pixel: previous / next event of g's pixel in arrival order
    const uint32_t NONE = 0xffffffffu;
    std::vector<uint32_t> prev(N, NONE), next(N, NONE);
    {
        uint32_t max_col = 0;
        for (uint64_t g = 0; g < N; ++g) max_col = hist_col[g] > max_col ? hist_col[g] : max_col;
        uint32_t max_row = 0;
        for (uint64_t g = 0; g < N; ++g) max_row = hist_row[g] > max_row ? hist_row[g] : max_row;
        std::vector<uint32_t> last((size_t)(max_row + 1) * (max_col + 1), NONE);
        for (uint64_t g = 0; g < N; ++g) {
            uint32_t &l = last[(size_t)hist_row[g] * (max_col + 1) + hist_col[g]];
            prev[g] = l;
            if (l != NONE) next[l] = (uint32_t)g;
            l = (uint32_t)g;
        }
    }
    std::vector<std::vector<uint8_t>> mark(K);
    size_t copies = 0;
    for (size_t i = 0; i < K; ++i) {
        mark[i].assign((size_t)kept[i].n, 0);
        copies += (size_t)kept[i].n;
        for (uint64_t p = 0; p < kept[i].n && hist_ts[kept[i].first + p] < kept[i].start_time; ++p)   // slice-local t == -1
            if (hist_ts[kept[i].first + p] + 1 == kept[i].start_time) mark[i][(size_t)p] = 1;
    }
    out.timestamp.reserve(N); out.row.reserve(N); out.col.reserve(N); out.u.reserve(N); out.v.reserve(N);
    (void)copies;
    auto mark_later = [&](size_t i, uint64_t c) {   // event c in the slices after i that hold it
        for (size_t j = i + 1; j < K && kept[j].first <= c; ++j)
            if (c < kept[j].first + kept[j].n) mark[j][(size_t)(c - kept[j].first)] = 1;
    };
    auto emit = [&](size_t i, uint64_t g, double eu, double ev) {   // an unmarked event of slice i: mark its later copies, write it
        const uint64_t t = hist_ts[g];
        if (i + 1 < K) {
            mark_later(i, g);
            for (uint32_t c = prev[g]; c != NONE && t - hist_ts[c] < 100000ull; c = prev[c]) mark_later(i, c);   // dt < 0.1 ms
            for (uint32_t c = next[g]; c != NONE && hist_ts[c] == t; c = next[c]) mark_later(i, c);             // same instant, arrived later
        }
        out.timestamp.push_back(t); out.row.push_back(hist_row[g]); out.col.push_back(hist_col[g]);
        out.u.push_back(eu); out.v.push_back(ev);
    };
    for (size_t i = 0; i < K; ++i) {
        const Kept &s = kept[i];
        // (its t is the raw timestamp -- it never saw set_local_time --, so the t == -1 mark does not apply to it)
        if (s.lead) emit(i, s.first - 1, 0.0, 0.0);
        for (uint64_t p = 0; p < s.n; ++p) {
            if (mark[i][(size_t)p]) continue;
            emit(i, s.first + p, s.block.get()[2 * (size_t)p], s.block.get()[2 * (size_t)p + 1]);
        }
    }
    return out;
}

// The reference's interface: ring size and span as template parameters, like DVS_flow<MAX_SZ, SPAN>.
template <size_t MAX_SZ, sll SPAN> class StreamFlow : public StreamEngine {
public:
    StreamFlow(ull on_ev_change_, ull on_time_change_, ull start_time = 0)
        : StreamEngine(MAX_SZ, SPAN, on_ev_change_, on_time_change_, start_time) {}
};

}  // namespace bf

#endif  // BF_HOST_STREAM_FLOW_H
