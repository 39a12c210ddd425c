// slice_farm.h -- the slice farm: a queue of (events, model) tasks solved by worker threads, one bf_ctx each,
// spread over the GPUs of the node.
//
// The reference already models its work this way -- DVS_flow::recompute builds a list of (event cloud, starting
// model) tasks and runs one OptimizerRolling per task (dvs_flow.h:200-231) -- but executes the list serially on the
// calling thread.  A slice is a complete optimisation problem, so here the tasks go to
// `contexts_per_device` workers per device, each with its own context, HIP stream and copy stream; no data is ever
// exchanged between workers (no collective: BASELINE.json north_star).  Results are delivered in SUBMISSION order
// through a callback, whatever order they finish in.
//
// Every worker keeps up to two uploads in flight ahead of the slice it is solving (the two staging slots of
// bf_upload_ring16_async / bf_upload_events_async), so the host-to-device copy of slice k + 1 overlaps the solve of
// slice k on the same worker.  With a single worker (a stream's chain) submit() itself starts the copy when a staging
// slot is free -- the worker is inside bf_run then and would not look at the queue before it returns; the upload /
// commit entry points touch only the context's staging state, and a mutex keeps them apart from each other.
//
// Warm starts.  Task::warm = FromPrevious is the reference's short-term memory (dvs_flow.h:218-224): start from the
// model the PREVIOUS task ended with.  That is a sequential chain, so it needs a farm with exactly one worker (the
// constructor argument `chained`); FromModel starts from Task::start, Cold from the zero model (--stm-disable).
#ifndef BF_HOST_SLICE_FARM_H
#define BF_HOST_SLICE_FARM_H

#include <better_flow/accel_lib.h>
#include <better_flow/common.h>
#include <better_flow/object_model.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

namespace bf {

class SliceFarm {
public:
    enum class Warm { Cold, FromPrevious, FromModel };

    struct Task {
        // -- the events: a slice of a structure-of-arrays ring with 16-bit addresses (pinned memory makes the copy a
        // true DMA), event i of the slice = ring slot (first + i) % cap, oldest -> newest; times become
        // timestamp - t0 on the device (Event::set_local_time, event.h:61-63) ...
        const uint16_t *ring_row = nullptr, *ring_col = nullptr;
        const uint64_t *ring_ts = nullptr;
        const uint8_t *ring_noise = nullptr;             // Event::noise ring, or null
        const std::atomic<uint64_t> *noise_live = nullptr;   // if set: the noise ring matters only while *noise_live > first_global
        uint64_t first_global = 0;
        int64_t cap = 0, first = 0, n = 0;
        uint64_t t0 = 0;
        // ... or three linear int32 arrays with slice-local times (the layout of bf_upload_events)
        const int32_t *fr_x = nullptr, *fr_y = nullptr, *t_ns = nullptr;
        // -- the optimisation
        int scale = 3, res_x = 180, res_y = 240, max_iter = -1;
        Warm warm = Warm::Cold;
        bf_model start;                                  // Warm::FromModel
        // -- per-event flow: interleaved (u, v) pairs into a ring (event i -> uv_ring[2 * ((uv_first + i) % uv_cap)]), or none
        double *uv_ring = nullptr;
        int64_t uv_cap = 0, uv_first = 0;
        uint64_t user = 0;                               // passed through to the result
    };

    struct Result {
        uint64_t id = 0, user = 0;
        int rc = 0;                                      // 0 optimised, 1 skipped by a guard, < 0 failed (error text below)
        bool window_guard = false;                       // skipped because the window is too small (optimizer_rolling.h:49-55)
        bf_run_info info;
        bf_model model;
        bf_window window;
        int worker = 0, device = 0;
        double ms = 0;                                   // upload issue -> results on the host
        std::string error;
    };

    typedef std::function<void(const Result &)> ResultFn;

    // devices: one entry per GPU to use (an ordinal may repeat: several workers on one GPU);
    // contexts_per_device: workers per entry; capacity of every context: max_events events, images of max_rows x max_cols.
    // chained: Warm::FromPrevious will be used -- demands exactly one worker.
    SliceFarm(const std::vector<int> &devices, int contexts_per_device, long long max_events, int max_rows, int max_cols,
              ResultFn on_result, bool chained = false)
        : on_result_(std::move(on_result)), chained_(chained) {
        if (devices.empty() || contexts_per_device < 1) throw AccelError(BF_ERR_ARG, "SliceFarm: no workers");
        const size_t nw = devices.size() * (size_t)contexts_per_device;
        if (chained && nw != 1) throw AccelError(BF_ERR_ARG, "SliceFarm: a warm-start chain is sequential and needs exactly one worker");
        for (size_t w = 0; w < nw; ++w) workers_.emplace_back();
        for (size_t w = 0; w < nw; ++w) {
            Worker &wk = workers_[w];
            wk.index = (int)w;
            wk.device = devices[w / (size_t)contexts_per_device];
            int rc = bf_create(wk.device, max_events, max_rows, max_cols, nullptr, &wk.ctx);
            if (rc != BF_OK) {
                for (Worker &o : workers_) if (o.ctx) bf_destroy(o.ctx);
                throw AccelError(rc, "SliceFarm: bf_create on device " + std::to_string(wk.device) + " failed (" + std::to_string(rc) +
                                         "): the motion-compensation path needs a HIP device (there is no CPU fallback)");
            }
            if (nw > 1) (void)bf_set_option(wk.ctx, "co_schedule", 1);   // contexts sharing a GPU: see include/bf_accel.h
        }
        for (Worker &wk : workers_) wk.thread = std::thread([this, &wk] { work(wk); });
    }

    ~SliceFarm() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stopping_ = true;
        }
        cv_work_.notify_all();
        for (Worker &wk : workers_) if (wk.thread.joinable()) wk.thread.join();
        for (Worker &wk : workers_) if (wk.ctx) bf_destroy(wk.ctx);
    }
    SliceFarm(const SliceFarm &) = delete;
    SliceFarm &operator=(const SliceFarm &) = delete;

    size_t workers() const { return workers_.size(); }
    bf_ctx *context(size_t worker) const { return workers_[worker].ctx; }   // e.g. for bf_host_alloc / bf_set_option before the first task

    // Queue a task; returns its id (0, 1, 2, ... in submission order).  The arrays it points to must stay valid and
    // unchanged until its result has been delivered.
    uint64_t submit(const Task &t) {
        Job job;
        job.task = t;
        job.t_issue = std::chrono::steady_clock::now();
        job.uploaded = t.n <= 0;   // nothing to copy
        if (workers_.size() == 1 && t.n > 0) {
            // a single worker: start the copy from here if a staging slot is free and no older task still waits for one
            Worker &wk = workers_[0];
            std::lock_guard<std::mutex> up(wk.up_mu);
            bool can;
            {
                std::lock_guard<std::mutex> g(mu_);
                job.id = next_id_++;
                can = waiting_for_slot_ == 0 && wk.slots < 2;
            }
            if (can) {
                job.upload_rc = issue_upload(wk.ctx, t);
                job.uploaded = true;
                if (job.upload_rc >= 0) ++wk.slots;
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                if (!job.uploaded) ++waiting_for_slot_;
                queue_.push_back(job);
            }
        } else {
            std::lock_guard<std::mutex> g(mu_);
            job.id = next_id_++;
            queue_.push_back(job);
        }
        cv_work_.notify_one();
        return job.id;
    }

    // Block until every task submitted so far has been delivered.
    void drain() {
        std::unique_lock<std::mutex> g(mu_);
        cv_done_.wait(g, [this] { return delivered_ == next_id_; });
    }

    uint64_t submitted() const { std::lock_guard<std::mutex> g(mu_); return next_id_; }
    uint64_t delivered() const { std::lock_guard<std::mutex> g(mu_); return delivered_; }

private:
    struct Job {
        uint64_t id = 0;
        Task task;
        bool uploaded = false;     // its copy has been issued (or it has no events)
        int upload_rc = 0;         // < 0: issuing the copy failed; reported when the task's turn comes
        std::chrono::steady_clock::time_point t_issue;
    };
    struct Worker {
        int index = 0, device = 0;
        bf_ctx *ctx = nullptr;
        std::thread thread;
        bf_model last_model;       // Warm::FromPrevious
        bool have_last = false;
        std::mutex up_mu;          // serialises bf_upload_*_async / bf_commit_upload on this context (worker and submit())
        int slots = 0;             // staging slots in use: copies issued and not yet committed (under up_mu)
    };
    typedef Job Staged;

    ResultFn on_result_;
    bool chained_;
    std::deque<Worker> workers_;   // (a Worker holds a mutex: no moves)
    mutable std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::deque<Job> queue_;
    std::map<uint64_t, Result> finished_;   // done, waiting for an earlier id
    uint64_t next_id_ = 0, delivered_ = 0;
    int waiting_for_slot_ = 0;              // queued tasks whose copy has not been issued yet (single-worker farms)
    bool stopping_ = false;

    static int issue_upload(bf_ctx *ctx, const Task &t) {
        if (t.ring_ts) {
            const bool noise = t.ring_noise && (!t.noise_live || t.noise_live->load(std::memory_order_acquire) > t.first_global);
            return bf_upload_ring16_async(ctx, t.ring_row, t.ring_col, t.ring_ts, noise ? t.ring_noise : nullptr, t.cap, t.first, t.n, t.t0);
        }
        return bf_upload_events_async(ctx, t.fr_x, t.fr_y, t.t_ns, t.n);
    }

    // deliver in submission order; called by the worker that finished `r`
    void publish(Result &&r) {
        std::unique_lock<std::mutex> g(mu_);
        finished_.emplace(r.id, std::move(r));
        while (!finished_.empty() && finished_.begin()->first == delivered_) {
            Result out = std::move(finished_.begin()->second);
            finished_.erase(finished_.begin());
            g.unlock();
            if (on_result_) on_result_(out);     // (outside the lock: the callback may submit)
            g.lock();
            ++delivered_;
        }
        g.unlock();
        cv_done_.notify_all();
    }

    Result solve(Worker &wk, const Staged &s) {
        const Task &t = s.task;
        Result r;
        r.id = s.id; r.user = t.user; r.worker = wk.index; r.device = wk.device;
        std::memset(&r.info, 0, sizeof(r.info));
        std::memset(&r.model, 0, sizeof(r.model));
        std::memset(&r.window, 0, sizeof(r.window));
        auto fail = [&](int rc, const char *what) {
            r.rc = rc;
            r.error = std::string("SliceFarm: ") + what + " failed (" + std::to_string(rc) + "): " + bf_last_error(wk.ctx);
            return r;
        };
        int rc;
        static const bool phase_timing = std::getenv("BF_FARM_TIMING") != nullptr;   // debug: where a slice's time goes
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return 1e6 * std::chrono::duration<double>(b - a).count();
        };
        if (s.upload_rc < 0) return fail(s.upload_rc, "upload");
        if (t.n > 0) {
            const auto t0 = now();
            {
                std::lock_guard<std::mutex> up(wk.up_mu);
                rc = bf_commit_upload(wk.ctx);
                --wk.slots;
            }
            if (rc < 0) return fail(rc, "commit_upload");
            const auto t1 = now();
            if ((rc = bf_set_cloud(wk.ctx, t.scale, t.res_x, t.res_y, &r.window)) < 0) return fail(rc, "set_cloud");
            const auto t2 = now();
            if (t.warm == Warm::FromModel) rc = bf_set_model(wk.ctx, &t.start);
            else if (t.warm == Warm::FromPrevious && wk.have_last) rc = bf_set_model(wk.ctx, &wk.last_model);
            else if (t.warm == Warm::FromPrevious) { bf_model zero; std::memset(&zero, 0, sizeof(zero)); rc = bf_set_model(wk.ctx, &zero); }
            if (rc < 0) return fail(rc, "set_model");
            bf_run_opts o;
            bf_run_opts_default(&o);
            o.max_iter = t.max_iter; o.res_x = t.res_x; o.res_y = t.res_y; o.want_uv = t.uv_ring ? 1 : 0;
            rc = bf_run(wk.ctx, &o, &r.model, &r.info);
            if (rc < 0) return fail(rc, "run");
            r.rc = rc;
            const auto t3 = now();
            if (t.uv_ring && (rc = bf_compute_uv_ring(wk.ctx, t.uv_ring, t.uv_cap, t.uv_first)) < 0) return fail(rc, "compute_uv_ring");
            if (phase_timing)
                std::fprintf(stderr, "farm worker %d slice %llu: since upload issue %.0f us | commit %.0f set_cloud %.0f run %.0f (%d it, %d launches, %d polls, %d re-bins) uv %.0f us\n", wk.index,
                             (unsigned long long)s.id, us(s.t_issue, t0), us(t0, t1), us(t1, t2), us(t2, t3), (int)r.info.iterations, (int)r.info.launches,
                             (int)r.info.polls, (int)r.info.rebins, us(t3, now()));
        } else {
            // the reference runs its optimizer on the empty cloud: x_min = RES_X, x_max = 0 (optimizer_rolling.h:252-260)
            // make a negative window, the guard of :49-55 skips it, and get_model() is the model set_model() stored
            r.rc = BF_SKIPPED;
            r.info.rc = BF_SKIPPED;
            r.info.x_divider = r.info.y_divider = 1.0f;
            r.info.rot_divider = r.info.div_divider = 10000.0f;
            r.window.scale = t.scale;
            r.window.x_min = t.res_x; r.window.y_min = t.res_y;
            r.window.metric_wsizex = t.scale * (0 - t.res_x); r.window.metric_wsizey = t.scale * (0 - t.res_y);
            r.window.scale_img_x = r.window.metric_wsizex + t.scale; r.window.scale_img_y = r.window.metric_wsizey + t.scale;
            if (t.warm == Warm::FromModel) r.model = t.start;
            else if (t.warm == Warm::FromPrevious && wk.have_last) r.model = wk.last_model;
        }
        r.window_guard = r.rc == BF_SKIPPED && (r.window.scale_img_x < t.scale * t.res_x / 15) && (r.window.scale_img_y < t.scale * t.res_y / 15);
        wk.last_model = r.model;
        wk.have_last = true;
        r.ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - s.t_issue).count();
        return r;
    }

    void work(Worker &wk) {
        // next to its GPU: the CPUs (and preferred memory) of the device's host NUMA node, where the process may use them
        (void)bf_bind_thread_to_device_numa(wk.device, nullptr);
        std::deque<Staged> staged;        // tasks this worker has taken, oldest first; their copies are in flight
        for (;;) {
            {   // sleep only when there is nothing to solve
                std::unique_lock<std::mutex> g(mu_);
                if (staged.empty()) cv_work_.wait(g, [this] { return stopping_ || !queue_.empty(); });
                if (stopping_ && queue_.empty() && staged.empty()) return;
            }
            {   // take tasks while staging slots last: those whose copy submit() already started, then copies of our own
                std::lock_guard<std::mutex> up(wk.up_mu);
                for (;;) {
                    Job job;
                    {
                        std::lock_guard<std::mutex> g(mu_);
                        if (queue_.empty()) break;
                        const Job &f = queue_.front();
                        if (!f.uploaded && wk.slots >= 2) break;
                        job = f;
                        queue_.pop_front();
                        if (!job.uploaded) --waiting_for_slot_;
                    }
                    if (!job.uploaded) {
                        job.t_issue = std::chrono::steady_clock::now();
                        job.upload_rc = issue_upload(wk.ctx, job.task);
                        job.uploaded = true;
                        if (job.upload_rc >= 0) ++wk.slots;
                    }
                    staged.push_back(job);
                    if (workers_.size() > 1 && staged.size() >= 2) break;   // one ahead is enough; leave the rest to the other workers
                }
            }
            if (staged.empty()) continue;
            Staged s = staged.front();
            staged.pop_front();
            Result r = solve(wk, s);
            const bool restage = chained_ && r.window_guard;
            publish(std::move(r));
            if (restage) {
                // The callback has just flagged this slice's events as noise (optimizer_rolling.h:52-53), and the slices
                // whose copies were already issued behind it were read without those flags: take them all, drain their
                // staging slots and copy them again.
                std::lock_guard<std::mutex> up(wk.up_mu);
                {
                    std::lock_guard<std::mutex> g(mu_);
                    while (!queue_.empty() && queue_.front().uploaded) { staged.push_back(queue_.front()); queue_.pop_front(); }
                }
                // (only the jobs that HOLD a slot: one whose first copy already failed never got one and is reported as it is)
                std::vector<Staged *> held;
                for (Staged &p : staged) if (p.task.n > 0 && p.upload_rc >= 0) held.push_back(&p);
                for (Staged *p : held) { (void)p; (void)bf_commit_upload(wk.ctx); }
                for (Staged *p : held) {
                    p->upload_rc = issue_upload(wk.ctx, p->task);
                    if (p->upload_rc < 0) --wk.slots;   // (a failed re-issue holds no slot)
                }
            }
        }
    }
};

}  // namespace bf

#endif  // BF_HOST_SLICE_FARM_H
