// bf_upload.cpp -- C-ABI: staging a slice on the device (AccelLib::init_gpu, accel_lib.h:71-115), blocking, from device arrays, and
// asynchronously on the copy stream from pinned arrays or from a structure-of-arrays event ring (DVS_flow::recompute hand-off).
#include <utility>

#include "bf_ctx.h"

static int stage_early(bf_ctx* c, int slot);

// The slice hand-off of DVS_flow::recompute (dvs_flow.h:185-216) for a structure-of-arrays ring in pinned
// memory: up to two contiguous pieces per array, no repacking on the host.  ADDR is int32_t (bf_upload_ring_async) or
// uint16_t (bf_upload_ring16_async: the addresses travel as 16-bit values and are widened by the staging kernel); TS is uint64_t
// (absolute nanoseconds) or uint32_t (their low 32 bits, bf_upload_ring16t32_async: 8 bytes per event over the link).
template <class ADDR, class TS>
static int upload_ring(bf_ctx* c, const ADDR* ring_x, const ADDR* ring_y, const TS* ring_ts, const uint8_t* ring_noise,
                       int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    if (!c) return BF_ERR_ARG;
    if (n <= 0 || cap <= 0 || first < 0 || first >= cap || n > cap || !ring_x || !ring_y || !ring_ts)
        return fail(c, BF_ERR_ARG, "bad ring slice (cap %lld, first %lld, n %lld)", (long long)cap, (long long)first, (long long)n);
    if (n > c->cap_events) return fail(c, BF_ERR_CAPACITY, "n=%lld exceeds capacity %lld", (long long)n, c->cap_events);
    if (c->pend_count >= 2) return fail(c, BF_ERR_STATE, "two uploads are already pending");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rc = streaming_setup(c);
        if (rc != BF_OK) return rc;
    }
    const int slot = (c->pend_head + c->pend_count) & 1;
    auto body = [=]() -> int {
    if (c->staged_valid[slot]) HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->staged[slot], 0));
    if (!c->d_in_ts[slot]) HIP_TRY(c, hipMalloc(&c->d_in_ts[slot], (size_t)c->cap_events * sizeof(unsigned long long)));
    const bool narrow = sizeof(ADDR) == 2;
    if (narrow && !c->d_in16[slot]) HIP_TRY(c, hipMalloc(&c->d_in16[slot], (size_t)c->cap_events * 2 * sizeof(uint16_t)));
    if (ring_noise && !c->d_in_noise[slot]) HIP_TRY(c, hipMalloc(&c->d_in_noise[slot], (size_t)c->cap_events));
    // destinations of the two address columns: the slot's int32 staging arrays, or (16-bit form) two halves of d_in16
    ADDR* dx = narrow ? reinterpret_cast<ADDR*>(c->d_in16[slot]) : reinterpret_cast<ADDR*>(slot ? c->d_in2[0] : c->d_in_x);
    ADDR* dy = narrow ? reinterpret_cast<ADDR*>(c->d_in16[slot] + c->cap_events) : reinterpret_cast<ADDR*>(slot ? c->d_in2[1] : c->d_in_y);
    const int64_t n0 = (first + n <= cap) ? n : cap - first, n1 = n - n0;   // [first, first + n0) then [0, n1)
    HIP_TRY(c, hipMemcpyAsync(dx, ring_x + first, (size_t)n0 * sizeof(ADDR), hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(dy, ring_y + first, (size_t)n0 * sizeof(ADDR), hipMemcpyHostToDevice, c->copy_stream));
    TS* dts = reinterpret_cast<TS*>(c->d_in_ts[slot]);   // (cap_events x 8 bytes: the 32-bit form uses half of it)
    HIP_TRY(c, hipMemcpyAsync(dts, ring_ts + first, (size_t)n0 * sizeof(TS), hipMemcpyHostToDevice, c->copy_stream));
    if (ring_noise) HIP_TRY(c, hipMemcpyAsync(c->d_in_noise[slot], ring_noise + first, (size_t)n0, hipMemcpyHostToDevice, c->copy_stream));
    if (n1 > 0) {
        HIP_TRY(c, hipMemcpyAsync(dx + n0, ring_x, (size_t)n1 * sizeof(ADDR), hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipMemcpyAsync(dy + n0, ring_y, (size_t)n1 * sizeof(ADDR), hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipMemcpyAsync(dts + n0, ring_ts, (size_t)n1 * sizeof(TS), hipMemcpyHostToDevice, c->copy_stream));
        if (ring_noise) HIP_TRY(c, hipMemcpyAsync(c->d_in_noise[slot] + n0, ring_noise, (size_t)n1, hipMemcpyHostToDevice, c->copy_stream));
    }
    HIP_TRY(c, hipEventRecord(c->copy_done[slot], c->copy_stream));
    c->pending_n[slot] = n;
    c->pending_ts64[slot] = true;
    c->pending_ts32[slot] = sizeof(TS) == 4;
    c->pending_addr16[slot] = narrow;
    c->pending_noise[slot] = ring_noise != nullptr;
    c->pending_t0[slot] = t0;
    c->pending_early[slot] = false;
    // (a noise ring goes through d_noise, which the running slice may still read: staged at the commit.  And only a context that
    // has the GPU to itself stages early: kernels on the copy stream need a hardware queue of their own, and the runtime gives a
    // process four -- with four "co_schedule"d chains in flight, eight kernel-carrying streams shared them and the warm regime
    // fell from 5.3 to 3.2 Gevents/s.)
    if (!ring_noise && !c->opt_co_schedule) {
        const int rc = stage_early(c, slot);
        if (rc != BF_OK) return rc;
    }
    return BF_OK;
    };
    c->pend_count++;
    if (c->opt_defer_uploads) { c->deferred[slot] = body; return BF_OK; }
    const int rc = body();
    if (rc != BF_OK) c->pend_count--;
    return rc;
}

// Early staging of the slice just copied into `slot` (no noise ring): its widening kernel and k_prepare on the COPY stream,
// behind the copies, into the slot's own event arrays (inc[slot]) and pinned statistics record.  The arrays may have been
// swapped out of set[0] at an earlier commit: the compute stream must be past that commit first (inc_free).
static int stage_early(bf_ctx* c, int slot) {
    // (the committed slice's statistics may still sit, unread, in this slot's record -- a caller that uploads two slices ahead
    // before bf_set_cloud: read them before k_prepare overwrites the record)
    if (c->uploaded && !c->stats_valid && c->stats_src == c->h_stats_slot[slot]) {
        const int rc = fold_stats(c);
        if (rc != BF_OK) return rc;
    }
    if (c->inc_free_valid[slot]) HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->inc_free[slot], 0));
    int32_t* dx = slot ? c->d_in2[0] : c->d_in_x;
    int32_t* dy = slot ? c->d_in2[1] : c->d_in_y;
    int32_t* dt = slot ? c->d_in2[2] : c->d_in_t;
    const long long n = c->pending_n[slot];
    if (c->pending_ts64[slot]) {
        if (c->pending_addr16[slot])
            launch_local_time16(c->d_in_ts[slot], c->pending_ts32[slot], c->d_in16[slot], c->d_in16[slot] + c->cap_events, c->pending_t0[slot],
                                dx, dy, dt, n, c->copy_stream);
        else
            launch_local_time(c->d_in_ts[slot], c->pending_t0[slot], dt, n, c->copy_stream);
    }
    const long long gran = (long long)kThreads * kEvPerThread;
    const long long n_pad = (n + gran - 1) / gran * gran;
    launch_prepare(dx, dy, dt, c->inc[slot].xy, c->inc[slot].t, c->inc[slot].p, n, n_pad, c->h_stats_slot[slot], c->copy_stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->prepared[slot], c->copy_stream));
    c->pending_early[slot] = true;
    return BF_OK;
}

extern "C" {

// ---- slice set-up ---------------------------------------------------------------------

static int stage_common(bf_ctx* c, const int32_t* dx, const int32_t* dy, const int32_t* dt, long long n) {
    const long long gran = (long long)kThreads * kEvPerThread;
    c->n_pad = (n + gran - 1) / gran * gran;
    {
        ProfScope ps(c, 3);
        launch_prepare(dx, dy, dt, c->set[0].xy, c->set[0].t, c->set[0].p, n, c->n_pad, c->d_stats,
                       c->stream);
        c->cs = 0;
        c->has_perm = false;
        c->stats_src = c->h_stats;
        c->stats_event = nullptr;
    }
    HIP_TRY(c, hipGetLastError());
    return after_upload(c, n);
}

int bf_upload_events(bf_ctx* c, const int32_t* fr_x, const int32_t* fr_y, const int32_t* t_ns,
                     const uint8_t* noise, int64_t n) {
    if (!c) return BF_ERR_ARG;
    if (n < 0 || (n > 0 && (!fr_x || !fr_y || !t_ns))) return fail(c, BF_ERR_ARG, "bad event arrays");
    if (n > c->cap_events) return fail(c, BF_ERR_CAPACITY, "n=%lld exceeds capacity %lld", (long long)n, c->cap_events);
    if (c->pend_count > 0)   // (the blocking upload stages through slot 0, which a pending asynchronous upload may own)
        return fail(c, BF_ERR_STATE, "bf_upload_events while %d asynchronous upload(s) are pending", c->pend_count);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->staged_valid[0]) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->staged[0], 0));
    const size_t nb = (size_t)n * sizeof(int32_t);
    if (n > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->d_in_x, fr_x, nb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_in_y, fr_y, nb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_in_t, t_ns, nb, hipMemcpyHostToDevice, c->stream));
    }
    c->has_noise = false;
    if (noise && n > 0) {
        const long long gran = (long long)kThreads * kEvPerThread;
        const long long n_pad = (n + gran - 1) / gran * gran;
        HIP_TRY(c, hipMemsetAsync(c->d_noise, 0, (size_t)n_pad, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_noise, noise, (size_t)n, hipMemcpyHostToDevice, c->stream));
        c->has_noise = true;
    }
    int rc = stage_common(c, c->d_in_x, c->d_in_y, c->d_in_t, n);
    if (rc != BF_OK) return rc;
    // host arrays are only borrowed for the duration of the call
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

int bf_host_alloc(bf_ctx* c, int64_t bytes, void** out) {
    if (!c || !out || bytes <= 0) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable));   // (portable: a slice farm uploads from one ring to several devices)
    return BF_OK;
}

int bf_host_free(bf_ctx* c, void* ptr) {
    if (!c) return BF_ERR_ARG;
    if (ptr) HIP_TRY(c, hipHostFree(ptr));
    return BF_OK;
}

int bf_upload_events_async(bf_ctx* c, const int32_t* fr_x, const int32_t* fr_y, const int32_t* t_ns, int64_t n) {
    if (!c) return BF_ERR_ARG;
    if (n <= 0 || !fr_x || !fr_y || !t_ns) return fail(c, BF_ERR_ARG, "bad event arrays");
    if (n > c->cap_events) return fail(c, BF_ERR_CAPACITY, "n=%lld exceeds capacity %lld", (long long)n, c->cap_events);
    if (c->pend_count >= 2) return fail(c, BF_ERR_STATE, "two uploads are already pending");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rc = streaming_setup(c);
        if (rc != BF_OK) return rc;
    }
    const int slot = (c->pend_head + c->pend_count) & 1;
    auto body = [=]() -> int {
    // the slot's previous content may still be waiting for its staging kernel on the compute stream
    if (c->staged_valid[slot]) HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->staged[slot], 0));
    int32_t* dx = slot ? c->d_in2[0] : c->d_in_x;
    int32_t* dy = slot ? c->d_in2[1] : c->d_in_y;
    int32_t* dt = slot ? c->d_in2[2] : c->d_in_t;
    const size_t nb = (size_t)n * sizeof(int32_t);
    HIP_TRY(c, hipMemcpyAsync(dx, fr_x, nb, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(dy, fr_y, nb, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(dt, t_ns, nb, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipEventRecord(c->copy_done[slot], c->copy_stream));
    c->pending_n[slot] = n;
    c->pending_ts64[slot] = false;
    c->pending_early[slot] = false;
    return c->opt_co_schedule ? BF_OK : stage_early(c, slot);
    };
    c->pend_count++;
    if (c->opt_defer_uploads) { c->deferred[slot] = body; return BF_OK; }
    const int rc = body();
    if (rc != BF_OK) c->pend_count--;
    return rc;
}

int bf_upload_ring_async(bf_ctx* c, const int32_t* ring_x, const int32_t* ring_y, const uint64_t* ring_ts, const uint8_t* ring_noise,
                         int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    return upload_ring<int32_t, uint64_t>(c, ring_x, ring_y, ring_ts, ring_noise, cap, first, n, t0);
}

int bf_upload_ring16_async(bf_ctx* c, const uint16_t* ring_row, const uint16_t* ring_col, const uint64_t* ring_ts,
                           const uint8_t* ring_noise, int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    return upload_ring<uint16_t, uint64_t>(c, ring_row, ring_col, ring_ts, ring_noise, cap, first, n, t0);
}

int bf_upload_ring16t32_async(bf_ctx* c, const uint16_t* ring_row, const uint16_t* ring_col, const uint32_t* ring_t32,
                              const uint8_t* ring_noise, int64_t cap, int64_t first, int64_t n, uint64_t t0) {
    return upload_ring<uint16_t, uint32_t>(c, ring_row, ring_col, ring_t32, ring_noise, cap, first, n, t0);
}

int bf_upload_events16_async(bf_ctx* c, const uint16_t* fr_x, const uint16_t* fr_y, const int32_t* t_ns, int64_t n) {
    // (slice-local times are their own low 32 bits relative to t0 = 0)
    return upload_ring<uint16_t, uint32_t>(c, fr_x, fr_y, reinterpret_cast<const uint32_t*>(t_ns), nullptr, n, 0, n, 0);
}

int bf_wait_uploads(bf_ctx* c) {
    if (!c) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rc = issue_deferred_uploads(c);
        if (rc != BF_OK) return rc;
    }
    if (c->copy_stream) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    return BF_OK;
}

int bf_commit_upload(bf_ctx* c) {
    if (!c) return BF_ERR_ARG;
    if (c->pend_count == 0) return fail(c, BF_ERR_STATE, "no upload is pending");
    HIP_TRY(c, hipSetDevice(c->device));
    {   // ("defer_uploads": whatever is still only recorded goes out now, oldest first)
        const int rc = issue_deferred_uploads(c);
        if (rc != BF_OK) return rc;
    }
    const int slot = c->pend_head & 1;
    if (c->pending_early[slot]) {
        // Staged already (copy stream): the compute stream waits for that, set[0] takes the slot's arrays -- a pointer swap; the
        // arrays that leave set[0] may still be in use by what the compute stream holds (the previous slice's last kernels), so
        // the next staging into them waits for this point of the compute stream -- and the statistics are the slot's record.
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->prepared[slot], 0));
        std::swap(c->set[0].xy, c->inc[slot].xy);
        std::swap(c->set[0].t, c->inc[slot].t);
        std::swap(c->set[0].p, c->inc[slot].p);
        HIP_TRY(c, hipEventRecord(c->inc_free[slot], c->stream));
        c->inc_free_valid[slot] = true;
        c->has_noise = false;
        const long long gran = (long long)kThreads * kEvPerThread;
        c->n_pad = (c->pending_n[slot] + gran - 1) / gran * gran;
        c->cs = 0;
        c->has_perm = false;
        c->stats_src = c->h_stats_slot[slot];
        c->stats_event = c->prepared[slot];
        c->pending_early[slot] = false;
        c->staged_valid[slot] = false;   // (the copy stream itself orders the slot's next copies behind its staging kernels)
        int rc = after_upload(c, c->pending_n[slot]);
        // the statistics are usually there already (the staging ran under the previous slice's solve): read them now, so that
        // the slot's record is free for the next upload whoever issues it, and bf_set_cloud has nothing to wait for
        if (rc == BF_OK && hipEventQuery(c->prepared[slot]) == hipSuccess) rc = fold_stats(c);
        c->pend_head++;
        c->pend_count--;
        return rc;
    }
    // the staging kernel (compute stream) waits for the copy; nothing blocks on the host
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_done[slot], 0));
    c->has_noise = false;
    if (c->pending_ts64[slot]) {   // absolute timestamps -> slice-local 32-bit times (Event::set_local_time)
        if (c->pending_addr16[slot])   // ... and 16-bit addresses -> the staging kernel's int32 columns, same pass
            launch_local_time16(c->d_in_ts[slot], c->pending_ts32[slot], c->d_in16[slot], c->d_in16[slot] + c->cap_events, c->pending_t0[slot],
                                slot ? c->d_in2[0] : c->d_in_x, slot ? c->d_in2[1] : c->d_in_y, slot ? c->d_in2[2] : c->d_in_t,
                                c->pending_n[slot], c->stream);
        else
            launch_local_time(c->d_in_ts[slot], c->pending_t0[slot], slot ? c->d_in2[2] : c->d_in_t, c->pending_n[slot], c->stream);
        if (c->pending_noise[slot]) {   // Event::noise of the slice (padding: not noise, like the blocking upload's)
            const long long gran = (long long)kThreads * kEvPerThread;
            const long long n_pad = (c->pending_n[slot] + gran - 1) / gran * gran;
            HIP_TRY(c, hipMemsetAsync(c->d_noise, 0, (size_t)n_pad, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->d_noise, c->d_in_noise[slot], (size_t)c->pending_n[slot], hipMemcpyDeviceToDevice, c->stream));
            c->has_noise = true;
        }
    }
    int rc = stage_common(c, slot ? c->d_in2[0] : c->d_in_x, slot ? c->d_in2[1] : c->d_in_y,
                          slot ? c->d_in2[2] : c->d_in_t, c->pending_n[slot]);
    // the slot may be refilled once the staging kernels above have read it
    HIP_TRY(c, hipEventRecord(c->staged[slot], c->stream));
    c->staged_valid[slot] = true;
    c->pend_head++;
    c->pend_count--;
    return rc;
}

int bf_upload_events_device(bf_ctx* c, const int32_t* d_fr_x, const int32_t* d_fr_y, const int32_t* d_t_ns,
                            int64_t n) {
    if (!c) return BF_ERR_ARG;
    if (n < 0 || (n > 0 && (!d_fr_x || !d_fr_y || !d_t_ns))) return fail(c, BF_ERR_ARG, "bad event arrays");
    if (n > c->cap_events) return fail(c, BF_ERR_CAPACITY, "n=%lld exceeds capacity %lld", (long long)n, c->cap_events);
    HIP_TRY(c, hipSetDevice(c->device));
    c->has_noise = false;
    return stage_common(c, d_fr_x, d_fr_y, d_t_ns, n);
}

}  // extern "C"
