// bf_context.cpp -- C-ABI (include/bf_accel.h): life cycle of a bf_ctx (AccelLib::AccelLib / ~AccelLib, accel_lib.h:44-69), options,
// diagnostics, raw device buffers and measurement.
#include "bf_ctx.h"

std::atomic<int> g_live_ctx[64];

extern "C" {

const char* bf_version(void) { return "bf_accel gfx950 r5"; }

int bf_device_count(int32_t* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    return (e == hipSuccess && n > 0) ? BF_OK : BF_ERR_NODEVICE;
}

int bf_abi_struct_sizes(int32_t* out, int32_t n) {
    const int32_t sz[8] = {(int32_t)sizeof(bf_model),     (int32_t)sizeof(bf_window),
                           (int32_t)sizeof(bf_run_opts),  (int32_t)sizeof(bf_run_info),
                           (int32_t)sizeof(bf_trace_rec), (int32_t)sizeof(bf_profile),
                           (int32_t)sizeof(bf_local_window), (int32_t)sizeof(bf_local_state)};
    for (int i = 0; out && i < n && i < 8; ++i) out[i] = sz[i];
    return 8;
}

void bf_run_opts_default(bf_run_opts* o) {
    if (!o) return;
    o->max_iter = -1;       // OptimizerRolling(): max_itercount(-1)
    o->min_events = 1000;   // optimizer_rolling.h:57
    o->res_x = 180;         // common.h:39
    o->res_y = 240;         // common.h:40
    o->hard_iter_cap = 100000;
    o->poll_interval = 8;
    o->trace_cap = 0;
    o->want_uv = 0;
}

int bf_create(int32_t device, int64_t max_events, int32_t max_rows, int32_t max_cols, void* hip_stream,
              bf_ctx** out) {
    if (!out || max_events <= 0 || max_rows <= 0 || max_cols <= 0) return BF_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BF_ERR_NODEVICE;
    if (device < 0 || device >= ndev) return BF_ERR_ARG;
    bf_ctx* c = new (std::nothrow) bf_ctx();
    if (!c) return BF_ERR_HIP;
    c->err[0] = 0;
    memset(&c->hst, 0, sizeof(c->hst));
    memset(&c->win, 0, sizeof(c->win));
    memset(&c->prof, 0, sizeof(c->prof));
    c->device = device;
    int rc = [&]() -> int {
        HIP_TRY(c, hipSetDevice(device));
        (void)hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device);
        if (hip_stream) {
            c->stream = (hipStream_t)hip_stream;
        } else {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->own_stream = true;
        }
        const long long gran = (long long)kThreads * kEvPerThread;
        c->cap_events = ((long long)max_events + gran - 1) / gran * gran;
        c->cap_px = (size_t)max_rows * (size_t)max_cols;
        int gx, gy;
        stencil_grid(max_rows, max_cols, &gx, &gy);
        // a window with the same pixel count but another aspect ratio can need more tiles
        c->cap_blocks = gx * gy * 2 + 64;
        const size_t ne = (size_t)c->cap_events;
        HIP_TRY(c, hipMalloc(&c->set[0].xy, ne * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->set[0].t, ne * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->set[0].p, ne * sizeof(float2)));
        HIP_TRY(c, hipMalloc(&c->d_noise, ne));
        HIP_TRY(c, hipMalloc(&c->d_in_x, ne * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->d_in_y, ne * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->d_in_t, ne * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->d_nxny, ne * sizeof(double2)));
        HIP_TRY(c, hipMalloc(&c->d_uv, ne * sizeof(double2)));
        for (int i = 0; i < 2; ++i)
            HIP_TRY(c, hipMalloc(&c->d_plane[i], c->cap_px * sizeof(unsigned long long)));
        HIP_TRY(c, hipMalloc(&c->d_time, c->cap_px * sizeof(float)));
        HIP_TRY(c, hipMalloc(&c->d_gx, c->cap_px * sizeof(float)));
        HIP_TRY(c, hipMalloc(&c->d_gy, c->cap_px * sizeof(float)));
        HIP_TRY(c, hipMalloc(&c->d_img, c->cap_px * sizeof(float)));
        HIP_TRY(c, hipMalloc(&c->d_count, c->cap_px * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_acc, (3 * kAccGroups + 1) * sizeof(MomentAcc)));
        HIP_TRY(c, hipMemsetAsync(c->d_acc, 0, (3 * kAccGroups + 1) * sizeof(MomentAcc), c->stream));
        HIP_TRY(c, hipMalloc(&c->d_ovf, 3 * kOvfSlotWords * sizeof(uint32_t)));   // (three slots of 17 lines: bf_device_fns.h)
        HIP_TRY(c, hipMemsetAsync(c->d_ovf, 0, 3 * kOvfSlotWords * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMalloc(&c->d_state, 2 * sizeof(DevState)));
        HIP_TRY(c, hipMalloc(&c->d_ticket, 16 * 64 * sizeof(unsigned int)));   // 1 + 32 counters, 64 B apart
        HIP_TRY(c, hipMemsetAsync(c->d_ticket, 0, 16 * 64 * sizeof(unsigned int), c->stream));

        HIP_TRY(c, hipHostMalloc(&c->h_state, 2 * sizeof(DevState) + 64, hipHostMallocDefault));
        // (behind the two snapshots: the sequence word a warm start's k_finish_update stores AFTER its snapshot -- bf_run.cpp)
        c->h_seq = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->h_state) + 2 * sizeof(DevState));
        *c->h_seq = 0ull;
        for (int i = 0; i < 2; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->poll_ev[i], hipEventDisableTiming));
        HIP_TRY(c, hipHostMalloc(&c->h_stats, kPrepBlocks * sizeof(SliceStats), hipHostMallocDefault));
        c->d_stats = c->h_stats;   // k_prepare writes its per-work-group records straight into pinned host memory: no copy command
        HIP_TRY(c, hipMemsetAsync(c->d_state, 0, 2 * sizeof(DevState), c->stream));
        // Test hooks and the phase-stamp dump exist only in the debug / timeline builds (`make debug`: debug/libbf_accel.so,
        // -DBF_DEBUG_HOOKS; `make tl`: -DBF_TIMELINE).  The release library reads nothing of this kind from its host's
        // environment: a stray variable cannot change a margin or make the persistent kernel give up.
#ifdef BF_DEBUG_HOOKS
        // (read here, once: bf_run may run on several threads -- no getenv there)
        if (const char* v = getenv("BF_DEBUG_PERSIST_ABORT")) c->dbg_persist_abort = atoi(v);
        if (const char* v = getenv("BF_DEBUG_PERSIST_MUTE")) c->dbg_persist_mute = atoi(v);
        if (const char* v = getenv("BF_DEBUG_PERSIST_SPLIT")) {
            c->dbg_persist_split = atoi(v);
            c->dbg_persist_split_late = strstr(v, ",late") ? 1 : 0;
        }
        if (const char* v = getenv("BF_DEBUG_MARGIN")) {   // tests only: see kBinMargin
            c->dbg_margin = atoi(v);
            if (c->dbg_margin < 1 || c->dbg_margin > 30) return fail(c, BF_ERR_ARG, "BF_DEBUG_MARGIN must be in [1, 30]");
        }
#endif
#ifdef BF_TIMELINE
        c->tl_path = getenv("BF_TIMELINE");
        if (c->tl_path && *c->tl_path) {
            HIP_TRY(c, hipMalloc(&c->d_tl, 3 * 64 * 2 * 16 * sizeof(unsigned long long)));
            HIP_TRY(c, hipMemsetAsync(c->d_tl, 0, 3 * 64 * 2 * 16 * sizeof(unsigned long long), c->stream));
        }
#endif
        int r = clear_planes(c);
        if (r != BF_OK) return r;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return BF_OK;
    }();
    if (rc != BF_OK) {
        fprintf(stderr, "bf_create: %s\n", c->err);
        bf_destroy(c);
        return rc;
    }
    // BF_ACCEL_OPTIONS="key=value,key=value": bf_set_option calls for every context of the process -- for A/B runs through a
    // host that has no flag for an option (the command line).  A bad entry fails the creation loudly.
    if (const char* env = getenv("BF_ACCEL_OPTIONS")) {
        std::string all(env);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(pos, end - pos);
            pos = end + 1;
            if (item.empty()) continue;
            const size_t eq = item.find('=');
            // (strtoll with an end pointer: "fused=abc" or "fused=" must not quietly become 0)
            bool syntax = eq == std::string::npos || eq + 1 >= item.size();
            long long val = 0;
            if (!syntax) {
                char* endp = nullptr;
                errno = 0;
                val = strtoll(item.c_str() + eq + 1, &endp, 10);
                syntax = errno != 0 || endp == item.c_str() + eq + 1 || *endp != '\0';
            }
            const int orc = syntax ? BF_ERR_ARG : bf_set_option(c, item.substr(0, eq).c_str(), val);
            if (orc != BF_OK) {
                fprintf(stderr, "bf_create: BF_ACCEL_OPTIONS entry '%s': %s\n", item.c_str(), syntax ? "expected key=<integer>" : c->err);
                bf_destroy(c);
                return orc;
            }
        }
    }
    g_live_ctx[device & 63].fetch_add(1);
    c->counted = true;
    *out = c;
    return BF_OK;
}

void bf_destroy(bf_ctx* c) {
    if (!c) return;
    if (c->counted) g_live_ctx[c->device & 63].fetch_sub(1);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_tl) {   // debug timeline dump: launch group slot ticks(100 MHz)
        std::vector<unsigned long long> tl(3 * 64 * 2 * 16);   // [kernel][launch][group][slot]
        (void)hipMemcpy(tl.data(), c->d_tl, tl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE* f = fopen(c->tl_path, "w")) {
            for (size_t i = 0; i < tl.size(); ++i)
                if (tl[i]) fprintf(f, "%zu %zu %zu %zu %llu\n", i / 2048, (i / 32) % 64, (i / 16) % 2, i % 16, tl[i]);
            fclose(f);
        }
        (void)hipFree(c->d_tl);
    }
    for (auto& r : c->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (c->poll_ev[i]) (void)hipEventDestroy(c->poll_ev[i]);
    for (int i = 0; i < 2; ++i) if (c->copy_done[i]) (void)hipEventDestroy(c->copy_done[i]);
    for (int i = 0; i < 2; ++i) if (c->staged[i]) (void)hipEventDestroy(c->staged[i]);
    for (int i = 0; i < 2; ++i) {
        if (c->prepared[i]) (void)hipEventDestroy(c->prepared[i]);
        if (c->inc_free[i]) (void)hipEventDestroy(c->inc_free[i]);
        if (c->inc[i].xy) (void)hipFree(c->inc[i].xy);
        if (c->inc[i].t) (void)hipFree(c->inc[i].t);
        if (c->inc[i].p) (void)hipFree(c->inc[i].p);
        if (c->h_stats_slot[i]) (void)hipHostFree(c->h_stats_slot[i]);
    }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (int i = 0; i < 3; ++i) if (c->d_in2[i]) (void)hipFree(c->d_in2[i]);
    for (int i = 0; i < 2; ++i) if (c->d_in_ts[i]) (void)hipFree(c->d_in_ts[i]);
    for (int i = 0; i < 2; ++i) if (c->d_in16[i]) (void)hipFree(c->d_in16[i]);
    for (int i = 0; i < 2; ++i) if (c->d_in_noise[i]) (void)hipFree(c->d_in_noise[i]);
    void* bufs[] = {c->d_xrec, c->d_xred, c->d_verdict, c->d_xscratch[0], c->d_xscratch[1], c->d_xscratch[2], c->d_xscratch[3], c->set[0].p2, c->set[1].p2, c->d_ftab, c->set[0].xy, c->set[0].t, c->set[0].p, c->set[0].perm, c->set[1].xy, c->set[1].t,
                    c->set[1].p, c->set[1].perm, c->d_binid, c->d_hist_cnt, c->d_bin_start,
                    c->d_cursor, c->d_slabs, c->d_cidx, c->d_chdr, c->d_mplane[0], c->d_mplane[1], c->d_mlist, c->d_mcount, c->d_ovf_bits[0], c->d_ovf_bits[1], c->d_armed, c->d_acc, c->d_ovf, c->d_out_tmp, c->d_lplane[0], c->d_lplane[1], c->d_lscore, c->d_limg, c->d_col_planes, c->d_col_img, c->d_tile_hist, c->d_tile_start, c->d_tile_cursor, c->d_tile_states, c->d_many_args, c->d_ltile,
                    c->d_noise, c->d_in_x, c->d_in_y, c->d_in_t, c->d_nxny,
                    c->d_uv, c->d_plane[0], c->d_plane[1], c->d_cplane[0], c->d_cplane[1], c->d_time,
                    c->d_gx, c->d_gy, c->d_img, c->d_count, c->d_ticket, c->d_state,
                    c->d_trace};   // (d_stats is h_stats: freed below)
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (c->h_many_args) (void)hipHostFree(c->h_many_args);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->h_lscore) (void)hipHostFree(c->h_lscore);
    if (c->h_broken) (void)hipHostFree(c->h_broken);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* bf_last_error(const bf_ctx* c) {
    if (!c) return "null ctx";
    // a copy private to the calling thread: another thread of the context's owner (the uploading one) may fail meanwhile
    static thread_local char text[sizeof(c->err)];
    std::lock_guard<std::mutex> g(const_cast<bf_ctx*>(c)->err_mu);
    memcpy(text, c->err, sizeof(text));
    text[sizeof(text) - 1] = 0;
    return text;
}

int bf_get_stat(bf_ctx* c, const char* key, int64_t* value) {
    if (!c || !key || !value) return BF_ERR_ARG;
    if (!strcmp(key, "scatter_format")) {
        *value = c->use_binned ? c->fmt : -1;
        return BF_OK;
    }
    if (!strcmp(key, "one_kernel")) {
        *value = (c->fused_ok && (!c->opt_co_schedule || c->fused_shared)) ? 1 : 0;
        return BF_OK;
    }
    if (!strcmp(key, "persistent")) {   // would bf_run, called now, take the persistent loop kernel?
        *value = (c->fused_ok && !c->opt_co_schedule && (c->opt_persist == 2 || (c->opt_persist == 1 && c->pending_warp)) &&
                  g_live_ctx[c->device & 63].load() == 1 &&
                  fused_loop_resident(c->win.scale / 2, c->fgrid.TSR, c->n_cus, c->fgrid.nbr * c->fgrid.nbc)) ? 1 : 0;
        return BF_OK;
    }
    if (!strcmp(key, "persist_giveups")) {   // launches of the persistent loop kernel that gave up and undid themselves
        *value = c->persist_giveups;
        return BF_OK;
    }
    return fail(c, BF_ERR_ARG, "unknown statistic '%s'", key);
}

int bf_set_option(bf_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return BF_ERR_ARG;
    if (!strcmp(key, "force_split")) {
        c->force_split = value != 0;
        return BF_OK;
    }
    if (!strcmp(key, "binned")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "binned must be 0, 1 or 2");
        c->opt_binned = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_pack_limit")) {
        if (value < 1 || value > 64) return fail(c, BF_ERR_ARG, "bin_pack_limit must be in [1, 64]");
        c->opt_bin_pack_limit = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "co_schedule")) {
        c->opt_co_schedule = value != 0;
        return BF_OK;
    }
    if (!strcmp(key, "stream_prealloc")) {   // everything the 16-bit ring hand-off allocates on first use, now
        if (!value) return BF_OK;
        HIP_TRY(c, hipSetDevice(c->device));
        const int rc = streaming_setup(c);
        if (rc != BF_OK) return rc;
        for (int slot = 0; slot < 2; ++slot) {
            if (!c->d_in_ts[slot]) HIP_TRY(c, hipMalloc(&c->d_in_ts[slot], (size_t)c->cap_events * sizeof(unsigned long long)));
            if (!c->d_in16[slot]) HIP_TRY(c, hipMalloc(&c->d_in16[slot], (size_t)c->cap_events * 2 * sizeof(uint16_t)));
            if (!c->d_in_noise[slot]) HIP_TRY(c, hipMalloc(&c->d_in_noise[slot], (size_t)c->cap_events));   // (else: first use, possibly in the middle of a solve)
        }
        return BF_OK;
    }
    if (!strcmp(key, "sep_update")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "sep_update must be 0 (never), 1 (auto) or 2 (always)");
        c->opt_sep_update = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "defer_uploads")) {
        if (!value) {   // (what was recorded goes out before the mode ends)
            const int rc = issue_deferred_uploads(c);
            if (rc != BF_OK) return rc;
        }
        c->opt_defer_uploads = value != 0;
        return BF_OK;
    }
    if (!strcmp(key, "watchdog_ms")) {
        if (value < 1) return fail(c, BF_ERR_ARG, "watchdog_ms must be >= 1");
        c->opt_watchdog_s = (double)value * 1e-3;
        return BF_OK;
    }
    if (!strcmp(key, "bin_compact")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "bin_compact must be 0 (never), 1 (auto) or 2 (always)");
        c->opt_bin_compact = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_split")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "bin_split must be 0, 1 or 2");
        c->opt_bin_split = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "bin_predict")) {
        c->opt_bin_predict = value != 0;
        return BF_OK;
    }
    if (!strcmp(key, "fused")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "fused must be 0, 1 (auto) or 2");
        c->opt_fused = (int)value;
        return BF_OK;
    }
    if (!strcmp(key, "persist")) {
        if (value < 0 || value > 2) return fail(c, BF_ERR_ARG, "persist must be 0, 1 (auto) or 2");
        c->opt_persist = (int)value;
        return BF_OK;
    }
    return fail(c, BF_ERR_ARG, "unknown option '%s'", key);
}

int bf_synchronize(bf_ctx* c) {
    if (!c) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BF_OK;
}

// ---- NUMA placement (one feeder thread per GPU, next to the GPU) ----------------------------------------------
// On an 8-GPU node the GPUs hang off different host NUMA nodes; a worker thread that polls a pinned snapshot, fills pinned
// staging buffers and launches ~100 000 kernels per second wants to run on the CPUs of ITS GPU's node, and memory it pins
// should come from there (first touch / preferred node).  Linux only, no libnuma: sysfs + sched_setaffinity + set_mempolicy.
}  // extern "C"
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
namespace {
bool read_text(const char* path, char* buf, size_t cap) {
    FILE* f = fopen(path, "r");
    if (!f) return false;
    const size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}
// "0-15,64-79" -> cpu set; false on a syntax error
bool parse_cpulist(const char* s, cpu_set_t* set, int* count) {
    CPU_ZERO(set);
    *count = 0;
    while (*s && *s != '\n') {
        char* e = nullptr;
        const long a = strtol(s, &e, 10);
        if (e == s || a < 0) return false;
        long b = a;
        s = e;
        if (*s == '-') {
            b = strtol(s + 1, &e, 10);
            if (e == s + 1 || b < a) return false;
            s = e;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, set); ++*count; }
        if (*s == ',') ++s;
        else if (*s && *s != '\n') return false;
    }
    return true;
}
}  // namespace
extern "C" {

int bf_device_numa_node(int32_t device, int32_t* node_out) {
    if (!node_out) return BF_ERR_ARG;
    *node_out = -1;
    char bus[64];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return BF_ERR_NODEVICE; }
    for (char* p = bus; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');   // sysfs spells the address in lower case
    char path[160], text[64];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    if (read_text(path, text, sizeof(text))) *node_out = (int32_t)strtol(text, nullptr, 10);   // (-1: the platform does not say)
    return BF_OK;
}

int bf_bind_thread_to_numa_node(int32_t node, int32_t* cpus_out) {
    if (cpus_out) *cpus_out = 0;
    if (node < 0) return BF_OK;   // unknown node: leave the thread where it is
    char path[96], text[4096];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    if (!read_text(path, text, sizeof(text))) return BF_OK;   // no such node here (not a NUMA system): nothing to do
    cpu_set_t want, have, both;
    int n = 0;
    if (!parse_cpulist(text, &want, &n)) return BF_ERR_ARG;
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return BF_OK;
    CPU_AND(&both, &want, &have);   // a container's cpuset wins: never ask for CPUs the process was not given
    const int m = CPU_COUNT(&both);
    if (m == 0) return BF_OK;       // the node's CPUs are not ours: stay
    if (sched_setaffinity(0, sizeof(both), &both) != 0) return BF_OK;
    // memory this thread touches first (and pins) from now on: that node if it has room, any other otherwise
    unsigned long mask[16] = {0};
    if (node < (int)(sizeof(mask) * 8)) {
        mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
        (void)syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, sizeof(mask) * 8);
    }
    if (cpus_out) *cpus_out = m;
    return BF_OK;
}

int bf_bind_thread_to_device_numa(int32_t device, int32_t* node_out) {
    int32_t node = -1;
    const int rc = bf_device_numa_node(device, &node);
    if (node_out) *node_out = node;
    if (rc != BF_OK) return rc;
    return bf_bind_thread_to_numa_node(node, nullptr);
}

// ---- raw device buffers -----------------------------------------------------------------------

int bf_device_malloc(bf_ctx* c, int64_t bytes, void** out) {
    if (!c || !out || bytes <= 0) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMalloc(out, (size_t)bytes));
    return BF_OK;
}

int bf_device_free(bf_ctx* c, void* ptr) {
    if (!c) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (ptr) HIP_TRY(c, hipFree(ptr));
    return BF_OK;
}

int bf_memcpy_h2d(bf_ctx* c, void* dst, const void* src, int64_t bytes) {
    if (!c || !dst || !src || bytes < 0) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return BF_OK;
}

// ---- measurement ---------------------------------------------------------------------------

int bf_profile_enable(bf_ctx* c, int32_t mode) {
    if (!c || mode < 0 || mode > 1) return BF_ERR_ARG;
    int rc = prof_fold(c);
    c->prof_mode = mode;
    return rc;
}

int bf_profile_reset(bf_ctx* c) {
    if (!c) return BF_ERR_ARG;
    int rc = prof_fold(c);
    memset(&c->prof, 0, sizeof(c->prof));
    return rc;
}

int bf_profile_get(bf_ctx* c, bf_profile* out) {
    if (!c || !out) return BF_ERR_ARG;
    int rc = prof_fold(c);
    *out = c->prof;
    return rc;
}

int bf_eval_sincos(bf_ctx* c, const double* x, int64_t n, int32_t table, double* sin_out, double* cos_out) {
    if (!c || !x || !sin_out || !cos_out || n < 0) return BF_ERR_ARG;
    if (n == 0) return BF_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    double* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, (size_t)n * 3 * sizeof(double)));
    int rc = [&]() -> int {
        HIP_TRY(c, hipMemcpyAsync(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        launch_eval_sincos(d, n, table ? 1 : 0, d + n, d + 2 * n, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(sin_out, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(cos_out, d + 2 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return BF_OK;
    }();
    (void)hipFree(d);
    return rc;
}

int bf_copy_bandwidth(bf_ctx* c, int64_t bytes, int32_t reps, double* gbps_out) {
    if (!c || !gbps_out || bytes < 4096 || reps < 1) return BF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    bytes &= ~(int64_t)15;
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double best = 0.0;
    const int rc = [&]() -> int {   // (whatever fails, the buffers and events below are released)
        HIP_TRY(c, hipMalloc(&a, (size_t)bytes));
        HIP_TRY(c, hipMalloc(&b, (size_t)bytes));
        HIP_TRY(c, hipMemsetAsync(a, 1, (size_t)bytes, c->stream));
        HIP_TRY(c, hipEventCreate(&e0));
        HIP_TRY(c, hipEventCreate(&e1));
        // the ceiling is the best of a few launch shapes (work-groups per CU, plain / non-temporal accesses), each warmed up
        for (int nt = 0; nt < 2; ++nt)
            for (int blocks : {1024, 2048, 4096, 8192}) {
                launch_copy(a, b, bytes, blocks, nt != 0, c->stream);   // warm-up
                for (int r = 0; r < reps; ++r) {
                    HIP_TRY(c, hipEventRecord(e0, c->stream));
                    launch_copy(a, b, bytes, blocks, nt != 0, c->stream);
                    HIP_TRY(c, hipEventRecord(e1, c->stream));
                    HIP_TRY(c, hipEventSynchronize(e1));
                    float ms = 0.f;
                    HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
                    const double g = 2.0 * (double)bytes / ((double)ms * 1e-3) / 1e9;
                    if (g > best) best = g;
                }
            }
        return BF_OK;
    }();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (rc != BF_OK) return rc;
    *gbps_out = best;
    return BF_OK;
}

}  // extern "C"
