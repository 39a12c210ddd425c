/*
 * bf_accel.h -- C-ABI of the MI355X (gfx950) motion-compensation path.
 *
 * This is the drop-in boundary for better-flow's `class AccelLib`
 * (better_flow_core/include/better_flow/accel_lib.h:14-616) as it is consumed by
 * `OptimizerRolling` (optimizer_rolling.h:19,269,294,308,322,327,340) plus the fused
 * on-device form of `OptimizerRolling::run` (optimizer_rolling.h:48-125,305-347).
 * Every entry point cites the reference interface it replaces.  All arguments are
 * plain pointers / sizes / PODs; outputs are caller-allocated; nothing throws.
 *
 * Conventions (same as the reference): fr_x is the sensor ROW, fr_y the COLUMN
 * (bf_motion_compensator.cpp:192,200); t is nanoseconds relative to the slice start
 * (Event::set_local_time, event.h:61-63) and must fit int32 (the reference's own
 * device layout, accel_lib.h:83-85); images are row-major R x C with
 * R = metric_wsizex + scale, C = metric_wsizey + scale.
 *
 * Return values: 0 = BF_OK, 1 = BF_SKIPPED (run() skipped by a guard,
 * optimizer_rolling.h:54,58 -- the reference's own "return 1"), < 0 = error.
 * There is NO CPU fallback: without a usable HIP device bf_create fails with
 * BF_ERR_NODEVICE and every other call fails with BF_ERR_ARG on a NULL ctx.
 *
 * Threading: one bf_ctx per (host thread, GPU).  A ctx is reusable across slices
 * (no per-slice allocation), owns all of its device memory, pinned staging and its HIP
 * stream, and is NOT thread-safe -- with one exception, which the stream engine
 * (better_flow/slice_farm.h) relies on: the asynchronous uploads (bf_upload_events_async,
 * bf_upload_ring_async, bf_upload_ring16_async, bf_upload_ring16t32_async, bf_upload_events16_async) and bf_wait_uploads touch only the
 * context's staging slots and copy stream and may be called from a SECOND thread while
 * the owning thread is inside bf_set_cloud, bf_set_model, bf_run or bf_compute_uv*.  The
 * caller serialises them against each other and against bf_commit_upload (one mutex),
 * and sets "stream_prealloc" first so that no upload allocates device memory in the
 * middle of a solve.  Nothing else may run concurrently on one ctx.  bf_last_error
 * returns a copy of the text private to the calling thread.
 */
#ifndef BF_ACCEL_H
#define BF_ACCEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_OK 0
#define BF_SKIPPED 1
#define BF_ERR_ARG (-1)
#define BF_ERR_HIP (-2)
#define BF_ERR_STATE (-3)
#define BF_ERR_NOCONV (-4)   /* hard iteration cap hit (the reference would spin) */
#define BF_ERR_NODEVICE (-5)
#define BF_ERR_CAPACITY (-6)

typedef struct bf_ctx bf_ctx;

/* ObjectModel, object_model.h:10-13. */
typedef struct bf_model {
    double cx, cy, dx, dy, rot, div;
    uint32_t cnt;
    uint32_t _pad;
    double total_dx, total_dy, total_rot, total_div;
} bf_model;

/* Window geometry of OptimizerRolling::set_cloud / set_scale,
 * optimizer_rolling.h:248-283 (members :22-31). */
typedef struct bf_window {
    int32_t scale;
    int32_t x_min, y_min, x_max, y_max;
    int32_t metric_wsizex, metric_wsizey;
    int32_t scale_img_x, scale_img_y;   /* R, C */
    int32_t _pad;
    double x_shift, y_shift;
} bf_window;

/* Options of the fused run; defaults (bf_run_opts_default) are the reference's. */
typedef struct bf_run_opts {
    int32_t max_iter;        /* OptimizerRolling::set_maxiter, :236-238; <= 0: unlimited */
    int32_t min_events;      /* literal 1000 at optimizer_rolling.h:57                  */
    int32_t res_x, res_y;    /* RES_X / RES_Y of the window guard, :49 (common.h:39-40) */
    int32_t hard_iter_cap;   /* not in the reference: give up with BF_ERR_NOCONV        */
    int32_t poll_interval;   /* iterations enqueued between host polls of `done`        */
    int32_t trace_cap;       /* per-iteration records kept for bf_get_trace (0 = none)  */
    int32_t want_uv;         /* 1: also compute per-event (u,v) (Event::compute_uv)     */
} bf_run_opts;

/* What run() leaves behind besides the model (optimizer_rolling.h:36,59). */
typedef struct bf_run_info {
    int32_t rc;              /* 0 optimised, 1 skipped, <0 error                        */
    int32_t iterations;      /* itercount                                               */
    float x_divider, y_divider, rot_divider, div_divider;
    int32_t launches;        /* kernel launches enqueued (diagnostic)                   */
    int32_t polls;           /* host polls of the done flag (diagnostic)                */
    int32_t rebins;          /* counting sorts of the events by image tile (diagnostic)  */
    int32_t overflow_events; /* events that left their bin's LDS tile, summed over the
                                iterations; they take an exact global-atomic path.  The one-kernel loop ("fused") has no
                                such path: there, the passes it repeated after a re-bin    */
} bf_run_info;

/* One record per iteration_step (trajectory tests). */
typedef struct bf_trace_rec {
    bf_model model;
    float x_divider, y_divider, rot_divider, div_divider;
    int32_t iteration;
    int32_t _pad;
} bf_trace_rec;

/* Accumulated per-kernel GPU time, measured with hipEvents on the ctx stream. */
typedef struct bf_profile {
    double warp_scatter_ms;  uint64_t warp_scatter_launches;
    double stencil_ms;       uint64_t stencil_launches;
    double update_ms;        uint64_t update_launches;
    double other_ms;         uint64_t other_launches;
    uint64_t warp_scatter_events;   /* sum over launches of events processed */
} bf_profile;

/* ---- life cycle -------------------------------------------------------------- */

/* Number of visible HIP devices (0 and BF_ERR_NODEVICE if the runtime has none). */
int bf_device_count(int32_t *count);

/* AccelLib::AccelLib (accel_lib.h:44-55) + the allocations of init_gpu (:86-92,
 * :127-144).  Capacity is fixed at creation: up to max_events events and images of
 * up to max_rows x max_cols pixels (R x C).  hip_stream: a hipStream_t to run on, or
 * NULL to let the ctx create (and own) a non-blocking stream. */
int bf_create(int32_t device, int64_t max_events, int32_t max_rows, int32_t max_cols,
              void *hip_stream, bf_ctx **out);

/* AccelLib::~AccelLib / clear_buffers (accel_lib.h:57-69). */
void bf_destroy(bf_ctx *ctx);

/* Text of the last error on this ctx ("" if none).  Never NULL.  Takes the place of the reference's
 * "OpenCL Error: <n>" prints (accel_lib.h:125,138,253,299,361): errors are returned, the text is kept here. */
const char *bf_last_error(const bf_ctx *ctx);

/* Library / build identification, e.g. "bf_accel gfx950 r5" (printed by `bf_motion_compensator --version`
 * next to the reference's version line, bf_motion_compensator.cpp:26-33). */
const char *bf_version(void);

/* The reference's settings of OptimizerRolling::run: no iteration cap (optimizer_rolling.h:25,43, max_itercount = -1),
 * guards of :49-58 (1000 events, window >= scale * RES / 15), sensor 180 x 240 (common.h:39-40). */
void bf_run_opts_default(bf_run_opts *opts);

/* No reference counterpart (the reference has no foreign-language boundary).
 * sizeof() of bf_model, bf_window, bf_run_opts, bf_run_info, bf_trace_rec, bf_profile,
 * bf_local_window, bf_local_state (in that order) as this library was compiled -- lets a
 * foreign-language binding verify its struct layouts.  Writes min(n, 8) entries; returns 8. */
int bf_abi_struct_sizes(int32_t *out, int32_t n);

/* Tuning / test knobs that have no counterpart in the reference (each takes effect at the
 * next bf_set_cloud).  Keys:
 *   "force_split"  1: keep the event-count and time-sum accumulators in separate planes
 *                  even when they fit one packed 64-bit word.
 *   "binned"       tile-binned LDS scatter inside bf_run: 1 (default) when it pays (fewer than ~12 image pixels per
 *                  event, or an image of at most 9 M pixels -- sparse slices then take its event-list form), 2 whenever
 *                  possible, 0 never (one global atomic per event).  Results are identical.
 *   "fused"        the one-kernel iteration inside bf_run (warp + LDS scatter + stencil + moment sums of one image tile
 *                  per work-group; the events of a tile's edge strips are also read by the neighbouring tiles' work-groups):
 *                  1 (default) where it is the faster loop -- slices of at most one event per two image pixels on images
 *                  of at most 1.2 M pixels, i.e. the reference's 50 000-event ring on a 240x180 or 346x260 sensor --, 2
 *                  whenever possible, 0 never.  A "co_schedule" context takes it only for sparser slices (at most one event
 *                  per eight image pixels: launch-bound even with eight contexts in flight).  Bit-identical to the two-kernel loop;
 *                  bf_run_info::overflow_events then counts the passes that were repeated after a re-bin because an
 *                  event had moved further than the loop's margin of 8 scaled pixels (detected exactly, never a wrong sum).
 *   "persist"      the persistent form of that loop (k_fused_loop, bf_loop.hip): the work-groups stay resident, keep their
 *                  events in registers and run iteration after iteration in ONE launch, exchanging the moment sums through
 *                  tagged records in memory instead of a launch boundary; the launch ends when the loop is over or a re-bin
 *                  is due.  0 never; 1 (default) for warm-started runs (bf_set_model: a stream's chain, where a slice is
 *                  ~100 iterations and one or two re-bins) on a context that is not "co_schedule"d, is the only context of
 *                  this process on its device (two resident kernels could wait for each other's CUs) and whose tiles can
 *                  all be resident at once; 2 for cold runs too (they re-bin a dozen times, each a host round trip: slower).
 *                  Bit-identical to the other loops.
 *   BF_ACCEL_OPTIONS (environment, read by bf_create): "key=value,key=value" applied to every context of the process.
 *   "bin_pack_limit"  bits the per-bin accumulator packing may use (default 64).  The counting sort sizes the
 *                  packed count / time-sum fields from the fullest bin; if they do not fit, every event takes the
 *                  exact unpacked path.  Lower values only serve to exercise that fallback in tests.
 *   "co_schedule"  1: this context shares the GPU with other slice contexts (threads / streams).  The model /
 *                  loop update of the tile-binned loop then runs in the last work-group of the stencil kernel (a
 *                  serial tail on one CU, which the other contexts' kernels fill) instead of at the head of the
 *                  next warp+scatter launch by every work-group (the shortest iteration for a context alone, but
 *                  ~1.5 us on all CUs).  Results are identical bit for bit.  Default 0.
 *   "stream_prealloc"  1: create now what the asynchronous uploads (bf_upload_events_async / bf_upload_ring*_async) create
 *                  on first use -- the copy stream, its events, both staging slots -- so that the first slice of a stream
 *                  does not pay ~20 ms of allocations.
 *   "sep_update"   where the model / loop update of a "co_schedule"d context's two-kernel loop runs: in the stencil kernel's last
 *                  work-group (every work-group drains its accumulator atomics and takes a ticket), or as a one-wave kernel of
 *                  its own ahead of every scatter launch (the stencil kernel then only accumulates).  0 the former, 2 the latter,
 *                  1 (default): the latter for event lists -- thousands of stencil work-groups per launch, each holding its slot
 *                  ~10 % longer for the ticket -- and the former for dense tiles (a few hundred work-groups: the extra launch costs
 *                  more).  Bit-identical results.
 *   "defer_uploads"  1: an asynchronous upload (bf_upload_events_async / bf_upload_events16_async / bf_upload_ring*_async) only takes
 *                  its staging slot and remembers its arguments; the HIP calls behind it -- three copies, the staging kernels, the
 *                  events: ~25 us of host time -- are issued by the next bf_run as soon as that run's first batch of kernels is
 *                  queued (or by bf_commit_upload / bf_wait_uploads, whichever needs the slot first).  For a caller that drives
 *                  ONE warm-started chain from ONE thread (dvs_flow.h:218-224) -- commit k, upload k + 1, set_cloud, run -- that
 *                  host time otherwise falls between two runs with the GPU idle.  The host arrays must stay valid until the
 *                  slice is committed (the asynchronous uploads' contract already).  Not for callers that upload from a second
 *                  thread.  Default 0.
 *   "watchdog_ms"  a cold bf_run whose device iteration counter has not advanced for this long (wall clock, default
 *                  40000) stops with BF_ERR_HIP "device loop makes no progress" instead of waiting for ever.
 *   "bin_predict"  1 (default): re-bin as soon as the model has moved events by 0.6 x the bins' margin (8 scaled pixels;
 *                  bounded analytically), i.e. before they leave their bin's LDS tile and take the exact overflow path;
 *                  0: re-bin only on observed overflow.
 *   "bin_compact"  what the scatter kernel hands to the stencil kernel: 0 dense tiles (the bin's events merged in an LDS
 *                  tile, one accumulator per tile pixel written), 2 event lists (one entry per event: tile-local pixel
 *                  index + packed accumulator, sorted by tile row; no LDS tile); 1 (default): lists when the slice has
 *                  fewer than one event per four pixels, dense tiles otherwise.  With lists, traffic and work follow the
 *                  events instead of the image area.  Bit-identical results in either form.
 * (Round 5 removed eight keys whose sweeps had been flat for two rounds or that only forced what bf_set_cloud / bf_run choose
 * per slice -- tile width / height, margins, scatter work-group size and events per thread, the one-kernel loop's tile rows, the
 * spinning poll -- and the merged-list scatter format; tests reach the paths they forced through geometry, "bin_predict" = 0 and
 * "bin_pack_limit".)
 *   "bin_split"    dense tiles only: 0 the bin writes its whole LDS tile, margin included, and the stencil kernel merges
 *                  up to 2 x 2 such slabs per pixel; 2 the bin writes its own pixels into a tiled image and ADDS the few
 *                  words its events left in the tile's margin to a margin plane (device atomics; the bin clears them again
 *                  at its next launch): 0.6 x the bytes, a quarter of the stencil kernel's loads; 1 (default): the second
 *                  form for a context that has the GPU to itself on an image of >= 1.5 M pixels (640x480 scale 3, 1M
 *                  events: 37.2 -> 32.9 us per iteration), the first otherwise.  Bit-identical results. */
int bf_set_option(bf_ctx *ctx, const char *key, int64_t value);

/* Diagnostics (no reference counterpart).  Keys:
 *   "scatter_format"  what the last bf_set_cloud chose for the tile-binned loop: 0 dense slabs, 2 event
 *                     lists, 3 own pixels + margin plane ("bin_split"); -1 when the slice does not take that loop.
 *   "one_kernel"      1 when bf_run would take the one-kernel iteration for the slice staged now, else 0.
 *   "persistent"      1 when bf_run, called now, would run it as the persistent kernel ("persist"), else 0.
 *   "persist_giveups" launches of the persistent kernel on this context that gave up (a work-group waited 0.2 s for others
 *                     that were not resident: something else holds part of the GPU) and undid themselves; the run they
 *                     belonged to went on with one launch per iteration, and the context leaves the kernel alone for the
 *                     next 1, 2, 4 ... 64 runs.
 * BF_ERR_ARG for an unknown key. */
int bf_get_stat(bf_ctx *ctx, const char *key, int64_t *value);

/* ---- slice set-up -------------------------------------------------------------- */

/* AccelLib::init_gpu (accel_lib.h:71-115): stage one slice on the device.  Host
 * arrays are borrowed for the duration of the call.  noise may be NULL (no event
 * is noise, the state DVS_flow hands over).  Also resets the per-event warp state
 * (Event::reset, event.h:54-59: pr <- fr, n <- 0).  BF_ERR_STATE while asynchronous
 * uploads (bf_upload_events_async / bf_upload_ring_async) are pending: commit them first. */
int bf_upload_events(bf_ctx *ctx, const int32_t *fr_x, const int32_t *fr_y, const int32_t *t_ns,
                     const uint8_t *noise, int64_t n);

/* Same (accel_lib.h:71-115 without the blocking enqueueWriteBuffer calls of :109-115), but the three arrays are
 * DEVICE pointers (slice already resident in HBM). */
int bf_upload_events_device(bf_ctx *ctx, const int32_t *d_fr_x, const int32_t *d_fr_y,
                            const int32_t *d_t_ns, int64_t n);

/* Streaming front end (BASELINE config 3): stage the NEXT slice while the current one is being
 * optimised -- the reference uploads with blocking writes on its one queue (accel_lib.h:109-115, CL_TRUE) and
 * re-allocates its staging arrays per slice (:82-89).  bf_host_alloc returns pinned host memory; bf_upload_events_async copies a slice from
 * pinned arrays into one of two device staging slots on the ctx's COPY stream and returns at once
 * (the arrays must stay untouched until the matching bf_commit_upload returns);
 * bf_commit_upload makes the compute stream wait for the oldest pending copy and stages it
 * (== bf_upload_events without the blocking copy).  At most two uploads may be pending. */
int bf_host_alloc(bf_ctx *ctx, int64_t bytes, void **out);
int bf_host_free(bf_ctx *ctx, void *ptr);
int bf_upload_events_async(bf_ctx *ctx, const int32_t *fr_x, const int32_t *fr_y, const int32_t *t_ns,
                           int64_t n);
int bf_commit_upload(bf_ctx *ctx);

/* OptimizerRolling::set_cloud + set_scale (optimizer_rolling.h:248-283): bounding
 * box over fr (device reduction), window geometry, Event::reset for every event.
 * res_x / res_y seed x_min / y_min (:252).  scale must be odd (:274). */
int bf_set_cloud(bf_ctx *ctx, int32_t scale, int32_t res_x, int32_t res_y, bf_window *window_out);

/* ---- AccelLib operators -------------------------------------------------------- */

/* AccelLib::project_4param (accel_lib.h:275-281; Event::project_4param, event.h:88-96; project_dn, event.h:72-76): the
 * INCREMENTAL form -- dn from the previous pr as in the reinit form, ADDED to the event's (nx, ny), then apply_project.  The
 * reference's only call is commented out (optimizer_rolling.h:333-339); exported so that every member of SURVEY 8(b)'s
 * signature list exists.  (nx, ny) start at 0 after bf_set_cloud (Event::reset). */
int bf_project_4param(bf_ctx *ctx, double dnx_, double dny_, double cx, double cy, double div, double crl);

/* AccelLib::project_4param_reinit (accel_lib.h:263-267; Event::project_4param_reinit,
 * event.h:99-110; apply_project, event.h:164-168).  cos/sin of crl are evaluated on
 * the host with libm, as the reference does. */
int bf_project_4param_reinit(bf_ctx *ctx, double dnx_, double dny_, double cx, double cy,
                             double div, double crl);

/* AccelLib::get_time_img (accel_lib.h:211-217 -> get_time_img_cpu :147-178) for the
 * current window; x_sh / y_sh are (int)x_shift, (int)y_shift as at :147.  time_out
 * (R*C floats) and count_out (R*C uint32, the event-count image the reference keeps
 * local at :149) may each be NULL.  The time image stays resident for bf_fast_model. */
int bf_get_time_img(bf_ctx *ctx, float *time_out, uint32_t *count_out);

/* AccelLib::Sobel / Sobel_cpu (accel_lib.h:400-432,513-543) on a host image. */
int bf_sobel(bf_ctx *ctx, const float *img, int32_t rows, int32_t cols, float *grad_x,
             float *grad_y);

/* AccelLib::fast_model (accel_lib.h:337-341) = ObjectModel::update
 * (object_model.h:31-34; object_model.cpp:4-39,103-126).  img == NULL: use the time
 * image left on the device by the last bf_get_time_img.  Sets cx, cy, dx, dy, rot,
 * div, cnt of *model; totals are left untouched. */
int bf_fast_model(bf_ctx *ctx, const float *img, int32_t rows, int32_t cols, bf_model *model);

/* AccelLib::writeout_events (accel_lib.h:310-329): copy per-event state back.  Any
 * pointer may be NULL.  Arrays hold n doubles in upload order. */
int bf_writeout_events(bf_ctx *ctx, double *pr_x, double *pr_y, double *nx, double *ny);

/* Event::compute_uv (event.h:135-142) for every event, from the current nx, ny. */
int bf_compute_uv(bf_ctx *ctx, double *u, double *v);

/* ---- fused optimizer ----------------------------------------------------------- */

/* OptimizerRolling::set_model (optimizer_rolling.h:289-299): warm start ("STM",
 * dvs_flow.h:218-219).  model->cx, cy are SENSOR coordinates (:345-346). */
int bf_set_model(bf_ctx *ctx, const bf_model *model);

/* OptimizerRolling::run (optimizer_rolling.h:48-125) with iteration_step (:305-347)
 * executed entirely on the device: per iteration a warp+scatter kernel and a
 * stencil+moments kernel (the scalar model / loop update rides in one of them), no host round trip except a
 * poll of the `done` word every opts->poll_interval iterations.  The starting model is
 * the zero ObjectModel of a fresh optimizer (after bf_set_cloud) or what bf_set_model
 * was given; model_out receives get_model() (:285-287).  opts == NULL: defaults.
 * Returns info->rc.  When bf_run returns, the model and info are final; the last warp of the events (and, with want_uv, their
 * per-event flow) may still be executing on the context's stream -- everything that reads per-event results (bf_compute_uv,
 * bf_compute_uv_ring, bf_writeout_events, the renderers) and every later operation of the context runs on that stream, behind it.
 *
 * Tolerance contract.  The fused loop is NOT bit-identical to the reference's CPU path, and cannot be: (i) the time
 * image is the exact integer-nanosecond sum of a pixel's events rounded to f32 once, where the reference adds f32 seconds
 * event by event in container order (accel_lib.h:162) -- its own result moves by ~1e-7 relative with the event order;
 * (ii) the moments are centred sums reduced in a fixed tree, where object_model.cpp:22-30 adds per pixel in row-major
 * order; (iii) sin / cos of the rotation angle are evaluated on the device (Taylor polynomials for |x| <= 0.25, <= 1 ulp
 * from libm), where event.h:102-103 calls std::cos / std::sin -- bf_project_4param_reinit, the one-to-one operator,
 * evaluates them on the host with libm like the reference, so the two entry points can differ by that ulp.
 * What holds: the event-count image is bit-exact at fixed warp parameters; the time image agrees to 1e-6 relative, the
 * moments to 1e-9; a trajectory agrees with the CPU restatement to rounding until the first event crosses a pixel
 * boundary differently and within the restatement's own event-order spread afterwards; converged per-event flow
 * agrees to 1e-4 relative or 0.02 px/s, iteration counts to +-1 (tests/test_gpu_parity.py, tests/test_gpu_geometries.py
 * bound all of these at full size).  Every scatter / loop mode of this library gives the same bits as every other, and
 * every run is bit-reproducible (integer accumulators). */
int bf_run(bf_ctx *ctx, const bf_run_opts *opts, bf_model *model_out, bf_run_info *info);

/* A grid of independent optimizers over the staged slice (BASELINE config 4; the reference's
 * queue of (events, model) tasks, dvs_flow.h:200-231, with one task per sensor tile).  Events are
 * bucketed by tile (row = fr_x * grid_rows / sensor_res_x, column likewise); every tile gets its
 * own OptimizerRolling: set_cloud (bounding box seeded with sensor_res, optimizer_rolling.h:252),
 * cold start, run() with the guards evaluated against guard_res_x / guard_res_y / min_events (the
 * reference's RES / 15 and 1000 would skip every small tile).  One work-group per tile runs the
 * whole loop on chip.  Needs bf_upload_events first (not bf_set_cloud).  models_out / infos_out
 * hold grid_rows * grid_cols entries (row-major); infos[i].rc is 0, 1 (skipped) or < 0.
 * Afterwards bf_compute_uv / bf_writeout_events return the per-event results of all tiles.
 * Throughput over many slices: a grid's launch lasts as long as its slowest tile (thousands of
 * iterations, while the mean is ~90), so a caller keeps several contexts' grids in flight, one host
 * thread each -- and starts the process with GPU_MAX_HW_QUEUES=16 (the HIP runtime reads it once, at
 * its first call; default 4): with four hardware queues a fifth grid waits behind another grid's
 * straggler (measured, 32 x 32 tiles over 1M-event slices: 189 Mevents/s with 4 queues, 510 with 16
 * queues and 16 grids in flight). */
typedef struct bf_tile_opts {
    int32_t grid_rows, grid_cols;
    int32_t scale;
    int32_t sensor_res_x, sensor_res_y;
    int32_t guard_res_x, guard_res_y;
    int32_t min_events;
    int32_t max_iter;        /* <= 0: unlimited */
    int32_t hard_iter_cap;
} bf_tile_opts;
int bf_run_tiles(bf_ctx *ctx, const bf_tile_opts *opts, bf_model *models_out, bf_run_info *infos_out);

/* The tile grids of n slices in ONE launch: the reference's queue of (events, model) tasks (dvs_flow.h:200-231) holding the
 * tiles of several slices at once.  ctxs: n DISTINCT contexts of one device, each holding its own uploaded slice; opts: the
 * grid, shared by all.  One resident grid of work-groups on ctxs[0]'s stream claims (slice, tile) pairs from a device
 * counter -- slice by slice, a slice's tiles in index order -- so the straggler tiles of the first slices (a grid alone
 * lasts as long as its slowest tile: thousands of iterations against a mean of ~90) run under the bulk of the later ones.
 * No stream and no hardware queue per slice: the sustained rate does not depend on the host process's GPU_MAX_HW_QUEUES
 * (a grid per context in flight needs 16 queues to pass ~180 Mevents/s; the runtime's default is 4).  Every (slice, tile)
 * is the computation of bf_run_tiles -- the same bits.  models_out / infos_out: n * grid_rows * grid_cols entries, slice
 * major (either may be NULL).  Afterwards every context is in the state bf_run_tiles leaves it in (bf_compute_uv etc.
 * read its per-event results).  Blocking; returns the first error (its text on ctxs[0]). */
int bf_run_tiles_many(bf_ctx *const *ctxs, int32_t n, const bf_tile_opts *opts, bf_model *models_out, bf_run_info *infos_out);

/* A batch of independent slices -- the reference's queue of (events, model) tasks (dvs_flow.h:200-231, executed there
 * one after the other) -- solved together: bf_run on each of the n contexts, all in flight at once (one host thread per
 * context inside the library; the contexts' streams share the GPU, set "co_schedule" on them when n > 1).  Every
 * context must hold its own slice (upload + bf_set_cloud [+ bf_set_model]).  models_out / infos_out: n entries (either
 * may be NULL); infos_out[i].rc is that slice's return code.  Returns BF_OK when every slice returned 0 or 1, else the
 * first error code (the text is on that context: bf_last_error).  For slices spread over several GPUs and for streams
 * of slices, see better_flow/slice_farm.h, which drives the same entry points. */
int bf_run_many(bf_ctx *const *ctxs, int32_t n, const bf_run_opts *opts, bf_model *models_out, bf_run_info *infos_out);

/* Records of the last bf_run (opts->trace_cap > 0): model and dividers after every iteration_step -- what the
 * reference prints under VERBOSE (optimizer_rolling.h:116-118) or shows in manual() (:212).  Returns the number written. */
int bf_get_trace(bf_ctx *ctx, bf_trace_rec *out, int32_t cap, int32_t *written);

/* ---- raw device buffers ----------------------------------------------------------- */

/* EventFile::projection_img(events, scale, show_final) (event_file.h:460-515): the 8-bit image of the
 * non-noise events on the full sensor, (res_x * scale) x (res_y * scale) -- at their current projected
 * positions pr (the motion-compensated image once bf_run has converged) or, with show_final != 0, at their
 * sensor positions.  Saturating count, this build's 8-bit Gaussian for scale > 1 (see above; the reference's
 * cv::GaussianBlur is unpinned), brightness normalised to a non-zero mean of 127 (cv::convertScaleAbs:
 * round-half-even of the float product, saturated).  img_out: host buffer of that many bytes. */
int bf_projection_img(bf_ctx *ctx, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final,
                      uint8_t *img_out);

/* EventFile::color_time_img(events, scale, show_final) (event_file.h:649-747): the colour-coded time image of
 * the non-noise events on the full sensor, (res_x * scale + scale) x (res_y * scale + scale) pixels of B, G, R
 * bytes (3 bytes per pixel, row-major).  Every event adds (1, cos a, sin a) to the scale x scale pixels it
 * covers, a = float(2 * 3.14 * (t - t_min) / (t_max - t_min)) with t_min = min t and t_max = max(0, max t)
 * (:659-662; a = 0 when they are equal -- the reference divides 0 by 0 there).  Per covered pixel: hue =
 * uchar((atan2(mean sin, mean cos) + 3.1416) * 180 / 3.1416 / 2), saturation = uchar(255 * |mean|), value = 255
 * (:712-723); uncovered pixels are black.  Positions are pr (compensated) or, with show_final != 0, the sensor
 * coordinates (:684-687).  scale = 0 means 11 (:650).
 * Two things are this build's own, both stated here: (i) the reference adds the f32 cos / sin in event order,
 * this build adds them as 2^-32 fixed point integers, so the image does not depend on event order and agrees
 * with the f32 sums to their own rounding error (~1e-7) before the 8-bit quantisation; (ii) HSV -> BGR follows the 8-bit convention of
 * cv::cvtColor (H in [0, 180), S and V in [0, 255]) in float32: h = H * (6 / 180), sector = floor(h), f = h -
 * sector, s = S / 255, v = V / 255, p = v (1 - s), q = v (1 - s f), t = v (1 - s (1 - f)), (r, g, b) = (v,t,p),
 * (q,v,p), (p,v,t), (p,q,v), (t,p,v), (v,p,q) for sectors 0..5, each channel rint(255 x) -- the reference's
 * cv::cvtColor comes from an un-versioned OpenCV (parity unpinned for that stage). */
int bf_color_time_img(bf_ctx *ctx, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final,
                      uint8_t *bgr_out);

/* The slice hand-off of DVS_flow::recompute (dvs_flow.h:185-216) for a structure-of-arrays event ring kept
 * in pinned memory (bf_host_alloc) -- no AoS -> SoA repack (accel_lib.h:91-99) and no per-slice allocation on
 * the host.  The slice is the n events starting at ring index `first` (wrapping at `cap`), stored oldest ->
 * newest; Event::set_local_time(t0) (event.h:61-63) is applied on the device.  ring_noise (may be NULL: no event
 * is noise) is the ring of Event::noise flags -- the reference sets them for every event of a slice that its
 * small-window guard rejects (optimizer_rolling.h:49-55) and get_time_img leaves flagged events out of later,
 * overlapping slices (accel_lib.h:152).  Asynchronous like bf_upload_events_async (same two staging slots;
 * bf_commit_upload makes the slice current).  The ring slots may be overwritten once bf_wait_uploads has
 * returned. */
int bf_upload_ring_async(bf_ctx *ctx, const int32_t *ring_fr_x, const int32_t *ring_fr_y,
                         const uint64_t *ring_timestamp_ns, const uint8_t *ring_noise, int64_t cap,
                         int64_t first, int64_t n, uint64_t t0_ns);

/* The same hand-off for a ring with 16-bit addresses: 12 bytes per event over the link instead of 16, and the
 * column layout of the binary event file (better_flow/event_reader.h: u64 t_ns[], u16 x[] = column, u16 y[] = row),
 * so that a file block read into the ring is uploaded as it lies.  ring_row = Event::fr_x, ring_col = Event::fr_y. */
int bf_upload_ring16_async(bf_ctx *ctx, const uint16_t *ring_row, const uint16_t *ring_col,
                           const uint64_t *ring_timestamp_ns, const uint8_t *ring_noise, int64_t cap,
                           int64_t first, int64_t n, uint64_t t0_ns);

/* ... and with 32-bit timestamps: 8 bytes per event over the link instead of 12 -- a warm-started stream (a handful of
 * iterations per slice) is bound by that link: 1M-event slices from pinned memory at 47-51 GB/s are 0.25 ms of copy against
 * 0.1-0.2 ms of solve.  ring_t32[i] holds the LOW 32 bits of event i's nanosecond timestamp; t0_ns is the slice start in full.
 * The device forms Event::set_local_time (event.h:61-63) as (int32)(ring_t32[i] - (uint32)t0_ns), which is the exact
 * difference while |timestamp - t0_ns| < 2^31 ns = 2.1 s (the caller's contract: a slice is at most SPAN long, dvs_flow.h:21,156 -- 0.2 s in the
 * command line, bf_motion_compensator.cpp:7,135; the 64-bit forms report a time that does not fit through bf_set_cloud, this one cannot). */
int bf_upload_ring16t32_async(bf_ctx *ctx, const uint16_t *ring_row, const uint16_t *ring_col,
                              const uint32_t *ring_t32, const uint8_t *ring_noise, int64_t cap,
                              int64_t first, int64_t n, uint64_t t0_ns);
/* bf_upload_events_async for a slice whose addresses are held as 16-bit values (AccelLib::init_gpu's int fr_x, fr_y, t,
 * accel_lib.h:83-85,101-103, with the two addresses narrowed: a sensor address fits 16 bits): 8 bytes per event. */
int bf_upload_events16_async(bf_ctx *ctx, const uint16_t *fr_x, const uint16_t *fr_y, const int32_t *t_ns,
                             int64_t n);

/* Event::compute_uv (event.h:135-142) of the current slice written straight into a ring of interleaved (u, v)
 * pairs: event i of the slice goes to uv_ring[2 * ((first + i) % cap)] and [... + 1].  No intermediate host copy:
 * with a pinned ring this is one or two DMA transfers.  Blocks until the data has arrived. */
int bf_compute_uv_ring(bf_ctx *ctx, double *uv_ring, int64_t cap, int64_t first);

/* Block until the host-to-device copies of every pending asynchronous upload have finished. */
int bf_wait_uploads(bf_ctx *ctx);

/* ---- contrast-score optimiser: OptimizerLocal (optimizer_sampler.h:12-68) ------------------
 * The secondary score of the path: saturating 8-bit event-count image, Gaussian blur, mean of
 * the non-zero pixels, coordinate descent on (nx, ny).  The blur is this build's own stated
 * 8-bit Gaussian (binomial taps {1,2,1}/4, {1,4,6,4,1}/16, {2,7,14,18,14,7,2}/64 for
 * scale 3 / 5 / 7, BORDER_REFLECT_101, exact integer sum, one rounding half up): the
 * reference calls cv::GaussianBlur of an unpinned OpenCV, so parity is unpinned there. */
typedef struct bf_local_window {
    int32_t scale;
    int32_t metric_wsizex, metric_wsizey;   /* optimizer_sampler.h:31-32,43-44 */
    int32_t scale_img_x, scale_img_y;       /* optimizer_sampler.cpp:208-209 */
    int32_t c_fr_x, c_fr_y;                 /* event_c, the window centre (optimizer_sampler.h:30,46) */
    int32_t pad_;
    int64_t c_t;                            /* event_c.t */
} bf_local_window;

typedef struct bf_local_state {
    double nx, ny, last_score, dnx, dny, dn_th;   /* optimizer_sampler.h:26-28 */
    int64_t evaluations;                          /* iteration_step calls */
} bf_local_state;

/* The two constructors (optimizer_sampler.h:29-48) + update_fields (optimizer_sampler.cpp:205-212).
 * wsz <= 0: OptimizerLocal(events, scale) -- window = bounding box of the uploaded cloud, centre
 * event in its middle with t = 0 (c_fr_x, c_fr_y, c_t are ignored).  wsz > 0:
 * OptimizerLocal(events, e, scale, wsz) with e = (c_fr_x, c_fr_y, c_t).  scale odd, <= 7. */
int bf_local_set_window(bf_ctx *ctx, int32_t scale, int32_t wsz, int32_t c_fr_x, int32_t c_fr_y,
                        int64_t c_t, bf_local_window *window_out);

/* OptimizerLocal::iteration_step (optimizer_sampler.cpp:120-153): Event::project(nx, ny) of every
 * event, count image, blur, score.  img_out (optional, host, scale_img_x * scale_img_y bytes)
 * receives project_img.  Does not touch the rolling optimizer's per-event state. */
int bf_local_iteration_step(bf_ctx *ctx, double nx, double ny, double *score, uint8_t *img_out);

/* OptimizerLocal::run (optimizer_sampler.cpp:4-38).  Returns BF_OK, BF_SKIPPED (window guard,
 * :9-13, with the sensor size res_x x res_y) or BF_ERR_NOCONV when max_evaluations (> 0) is
 * reached (the reference has no cap). */
int bf_local_run(bf_ctx *ctx, int32_t res_x, int32_t res_y, int64_t max_evaluations, bf_local_state *out);

/* A GRID of OptimizerLocal windows over the uploaded slice -- SURVEY f1's formulation of BASELINE config 4 (per-tile local flow):
 * the sensor is cut into grid_rows x grid_cols tiles (an event belongs to tile (fr_x * grid_rows / sensor_res_x, fr_y *
 * grid_cols / sensor_res_y), as in bf_run_tiles); every tile gets OptimizerLocal(events of the tile, e, scale, wsz)
 * (optimizer_sampler.h:31-34) with the centre event e = (middle row of the tile, middle column of the tile, t = 0) -- "middle"
 * = (first + last) / 2 of the tile's rows / columns, integer division -- and its run() (optimizer_sampler.cpp:4-38): coordinate
 * descent on (nx, ny) for the largest contrast score.  One work-group per window runs the whole descent on chip.  The window
 * guard (:9-13) uses guard_res_x / guard_res_y as RES; max_evaluations > 0 caps a window's iteration_step calls (checked after
 * each ny step, as bf_local_run does).  states_out / rc_out: grid_rows * grid_cols entries (either may be NULL); rc_out[k] is 0,
 * BF_SKIPPED or BF_ERR_NOCONV.  The blur is this build's stated 8-bit Gaussian (above).  Afterwards the slice's events are
 * sorted by tile (as after bf_run_tiles) and hold no per-event result. */
typedef struct bf_local_tile_opts {
    int32_t grid_rows, grid_cols;
    int32_t scale;                        /* odd, <= 7 */
    int32_t wsz;                          /* window side in sensor pixels: metric_wsize = scale * wsz (optimizer_sampler.h:32) */
    int32_t sensor_res_x, sensor_res_y;   /* sensor rows / columns */
    int32_t guard_res_x, guard_res_y;     /* RES_X / RES_Y of the window guard */
    int64_t max_evaluations;              /* <= 0: unlimited */
} bf_local_tile_opts;
int bf_local_run_tiles(bf_ctx *ctx, const bf_local_tile_opts *opts, bf_local_state *states_out, int32_t *rc_out);

/* NUMA placement of a feeder thread (no reference counterpart: the reference is single-threaded, SURVEY 8(b) "Threading"; the
 * 8-GPU farm of SURVEY 8(e) wants one feeder thread per GPU with NUMA-local pinned buffers).
 *   bf_device_numa_node          host NUMA node of HIP device `device` (sysfs numa_node of its PCI function); -1: unknown.
 *   bf_bind_thread_to_numa_node  restricts the CALLING thread to the CPUs of `node` that the process may use (a container's
 *                                cpuset wins; no such CPUs, no such node or node < 0: nothing happens) and makes that node the
 *                                preferred one for memory the thread touches or pins from now on.  cpus_out (may be NULL): CPUs
 *                                the thread is bound to, 0 when nothing was done.
 *   bf_bind_thread_to_device_numa  the two together; node_out may be NULL.
 * Call it on a worker thread before the thread's first bf_create / bf_host_alloc.  bf::SliceFarm's workers, the lanes of
 * better_flow_amd/farm.py and every rank of bench.py do. */
int bf_device_numa_node(int32_t device, int32_t *node_out);
int bf_bind_thread_to_numa_node(int32_t node, int32_t *cpus_out);
int bf_bind_thread_to_device_numa(int32_t device, int32_t *node_out);

/* For callers that keep slices resident in HBM (bench.py, the streaming front end) and
 * hand them over with bf_upload_events_device.  bf_memcpy_h2d is synchronous.  (The reference's device buffers are
 * private members of AccelLib, accel_lib.h:15-27; no counterpart.) */
int bf_device_malloc(bf_ctx *ctx, int64_t bytes, void **out);
int bf_device_free(bf_ctx *ctx, void *ptr);
int bf_memcpy_h2d(bf_ctx *ctx, void *dst, const void *src, int64_t bytes);

/* ---- measurement ---------------------------------------------------------------- */

/* Per-kernel hipEvent timing (the reference times whole slices with std::clock in its driver,
 * bf_motion_compensator.cpp:158-177, and the minimiser under VERBOSE, optimizer_rolling.h:116-118).  mode 0: off (default).  mode 1: bracket every kernel
 * launch with events on the ctx stream; totals are read with bf_profile_get, which
 * synchronises the stream. */
int bf_profile_enable(bf_ctx *ctx, int32_t mode);
int bf_profile_reset(bf_ctx *ctx);
int bf_profile_get(bf_ctx *ctx, bf_profile *out);

/* Block until everything enqueued on the ctx stream has finished (the reference's calls are all blocking:
 * CL_TRUE reads / writes and queue->finish(), accel_lib.h:109-115,317-321). */
int bf_synchronize(bf_ctx *ctx);

/* No reference counterpart.  Streaming-copy bandwidth probe on this device: copies `bytes` bytes `reps` times
 * with a float4 kernel and returns the best GB/s (read + write counted).  Used by
 * bench.py to report the measured HBM ceiling next to the 8 TB/s nominal peak. */
int bf_copy_bandwidth(bf_ctx *ctx, int64_t bytes, int32_t reps, double *gbps_out);

/* No reference counterpart.  The sine and cosine the device loops use for the warp's rotation angle
 * (Event::project_4param_reinit evaluates std::cos / std::sin, event.h:102-103; bf_run's update computes them on the device:
 * a polynomial for |x| <= 0.25, the device library beyond), evaluated for `n` host arguments.  table != 0: the variant of the
 * persistent loop kernel, coefficients read from an LDS table -- the same operations in the same order.  For measuring the
 * distance to the host's libm (tests/test_gpu_parity.py; the bound is stated in DESIGN.md, "Oracle"). */
int bf_eval_sincos(bf_ctx *ctx, const double *x, int64_t n, int32_t table, double *sin_out, double *cos_out);

#ifdef __cplusplus
}
#endif
#endif /* BF_ACCEL_H */
