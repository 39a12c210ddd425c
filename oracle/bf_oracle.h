/*
 * bf_oracle.h -- CPU restatement of better-flow's motion-compensation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * there only as the checker / the timed CPU baseline.  The product path
 * (better_flow_amd/) never links, imports or falls back to this code.
 *
 * PARITY UNPINNED.  The reference has no tests, golden vectors or fixtures
 *  and it cannot be compiled in this image: every header on the
 * path includes OpenCV (common.h:17-19) and accel_lib.h includes TBB (accel_lib.h:7-8);
 * neither is installed, and building it against stand-in headers is not allowed.
 * This file is therefore a line-by-line restatement of the reference arithmetic,
 * each function citing the reference lines it follows, checked by review and by
 * analytic / known-answer properties -- not by outputs of the reference itself.
 *
 * All paths below are relative to /root/reference/better_flow_core/ :
 *   event.h            = include/better_flow/event.h
 *   accel_lib.h        = include/better_flow/accel_lib.h
 *   optimizer_rolling.h= include/better_flow/optimizer_rolling.h
 *   object_model.h/.cpp= include/better_flow/object_model.h, src/object_model.cpp
 *
 * Conventions: fr_x is the ROW, fr_y the COLUMN (bf_motion_compensator.cpp:192,200
 * swaps the file's x/y).  Images are row-major R x C float, R = wsx + scale.
 */
#ifndef BF_ORACLE_H
#define BF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Event cloud, structure-of-arrays view of the live fields of `class Event`
 * (event.h:9-24).  All arrays have length n and are owned by the caller. */
typedef struct {
    int64_t n;
    const int32_t *fr_x;   /* row    (uint fr_x, event.h:9)  */
    const int32_t *fr_y;   /* column (uint fr_y, event.h:9)  */
    const int64_t *t;      /* ns relative to slice start (sll t, event.h:10,61-63) */
    uint8_t *noise;        /* bool noise (event.h:12) */
    double *pr_x, *pr_y;   /* projected position (event.h:15) */
    double *nx, *ny;       /* direction vector (event.h:16); nz == 127 (common.h:60) */
} bfo_cloud;

/* Window geometry computed by OptimizerRolling::set_cloud / set_scale
 * (optimizer_rolling.h:248-283). */
typedef struct {
    int32_t scale;
    int32_t x_min, y_min, x_max, y_max;
    int32_t metric_wsizex, metric_wsizey;   /* scale * (max - min)           */
    int32_t scale_img_x, scale_img_y;       /* R, C = metric_wsize + scale   */
    double x_shift, y_shift;
} bfo_window;

/* ObjectModel (object_model.h:10-13). */
typedef struct {
    double cx, cy, dx, dy, rot, div;
    uint32_t cnt;
    double total_dx, total_dy, total_rot, total_div;
} bfo_model;

/* Loop state of OptimizerRolling::run (optimizer_rolling.h:36,61-63). */
typedef struct {
    float x_divider, y_divider, rot_divider, div_divider;
    int64_t itercount;
} bfo_loop;

/* One record per iteration_step, for trajectory comparisons. */
typedef struct {
    bfo_model model;       /* after update_accumulators, cx/cy in sensor coords */
    bfo_loop loop;         /* dividers AFTER the sign-flip update of this pass  */
} bfo_trace_rec;

void bfo_model_init(bfo_model *m);

/* event.h:61-63 */
void bfo_set_local_time(const uint64_t *timestamp, int64_t n, uint64_t t0, int64_t *t_out);

/* optimizer_rolling.h:248-283 (+ Event::reset, event.h:54-59).  res_x/res_y are the
 * compile-time RES_X/RES_Y (common.h:39-40) made runtime. */
void bfo_set_cloud(bfo_cloud *ev, int32_t scale, int32_t res_x, int32_t res_y, bfo_window *w);

/* accel_lib.h:263-267 -> event.h:99-110,164-168 */
void bfo_project_4param(bfo_cloud *ev, double dnx_, double dny_, double cx, double cy, double div, double crl);
void bfo_project_4param_reinit(bfo_cloud *ev, double dnx_, double dny_, double cx, double cy,
                               double div, double crl);

/* accel_lib.h:147-178.  time_img and cnt_img are (w+scale) x (h+scale) floats.
 * cnt_img may be NULL.  x_sh / y_sh are the int-truncated shifts. */
void bfo_get_time_img(const bfo_cloud *ev, int32_t w, int32_t h, int32_t scale, int32_t x_sh,
                      int32_t y_sh, float *time_img, float *cnt_img);

/* object_model.cpp:103-126 */
void bfo_center_of_mass(const float *img, int32_t rows, int32_t cols, bfo_model *m);

/* accel_lib.h:513-615 */
void bfo_sobel(const float *img, int32_t rows, int32_t cols, float *grad_x, float *grad_y);

/* object_model.cpp:4-39 (grad planes are scratch of size rows*cols, may be NULL) */
void bfo_model_compute(const float *img, int32_t rows, int32_t cols, bfo_model *m,
                       float *grad_x, float *grad_y);

/* accel_lib.h:337-341 (CPU branch) -> object_model.h:31-34 */
void bfo_fast_model(const float *img, int32_t rows, int32_t cols, bfo_model *m);

/* optimizer_rolling.h:305-347 ; scratch = 4 planes of R*C floats or NULL */
void bfo_iteration_step(bfo_cloud *ev, const bfo_window *w, bfo_model *m, const bfo_loop *lp,
                        float *scratch);

/* optimizer_rolling.h:289-299 */
void bfo_set_model(bfo_cloud *ev, bfo_model *m, const bfo_model *last);

/* optimizer_rolling.h:48-125.  Returns 0 (optimised), 1 (skipped by a guard), or
 * -2 when hard_iter_cap (>0) was hit (the reference would spin forever, e.g. on a
 * NaN model).  min_events is the reference's literal 1000 (:57) made runtime.
 * trace (may be NULL) receives up to trace_cap records. */
int bfo_run(bfo_cloud *ev, const bfo_window *w, bfo_model *m, int32_t max_itercount,
            int32_t res_x, int32_t res_y, int32_t min_events, int64_t hard_iter_cap,
            bfo_loop *loop_out, bfo_trace_rec *trace, int64_t trace_cap);

/* event.h:135-142 */
void bfo_compute_uv(const double *nx, const double *ny, int64_t n, double *u, double *v);

/* std::sin / std::cos of event.h:102-103 on this host's libm */
void bfo_sincos(const double *x, int64_t n, double *sn, double *cs);


/* ---- Contrast-score optimiser: OptimizerLocal (optimizer_sampler.h:12-68, optimizer_sampler.cpp) ----
 * PARITY UNPINNED for the blur stage: the reference calls cv::GaussianBlur(CV_8UC1, ksize = scale,
 * sigma = 0) from an un-vendored, un-versioned OpenCV; bfo_gauss_u8 below is this build's own stated
 * 8-bit Gaussian (see its comment).  Everything before the blur restates the reference exactly. */
typedef struct {
    int32_t scale;
    int32_t metric_wsizex, metric_wsizey;   /* optimizer_sampler.h:31-32,43-44 */
    int32_t scale_img_x, scale_img_y;       /* :208-209 */
    int32_t c_fr_x, c_fr_y;                 /* event_c (the window centre), :30,46 */
    int64_t c_t;                            /* event_c.t (0 for the whole-cloud constructor) */
} bfo_local_window;

/* OptimizerLocal(events, scale) (optimizer_sampler.h:35-48): window = bounding box of the cloud
 * (LinearEventCloud x_min.. seeded INT_MAX / INT_MIN, datastructures.h:127,143-147). */
void bfo_local_window_cloud(const bfo_cloud *ev, int32_t scale, bfo_local_window *w);
/* OptimizerLocal(events, e, scale, wsz) (optimizer_sampler.h:29-33). */
void bfo_local_window_at(int32_t scale, int32_t wsz, int32_t c_fr_x, int32_t c_fr_y, int64_t c_t,
                         bfo_local_window *w);

/* Event::project for every event (event.h:65-70,164-168): absolute projection from fr. */
void bfo_project(bfo_cloud *ev, double nx_, double ny_);

/* The saturating u8 count image of iteration_step (optimizer_sampler.cpp:120-146), before the blur. */
void bfo_local_count_img(bfo_cloud *ev, const bfo_local_window *w, double nx_, double ny_, uint8_t *img);

/* This build's 8-bit Gaussian, in place, ksize in {1, 3, 5, 7}: the binomial taps OpenCV tabulates for
 * sigma <= 0 ({1,2,1}/4, {1,4,6,4,1}/16, {2,7,14,18,14,7,2}/64), separable, exact integer arithmetic,
 * border BORDER_REFLECT_101, one final rounding (half up).  Returns -1 for any other ksize. */
int bfo_gauss_u8(uint8_t *img, int32_t rows, int32_t cols, int32_t ksize, uint8_t *scratch);

/* get_event_score (optimizer_sampler.cpp:192-205): mean of the non-zero pixels. */
double bfo_nonzero_average(const uint8_t *img, int64_t n);

/* iteration_step (optimizer_sampler.cpp:120-153): project, count image, blur (scale > 1), score.
 * img: scale_img_x * scale_img_y bytes, receives project_img. */
double bfo_local_iteration_step(bfo_cloud *ev, const bfo_local_window *w, double nx_, double ny_, uint8_t *img,
                                uint8_t *scratch);

typedef struct {
    double nx, ny, last_score, dnx, dny, dn_th;
    int64_t evaluations;
} bfo_local_state;

/* run (optimizer_sampler.cpp:4-38).  Returns 0, or 1 when the window guard skips the cloud (:9-13);
 * -2 if max_evaluations (>0) was reached (the reference has no cap). */
int bfo_local_run(bfo_cloud *ev, const bfo_local_window *w, int32_t res_x, int32_t res_y, int64_t max_evaluations,
                  bfo_local_state *out, uint8_t *img, uint8_t *scratch);


/* ---- Motion-compensated event image: EventFile::projection_img (event_file.h:460-515) ----
 * Saturating 8-bit count image of the non-noise events at pr * scale (or fr * scale when show_final) on the
 * full sensor (res_x * scale) x (res_y * scale), Gaussian blur for scale > 1 (bfo_gauss_u8: PARITY UNPINNED,
 * see above), then cv::convertScaleAbs(img, img, 127.0 / nonzero_average, 0), restated as
 * saturate_u8(round-half-even((float)v * (float)alpha)) -- OpenCV's 8-bit path computes in float and rounds
 * with cvRound.  The min_t / max_t window of the reference (:463-478) is not used by its callers with
 * non-default values and is not restated.  img: res_x*scale x res_y*scale bytes; scratch same size. */
void bfo_projection_img(const bfo_cloud *ev, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final,
                        uint8_t *img, uint8_t *scratch);

/* ---- Colour-coded time image: EventFile::color_time_img (event_file.h:649-747) ----
 * (res_x*scale + scale) x (res_y*scale + scale) pixels of B, G, R bytes.  Restated as written: f32 running sums
 * of cos / sin of the f32 phase in EVENT ORDER (the unqualified cos / sin / hypot / atan2 calls are taken as the
 * double overloads), f32 means, hue / saturation through double -> uchar truncation, value 255.  t_max == t_min
 * (0 / 0 in the reference) is taken as phase 0.  The last step, cv::cvtColor(HSV2BGR), is third-party
 * arithmetic of an un-versioned OpenCV: PARITY UNPINNED; restated as the float formula given in
 * include/bf_accel.h (bf_color_time_img).  scratch: 3 floats per pixel. */
void bfo_color_time_img(const bfo_cloud *ev, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final,
                        uint8_t *bgr, float *scratch);
void bfo_hsv_to_bgr_u8(int32_t H, int32_t S, int32_t V, uint8_t *bgr);

#ifdef __cplusplus
}
#endif
#endif
