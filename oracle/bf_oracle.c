/*
 * bf_oracle.c -- CPU restatement of better-flow's motion-compensation hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see bf_oracle.h.  PARITY UNPINNED: the reference
 * cannot be built in this image (needs OpenCV + TBB headers) and ships no tests or
 * golden vectors, so every function below is a restatement of the cited reference
 * lines, not something validated against the reference's own output.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off (oracle/Makefile).  The reference is
 * built for baseline x86-64 without FMA (better_flow_core/CMakeLists.txt:4,10), so no
 * operation here may be contracted; every mixed float/double expression is written
 * with the conversions C++ performs implicitly in the reference made explicit.
 */
#include "bf_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BFO_NZ 127.0 /* common.h:60, Event::nz is double (event.h:16) */

/* `int x = <double expr>;` on x86-64 is cvttsd2si: truncation toward zero, and the
 * "integer indefinite" value INT_MIN for NaN or out-of-range inputs.  Out-of-range
 * conversion is UB in C, so the x86 behaviour the reference binary has is spelled
 * out here (it only matters for NaN / absurd models; such events are then rejected
 * by the bounds test at accel_lib.h:157). */
static int32_t trunc_to_int_x86(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

void bfo_model_init(bfo_model *m) { /* object_model.h:15-17 */
    memset(m, 0, sizeof(*m));
}

/* Event::set_local_time, event.h:61-63:
 *   t = (timestamp > t_) ? timestamp - t_ : -sll(t_ - timestamp);               */
void bfo_set_local_time(const uint64_t *timestamp, int64_t n, uint64_t t0, int64_t *t_out) {
    for (int64_t i = 0; i < n; ++i)
        t_out[i] = (timestamp[i] > t0) ? (int64_t)(timestamp[i] - t0)
                                       : -(int64_t)(t0 - timestamp[i]);
}

/* OptimizerRolling::set_cloud + set_scale, optimizer_rolling.h:248-283, and
 * Event::reset, event.h:54-59 (pr <- fr, n <- 0). */
void bfo_set_cloud(bfo_cloud *ev, int32_t scale, int32_t res_x, int32_t res_y, bfo_window *w) {
    w->scale = scale;
    w->x_min = res_x; w->y_min = res_y;            /* :252 */
    w->x_max = 0;     w->y_max = 0;                /* :253 */
    for (int64_t i = 0; i < ev->n; ++i) {          /* :255-261 */
        int32_t fx = ev->fr_x[i], fy = ev->fr_y[i];
        if (fx > w->x_max) w->x_max = fx;
        if (fy > w->y_max) w->y_max = fy;
        if (fx < w->x_min) w->x_min = fx;
        if (fy < w->y_min) w->y_min = fy;
        ev->pr_x[i] = (double)fx;                  /* reset(), event.h:55-57 */
        ev->pr_y[i] = (double)fy;
        ev->nx[i] = 0.0;
        ev->ny[i] = 0.0;
    }
    w->metric_wsizex = scale * (w->x_max - w->x_min);   /* :263 */
    w->metric_wsizey = scale * (w->y_max - w->y_min);   /* :264 */
    /* set_scale, :272-283.  (x_max - x_min) / 2 and scale / 2 are INTEGER divisions. */
    w->scale_img_x = w->metric_wsizex + scale;
    w->scale_img_y = w->metric_wsizey + scale;
    w->x_shift = -(double)((w->x_max - w->x_min) / 2 + w->x_min) * (double)scale
                 + (double)w->metric_wsizex / 2.0 + (double)(scale / 2);
    w->y_shift = -(double)((w->y_max - w->y_min) / 2 + w->y_min) * (double)scale
                 + (double)w->metric_wsizey / 2.0 + (double)(scale / 2);
}

/* AccelLib::project_4param (accel_lib.h:275-281) applying Event::project_4param (event.h:88-96): the dn of the reinit form,
 * ADDED to the event's (nx, ny) by project_dn (event.h:72-76), then apply_project (event.h:164-168). */
void bfo_project_4param(bfo_cloud *ev, double dnx_, double dny_, double cx, double cy, double div, double crl) {
    for (int64_t i = 0; i < ev->n; ++i) {
        double rx = ev->pr_x[i] - cx;                      /* event.h:89 */
        double ry = ev->pr_y[i] - cy;
        double qx = cos(crl) * rx - sin(crl) * ry;         /* :91-92 */
        double qy = sin(crl) * rx + cos(crl) * ry;
        double dnx = (-qx) * div + (qx - rx);              /* :94 */
        double dny = (-qy) * div + (qy - ry);
        ev->nx[i] += dnx + dnx_;                           /* :95 -> project_dn, :73-74 */
        ev->ny[i] += dny + dny_;
        float kx = (float)((double)(float)ev->nx[i] / BFO_NZ);   /* apply_project, :164-168 */
        float ky = (float)((double)(float)ev->ny[i] / BFO_NZ);
        float ft = (float)ev->t[i];
        float px = kx * ft;
        float py = ky * ft;
        ev->pr_x[i] = (double)(float)ev->fr_x[i] - (double)px / 10000.0;
        ev->pr_y[i] = (double)(float)ev->fr_y[i] - (double)py / 10000.0;
    }
}

/* AccelLib::project_4param_reinit (accel_lib.h:263-267) applying
 * Event::project_4param_reinit (event.h:99-110) and Event::apply_project
 * (event.h:164-168) to every event, noise or not. */
void bfo_project_4param_reinit(bfo_cloud *ev, double dnx_, double dny_, double cx, double cy,
                               double div, double crl) {
    for (int64_t i = 0; i < ev->n; ++i) {
        /* event.h:100  cv::Point2d r(pr_x - cx, pr_y - cy);  -- the PREVIOUS pr */
        double rx = ev->pr_x[i] - cx;
        double ry = ev->pr_y[i] - cy;
        /* event.h:102-103 */
        double qx = cos(crl) * rx - sin(crl) * ry;
        double qy = sin(crl) * rx + cos(crl) * ry;
        /* event.h:105  dn = - r_ * div + (r_ - r)  (unary minus, then * double, then +) */
        double dnx = (-qx) * div + (qx - rx);
        double dny = (-qy) * div + (qy - ry);
        /* event.h:107-108 */
        double nx = dnx + dnx_;
        double ny = dny + dny_;
        ev->nx[i] = nx;
        ev->ny[i] = ny;
        /* event.h:164-165  float kx = float(nx) / nz;   (nz is double 127) */
        float kx = (float)((double)(float)nx / BFO_NZ);
        float ky = (float)((double)(float)ny / BFO_NZ);
        /* event.h:167-168  pr_x = float(fr_x) - kx * float(t) / 10000.0;
         * float*float is a float product; the divide and subtract are double. */
        float ft = (float)ev->t[i];
        float px = kx * ft;
        float py = ky * ft;
        ev->pr_x[i] = (double)(float)ev->fr_x[i] - (double)px / 10000.0;
        ev->pr_y[i] = (double)(float)ev->fr_y[i] - (double)py / 10000.0;
    }
}

/* AccelLib::get_time_img_cpu, accel_lib.h:147-178.  Two zeroed float planes of
 * (w+scale) x (h+scale); per non-noise event an s x s splat of t/1e9 and of 1;
 * then avg /= cnt where cnt >= 1.  The reference returns only avg; cnt (the
 * "event-count image") is exposed here because north_star pins it bit-exact. */
void bfo_get_time_img(const bfo_cloud *ev, int32_t w, int32_t h, int32_t scale, int32_t x_sh,
                      int32_t y_sh, float *time_img, float *cnt_img) {
    const int32_t R = w + scale, C = h + scale;
    const size_t P = (size_t)R * (size_t)C;
    float *cnt = cnt_img ? cnt_img : (float *)malloc(P * sizeof(float));
    memset(time_img, 0, P * sizeof(float));        /* :148 */
    memset(cnt, 0, P * sizeof(float));             /* :149 */

    for (int64_t i = 0; i < ev->n; ++i) {          /* :151 container order */
        if (ev->noise && ev->noise[i]) continue;   /* :152 */
        /* :154-155  int x = e.pr_x * scale + x_sh;  (double * int + int -> double -> int) */
        int32_t x = trunc_to_int_x86(ev->pr_x[i] * (double)scale + (double)x_sh);
        int32_t y = trunc_to_int_x86(ev->pr_y[i] * (double)scale + (double)y_sh);
        /* :157-158 */
        if ((x >= w + scale / 2) || (x < scale / 2) || (y >= h + scale / 2) || (y < scale / 2))
            continue;
        for (int32_t jx = x - scale / 2; jx <= x + scale / 2; ++jx) {       /* :160 */
            for (int32_t jy = y - scale / 2; jy <= y + scale / 2; ++jy) {   /* :161 */
                size_t k = (size_t)jx * (size_t)C + (size_t)jy;
                /* :162  float += double  ==  (float)((double)avg + double(t)/1e9) */
                time_img[k] = (float)((double)time_img[k] + (double)ev->t[i] / 1000000000.0);
                cnt[k] = cnt[k] + 1.0f;                                     /* :163 */
            }
        }
    }
    for (int32_t jx = 0; jx < R; ++jx) {           /* :168-175 (tbb rows; disjoint) */
        for (int32_t jy = 0; jy < C; ++jy) {
            size_t k = (size_t)jx * (size_t)C + (size_t)jy;
            if (cnt[k] < 1.0f) continue;
            time_img[k] = time_img[k] / cnt[k];
        }
    }
    if (!cnt_img) free(cnt);
}

/* ObjectModel::center_of_mass, object_model.cpp:103-126.  `p[j] > 0.000001`
 * compares a float against a double literal. */
void bfo_center_of_mass(const float *img, int32_t rows, int32_t cols, bfo_model *m) {
    m->cx = 0; m->cy = 0; m->cnt = 0;
    for (int32_t i = 0; i < rows; ++i) {
        const float *p = img + (size_t)i * (size_t)cols;
        for (int32_t j = 0; j < cols; ++j) {
            if ((double)p[j] > 0.000001) {
                m->cx += (double)i;
                m->cy += (double)j;
                m->cnt++;
            }
        }
    }
    /* assert(cnt > 0) is compiled out (-DNDEBUG); cnt == 0 gives 0/0 = NaN. */
    m->cx /= (double)m->cnt;
    m->cy /= (double)m->cnt;
}

/* AccelLib::sobel_point, accel_lib.h:545-615, for the pixel at (row, col).
 * The reference calls sobel_point(img, j = col, i = row, ...) (:536) and reads
 * img.at<float>(l + row - 1, k + col - 1) with k outer, l inner, idx = 3k + l
 * (:594-604).  mask_* / *_norm (:548-591) are computed but never used. */
static int sobel_point(const float *img, int32_t cols, int32_t row, int32_t col, float *dx,
                       float *dy) {
    static const int sharr_x[9] = {3, 0, -3, 10, 0, -10, 3, 0, -3};   /* :546 */
    static const int sharr_y[9] = {3, 10, 3, 0, 0, 0, -3, -10, -3};   /* :547 */
    int idx = 0;
    float ax = 0.0f, ay = 0.0f;
    for (int k = 0; k < 3; ++k) {
        for (int l = 0; l < 3; ++l) {
            float val = img[(size_t)(l + row - 1) * (size_t)cols + (size_t)(k + col - 1)];
            if ((double)val <= 0.000001) return 0;                    /* :599 */
            ax = ax + val * (float)sharr_x[idx];                      /* :601 */
            ay = ay + val * (float)sharr_y[idx];                      /* :602 */
            idx++;
        }
    }
    *dx = ax;
    *dy = ay;
    return 1;
}

/* AccelLib::Sobel_cpu, accel_lib.h:513-543: zero planes; interior pixels with a
 * valid centre get the gated 3x3 Scharr response. */
void bfo_sobel(const float *img, int32_t rows, int32_t cols, float *grad_x, float *grad_y) {
    const size_t P = (size_t)rows * (size_t)cols;
    memset(grad_x, 0, P * sizeof(float));          /* :522 */
    memset(grad_y, 0, P * sizeof(float));          /* :523 */
    for (int32_t i = 1; i < rows - 1; ++i) {       /* :528 */
        const float *p = img + (size_t)i * (size_t)cols;
        for (int32_t j = 1; j < cols - 1; ++j) {   /* :533 */
            if ((double)p[j] <= 0.000001) continue;   /* :534 */
            float dx = 0, dy = 0;
            if (sobel_point(img, cols, i, j, &dx, &dy)) {
                grad_x[(size_t)i * (size_t)cols + (size_t)j] = dx;
                grad_y[(size_t)i * (size_t)cols + (size_t)j] = dy;
            }
        }
    }
}

/* ObjectModel::compute(cv::Mat&), object_model.cpp:4-39.  cross / ddot are
 * cv::Point2d::cross = x*pt.y - y*pt.x and ddot = x*pt.x + y*pt.y in double. */
void bfo_model_compute(const float *img, int32_t rows, int32_t cols, bfo_model *m,
                       float *grad_x, float *grad_y) {
    const size_t P = (size_t)rows * (size_t)cols;
    float *gx = grad_x ? grad_x : (float *)malloc(P * sizeof(float));
    float *gy = grad_y ? grad_y : (float *)malloc(P * sizeof(float));
    bfo_sobel(img, rows, cols, gx, gy);            /* :6 */
    m->dx = 0; m->dy = 0; m->rot = 0; m->div = 0; m->cnt = 0;   /* :8-12 */
    for (int32_t i = 0; i < rows; ++i) {           /* :17 */
        const float *px = gx + (size_t)i * (size_t)cols;
        const float *py = gy + (size_t)i * (size_t)cols;
        const float *p = img + (size_t)i * (size_t)cols;
        for (int32_t j = 0; j < cols; ++j) {
            if ((double)p[j] > 0.000001) {         /* :22 */
                double rx = (double)i - m->cx;     /* :23 */
                double ry = (double)j - m->cy;
                double gxd = (double)px[j];        /* :24 */
                double gyd = (double)py[j];
                m->dx += gxd;                      /* :26 */
                m->dy += gyd;                      /* :27 */
                m->rot += rx * gyd - ry * gxd;     /* :28 r.cross(g) */
                m->div += rx * gxd + ry * gyd;     /* :29 r.ddot(g)  */
                m->cnt++;
            }
        }
    }
    m->rot /= (double)m->cnt;                      /* :35-38 */
    m->div /= (double)m->cnt;
    m->dx /= (double)m->cnt;
    m->dy /= (double)m->cnt;
    if (!grad_x) free(gx);
    if (!grad_y) free(gy);
}

/* AccelLib::fast_model CPU branch (accel_lib.h:337-341) -> ObjectModel::update
 * (object_model.h:31-34): center_of_mass then compute. */
void bfo_fast_model(const float *img, int32_t rows, int32_t cols, bfo_model *m) {
    bfo_center_of_mass(img, rows, cols, m);
    bfo_model_compute(img, rows, cols, m, NULL, NULL);
}

/* OptimizerRolling::iteration_step, optimizer_rolling.h:305-347. */
void bfo_iteration_step(bfo_cloud *ev, const bfo_window *w, bfo_model *m, const bfo_loop *lp,
                        float *scratch) {
    const int32_t R = w->scale_img_x, C = w->scale_img_y;
    const size_t P = (size_t)R * (size_t)C;
    float *buf = scratch ? scratch : (float *)malloc(4 * P * sizeof(float));
    float *time_img = buf, *cnt = buf + P, *gx = buf + 2 * P, *gy = buf + 3 * P;

    /* :322-324  the double shifts are passed to `int x_sh, int y_sh` parameters:
     * implicit double -> int conversion truncates toward zero (accel_lib.h:147). */
    bfo_get_time_img(ev, w->metric_wsizex, w->metric_wsizey, w->scale,
                     trunc_to_int_x86(w->x_shift), trunc_to_int_x86(w->y_shift), time_img, cnt);
    /* :327 fast_model -> update */
    bfo_center_of_mass(time_img, R, C, m);
    bfo_model_compute(time_img, R, C, m, gx, gy);
    /* :328 update_accumulators(rot_divider, div_divider, x_divider, y_divider),
     * object_model.h:48-53 (double / float -> double) */
    m->total_rot += m->rot / (double)lp->rot_divider;
    m->total_div += m->div / (double)lp->div_divider;
    m->total_dx += m->dx / (double)lp->x_divider;
    m->total_dy += m->dy / (double)lp->y_divider;
    /* :330-331  image -> sensor coordinates with the DOUBLE shift */
    double cx = (m->cx - w->x_shift) / (double)w->scale;
    double cy = (m->cy - w->y_shift) / (double)w->scale;
    /* :340-344 */
    bfo_project_4param_reinit(ev, -m->total_dx, -m->total_dy, cx, cy, m->total_div,
                              -m->total_rot);
    m->cx = cx;                                    /* :345-346 */
    m->cy = cy;
    if (!scratch) free(buf);
}

/* OptimizerRolling::set_model, optimizer_rolling.h:289-299 (warm start, "STM"). */
void bfo_set_model(bfo_cloud *ev, bfo_model *m, const bfo_model *last) {
    *m = *last;
    bfo_project_4param_reinit(ev, -m->total_dx, -m->total_dy, m->cx, m->cy, m->total_div,
                              -m->total_rot);
}

/* OptimizerRolling::run, optimizer_rolling.h:48-125. */
int bfo_run(bfo_cloud *ev, const bfo_window *w, bfo_model *m, int32_t max_itercount,
            int32_t res_x, int32_t res_y, int32_t min_events, int64_t hard_iter_cap,
            bfo_loop *loop_out, bfo_trace_rec *trace, int64_t trace_cap) {
    bfo_loop lp;
    lp.x_divider = lp.y_divider = 1.0f;
    lp.rot_divider = lp.div_divider = 10000.0f;
    lp.itercount = 0;
    if (loop_out) *loop_out = lp;
    /* :49-55  integer arithmetic scale * RES / 15 */
    if ((w->scale_img_x < w->scale * res_x / 15) && (w->scale_img_y < w->scale * res_y / 15)) {
        if (ev->noise)
            for (int64_t i = 0; i < ev->n; ++i) ev->noise[i] = 1;
        return 1;
    }
    if (ev->n < (int64_t)min_events) return 1;     /* :57-58 */

    const size_t P = (size_t)w->scale_img_x * (size_t)w->scale_img_y;
    float *scratch = (float *)malloc(4 * P * sizeof(float));
    int rc = 0;

    bfo_iteration_step(ev, w, m, &lp, scratch);    /* :73 */
    lp.itercount++;
    if (trace && lp.itercount <= trace_cap) {
        trace[lp.itercount - 1].model = *m;
        trace[lp.itercount - 1].loop = lp;
    }
    while (lp.x_divider < 32 * 10 || lp.y_divider < 32 * 10 || lp.rot_divider < 32 * 1000 ||
           lp.div_divider < 32 * 1000) {           /* :76-79 */
        /* :81-84  double / float -> double */
        if (fabs(m->dx / (double)lp.x_divider) < 1e-5 &&
            fabs(m->dy / (double)lp.y_divider) < 1e-5 &&
            fabs(m->rot / (double)lp.rot_divider) < 1e-4 &&
            fabs(m->div / (double)lp.div_divider) < 1e-1)
            break;
        float old_dx = (float)m->dx;               /* :86-89 */
        float old_dy = (float)m->dy;
        float old_rot = (float)m->rot;
        float old_div = (float)m->div;

        bfo_iteration_step(ev, w, m, &lp, scratch);   /* :91 */
        lp.itercount++;
        if (max_itercount > 0 && (int)lp.itercount > max_itercount) {   /* :94-96 */
            if (trace && lp.itercount <= trace_cap) {
                trace[lp.itercount - 1].model = *m;
                trace[lp.itercount - 1].loop = lp;
            }
            break;
        }
        /* :98-101  double * float -> double; float dividers *= 2 */
        if (m->dx * (double)old_dx < 0) lp.x_divider *= 2;
        if (m->dy * (double)old_dy < 0) lp.y_divider *= 2;
        if (m->rot * (double)old_rot < 0) lp.rot_divider *= 2;
        if (m->div * (double)old_div < 0) lp.div_divider *= 2;
        if (trace && lp.itercount <= trace_cap) {
            trace[lp.itercount - 1].model = *m;
            trace[lp.itercount - 1].loop = lp;
        }
        if (hard_iter_cap > 0 && lp.itercount >= hard_iter_cap) {
            rc = -2;                               /* not in the reference: it would spin */
            break;
        }
    }
    free(scratch);
    if (loop_out) *loop_out = lp;
    return rc;                                     /* :124 */
}

/* Event::compute_uv, event.h:135-142.  1000000000 / (T_DIVIDER * 10000) is the
 * INTEGER 100000 (common.h:64), nz / 100000 the double 0.00127. */
void bfo_compute_uv(const double *nx, const double *ny, int64_t n, double *u, double *v) {
    for (int64_t i = 0; i < n; ++i) {
        double xy_len = hypot(nx[i], ny[i]);
        double speed = xy_len / (BFO_NZ / (double)(1000000000 / (1 * 10000)));
        u[i] = (xy_len == 0) ? 0 : speed * nx[i] / xy_len;
        v[i] = (xy_len == 0) ? 0 : speed * ny[i] / xy_len;
    }
}

/* The std::cos / std::sin of Event::project_4param_reinit (event.h:102-103) as this host's libm evaluates them, for `n`
 * arguments: the yardstick of tests/test_gpu_parity.py for the device loops' own sine / cosine. */
void bfo_sincos(const double *x, int64_t n, double *sn, double *cs) {
    for (int64_t i = 0; i < n; ++i) {
        sn[i] = sin(x[i]);
        cs[i] = cos(x[i]);
    }
}


/* ===================== OptimizerLocal: the contrast-score optimiser ===================== */

void bfo_local_window_cloud(const bfo_cloud *ev, int32_t scale, bfo_local_window *w) {
    /* datastructures.h:127,143-147: min seeded INT_MAX, max INT_MIN */
    int32_t x_min = INT32_MAX, y_min = INT32_MAX, x_max = INT32_MIN, y_max = INT32_MIN;
    for (int64_t i = 0; i < ev->n; ++i) {
        if (ev->fr_x[i] > x_max) x_max = ev->fr_x[i];
        if (ev->fr_y[i] > y_max) y_max = ev->fr_y[i];
        if (ev->fr_x[i] < x_min) x_min = ev->fr_x[i];
        if (ev->fr_y[i] < y_min) y_min = ev->fr_y[i];
    }
    w->scale = scale;
    w->metric_wsizex = scale * (x_max - x_min);   /* optimizer_sampler.h:43 */
    w->metric_wsizey = scale * (y_max - y_min);   /* :44 */
    w->c_fr_x = (x_max - x_min) / 2 + x_min;      /* :46 */
    w->c_fr_y = (y_max - y_min) / 2 + y_min;
    w->c_t = 0;
    w->scale_img_x = w->metric_wsizex + scale;    /* optimizer_sampler.cpp:208-209 */
    w->scale_img_y = w->metric_wsizey + scale;
}

void bfo_local_window_at(int32_t scale, int32_t wsz, int32_t c_fr_x, int32_t c_fr_y, int64_t c_t,
                         bfo_local_window *w) {
    w->scale = scale;
    w->metric_wsizex = scale * wsz;   /* optimizer_sampler.h:32 */
    w->metric_wsizey = scale * wsz;
    w->c_fr_x = c_fr_x;
    w->c_fr_y = c_fr_y;
    w->c_t = c_t;
    w->scale_img_x = w->metric_wsizex + scale;
    w->scale_img_y = w->metric_wsizey + scale;
}

/* Event::project -> apply_project (event.h:65-70,164-168) for one event. */
static void project_one(int32_t fr_x, int32_t fr_y, int64_t t, double nx_, double ny_, double *pr_x, double *pr_y) {
    float kx = (float)((double)(float)nx_ / BFO_NZ);
    float ky = (float)((double)(float)ny_ / BFO_NZ);
    float ft = (float)t;
    float px = kx * ft;
    float py = ky * ft;
    *pr_x = (double)(float)fr_x - (double)px / 10000.0;
    *pr_y = (double)(float)fr_y - (double)py / 10000.0;
}

void bfo_project(bfo_cloud *ev, double nx_, double ny_) {
    for (int64_t i = 0; i < ev->n; ++i) {
        ev->nx[i] = nx_;
        ev->ny[i] = ny_;
        project_one(ev->fr_x[i], ev->fr_y[i], ev->t[i], nx_, ny_, &ev->pr_x[i], &ev->pr_y[i]);
    }
}

void bfo_local_count_img(bfo_cloud *ev, const bfo_local_window *w, double nx_, double ny_, uint8_t *img) {
    const int32_t s = w->scale, C = w->scale_img_y;
    bfo_project(ev, nx_, ny_);                                  /* :121 */
    double cpx, cpy;
    project_one(w->c_fr_x, w->c_fr_y, w->c_t, nx_, ny_, &cpx, &cpy);   /* :122 */
    memset(img, 0, (size_t)w->scale_img_x * (size_t)w->scale_img_y);   /* :124 */
    const double x_shift = -cpx * s + (double)w->metric_wsizex / 2.0;  /* :126 */
    const double y_shift = -cpy * s + (double)w->metric_wsizey / 2.0;  /* :127 */
    for (int64_t i = 0; i < ev->n; ++i) {
        int32_t x = trunc_to_int_x86(ev->pr_x[i] * s + x_shift);  /* :130 */
        int32_t y = trunc_to_int_x86(ev->pr_y[i] * s + y_shift);
        if ((x >= w->metric_wsizex) || (x < 0) || (y >= w->metric_wsizey) || (y < 0)) continue;   /* :133 */
        x += s / 2;
        y += s / 2;
        for (int32_t jx = x - s / 2; jx <= x + s / 2; ++jx)
            for (int32_t jy = y - s / 2; jy <= y + s / 2; ++jy)
                if (img[(size_t)jx * C + jy] < 255) img[(size_t)jx * C + jy]++;   /* :141-143 */
    }
}

static int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

int bfo_gauss_u8(uint8_t *img, int32_t rows, int32_t cols, int32_t ksize, uint8_t *scratch) {
    static const int k3[3] = {1, 2, 1}, k5[5] = {1, 4, 6, 4, 1}, k7[7] = {2, 7, 14, 18, 14, 7, 2};
    const int *k;
    int norm;   /* sum of the taps of one pass */
    if (ksize == 1) return 0;
    if (ksize == 3) { k = k3; norm = 4; }
    else if (ksize == 5) { k = k5; norm = 16; }
    else if (ksize == 7) { k = k7; norm = 64; }
    else return -1;
    const int h = ksize / 2;
    /* two-dimensional tap sum in exact integers, one rounding: (sum + norm^2 / 2) / norm^2 */
    const int n2 = norm * norm;
    memcpy(scratch, img, (size_t)rows * (size_t)cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            int acc = 0;
            for (int a = -h; a <= h; ++a) {
                const int rr = reflect101(r + a, rows);
                int row = 0;
                for (int b = -h; b <= h; ++b) row += k[b + h] * scratch[(size_t)rr * cols + reflect101(c + b, cols)];
                acc += k[a + h] * row;
            }
            img[(size_t)r * cols + c] = (uint8_t)((acc + n2 / 2) / n2);
        }
    return 0;
}

double bfo_nonzero_average(const uint8_t *img, int64_t n) {
    double nz_avg = 0;
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (img[i] == 0) continue;
        cnt++;
        nz_avg += img[i];
    }
    return (cnt == 0) ? 0 : nz_avg / (double)cnt;
}

double bfo_local_iteration_step(bfo_cloud *ev, const bfo_local_window *w, double nx_, double ny_, uint8_t *img,
                                uint8_t *scratch) {
    bfo_local_count_img(ev, w, nx_, ny_, img);
    if (w->scale > 1) bfo_gauss_u8(img, w->scale_img_x, w->scale_img_y, w->scale, scratch);   /* :148-150 */
    return bfo_nonzero_average(img, (int64_t)w->scale_img_x * w->scale_img_y);
}

int bfo_local_run(bfo_cloud *ev, const bfo_local_window *w, int32_t res_x, int32_t res_y, int64_t max_evaluations,
                  bfo_local_state *st, uint8_t *img, uint8_t *scratch) {
    st->nx = 0; st->ny = 0;                       /* :5 */
    st->last_score = 0;
    double dscore = 0;
    st->dnx = 0.01; st->dny = 0.01;               /* :7 */
    /* (NZ * T_DIVIDER * 1000.0) / (10 * scale * FROM_MS(MAX_TIME_MS)); FROM_MS(100) = ull(1e8) */
    st->dn_th = (127 * 1 * 1000.0) / (double)(10ull * (unsigned long long)w->scale * 100000000ull);
    st->evaluations = 0;
    if ((w->scale_img_x < w->scale * res_x / 15) && (w->scale_img_y < w->scale * res_y / 15)) return 1;   /* :9-13 */
    st->last_score = bfo_local_iteration_step(ev, w, st->nx, st->ny, img, scratch);   /* :16 */
    st->evaluations = 1;
    while (hypot(st->dnx, st->dny) > st->dn_th) {  /* :20 */
        {   /* compute_new_nx, :90-102 */
            const double nx_new = st->nx + st->dnx;
            const double new_score = bfo_local_iteration_step(ev, w, nx_new, st->ny, img, scratch);
            dscore = new_score - st->last_score;
            st->last_score = new_score;
            if (dscore <= 0) st->dnx = -st->dnx / 2.0;
            st->nx = nx_new;
        }
        {   /* compute_new_ny, :105-117 */
            const double ny_new = st->ny + st->dny;
            const double new_score = bfo_local_iteration_step(ev, w, st->nx, ny_new, img, scratch);
            dscore = new_score - st->last_score;
            st->last_score = new_score;
            if (dscore <= 0) st->dny = -st->dny / 2.0;
            st->ny = ny_new;
        }
        st->evaluations += 2;
        if (max_evaluations > 0 && st->evaluations >= max_evaluations) return -2;
    }
    return 0;
}


/* ===================== EventFile::projection_img (event_file.h:460-515) ===================== */
void bfo_projection_img(const bfo_cloud *ev, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final,
                        uint8_t *img, uint8_t *scratch) {
    const int32_t R = res_x * scale, C = res_y * scale;   /* :463-464 */
    memset(img, 0, (size_t)R * (size_t)C);
    for (int64_t i = 0; i < ev->n; ++i) {
        if (ev->noise[i]) continue;                        /* :472 */
        int32_t x = trunc_to_int_x86(ev->pr_x[i] * scale); /* :482-483 */
        int32_t y = trunc_to_int_x86(ev->pr_y[i] * scale);
        if (show_final) {                                  /* :485-488 */
            x = ev->fr_x[i] * scale;
            y = ev->fr_y[i] * scale;
        }
        if ((x >= scale * (res_x - 1)) || (x < 0) || (y >= scale * (res_y - 1)) || (y < 0)) continue;   /* :490 */
        x += scale / 2;
        y += scale / 2;
        const int32_t lx = x - scale / 2 > 0 ? x - scale / 2 : 0, ly = y - scale / 2 > 0 ? y - scale / 2 : 0;   /* :498 */
        const int32_t rx = x + scale / 2 < R ? x + scale / 2 : R, ry = y + scale / 2 < C ? y + scale / 2 : C;   /* :499 */
        for (int32_t jx = lx; jx <= rx; ++jx)
            for (int32_t jy = ly; jy <= ry; ++jy)
                if (img[(size_t)jx * C + jy] < 255) img[(size_t)jx * C + jy]++;   /* :502-503 */
    }
    if (scale > 1) bfo_gauss_u8(img, R, C, scale, scratch);   /* :508-510 */
    const double avg = bfo_nonzero_average(img, (int64_t)R * C);
    const double img_scale = 127.0 / avg;                      /* :512; avg == 0 -> +inf, 0 * inf = NaN -> 0 below */
    const float a = (float)img_scale;
    for (size_t k = 0; k < (size_t)R * (size_t)C; ++k) {
        const float v = fabsf((float)img[k] * a);
        int r = (v != v) ? 0 : (v >= 255.0f ? 255 : (int)lrintf(v));   /* cvRound + saturate_cast<uchar>; NaN -> 0 */
        img[k] = (uint8_t)r;
    }
}

/* ---- EventFile::color_time_img (event_file.h:649-747) ---- */
static uint8_t unit_to_u8(float x) {
    const float v = x * 255.0f;
    return (uint8_t)(v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (int)lrintf(v)));
}

void bfo_hsv_to_bgr_u8(int32_t H, int32_t S, int32_t V, uint8_t *bgr) {
    static const int map[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const float s = (float)S * (1.0f / 255.0f), v = (float)V * (1.0f / 255.0f);
    float h = (float)H * (6.0f / 180.0f);
    int sector = (int)floorf(h);
    h -= (float)sector;
    sector = ((sector % 6) + 6) % 6;
    float tab[4];
    tab[0] = v;
    tab[1] = v * (1.0f - s);
    tab[2] = v * (1.0f - s * h);
    tab[3] = v * (1.0f - s * (1.0f - h));
    bgr[0] = unit_to_u8(tab[map[sector][0]]);
    bgr[1] = unit_to_u8(tab[map[sector][1]]);
    bgr[2] = unit_to_u8(tab[map[sector][2]]);
}

void bfo_color_time_img(const bfo_cloud *ev, int32_t scale, int32_t res_x, int32_t res_y, int32_t show_final,
                        uint8_t *bgr, float *scratch) {
    if (scale == 0) scale = 11;                                         /* :650 */
    uint64_t t_min = (uint64_t)LLONG_MAX, t_max = 0;                    /* :652 */
    for (int64_t i = 0; i < ev->n; ++i) {                               /* :661-664 */
        if (ev->t[i] < (int64_t)t_min) t_min = (uint64_t)ev->t[i];
        if (ev->t[i] > (int64_t)t_max) t_max = (uint64_t)ev->t[i];
    }
    const int32_t mx = scale * res_x, my = scale * res_y;               /* :666-676 with the fixed full-sensor box */
    const int32_t R = mx + scale, C = my + scale;
    const size_t px = (size_t)R * (size_t)C;
    float *sum_c = scratch, *sum_s = scratch + px, *cnt = scratch + 2 * px;
    for (size_t k = 0; k < 3 * px; ++k) scratch[k] = 0.0f;
    const double x_shift = -(double)(res_x / 2) * (double)scale + (double)mx / 2.0;   /* :677-678 */
    const double y_shift = -(double)(res_y / 2) * (double)scale + (double)my / 2.0;
    for (int64_t i = 0; i < ev->n; ++i) {
        if (ev->noise[i]) continue;                                     /* :679 */
        int32_t x = trunc_to_int_x86(ev->pr_x[i] * scale + x_shift);    /* :681-682 */
        int32_t y = trunc_to_int_x86(ev->pr_y[i] * scale + y_shift);
        if (show_final) {                                               /* :684-687, unsigned product */
            x = trunc_to_int_x86((double)((uint32_t)ev->fr_x[i] * (uint32_t)scale) + x_shift);
            y = trunc_to_int_x86((double)((uint32_t)ev->fr_y[i] * (uint32_t)scale) + y_shift);
        }
        if ((x >= mx) || (x < 0) || (y >= my) || (y < 0)) continue;     /* :689-692 */
        const uint64_t num = (uint64_t)ev->t[i] - t_min, den = t_max - t_min;   /* unsigned arithmetic, :694 */
        const double ratio = den != 0 ? (double)num / (double)den : 0.0;
        const float angle = (float)(2 * 3.14 * ratio);
        x += scale / 2;                                                 /* :696-697 */
        y += scale / 2;
        for (int32_t jx = x - scale / 2; jx <= x + scale / 2; ++jx)     /* :699-705 */
            for (int32_t jy = y - scale / 2; jy <= y + scale / 2; ++jy) {
                const size_t at = (size_t)jx * C + jy;
                sum_c[at] = (float)((double)sum_c[at] + cos((double)angle));
                sum_s[at] = (float)((double)sum_s[at] + sin((double)angle));
                cnt[at] = cnt[at] + 1;
            }
    }
    for (size_t k = 0; k < px; ++k) {                                   /* :708-725 */
        uint8_t *o = bgr + 3 * k;
        if (cnt[k] < 1) { o[0] = o[1] = o[2] = 0; continue; }            /* HSV (0,0,0) -> black */
        const float vx = sum_c[k] / cnt[k], vy = sum_s[k] / cnt[k];
        const double speed = hypot((double)vx, (double)vy);
        double angle = 0;
        if (speed != 0) angle = (atan2((double)vy, (double)vx) + 3.1416) * 180 / 3.1416;
        const int32_t H = (int32_t)(unsigned char)trunc_to_int_x86(angle / 2);
        const int32_t S = (int32_t)(unsigned char)trunc_to_int_x86(speed * 255);
        bfo_hsv_to_bgr_u8(H, S, 255, o);
    }
}
