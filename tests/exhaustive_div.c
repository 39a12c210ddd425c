/* Exhaustive proof (all 2^32 float bit patterns) that the 3-operation sequences used by the
 * HIP kernels are bit-identical to the IEEE divisions of the reference:
 *   (a)  (double)f / 10000.0        event.h:167-168   (f = the f32 product kx * float(t))
 *   (b)  f / 127.0f                 event.h:164-165   (== (float)((double)f / 127.0), see kernels)
 * Sequence:  q0 = x * R;  r = fma(-q0, d, x);  q = fma(r, R, q0),  R = RN(1 / d).
 * Exactly three inputs differ: -0 (the sequence gives +0) and +-inf (it gives NaN).  The kernels use the bare sequence:
 * the quotient is only ever subtracted from a coordinate, so the sign of a zero is immaterial, and an infinite dividend
 * means a diverged model, whose events are rejected either way (bf_device_fns.h).  Exit status 0 iff every OTHER input
 * agrees bit for bit and those three behave as stated.  Build: gcc -O2 -mfma -fopenmp -ffp-contract=off.  Test-only.
 *   (d)  f / (float)b               accel_lib.h:172   (mean time of a pixel: f32 sum / f32 count, b the event count)
 * as (float)((double)f * RN64(1.0 / b)) -- the stencil kernel's table form (bf_device_fns.h: time_from_sums) -- for EVERY finite
 * f32 f and every integer b in [1, 255]: identical whenever the quotient is a normal f32 (the dividend is a sum of integer
 * nanoseconds in seconds: 0 or >= 1e-9, quotient >= 3.9e-12); the 573 364 mismatches all have subnormal quotients (|f| <= 3.6e-38).
 * Round 5: 9 min 40 s on 8 cores.  Run with any argument to include it (it is 255 x the work of (a) + (b)). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline double div10000(double x) {
    const double R = 1.0 / 10000.0;
    double q0 = x * R;
    double r = fma(-q0, 10000.0, x);
    return fma(r, R, q0);
}
static inline float div127(float x) {
    const float R = 1.0f / 127.0f;
    float q0 = x * R;
    float r = fmaf(-q0, 127.0f, x);
    return fmaf(r, R, q0);
}

/* (c)  (double)ts / 1e9 with ts an integer number of nanoseconds (accel_lib.h:162): the dividend
 * is an integer |ts| < 2^53, too many to enumerate; checked on 2^32 pseudo-random integers over
 * all magnitudes plus every integer in [0, 2^26) (the one-event-per-pixel range of a 67 ms slice). */
static inline double div1e9(double x) {
    const double R = 1.0 / 1000000000.0;
    double q0 = x * R;
    double r = fma(-q0, 1000000000.0, x);
    return fma(r, R, q0);
}

int main(int argc, char **argv) {
    unsigned long long bad_c = 0;
#pragma omp parallel for reduction(+ : bad_c)
    for (long long i = 0; i < (1LL << 32); ++i) {
        unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0xD1342543DE82EF95ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const int sh = (int)(i & 63) % 52;                 /* all magnitudes up to 2^52 */
        long long v = (long long)(z >> 11) >> sh;
        if (i & 64) v = -v;
        double x = (double)v;
        double a = x / 1000000000.0, b = div1e9(x);
        if (memcmp(&a, &b, 8) != 0) bad_c++;
        if (i < (1LL << 26)) {
            x = (double)i;
            a = x / 1000000000.0; b = div1e9(x);
            if (memcmp(&a, &b, 8) != 0) bad_c++;
        }
    }
    printf("div1e9   (f64, integer dividends): %llu mismatches in 2^32 samples + [0, 2^26)\n", bad_c);
    unsigned long long bad_a = 0, bad_b = 0;
    float max_bad_b = 0.f, min_bad_b = INFINITY, max_bad_a = 0.f;
#pragma omp parallel for reduction(+ : bad_a, bad_b) reduction(max : max_bad_b, max_bad_a) reduction(min : min_bad_b)
    for (long long hi = 0; hi < 65536; ++hi) {
        for (uint32_t lo = 0; lo < 65536; ++lo) {
            uint32_t bits = ((uint32_t)hi << 16) | lo;
            float f;
            memcpy(&f, &bits, 4);
            if (isnan(f)) continue;
            if (isinf(f) || (f == 0.0f && signbit(f))) {   /* the three stated exceptions */
                double g = div10000((double)f);
                float h = div127(f);
                if (isinf(f) ? !(isnan(g) && isnan(h)) : !(g == 0.0 && h == 0.0f)) bad_a++;
                continue;
            }
            double x = (double)f;
            double qa = x / 10000.0, ga = div10000(x);
            if (memcmp(&qa, &ga, 8) != 0) {
                bad_a++;
                if (fabsf(f) > max_bad_a) max_bad_a = fabsf(f);
            }
            float qb = f / 127.0f, gb = div127(f);
            if (memcmp(&qb, &gb, 4) != 0 && !(isnan(qb) && isnan(gb))) {
                bad_b++;
                if (fabsf(f) > max_bad_b) max_bad_b = fabsf(f);
                if (fabsf(f) < min_bad_b) min_bad_b = fabsf(f);
            }
        }
    }
    printf("div10000 (f64): %llu mismatches (max |f| %.9g)\n", bad_a, (double)max_bad_a);
    printf("div127   (f32): %llu mismatches (|f| in [%.9g, %.9g])\n", bad_b, (double)min_bad_b, (double)max_bad_b);
    unsigned long long bad_d = 0, bad_d_normal = 0;
    if (argc > 1) {
        float worst = 0.f;
#pragma omp parallel for reduction(+ : bad_d, bad_d_normal) reduction(max : worst) schedule(dynamic, 64)
        for (long long hi = 0; hi < 65536; ++hi) {
            for (uint32_t lo = 0; lo < 65536; ++lo) {
                uint32_t bits = ((uint32_t)hi << 16) | lo;
                float f;
                memcpy(&f, &bits, 4);
                if (isnan(f) || isinf(f)) continue;
                const double fd = (double)f;
                for (int b = 1; b < 256; ++b) {
                    const float q = f / (float)b;
                    const float t = (float)(fd * (1.0 / (double)b));
                    if (memcmp(&q, &t, 4) != 0) {
                        bad_d++;
                        if (fabsf(q) >= 1.17549435e-38f) bad_d_normal++;
                        if (fabsf(f) > worst) worst = fabsf(f);
                    }
                }
            }
        }
        printf("f / count as (double)f * RN64(1 / count), count 1..255: %llu mismatches, %llu with a normal quotient (largest |f| %.9g)\n",
               bad_d, bad_d_normal, (double)worst);
    }
    return (bad_a || bad_b || bad_c || bad_d_normal) ? 1 : 0;
}
