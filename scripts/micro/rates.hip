// Issue cost of the vector instructions the loop kernels are made of (gfx950): one wave per SIMD, 8 independent
// chains per lane, clock64() around 8 x 512 instructions.  Prints cycles per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/micro/rates.hip -o scripts/micro/rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define BODY(NAME, DECL, OP)                                                                  \
    __global__ void k_##NAME(long long* out, double seed) {                                   \
        DECL;                                                                                 \
        const long long t0 = clock64();                                                       \
        for (int i = 0; i < 512; ++i) { OP; }                                                 \
        const long long t1 = clock64();                                                       \
        if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&out[1100], (unsigned long long)t0); atomicMax((unsigned long long*)&out[1101], (unsigned long long)t1); } \
        out[1 + threadIdx.x] = (long long)SINK;                                               \
    }

#define D8 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7
#define F8 float a0 = (float)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7
#define U8 uint32_t a0 = (uint32_t)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7
#define ALL(X) X(a0); X(a1); X(a2); X(a3); X(a4); X(a5); X(a6); X(a7)

#define SINK (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define ADD64(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(seed))
#define MUL64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(seed))
#define FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(seed))
#define ADD32(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(fs))
#define FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(fs))
#define MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(us))
#define MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(us))
#define MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(us))
#define ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(us))
#define CNDM(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(us))
BODY(add_f64, D8, ALL(ADD64))
BODY(mul_f64, D8, ALL(MUL64))
BODY(fma_f64, D8, ALL(FMA64))
BODY(add_f32, F8; float fs = (float)seed, ALL(ADD32))
BODY(fma_f32, F8; float fs = (float)seed, ALL(FMA32))
BODY(mul_lo_u32, U8; uint32_t us = (uint32_t)seed | 3, ALL(MULLO))
BODY(mul_hi_u32, U8; uint32_t us = (uint32_t)seed | 3, ALL(MULHI))
BODY(mul_u32_u24, U8; uint32_t us = (uint32_t)seed | 3, ALL(MUL24))
BODY(add_u32, U8; uint32_t us = (uint32_t)seed | 3, ALL(ADDU))
BODY(cndmask, U8; uint32_t us = (uint32_t)seed | 3, ALL(CNDM))
#undef SINK
// conversions: chains through a second register of the other type
#define SINK (double)(b0 + b1 + b2 + b3)
#define CVT_F64_F32(x, y) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(y))
#define CVT_F32_F64(y, x) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(y) : "v"(x))
__global__ void k_cvt_f64_f32_and_back(long long* out, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    float b0 = 1, b1 = 2, b2 = 3, b3 = 4;
    const long long t0 = clock64();
    for (int i = 0; i < 512; ++i) {
        CVT_F32_F64(b0, a0); CVT_F32_F64(b1, a1); CVT_F32_F64(b2, a2); CVT_F32_F64(b3, a3);
        CVT_F64_F32(a0, b0); CVT_F64_F32(a1, b1); CVT_F64_F32(a2, b2); CVT_F64_F32(a3, b3);
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&out[1100], (unsigned long long)t0); atomicMax((unsigned long long*)&out[1101], (unsigned long long)t1); }
    out[1 + threadIdx.x] = (long long)SINK;
}
__global__ void k_cvt_i32_f64_and_back(long long* out, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    int b0 = 1, b1 = 2, b2 = 3, b3 = 4;
    const long long t0 = clock64();
    for (int i = 0; i < 512; ++i) {
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b0) : "v"(a0)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b1) : "v"(a1));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b2) : "v"(a2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b3) : "v"(a3));
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a0) : "v"(b0)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a1) : "v"(b1));
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a2) : "v"(b2)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a3) : "v"(b3));
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&out[1100], (unsigned long long)t0); atomicMax((unsigned long long*)&out[1101], (unsigned long long)t1); }
    out[1 + threadIdx.x] = (long long)SINK;
}
__global__ void k_mad_i64_i32(long long* out, double seed) {
    long long a0 = (long long)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    int b0 = threadIdx.x | 1, b1 = 7;
    const long long t0 = clock64();
    for (int i = 0; i < 512; ++i) {
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(b0), "v"(b1) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(b0), "v"(b1) : "vcc");
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(b0), "v"(b1) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&out[1100], (unsigned long long)t0); atomicMax((unsigned long long*)&out[1101], (unsigned long long)t1); }
    out[1 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ void k_cmp_class_f64(long long* out, double seed) {
    double a0 = seed, a1 = seed + 1;
    int m = 0x207;
    const long long t0 = clock64();
    for (int i = 0; i < 512; ++i) {
        asm volatile("v_cmp_class_f64 vcc, %0, %1" :: "v"(a0), "v"(m) : "vcc"); asm volatile("v_cmp_class_f64 vcc, %0, %1" :: "v"(a1), "v"(m) : "vcc");
        asm volatile("v_cmp_class_f64 vcc, %0, %1" :: "v"(a0), "v"(m) : "vcc"); asm volatile("v_cmp_class_f64 vcc, %0, %1" :: "v"(a1), "v"(m) : "vcc");
        asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a0), "v"(a1) : "vcc"); asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a1), "v"(a0) : "vcc");
        asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a0), "v"(a1) : "vcc"); asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a1), "v"(a0) : "vcc");
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&out[1100], (unsigned long long)t0); atomicMax((unsigned long long*)&out[1101], (unsigned long long)t1); }
    out[1 + threadIdx.x] = (long long)(a0 + a1);
}

template <class K> static double once(K k, int threads, long long* d) {
    long long init[2] = {0x7fffffffffffffffll, 0}, h[2];
    (void)hipMemcpy(d + 1100, init, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, 1.25);
    (void)hipMemcpy(h, d + 1100, 16, hipMemcpyDeviceToHost);
    return (double)(h[1] - h[0]);
}
template <class K> static void run(const char* name, K k, int per_iter, long long* d) {
    once(k, 64, d);
    const double one = once(k, 64, d) / (512.0 * per_iter);
    const double four = once(k, 1024, d) / (512.0 * per_iter * 4);   // 16 waves = 4 per SIMD, first start to last end
    std::printf("%-28s one wave: %5.2f cycles / instruction;  4 waves per SIMD: %5.2f cycles / wave-instruction per SIMD\n", name, one, four);
}

int main() {
    long long* d;
    (void)hipMalloc(&d, 8 * 1200);
    run("v_add_f64", k_add_f64, 8, d); run("v_mul_f64", k_mul_f64, 8, d); run("v_fma_f64", k_fma_f64, 8, d);
    run("v_add_f32", k_add_f32, 8, d); run("v_fma_f32", k_fma_f32, 8, d);
    run("v_add_u32", k_add_u32, 8, d); run("v_mul_u32_u24", k_mul_u32_u24, 8, d);
    run("v_mul_lo_u32", k_mul_lo_u32, 8, d); run("v_mul_hi_u32", k_mul_hi_u32, 8, d);
    run("v_cndmask_b32", k_cndmask, 8, d);
    run("v_cvt f64<->f32 (pair avg)", k_cvt_f64_f32_and_back, 8, d);
    run("v_cvt f64<->i32 (pair avg)", k_cvt_i32_f64_and_back, 8, d);
    run("v_mad_i64_i32", k_mad_i64_i32, 4, d);
    run("v_cmp_class/lt_f64", k_cmp_class_f64, 8, d);
    return 0;
}
