// How many 256-thread work-groups does a CU really hold, and how fast are work-groups dispatched?  A kernel whose
// work-groups do nothing but stay alive for a fixed time (wall clock) with a given LDS size: duration / lifetime = rounds,
// work-groups / rounds = resident work-groups.  And with lifetime 0: the dispatcher's own rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/residency.hip -o scripts/micro/residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int LDS>
__global__ __launch_bounds__(256) void k_hold(unsigned long long* out, int ticks /* 100 MHz */) {
    __shared__ unsigned long long s[LDS / 8];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(4);
    if (s[(threadIdx.x + 1) & 255] == 12345ull) out[0] = 1;
}

// the same with a private (scratch) array and a real exit at a per-work-group time: lifetimes 0.75 .. 1.25 x nominal
template <int LDS, bool SCRATCH>
__global__ __launch_bounds__(256) void k_hold2(unsigned long long* out, int ticks, int idx) {
    __shared__ unsigned long long s[LDS / 8];
    volatile unsigned long long priv[8];
    s[threadIdx.x] = threadIdx.x;
    if (SCRATCH) { for (int i = 0; i < 8; ++i) priv[i] = i; }
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    const unsigned long long life = (unsigned long long)ticks * (48 + (blockIdx.x * 7 % 33)) / 64;
    while (wall_clock64() - t0 < life) __builtin_amdgcn_s_sleep(4);
    if (s[(threadIdx.x + 1) & 255] == 12345ull) out[0] = SCRATCH ? priv[idx & 7] : 1;
}
template <int LDS, bool SCRATCH>
static void run2(int nwg, int ticks, unsigned long long* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_hold2<LDS, SCRATCH>), dim3(nwg), dim3(256), 0, 0, d, ticks, r);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 1e3, life = ticks / 100.0;
    printf("uneven lifetimes, scratch %d: LDS %6d B  %5d work-groups  lifetime %5.1f us : %7.1f us -> %.1f resident per CU\n", (int)SCRATCH, LDS, nwg, life, us, nwg * life / (us - 5.0) / 256.0);
}

template <int LDS>
static void run(int nwg, int ticks, unsigned long long* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_hold<LDS>, dim3(nwg), dim3(256), 0, 0, d, ticks);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 1e3, life = ticks / 100.0;
    printf("LDS %6d B  %5d work-groups  lifetime %5.1f us : %7.1f us", LDS, nwg, life, us);
    if (ticks > 0) printf("  -> %.1f rounds, %.1f resident per CU", us / life, nwg / (us / life) / 256.0);
    else printf("  -> %.1f work-groups per us", nwg / us);
    printf("\n");
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    for (int nwg : {833, 6240, 26000}) {
        run<2048>(nwg, 0, d);
        run<16672>(nwg, 0, d);
    }
    for (int ticks : {200, 900}) {
        run<2048>(6240, ticks, d);
        run<16672>(6240, ticks, d);
        run<20480>(6240, ticks, d);
        run<32768>(6240, ticks, d);
    }
    run2<16672, false>(6240, 900, d);
    run2<16672, true>(6240, 900, d);
    run2<16672, false>(833, 900, d);
    run2<16672, true>(833, 900, d);
    return 0;
}
