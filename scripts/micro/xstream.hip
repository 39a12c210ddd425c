// Cost of a dependent kernel chain A(k) -> B(k) -> A(k+1) on gfx950 under three schedules:
//   1. one stream (in-order queue, the loop of today: scatter -> stencil -> scatter ...)
//   2. two streams, every hop an event (hipEventRecord / hipStreamWaitEvent)
//   3. B(k) launched EARLY on a second stream and spinning on a device flag that A(k) sets when it ends (B's launch ramp
//      and prologue overlap A); A(k+1) waits for B(k) with an event
// Each kernel busy-waits ~T us on one work-group per CU.   hipcc --offload-arch=gfx950 -O2 xstream.hip -o xstream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_work(const int* wait_flag, int wait_val, int* set_flag, int set_val, long long ticks, int* sink) {
    if (wait_flag) {
        while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_val) __builtin_amdgcn_s_sleep(2);
    }
    const long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) acc += 1;
    if (acc == -1) *sink = acc;
    if (set_flag) {
        __syncthreads();
        if (threadIdx.x == 0) {
            // last work-group to finish publishes (a ticket)
            const int t = atomicAdd(set_flag + 1, 1);
            if (t == (int)gridDim.x - 1) { set_flag[1] = 0; __hip_atomic_store(set_flag, set_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        }
    }
}
int main() {
    const int N = 400, WG = 256, TH = 256;
    const long long ticks = 800;   // wall_clock64: 100 MHz -> 8 us
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    int *flags, *sink;
    hipMalloc(&flags, 64); hipMalloc(&sink, 4); hipMemset(flags, 0, 64);
    std::vector<hipEvent_t> ev(2 * N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return 1e6 * std::chrono::duration<double>(b - a).count(); };
    for (int rep = 0; rep < 2; ++rep) {
        // 1: one stream
        hipDeviceSynchronize();
        auto t0 = now();
        for (int k = 0; k < N; ++k) {
            hipLaunchKernelGGL(k_work, dim3(WG), dim3(TH), 0, s1, nullptr, 0, nullptr, 0, ticks, sink);
            hipLaunchKernelGGL(k_work, dim3(WG), dim3(TH), 0, s1, nullptr, 0, nullptr, 0, ticks, sink);
        }
        hipStreamSynchronize(s1);
        const double one = us(t0, now()) / N;
        // 2: two streams, events
        hipDeviceSynchronize();
        t0 = now();
        for (int k = 0; k < N; ++k) {
            if (k) hipStreamWaitEvent(s1, ev[2 * k - 1], 0);
            hipLaunchKernelGGL(k_work, dim3(WG), dim3(TH), 0, s1, nullptr, 0, nullptr, 0, ticks, sink);
            hipEventRecord(ev[2 * k], s1);
            hipStreamWaitEvent(s2, ev[2 * k], 0);
            hipLaunchKernelGGL(k_work, dim3(WG), dim3(TH), 0, s2, nullptr, 0, nullptr, 0, ticks, sink);
            hipEventRecord(ev[2 * k + 1], s2);
        }
        hipStreamSynchronize(s1); hipStreamSynchronize(s2);
        const double two = us(t0, now()) / N;
        // 3: B early on stream 2, spinning on A's flag; A(k+1) waits for B(k) by event
        hipMemset(flags, 0, 64);
        hipDeviceSynchronize();
        t0 = now();
        for (int k = 0; k < N; ++k) {
            if (k) hipStreamWaitEvent(s1, ev[2 * k - 1], 0);
            hipLaunchKernelGGL(k_work, dim3(WG), dim3(TH), 0, s1, nullptr, 0, flags, k + 1, ticks, sink);          // A(k) sets flag = k + 1
            hipLaunchKernelGGL(k_work, dim3(WG), dim3(TH), 0, s2, flags, k + 1, nullptr, 0, ticks, sink);          // B(k) spins for it
            hipEventRecord(ev[2 * k + 1], s2);
        }
        hipStreamSynchronize(s1); hipStreamSynchronize(s2);
        const double three = us(t0, now()) / N;
        printf("per A+B pair (2 x %.0f us of work): one stream %.2f us | two streams + events %.2f us | early launch + flag %.2f us\n",
               ticks / 100.0, one, two, three);
    }
    return 0;
}
