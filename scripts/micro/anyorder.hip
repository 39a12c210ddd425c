// Does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let the NEXT kernel of a stream become resident while the
// previous one still runs on gfx950?  Kernel A: every work-group spends ~T us, the last one to finish publishes a sequence
// number.  Kernel B: every work-group spins (bounded: 20 ms, then gives up and counts a timeout) until the number is
// published, then spends ~T us.  N iterations of A, B on one stream: (a) both launched normally, B's wait is always over
// when it starts; (b) B launched with hipExtAnyOrderLaunch.  If (b) is shorter per iteration, B's launch overlapped A.
// build: hipcc --offload-arch=gfx950 -O3 anyorder.hip -o anyorder ; run: ./anyorder [T_us] [blocks]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ void burn(long long ticks) {   // 100 MHz clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
__global__ void kA(unsigned* ticket, unsigned* seq, unsigned j, long long ticks, unsigned* early) {
    burn(ticks);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            *ticket = 0;
            __threadfence();
            __hip_atomic_store(seq, j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ void kB(const unsigned* seq, unsigned j, long long ticks, unsigned* timeouts, unsigned* early) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int good = 0, waited = 0;
        for (;;) {
            if (__hip_atomic_load(seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= j + 1) { good = 1; break; }
            waited = 1;
            if (wall_clock64() - t0 > 2000000) break;   // 20 ms
            __builtin_amdgcn_s_sleep(8);
        }
        if (!good) atomicAdd(timeouts, 1u);
        if (waited && blockIdx.x == 0) atomicAdd(early, 1u);   // this launch really started before A had finished
        ok = good;
    }
    __syncthreads();
    burn(ticks);
}
int main(int argc, char** argv) {
    const double T = argc > 1 ? atof(argv[1]) : 8.0;
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    const int N = 500;
    const long long ticks = (long long)(T * 100);
    unsigned *ticket, *seq, *timeouts, *early;
    CK(hipMalloc(&ticket, 4)); CK(hipMalloc(&seq, 4)); CK(hipMalloc(&timeouts, 4)); CK(hipMalloc(&early, 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(ticket, 0, 4)); CK(hipMemset(seq, 0, 4)); CK(hipMemset(timeouts, 0, 4)); CK(hipMemset(early, 0, 4));
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int j = 0; j < N; ++j) {
                hipLaunchKernelGGL(kA, dim3(blocks), dim3(256), 0, s, ticket, seq, (unsigned)j, ticks, early);
                if (mode == 0) hipLaunchKernelGGL(kB, dim3(blocks), dim3(512), 0, s, (const unsigned*)seq, (unsigned)j, ticks, timeouts, early);
                else hipExtLaunchKernelGGL(kB, dim3(blocks), dim3(512), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (const unsigned*)seq, (unsigned)j, ticks, timeouts, early);
            }
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            unsigned to = 0, ea = 0;
            CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ea, early, 4, hipMemcpyDeviceToHost));
            printf("%s: %.2f us per A+B pair (2 x %.1f us of work), B started early in %u of %d launches, timeouts %u\n",
                   mode ? "B any-order" : "B ordered  ", us / N, T, ea, N, to);
        }
    }
    return 0;
}
