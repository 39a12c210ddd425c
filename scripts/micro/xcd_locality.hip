// Microbenchmark: does data written by one kernel come back faster in the NEXT kernel when the reading work-group
// sits on the same XCD as the writer (work-group id mod 8) than when it sits on another one?  Decides whether an
// XCD-aware tile order for the stencil kernel (reading the slabs the scatter kernel just flushed) can pay.
//   writer:  work-group i writes chunk i (plain stores, or write-through agent-scope stores like the slab flush)
//   reader:  work-group i reads chunk (i + shift) mod n with a chain of DEPENDENT loads (latency bound), 1 wave
// Reported: reader kernel time for shift = 0 (same XCD), 1 (next XCD), 8 (same XCD, other CU), plus a cold read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

constexpr int kChunk = 4096;   // u64 words per work-group (32 KB)

template <bool WT>
__global__ void k_write(unsigned long long* buf, unsigned long long salt) {
    unsigned long long* o = buf + (size_t)blockIdx.x * kChunk;
    for (int i = threadIdx.x; i < kChunk; i += blockDim.x) {
        // each word holds the index of the next word to visit (a stride-67 walk inside the chunk)
        const unsigned long long v = (unsigned long long)((i * 67 + 1 + (int)(salt & 1)) % kChunk);
        if (WT) __hip_atomic_store(&o[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else o[i] = v;
    }
}

__global__ void k_chase(const unsigned long long* buf, int shift, int steps, unsigned long long* out, unsigned long long* clk) {
    const int src = (int)((blockIdx.x + (unsigned)shift) % gridDim.x);
    const unsigned long long* p = buf + (size_t)src * kChunk;
    unsigned long long idx = threadIdx.x;   // 64 lanes, 64 independent chains (one wave)
    const unsigned long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) idx = p[idx];
    const unsigned long long t1 = wall_clock64();
    if (idx == 0xdeadbeefull) out[0] = idx;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    const int n = 256, steps = 8;
    unsigned long long *buf, *out, *clk;
    CK(hipMalloc(&buf, (size_t)n * kChunk * 8));
    CK(hipMalloc(&out, 8));
    CK(hipMalloc(&clk, n * 8));
    unsigned long long h[n];
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wt = 0; wt < 2; ++wt)
        for (int shift : {0, 1, 8, 3, 0}) {
            double tot = 0, cyc = 0;
            const int reps = 20;
            for (int r = 0; r < reps; ++r) {
                if (wt) hipLaunchKernelGGL(k_write<true>, dim3(n), dim3(256), 0, 0, buf, (unsigned long long)r);
                else hipLaunchKernelGGL(k_write<false>, dim3(n), dim3(256), 0, 0, buf, (unsigned long long)r);
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_chase, dim3(n), dim3(64), 0, 0, buf, shift, steps, out, clk);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                tot += ms;
                CK(hipMemcpy(h, clk, n * 8, hipMemcpyDeviceToHost));
                double c = 0; for (int i = 0; i < n; ++i) c += (double)h[i];
                cyc += c / n;
            }
            printf("%s stores, reader shift %d: kernel %.2f us, %.0f ns per dependent load (in-kernel clock, 100 MHz)\n",
                   wt ? "write-through" : "plain", shift, 1e3 * tot / reps, cyc / reps * 10.0 / steps);
        }
    // re-read without a writer in between (same kernel twice): L2-resident?
    for (int shift : {0, 1}) {
        hipLaunchKernelGGL(k_chase, dim3(n), dim3(64), 0, 0, buf, shift, steps, out, clk);
        hipLaunchKernelGGL(k_chase, dim3(n), dim3(64), 0, 0, buf, shift, steps, out, clk);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, clk, n * 8, hipMemcpyDeviceToHost));
        double c = 0; for (int i = 0; i < n; ++i) c += (double)h[i];
        printf("second read in a row, shift %d: %.0f ns per dependent load\n", shift, c / n * 10.0 / steps);
    }
    return 0;
}
