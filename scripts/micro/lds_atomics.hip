// LDS atomic throughput on gfx950 (what the stencil kernel's entry splat and the scatter kernels' accumulates cost):
// every CU runs WG work-groups of 256 threads; each thread issues N atomic adds to pseudo-random words of a 1188-entry LDS
// array (the stencil tile's box plane).  Prints nanoseconds per wave-instruction per CU and lanes per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_atomics.hip -o scripts/micro/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>   // 0: ds_add_u64 random; 1: ds_add_u32 random; 2: ds_write_b64 random; 3: ds_add_u64 3x3 splat pattern; 4: ds_add_rtn_u64 random
__global__ __launch_bounds__(256) void k(unsigned long long* out, int n, uint32_t seed) {
    __shared__ unsigned long long s[1188];
    for (int i = threadIdx.x; i < 1188; i += 256) s[i] = 0;
    __syncthreads();
    uint32_t r = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    unsigned long long acc = 0;
    for (int i = 0; i < n; ++i) {
        r = r * 1664525u + 1013904223u;
        const uint32_t idx = (r >> 8) % 1188u;
        if (MODE == 0) atomicAdd(&s[idx], (unsigned long long)r);
        else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(s) + idx, r);
        else if (MODE == 2) s[idx] = r;
        else if (MODE == 4) acc += atomicAdd(&s[idx], (unsigned long long)r);
        else {
            const uint32_t base = (r >> 8) % (1188u - 2 * 66 - 2);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) atomicAdd(&s[base + a * 66 + b], (unsigned long long)r);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[5] + acc;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 1 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"ds_add_u64 random", "ds_add_u32 random", "ds_write_b64 random", "ds_add_u64 3x3 splat (9 per step)", "ds_add_rtn_u64 random"};
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
        const int blocks = 256 * wg_per_cu, n = 2000;
        for (int mode = 0; mode < 5; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, n, 7u);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, n, 7u);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, n, 7u);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, n / 9, 7u);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, n, 7u);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double ops_per_cu = (double)wg_per_cu * 256.0 * (mode == 3 ? (n / 9) * 9 : n);   // lane-operations per CU
            printf("%d work-groups per CU  %-36s %8.1f us  -> %6.2f lane-ops per ns per CU (%.2f per clock at 2.4 GHz)\n", wg_per_cu, names[mode],
                   ms * 1e3, ops_per_cu / (ms * 1e6), ops_per_cu / (ms * 1e6) / 2.4);
        }
    }
    return 0;
}
