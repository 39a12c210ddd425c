// Microbenchmark: cost of 1M 64-bit no-return global atomics on gfx950 by address pattern,
// plus LDS-atomic + dense flush variants.  Decides the scatter design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ void k_atomic(unsigned long long* plane, const uint32_t* idx, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&plane[idx[i]], 0x100000001ull);
}
__global__ void k_atomic32(uint32_t* plane, const uint32_t* idx, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&plane[idx[i]], 1u);
}
__global__ void k_store(unsigned long long* plane, const uint32_t* idx, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) plane[idx[i]] = 0x100000001ull;
}
// workgroup-scope atomics (executed in the XCD's L2?)
__global__ void k_atomic_wg(unsigned long long* plane, const uint32_t* idx, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __hip_atomic_fetch_add(&plane[idx[i]], 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS accumulate (random within a tile) then dense flush with plain stores / atomics
template <int MODE>
__global__ void k_lds(unsigned long long* out, const uint32_t* idx, int per_block, int tile) {
    extern __shared__ unsigned long long s[];
    for (int i = threadIdx.x; i < tile; i += blockDim.x) s[i] = 0;
    __syncthreads();
    const uint32_t* my = idx + (size_t)blockIdx.x * per_block;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) atomicAdd(&s[my[i] % tile], 0x100000001ull);
    __syncthreads();
    unsigned long long* o = out + (size_t)blockIdx.x * tile;
    for (int i = threadIdx.x; i < tile; i += blockDim.x) {
        unsigned long long v = s[i];
        if (MODE == 0) o[i] = v;
        else if (MODE == 1) { if (v) atomicAdd(&out[i], v); }      // all blocks -> same dense region
        else { atomicAdd(&o[i], v); }                               // dense atomics, private region
    }
}

int main() {
    const int N = 1 << 20, P = 780 * 1038;
    std::vector<uint32_t> h(N);
    unsigned long long* plane; uint32_t* idx; uint32_t* plane32;
    CK(hipMalloc(&plane, (size_t)64 << 20)); CK(hipMalloc(&plane32, (size_t)64 << 20)); CK(hipMalloc(&idx, N * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto fill = [&](int mode) {
        uint64_t s = 88172645463325252ull;
        for (int i = 0; i < N; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            uint32_t r = (uint32_t)(s >> 20);
            switch (mode) {
                case 0: h[i] = r % P; break;                         // random over the image
                case 1: h[i] = i % P; break;                         // perfectly coalesced
                case 2: h[i] = r % 8192; break;                      // random in 64 KB
                case 3: h[i] = ((i / 64) * 1031 % (P / 64)) * 64 + (r % 64); break;  // per-wave: random within a 512 B window
                case 4: h[i] = ((i / 64) * 1031 % (P / 512)) * 512 + (r % 512); break; // per-wave: random in 4 KB
                case 5: h[i] = ((i / 4096) * 37 % (P / 16384)) * 16384 + (r % 16384); break; // per-block(4096): random in 128 KB
            }
        }
        CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice));
    };
    const char* names[] = {"random-image", "coalesced", "random-64KB", "wave-in-512B", "wave-in-4KB", "block-in-128KB"};
    for (int mode = 0; mode < 6; ++mode) {
        fill(mode);
        float best[4] = {1e9, 1e9, 1e9, 1e9};
        for (int rep = 0; rep < 6; ++rep) {
            float ms;
            CK(hipEventRecord(a)); hipLaunchKernelGGL(k_atomic, dim3(N / 256), dim3(256), 0, 0, plane, idx, N); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b)); if (ms < best[0]) best[0] = ms;
            CK(hipEventRecord(a)); hipLaunchKernelGGL(k_atomic32, dim3(N / 256), dim3(256), 0, 0, plane32, idx, N); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b)); if (ms < best[1]) best[1] = ms;
            CK(hipEventRecord(a)); hipLaunchKernelGGL(k_store, dim3(N / 256), dim3(256), 0, 0, plane, idx, N); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b)); if (ms < best[2]) best[2] = ms;
            CK(hipEventRecord(a)); hipLaunchKernelGGL(k_atomic_wg, dim3(N / 256), dim3(256), 0, 0, plane, idx, N); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b)); if (ms < best[3]) best[3] = ms;
        }
        printf("%-16s  atomic64 %7.1f us  atomic32 %7.1f us  store64 %7.1f us  atomic64-wgscope %7.1f us\n", names[mode], best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3);
    }
    // LDS variants: 256 blocks x 4096 events, tile = 16384 entries (128 KB)
    fill(0);
    for (int tile : {4096, 16384}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                float ms;
                CK(hipEventRecord(a));
                if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(256), dim3(256), tile * 8, 0, plane, idx, 4096, tile);
                if (mode == 1) hipLaunchKernelGGL(k_lds<1>, dim3(256), dim3(256), tile * 8, 0, plane, idx, 4096, tile);
                if (mode == 2) hipLaunchKernelGGL(k_lds<2>, dim3(256), dim3(256), tile * 8, 0, plane, idx, 4096, tile);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            printf("lds tile %5d entries, flush mode %d (0 store,1 atomic-shared-nonzero,2 atomic-private-dense): %7.1f us\n", tile, mode, best * 1e3);
        }
    }
    CK(hipGetLastError());
    return 0;
}
