// Microbenchmark: ~5500 waves (690 work-groups x 8) each add once to an overflow counter -- one word, 16 words of one
// cache line, 16 words on 16 lines.  What the scatter kernel's per-wave overflow count costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template <int MODE>
__global__ void k(unsigned int* ctr, int frac) {
    // a little work first so that the waves do not arrive in lock step
    float x = threadIdx.x;
    for (int i = 0; i < 64 + (int)(blockIdx.x & 63); ++i) x = x * 1.0001f + 0.5f;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0 && (wave % frac) == 0) {
        unsigned int* p = MODE == 0 ? ctr : (MODE == 1 ? ctr + (blockIdx.x & 15) : ctr + (blockIdx.x & 15) * 32);
        atomicAdd(p, 1u + (x < 0.f));
    }
}
template <int MODE>
void run(const char* name, unsigned int* d, int frac) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(690), dim3(512), 0, 0, d, frac);
    CK(hipEventRecord(a));
    for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(690), dim3(512), 0, 0, d, frac);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s 1 of %d waves adds: %.2f us per launch\n", name, frac, 1e3 * ms / 50);
}
int main() {
    unsigned int* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    for (int frac : {1000000, 4, 2, 1}) {
        run<0>("one word", d, frac);
        run<1>("16 words, one line", d, frac);
        run<2>("16 words, 16 lines", d, frac);
    }
    return 0;
}
