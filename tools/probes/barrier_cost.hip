// Probe: what an s_barrier costs a 1024-thread workgroup (16 waves, one workgroup per CU, every CU busy): cycles per barrier for a loop of
// nothing but barriers, with a little scalar / vector work between them, and with the staggered four-tile beat of conv4.hip / conv4_ups.hip
// (tile q does `work` dependent VALU instructions in every second interval, offset by q).   usage: barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int work) {
    const int tile = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    float v = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1) { for (int j = 0; j < work; ++j) v = v * 1.0001f + 0.5f; }
        if (MODE == 2) { if (((i + tile) & 1) == 0) for (int j = 0; j < work; ++j) v = v * 1.0001f + 0.5f; }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 1024 + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / iters;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;
    float* out; hipMalloc(&out, (size_t)blocks * 1024 * 4);
    const int iters = 20000;
    auto run = [&](int mode, int work) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, out, iters, work);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, out, iters, work);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, out, iters, work);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float cyc; hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
        printf("mode %d work %4d: %.1f ns per barrier interval (%.0f shader-clock ticks of 100 MHz x?)\n", mode, work, ms * 1e6 / iters, cyc);
    };
    run(0, 0);
    for (int w : {16, 64, 256}) run(1, w);
    for (int w : {16, 64, 256, 1024}) run(2, w);
    return 0;
}
