// Do matrix and vector work of the waves of one SIMD overlap?  conv1s.hip's time is close to (vector-only time) + (matrix-only time)
// although two waves share each SIMD.  Per wave and iteration: NM products v_mfma_f32_32x32x16_f16 (three accumulators, each product
// depending on the one three back) and NV independent v_add_f32, either one block after the other ("seq", conv1s.hip's shape) or four
// vector instructions behind every product ("mix").  One 8-wave block per CU on every CU (2 waves per SIMD), operands in registers.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/mfma_valu_overlap tools/probes/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// SHAPE 1: every 32x32x16 product as two v_mfma_f32_16x16x32_f16 on two quarters of the tile (the halves change places: all registers are results)
template <int SHAPE>
__device__ __forceinline__ f32x16 prod(const f16x8& w, const f16x8& x, const f32x16& c) {
    if constexpr (SHAPE == 0) return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0);
    f32x4 q0 = {c[0], c[1], c[2], c[3]}, q1 = {c[4], c[5], c[6], c[7]};
    q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, q0, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, q1, 0, 0, 0);
    f32x16 r;
    for (int i = 0; i < 8; ++i) r[i] = c[8 + i];
    for (int i = 0; i < 4; ++i) { r[8 + i] = q0[i]; r[12 + i] = q1[i]; }
    return r;
}
template <int MODE, int SHAPE>   // 0 matrix only, 1 vector only, 2 seq, 3 mix
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0};
    f16x8 x, w;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.001f * (threadIdx.x + i)); w[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const float c = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            if (MODE != 1) {
                a0 = prod<SHAPE>(w, x, a0);
                a1 = prod<SHAPE>(w, x, a1);
                a2 = prod<SHAPE>(w, x, a2);
            }
            if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 12; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 216; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE, int SHAPE>
static float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, SHAPE>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, SHAPE>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 2000;
    {
        const float t0 = run<0, 0>(d, iters), t1 = run<1, 0>(d, iters), t2 = run<2, 0>(d, iters), t3 = run<3, 0>(d, iters);
        printf("32x32x16: per iteration and SIMD (2 waves x 54 products, 2 x 216 v_add): matrix only %.3f us | vector only %.3f us | one block after the other %.3f us | 12 adds behind every 3 products %.3f us\n",
               t0 / iters, t1 / iters, t2 / iters, t3 / iters);
        printf("cycles at 2.4 GHz: matrix %.0f (54 x 2 x 32 = 3456) | vector %.0f | seq %.0f | mix %.0f\n", t0 / iters * 2400, t1 / iters * 2400, t2 / iters * 2400, t3 / iters * 2400);
    }
    {
        const float t0 = run<0, 1>(d, iters), t2 = run<2, 1>(d, iters), t3 = run<3, 1>(d, iters);
        printf("the same products as pairs of 16x16x32: matrix only %.3f us | one block after the other %.3f us | mixed %.3f us\n", t0 / iters, t2 / iters, t3 / iters);
    }
    return 0;
}
