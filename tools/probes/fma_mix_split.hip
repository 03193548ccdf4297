// Probe: hi = f16(x), lo = f16(x - hi) of a pair as v_cvt_pk_f16_f32 + v_fma_mixlo_f16 + v_fma_mixhi_f16 (three instructions) against the
// six-instruction form (two conversions back, two subtractions, one more pack): bit for bit the same over 4 M random pairs, f16 range.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2)); }
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = pack_f16(x0, x1);
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
    lo = l;
}
__global__ void k(const float* in, uint32_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = in[i * 2], x1 = in[i * 2 + 1];
    uint32_t h, l; split_pair(x0, x1, h, l);
    // reference
    const f16x2 hv = __builtin_bit_cast(f16x2, pack_f16(x0, x1));
    const uint32_t lr = pack_f16(x0 - (float)hv[0], x1 - (float)hv[1]);
    out[i * 4] = h; out[i * 4 + 1] = l; out[i * 4 + 2] = lr; out[i * 4 + 3] = (l == lr);
}
int main() {
    const int n = 1 << 22;
    float* in; uint32_t* out; hipMalloc(&in, n * 8); hipMalloc(&out, n * 16);
    float* h = (float*)malloc(n * 8);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) { uint32_t u = ((uint32_t)rand() << 16) ^ rand(); u &= 0xffffffffu; float f; memcpy(&f, &u, 4); int e = (u >> 23) & 255; if (e > 142 || e < 90) { u = (u & 0x807fffffu) | ((100u + (u >> 5) % 40) << 23); memcpy(&f, &u, 4); } h[i] = (i % 97 == 0) ? 0.f : f; }
    hipMemcpy(in, h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, in, out, n);
    uint32_t* o = (uint32_t*)malloc(n * 16); hipMemcpy(o, out, n * 16, hipMemcpyDeviceToHost);
    long bad = 0; for (int i = 0; i < n; ++i) if (!o[i * 4 + 3]) { if (bad < 5) printf("mismatch %d: x=%g,%g h=%08x l=%08x ref=%08x\n", i, h[2*i], h[2*i+1], o[i*4], o[i*4+1], o[i*4+2]); ++bad; }
    printf("pairs %d mismatches %ld\n", n, bad);
    return bad != 0;
}
