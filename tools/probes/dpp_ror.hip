#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    const float m = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)l), 0x140, 0xf, 0xf, false));
    out[2 * l] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)l), 0x121, 0xf, 0xf, false));
    out[2 * l + 1] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, m), 0x121, 0xf, 0xf, false));
}
int main() {
    float* d; float h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 20; ++l) printf("lane %2d: row_ror:1 -> %g ; ror1(mirror) -> %g (want %d)\n", l, h[2 * l], h[2 * l + 1], (l & 48) + ((16 - (l & 15)) & 15));
    return 0;
}
