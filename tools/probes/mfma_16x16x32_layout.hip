// Operand and result layout of v_mfma_f32_16x16x32_f16 on gfx950, as conv1s.hip's 16-pixel form assumes it:
//   A (16 x 32): lane l holds row l & 15, k = 8 (l >> 4) + j, j = 0..7        B (32 x 16): lane l holds column l & 15, k = 8 (l >> 4) + j
//   D (16 x 16): lane l holds column l & 15, rows 4 (l >> 4) + r, r = 0..3
// and that f16 subnormal operands are kept (the low halves of the f16x2 split may be subnormal).
// build: hipcc -O2 --offload-arch=gfx950 -o tools/_bin/mfma_16x16x32_layout tools/probes/mfma_16x16x32_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const _Float16* A, const _Float16* B, float* D) {   // A [16][32], B [32][16] row-major, D [16][16]
    const int l = threadIdx.x, g = l >> 4, c = l & 15;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[c * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + c]; }
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + c] = d[r];
}
int main() {
    _Float16 hA[16 * 32], hB[32 * 16];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 32; ++kk) hA[i * 32 + kk] = (_Float16)(float)((i * 7 + kk * 3) % 11 - 5);
    for (int kk = 0; kk < 32; ++kk) for (int n = 0; n < 16; ++n) hB[kk * 16 + n] = (_Float16)(float)((kk * 5 + n * 2) % 13 - 6);
    // subnormal check: A[0][0] = 2^-20 (an f16 subnormal), B[0][0] = 1024 -> contributes 2^-10 to D[0][0] if kept
    const float sub = ldexpf(1.f, -20);
    hA[0] = (_Float16)sub; hB[0] = (_Float16)1024.f;
    _Float16 *dA, *dB; float* dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, 256 * 4);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    float hD[256]; (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) {
        double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)(float)hA[i * 32 + kk] * (double)(float)hB[kk * 16 + n];
        if (fabs(s - hD[i * 16 + n]) > 1e-6 + 2e-7 * fabs(s)) { if (bad < 5) printf("D[%d][%d] = %g, want %g\n", i, n, hD[i * 16 + n], s); ++bad; }
    }
    printf(bad ? "mfma_16x16x32_layout: %d elements differ\n" : "mfma_16x16x32_layout: ok (A row l&15 k 8(l>>4)+j, B column l&15, D rows 4(l>>4)+r; f16 subnormal operand kept)\n", bad);
    return bad != 0;
}
