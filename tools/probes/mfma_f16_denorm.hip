// Probe: does v_mfma_f32_32x32x16_f16 keep f16 subnormal A/B inputs (default hipcc float mode)?
// A = one subnormal / tiny normal value per row, B = 1.0 -> D = sum_k A[i][k] * B[k][j].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const _Float16* vals, float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (lane >> 5) == 0 && j == 0 ? vals[lane & 31] : (_Float16)0.f; b[j] = (_Float16)1.0f; }
    f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    // D[row][col = lane & 31], row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if ((lane & 31) == 0) for (int r = 0; r < 16; ++r) out[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = c[r];
    // and as the B operand (pixels): swap roles
    f32x16 d; for (int r = 0; r < 16; ++r) d[r] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, d, 0, 0, 0);    // D[row][col]: sum_k 1 * A'[k][col=lane&31]
    if (lane < 32) out[32 + lane] = d[0];
}
int main() {
    _Float16 h[32]; float expect[32];
    for (int i = 0; i < 32; ++i) {
        uint16_t bits = (uint16_t)(i < 16 ? (1u << (i % 10)) + (i >= 10 ? 0x0400u : 0u) : 0x0001u * (i - 15) * 37u);   // subnormals and small normals
        memcpy(&h[i], &bits, 2); expect[i] = (float)h[i];
    }
    _Float16* dv; float* dout; float out[64];
    hipMalloc(&dv, sizeof h); hipMalloc(&dout, sizeof out);
    hipMemcpy(dv, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dv, dout);
    hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) {
        const bool ok = out[i] == expect[i] && out[32 + i] == expect[i];
        if (!ok) ++bad;
        printf("%2d in %.9g  A-side %.9g  B-side %.9g %s\n", i, expect[i], out[i], out[32 + i], ok ? "" : "<-- differs");
    }
    printf("F16_SUBNORMALS_%s\n", bad ? "FLUSHED_OR_WRONG" : "KEPT");
    return 0;
}
