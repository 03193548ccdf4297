// Probe: v_pk_mul_f32 / v_pk_fma_f32 op_sel semantics used by frontend.hip's cmul2; DPP row_mirror; ds_bpermute.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 cmul2(f2 a, f2 s) {
    f2 t, o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(s));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(o) : "v"(a), "v"(s), "v"(t));
    return o;
}
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f2 a = f2{1.0f + l, 0.5f * l - 3.0f}, s = f2{0.25f * l + 0.125f, -1.5f + 0.0625f * l};
    f2 o = cmul2(a, s);
    out[4 * l] = o.x; out[4 * l + 1] = o.y;
    out[4 * l + 2] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)l), 0x140, 0xf, 0xf, false));
    out[4 * l + 3] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((l & 48) + ((16 - (l & 15)) & 15)) * 4, __builtin_bit_cast(int, (float)l)));
}
int main() {
    float* d; float h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const float ax = 1.0f + l, ay = 0.5f * l - 3.0f, c = 0.25f * l + 0.125f, sn = -1.5f + 0.0625f * l;
        const float wr = ax * c - ay * sn, wi = ay * c + ax * sn;
        const int mir = (l & 48) + 15 - (l & 15), bp = (l & 48) + ((16 - (l & 15)) & 15);
        const bool ok = fabsf(h[4 * l] - wr) < 1e-4f * (1 + fabsf(wr)) && fabsf(h[4 * l + 1] - wi) < 1e-4f * (1 + fabsf(wi)) && h[4 * l + 2] == (float)mir && h[4 * l + 3] == (float)bp;
        if (!ok) { ++bad; if (bad < 6) printf("lane %d: cmul (%g, %g) want (%g, %g); mirror %g want %d; bperm %g want %d\n", l, h[4 * l], h[4 * l + 1], wr, wi, h[4 * l + 2], mir, h[4 * l + 3], bp); }
    }
    printf("PROBE_%s\n", bad ? "MISMATCH" : "OK");
    return 0;
}
