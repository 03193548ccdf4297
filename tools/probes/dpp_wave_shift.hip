// Does gfx950 execute the wavefront-wide DPP shifts (wave_shr:1 / wave_shl:1, GFX8/GFX9 encodings 0x138 / 0x130)?  conv1s.hip shifts
// accumulator columns by one lane with them.  Expected: lane i of `shr` holds i - 1 (lane 0: 0, bound_ctrl), of `shl` i + 1 (lane 63: 0).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/dpp_wave_shift tools/probes/dpp_wave_shift.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    { const int x = threadIdx.x * 10; out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x104, 0xf, 0xf, false); }
    const int v = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);
    // fused form: v_add_f32 with a DPP source
    float a = (float)threadIdx.x, b = 1000.f;
    float r;
    asm volatile("v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b));
    out[128 + threadIdx.x] = (int)r;
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int e_shr = i ? 100 + i - 1 : 0, e_shl = i < 63 ? 100 + i + 1 : 0, e_add = (i ? i - 1 : 0) + 1000;
        if (h[i] != e_shr || h[64 + i] != e_shl || h[128 + i] != e_add) { ++bad; printf("lane %d: shr %d (want %d) shl %d (want %d) add %d (want %d)\n", i, h[i], e_shr, h[64 + i], e_shl, h[128 + i], e_add); }
    }
    printf("row_shl:4 -- lanes 0, 1, 11, 12, 15, 16 hold %d %d %d %d %d %d (10 x source lane, -1 = kept)\n", h[192], h[193], h[203], h[204], h[207], h[208]);
    printf(bad ? "dpp_wave_shift: %d lanes differ\n" : "dpp_wave_shift: ok (wave_shr:1, wave_shl:1, v_add_f32_dpp)\n", bad);
    return bad != 0;
}
