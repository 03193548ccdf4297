// Measured HBM ceilings of the box the numbers in DESIGN.md are taken on: read-only, write-only and copy streams with 16-byte
// accesses per lane, sized well past the 256 MB Infinity Cache.  Development aid, not part of the library.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_ceiling.hip -o tools/_bin/hbm_ceiling && tools/_bin/hbm_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ in, float* __restrict__ sink, size_t n) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ out, size_t n, float v) {
    const f32x4 x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = x;
}
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// the conv epilogue's store shape: a wave writes 32 pixels x 64 B with two instructions, each lane 16 B of its pixel per instruction
// (mode 0: bytes [16 hh, +16) then [32 + 16 hh, +16) -- 32-byte pieces at a 64-byte stride; mode 1: whole pixels per instruction)
__global__ __launch_bounds__(512) void k_write_px(char* __restrict__ out, size_t npx, int mode, float v) {
    const f32x4 x = {v, v, v, v};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, hh = lane >> 5;
    for (size_t p0 = ((size_t)blockIdx.x * 8 + wave) * 32; p0 < npx; p0 += (size_t)gridDim.x * 256) {
        if (mode == 0) {
            char* q = out + (p0 + m) * 64 + hh * 16;
            *(f32x4*)q = x; *(f32x4*)(q + 32) = x;
        } else {
            char* q = out + (p0 + (m & ~1)) * 64 + (m & 1) * 32 + hh * 16;      // four lanes cover one pixel
            *(f32x4*)q = x; *(f32x4*)(q + 64) = x;
        }
    }
}
// r reads per w writes in one kernel (the conv launches are mixes like 1:18, 3:2, 130:1)
__global__ __launch_bounds__(256) void k_mix(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n, int r) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f32x4 a = {0, 0, 0, 0};
        for (int j = 0; j < r; ++j) a += in[i + (size_t)j * n];
        out[i] = a;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) * (1ull << 30);
    const size_t n = bytes / 16;
    f32x4 *a, *b; float* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {256 * 3, 256 * 4, 256 * 8, 256 * 16, 256 * 64};
    for (int g : grids) {
        float ms;
        auto run = [&](const char* name, auto&& f, double moved) {
            f(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("grid %6d  %-10s %7.1f us  %6.2f TB/s\n", g, name, ms / 5 * 1e3, moved / (ms / 5 * 1e-3) / 1e12);
        };
        run("read", [&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, sink, n); }, (double)bytes);
        run("write", [&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n, 1.0f); }, (double)bytes);
        run("write px0", [&] { hipLaunchKernelGGL(k_write_px, dim3(g / 2), dim3(512), 0, 0, (char*)b, bytes / 64, 0, 1.0f); }, (double)bytes);
        run("write px1", [&] { hipLaunchKernelGGL(k_write_px, dim3(g / 2), dim3(512), 0, 0, (char*)b, bytes / 64, 1, 1.0f); }, (double)bytes);
        run("copy", [&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        run("mix 3r:1w", [&] { hipLaunchKernelGGL(k_mix, dim3(g), dim3(256), 0, 0, a, b, n / 4, 3); }, (double)bytes);
    }
    return 0;
}
