// Issue cost of the vector instructions conv1s.hip's epilogue is made of, one wave alone on its SIMD: cycles per instruction over
// 512 independent instances (s_memtime around an unrolled block).  build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/valu_cost tools/probes/valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
// body(i): one instruction writing destination i of 8 (independent of its neighbours: throughput, not latency)
#define EIGHT(body) body(0) body(1) body(2) body(3) body(4) body(5) body(6) body(7)
#define BLOCK(name, body)                                                                          \
    {                                                                                              \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) { REP8(EIGHT(body)) }                        \
        asm volatile("s_nop 0" ::: "memory");                                                      \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        if (threadIdx.x == 0) out[k++] = (float)(t1 - t0) / 512.0f;                                \
    }
__global__ void k_cost(float* out, float* sink) {
    int k = 0;
    float a = threadIdx.x * 0.5f, b = 1.25f, c = 3.f, d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t u = threadIdx.x, w = 7u, z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define B_ADD(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d[i]) : "v"(a), "v"(b));
#define B_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d[i]) : "v"(a), "v"(b), "v"(c));
#define B_MAXI(i) asm volatile("v_max_i32 %0, %1, %2" : "=v"(z[i]) : "v"(u), "v"(w));
#define B_CVT(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(z[i]) : "v"(a), "v"(b));
#define B_MIX(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(z[i]) : "v"(u), "v"(a));
#define B_PKMAX(i) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(z[i]) : "v"(u), "v"(w));
#define B_DPPW(i) asm volatile("v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d[i]) : "v"(a), "v"(b));
#define B_DPPR(i) asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d[i]) : "v"(a), "v"(b));
#define B_DPPQ(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(z[i]) : "v"(u));
#define B_AND(i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(z[i]) : "v"(u), "v"(w));
#define B_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(pa), "v"(pb));
#define B_LOG(i) asm volatile("v_log_f32 %0, %1" : "=v"(d[i]) : "v"(a));
#define B_MAXI_DPP(i) asm volatile("v_max_i32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(z[i]) : "v"(u), "v"(w));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa = {a, b}, pb = {c, a}, pk[8];
    BLOCK("v_add_f32", B_ADD)
    BLOCK("v_fma_f32", B_FMA)
    BLOCK("v_max_i32", B_MAXI)
    BLOCK("v_cvt_pk_f16_f32", B_CVT)
    BLOCK("v_fma_mixlo_f16", B_MIX)
    BLOCK("v_pk_max_u16", B_PKMAX)
    BLOCK("v_add_f32_dpp wave_shr", B_DPPW)
    BLOCK("v_add_f32_dpp row_shr", B_DPPR)
    BLOCK("v_mov_b32_dpp quad_perm", B_DPPQ)
    BLOCK("v_and_b32", B_AND)
    BLOCK("v_pk_mul_f32", B_PKMUL)
    BLOCK("v_log_f32", B_LOG)
    {   // permlane32_swap modifies both operands
        uint64_t t0 = __builtin_amdgcn_s_memtime();
        _Pragma("unroll") for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));) }
        asm volatile("s_nop 0" ::: "memory");
        uint64_t t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[k++] = (float)(t1 - t0) / 512.0f;
    }
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += d[i] + (float)z[i] + pk[i].x + pk[i].y;
    sink[threadIdx.x] = acc + (float)u + (float)w;
}
int main() {
    float *d, *s; hipMalloc(&d, 64 * 4); hipMalloc(&s, 4096);
    hipMemset(s, 0, 4096);
    const char* names[] = {"v_add_f32", "v_fma_f32", "v_max_i32", "v_cvt_pk_f16_f32", "v_fma_mixlo_f16", "v_pk_max_u16", "v_add_f32_dpp wave_shr:1",
                           "v_add_f32_dpp row_shr:1", "v_mov_b32_dpp quad_perm", "v_and_b32", "v_pk_mul_f32", "v_log_f32", "v_permlane32_swap_b32 (dependent chain)"};
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_cost, dim3(1), dim3(64), 0, 0, d, s);
    float h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 13; ++i) printf("%-44s %6.2f cycles\n", names[i], h[i]);
    return 0;
}
