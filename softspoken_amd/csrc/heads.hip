// Small kernels at the end of the conv stack: the 1-D mask head and the spec head's tail.
// Reference: root/code/backend/pytorch_neural_nets.py:126-140.
#include "kernels.h"

namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// mask head: conv_flatten's row-group partial sums [N][n_parts][4][256] summed in a fixed order, + bias, ReLU (relu_flatten),
// then ResBlock1D(4,4) + Conv1d(4,1,1) over the 256 time bins (pytorch_neural_nets.py:133-140,188-195).
// Output = raw logits (no sigmoid in the reference).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_head_parts_kernel(const float* __restrict__ parts, int n_parts, const float* __restrict__ fbias,
                                                              Head1dWeights hw, float* __restrict__ logits) {
    __shared__ float sx[4][258], sh[4][258];
    const int n = blockIdx.x, t = threadIdx.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float s = 0.f;
        for (int g = 0; g < n_parts; ++g) s += parts[(((size_t)n * n_parts + g) * 4 + c) * 256 + t];   // fixed order
        sx[c][t + 1] = fmaxf(fmaf(s, hw.fscale, fbias[c]), 0.f);       // (x 2^k: exact) + conv_flatten bias, relu_flatten
    }
    if (t < 4) { sx[t][0] = 0.f; sx[t][257] = 0.f; sh[t][0] = 0.f; sh[t][257] = 0.f; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float s = hw.b1[c];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int k = 0; k < 3; ++k) s = fmaf(hw.w1[c][ci][k], sx[ci][t + k], s);
        sh[c][t + 1] = fmaxf(s, 0.f);
    }
    __syncthreads();
    float o = hw.bo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float s = hw.b2r[c];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
#pragma unroll
            for (int k = 0; k < 3; ++k) s = fmaf(hw.w2[c][ci][k], sh[ci][t + k], s);
            s = fmaf(hw.wr[c][ci], sx[ci][t + 1], s);
        }
        o = fmaf(hw.wo[c], fmaxf(s, 0.f), o);
    }
    logits[(size_t)n * 256 + t] = o;
}

hipError_t launch_mask_head_parts(const float* parts, int n_parts, const float* flat_bias, const Head1dWeights& hw, float* logits, int N,
                                  hipStream_t s) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(mask_head_parts_kernel, dim3(N), dim3(256), 0, s, parts, n_parts, flat_bias, hw, logits);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// spec head tail: Conv2d(32,2,1) + bias + ReLU -> NCHW fp32 (the reference's spec_output layout, :128,184-185).
// PREC 0: fp32 activations; 1: bf16; 2: two f16 planes (value = hi + lo).
// ---------------------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(256) void spec_tail_kernel(const void* __restrict__ x, const void* __restrict__ x_lo, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ spec, size_t total) {
    const size_t gp = (size_t)blockIdx.x * 256 + threadIdx.x;   // pixel index over [N][128][256]
    if (gp >= total) return;
    float s0 = bias[0], s1 = bias[1];
#pragma unroll
    for (int ci = 0; ci < 32; ++ci) {
        float xv;
        if constexpr (PREC == 1) xv = (float)((const __bf16*)x)[gp * 32 + ci];
        else if constexpr (PREC == 2) xv = (float)((const _Float16*)x)[gp * 32 + ci] + (float)((const _Float16*)x_lo)[gp * 32 + ci];
        else xv = ((const float*)x)[gp * 32 + ci];
        s0 = fmaf(w[ci], xv, s0);
        s1 = fmaf(w[32 + ci], xv, s1);
    }
    const size_t n = gp / 32768, rem = gp % 32768;
    spec[(n * 2 + 0) * 32768 + rem] = fmaxf(s0, 0.f);
    spec[(n * 2 + 1) * 32768 + rem] = fmaxf(s1, 0.f);
}

hipError_t launch_spec_tail(const void* x, int64_t lo_delta, const float* w, const float* bias, float* spec, int N, int prec, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    const size_t total = (size_t)N * 32768;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    const void* xl = (const char*)x + lo_delta;
    if (prec == 1) hipLaunchKernelGGL(spec_tail_kernel<1>, dim3(blocks), dim3(256), 0, s, x, xl, w, bias, spec, total);
    else if (prec == 2) hipLaunchKernelGGL(spec_tail_kernel<2>, dim3(blocks), dim3(256), 0, s, x, xl, w, bias, spec, total);
    else hipLaunchKernelGGL(spec_tail_kernel<0>, dim3(blocks), dim3(256), 0, s, x, xl, w, bias, spec, total);
    return hipGetLastError();
}

}  // namespace ss
