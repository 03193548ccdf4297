// SpecUNet_2D conv stack on CDNA4 matrix cores (gfx950 only).
//
// Reference: root/code/backend/pytorch_neural_nets.py:7-41 (ResBlock), :142-197 (forward), eval mode.
// BatchNorm is folded into weights/bias on the host (engine.cpp), Dropout is identity, so a ResBlock is
//   h = relu(conv3x3(x, W1') + b1')                                   -> launch A
//   y = relu(conv3x3(h, W2') + conv1x1(x, Wr') + b2' + br')           -> launch B (the 1x1 is extra K)
// MaxPool2d(2,2) is an optional second output of launch B; Upsample(nearest, x2) + torch.cat are
// folded into the operand loader (virtual concat of a full-resolution and a half-resolution source).
//
// GEMM view: M = pixels, N = output channels, K = (tap, input channel).  A block owns a 16x16 pixel
// tile of one window and 32*NT output channels; 4 waves, wave w owns rows 4w..4w+3 as two 32-pixel
// MFMA M-tiles (2 rows x 16 columns each), so 2x2 pooling partners sit in one lane's accumulator.
// K is walked in chunks of 64 bytes of channels (32 bf16 / 16 fp32): the 18x18 halo patch of the
// chunk and all 9 taps of its weights go to LDS, then 9 taps x 2 sub-steps of MFMA per chunk.
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//   fp32: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) -- the parity path.
#include "kernels.h"

namespace ss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr int kPixPitch = 80;     // bytes per patch pixel in LDS: 64 B of channels + 16 B pad
static constexpr int kRowPitch = 1664;   // bytes per patch row: 104 x 16 B, == 8 (mod 16) slots -> conflict-free b128 reads
static constexpr int kPatch = 18;        // 16 + halo
static constexpr int kABytes = kPatch * kRowPitch;

size_t conv_lds_bytes(int NT) { return (size_t)kABytes + (size_t)9 * 2 * NT * 1024; }

template <bool BF16>
__device__ __forceinline__ void mma_step(f32x16& acc, const u32x4& a, const u32x4& b) {
    if constexpr (BF16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    } else {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
    }
}

template <bool BF16>
__device__ __forceinline__ void store_elem(void* base, size_t idx, float v) {
    if constexpr (BF16) ((__bf16*)base)[idx] = (__bf16)v;
    else ((float*)base)[idx] = v;
}

template <bool BF16, int NT>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(ConvArgs a) {
    constexpr int KC = BF16 ? 32 : 16;   // channels per 64-byte chunk
    constexpr int ES = BF16 ? 2 : 4;
    constexpr int kTapBytes = 2 * NT * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + kABytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, m = lane & 31;
    // lane's pixel inside a 2x16 M-tile: m = (x&1) | (y<<1) | ((x>>1)<<2)
    const int py = (m >> 1) & 1, px = (m & 1) | ((m >> 2) << 1);

    int bid = blockIdx.x;
    const int ngroups = a.Cout / (32 * NT);
    const int g = bid % ngroups; bid /= ngroups;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int y0 = ty * 16, x0 = tx * 16;
    const int H = a.H, W = a.W;

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nmain = (a.C0 + a.C1) / KC, nres = (a.R0 + a.R1) / KC;
    const char* wg = (const char*)a.wpk + (size_t)g * (size_t)(nmain * 9 + nres) * kTapBytes;

    // per-lane LDS byte offsets (tap (0,0), sub 0)
    const int aoff0 = (4 * wave + py) * kRowPitch + px * kPixPitch + (BF16 ? hh * 16 : hh * 32);
    const int boff0 = lane * 16;

    for (int ci = 0; ci < nmain + nres; ++ci) {
        const bool is_res = ci >= nmain;
        const int ch = (is_res ? ci - nmain : ci) * KC;
        const char* src; int Cs, up, c0;
        if (!is_res) {
            if (ch < a.C0) { src = (const char*)a.src0; Cs = a.C0; up = 0; c0 = ch; }
            else { src = (const char*)a.src1; Cs = a.C1; up = 1; c0 = ch - a.C0; }
        } else {
            if (ch < a.R0) { src = (const char*)a.res0; Cs = a.R0; up = 0; c0 = ch; }
            else { src = (const char*)a.res1; Cs = a.R1; up = 1; c0 = ch - a.R0; }
        }
        const int Hs = up ? (H >> 1) : H, Ws = up ? (W >> 1) : W;
        __syncthreads();                      // everyone is done reading the previous chunk
        if (!((a.dbg & 2) && ci > 0)) {
        // ---- stage the halo patch of this channel chunk: 18*18 pixels x 4 x 16 B ----
        for (int p = tid; p < kPatch * kPatch * 4; p += 256) {
            const int part = p & 3, pix = p >> 2;
            const int pyy = pix / kPatch, pxx = pix - pyy * kPatch;
            const int Y = y0 - 1 + pyy, X = x0 - 1 + pxx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (Y >= 0 && Y < H && X >= 0 && X < W) {
                const int Ys = up ? (Y >> 1) : Y, Xs = up ? (X >> 1) : X;
                const size_t e = (((size_t)n * Hs + Ys) * Ws + Xs) * Cs + c0;
                v = *(const u32x4*)(src + e * ES + part * 16);
            }
            *(u32x4*)(sA + pyy * kRowPitch + pxx * kPixPitch + part * 16) = v;
        }
        // ---- stage the weights of this chunk (already in fragment order: straight copy) ----
        const int ntaps = is_res ? 1 : 9;
        const char* wsrc = wg + (size_t)(is_res ? nmain * 9 + (ci - nmain) : ci * 9) * kTapBytes;
        for (int p = tid; p < ntaps * (kTapBytes / 16); p += 256)
            *(u32x4*)(sB + p * 16) = *(const u32x4*)(wsrc + (size_t)p * 16);
        }
        __syncthreads();
        // ---- MFMA ----
        if (!is_res) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    u32x4 af[2], bfr[NT];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        af[mt] = *(const u32x4*)(sA + aoff0 + (2 * mt + dy) * kRowPitch + dx * kPixPitch + (BF16 ? sub * 32 : sub * 16));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        bfr[nt] = *(const u32x4*)(sB + boff0 + tap * kTapBytes + (sub * NT + nt) * 1024);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) mma_step<BF16>(acc[mt][nt], af[mt], bfr[nt]);
                }
            }
        } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                u32x4 af[2], bfr[NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    af[mt] = *(const u32x4*)(sA + aoff0 + (2 * mt + 1) * kRowPitch + 1 * kPixPitch + (BF16 ? sub * 32 : sub * 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bfr[nt] = *(const u32x4*)(sB + boff0 + (sub * NT + nt) * 1024);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma_step<BF16>(acc[mt][nt], af[mt], bfr[nt]);
            }
        }
    }

    // ---- epilogue: bias (+ rank-1 residual) + ReLU, store NHWC, optional 2x2 max-pool ----
    // C/D map of the 32x32 MFMA: column = lane&31 (output channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5);
    // with the pixel order above: y = (r>>1)&1, x = (r&1) + 2*(lane>>5) + 4*(r>>2).
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int Yb = y0 + 4 * wave + 2 * mt;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = g * 32 * NT + nt * 32 + m;
            const float b = a.bias[co];
            const float r1w = a.rank1_src ? a.rank1_w[co] : 0.f;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int Y = Yb + ((r >> 1) & 1), X = x0 + (r & 1) + 2 * hh + 4 * (r >> 2);
                float t = acc[mt][nt][r] + b;
                if (a.rank1_src && Y < H) t += r1w * a.rank1_src[((size_t)n * H + Y) * W + X];
                if (a.relu) t = fmaxf(t, 0.f);
                v[r] = t;
                if (Y < H && (!(a.dbg & 1) || t == 12345.678f)) store_elem<BF16>(a.out, (((size_t)n * H + Y) * W + X) * a.Cout + co, t);
            }
            if (a.pool_out && Yb < H && !(a.dbg & 1)) {
                const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float mx = fmaxf(fmaxf(v[4 * q], v[4 * q + 1]), fmaxf(v[4 * q + 2], v[4 * q + 3]));
                    const int Xp = (x0 >> 1) + hh + 2 * q, Yp = Yb >> 1;
                    store_elem<BF16>(a.pool_out, (((size_t)n * Hp + Yp) * Wp + Xp) * a.Cout + co, mx);
                }
            }
        }
    }
}

template <bool BF16, int NT>
static hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
    const size_t lds = conv_lds_bytes(NT);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<BF16, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int blocks = a.N * a.tiles_y * a.tiles_x * (a.Cout / (32 * NT));
    hipLaunchKernelGGL((conv3x3_mfma_kernel<BF16, NT>), dim3(blocks), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_conv3x3(const ConvArgs& a, bool bf16, int NT, hipStream_t s) {
    if (a.W % 16 != 0 || a.Cout % (32 * NT) != 0) return hipErrorInvalidValue;
    const int kc = bf16 ? 32 : 16;
    if (a.C0 % kc || a.C1 % kc || a.R0 % kc || a.R1 % kc) return hipErrorInvalidValue;
    if (bf16) {
        switch (NT) {
            case 1: return launch_t<true, 1>(a, s);
            case 2: return launch_t<true, 2>(a, s);
            case 3: return launch_t<true, 3>(a, s);
        }
    } else {
        switch (NT) {
            case 1: return launch_t<false, 1>(a, s);
            case 2: return launch_t<false, 2>(a, s);
            case 3: return launch_t<false, 3>(a, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------
// conv1_1.conv1: Conv2d(1, 32, 3, padding=1) + folded BN + ReLU.  K = 9: VALU, not MFMA.
// thread = (pixel, 8 output channels); 4 lanes cover a pixel's 32 channels -> 16/32-byte stores.
// ---------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ feat, const float* __restrict__ w,
                                                         const float* __restrict__ bias, void* out, int N, int H, int W) {
    const size_t gp = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const int cg = threadIdx.x & 3;
    const size_t total = (size_t)N * H * W;
    if (gp >= total) return;
    const int x = (int)(gp % W);
    const int y = (int)((gp / W) % H);
    const size_t n = gp / ((size_t)W * H);
    float f[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        f[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? feat[(n * H + yy) * W + xx] : 0.f;
    }
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float s = bias[cg * 8 + c];
#pragma unroll
        for (int t = 0; t < 9; ++t) s = fmaf(w[t * 32 + cg * 8 + c], f[t], s);
        o[c] = fmaxf(s, 0.f);
    }
    if constexpr (BF16) {
        bf16x8 v;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (__bf16)o[c];
        *(bf16x8*)((__bf16*)out + gp * 32 + cg * 8) = v;
    } else {
        float* p = (float*)out + gp * 32 + cg * 8;
        *(f32x4*)p = f32x4{o[0], o[1], o[2], o[3]};
        *(f32x4*)(p + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
}

hipError_t launch_conv_first(const float* feat, const float* w, const float* bias, void* out, int N, int H, int W, bool bf16,
                             hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    const unsigned blocks = (unsigned)((total + 63) / 64);
    if (bf16) hipLaunchKernelGGL(conv_first_kernel<true>, dim3(blocks), dim3(256), 0, s, feat, w, bias, out, N, H, W);
    else hipLaunchKernelGGL(conv_first_kernel<false>, dim3(blocks), dim3(256), 0, s, feat, w, bias, out, N, H, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// conv_flatten: Conv2d(32, 4, (128,1)) + bias + ReLU (pytorch_neural_nets.py:133-134,188-192).
// Per time column t a K = 128*32 dot product for each of 4 outputs.  Block = (window, 64 columns);
// wave w reduces mel rows 32w..32w+31, lanes are columns; cross-wave sum through LDS.
// ---------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void flatten_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ flat) {
    __shared__ float red[4][4][64];
    const int n = blockIdx.x >> 2, t0 = (blockIdx.x & 3) * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int hrow = wave * 32; hrow < wave * 32 + 32; ++hrow) {
        const size_t e = (((size_t)n * 128 + hrow) * 256 + t0 + lane) * 32;
        float xv[32];
        if constexpr (BF16) {
            const bf16x8* p = (const bf16x8*)((const __bf16*)x + e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x8 v = p[q];
#pragma unroll
                for (int c = 0; c < 8; ++c) xv[q * 8 + c] = (float)v[c];
            }
        } else {
            const f32x4* p = (const f32x4*)((const float*)x + e);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 v = p[q];
#pragma unroll
                for (int c = 0; c < 4; ++c) xv[q * 4 + c] = v[c];
            }
        }
        const float* wr = w + (size_t)hrow * 128;   // [ci][4], wave-uniform -> scalar loads
#pragma unroll
        for (int ci = 0; ci < 32; ++ci) {
            s0 = fmaf(wr[ci * 4 + 0], xv[ci], s0);
            s1 = fmaf(wr[ci * 4 + 1], xv[ci], s1);
            s2 = fmaf(wr[ci * 4 + 2], xv[ci], s2);
            s3 = fmaf(wr[ci * 4 + 3], xv[ci], s3);
        }
    }
    red[wave][0][lane] = s0; red[wave][1][lane] = s1; red[wave][2][lane] = s2; red[wave][3][lane] = s3;
    __syncthreads();
    {
        const int c = wave;   // wave w finishes output channel w
        const float v = ((red[0][c][lane] + red[1][c][lane]) + (red[2][c][lane] + red[3][c][lane])) + bias[c];
        flat[((size_t)n * 4 + c) * 256 + t0 + lane] = fmaxf(v, 0.f);
    }
}

hipError_t launch_flatten(const void* x, const float* w, const float* bias, float* flat, int N, bool bf16, hipStream_t s) {
    if (bf16) hipLaunchKernelGGL(flatten_kernel<true>, dim3(N * 4), dim3(256), 0, s, x, w, bias, flat);
    else hipLaunchKernelGGL(flatten_kernel<false>, dim3(N * 4), dim3(256), 0, s, x, w, bias, flat);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// mask head: ResBlock1D(4,4) + Conv1d(4,1,1) over the 256 time bins (pytorch_neural_nets.py:137-140,195).
// Output = raw logits (no sigmoid in the reference).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_head_kernel(const float* __restrict__ flat, Head1dWeights hw, float* __restrict__ logits) {
    __shared__ float sx[4][258], sh[4][258];
    const int n = blockIdx.x, t = threadIdx.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) sx[c][t + 1] = flat[((size_t)n * 4 + c) * 256 + t];
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

__global__ __launch_bounds__(256) void mask_head_parts_kernel(const float* __restrict__ parts, int n_parts, const float* __restrict__ fbias,
                                                              Head1dWeights hw, float* __restrict__ logits) {
    __shared__ float sx[4][258], sh[4][258];
    const int n = blockIdx.x, t = threadIdx.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float s = 0.f;
        for (int g = 0; g < n_parts; ++g) s += parts[(((size_t)n * n_parts + g) * 4 + c) * 256 + t];   // fixed order
        sx[c][t + 1] = fmaxf(s + fbias[c], 0.f);       // conv_flatten bias + relu_flatten
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
    hipLaunchKernelGGL(mask_head_parts_kernel, dim3(N), dim3(256), 0, s, parts, n_parts, flat_bias, hw, logits);
    return hipGetLastError();
}

hipError_t launch_mask_head(const float* flat, const Head1dWeights& hw, float* logits, int N, hipStream_t s) {
    hipLaunchKernelGGL(mask_head_kernel, dim3(N), dim3(256), 0, s, flat, hw, logits);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// spec head tail: Conv2d(32,2,1) + bias + ReLU -> NCHW fp32 (the reference's spec_output layout).
// ---------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void spec_tail_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ spec, size_t total) {
    const size_t gp = (size_t)blockIdx.x * 256 + threadIdx.x;   // pixel index over [N][128][256]
    if (gp >= total) return;
    float s0 = bias[0], s1 = bias[1];
#pragma unroll
    for (int ci = 0; ci < 32; ++ci) {
        float xv;
        if constexpr (BF16) xv = (float)((const __bf16*)x)[gp * 32 + ci];
        else xv = ((const float*)x)[gp * 32 + ci];
        s0 = fmaf(w[ci], xv, s0);
        s1 = fmaf(w[32 + ci], xv, s1);
    }
    const size_t n = gp / 32768, rem = gp % 32768;
    spec[(n * 2 + 0) * 32768 + rem] = fmaxf(s0, 0.f);
    spec[(n * 2 + 1) * 32768 + rem] = fmaxf(s1, 0.f);
}

hipError_t launch_spec_tail(const void* x, const float* w, const float* bias, float* spec, int N, bool bf16, hipStream_t s) {
    const size_t total = (size_t)N * 32768;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (bf16) hipLaunchKernelGGL(spec_tail_kernel<true>, dim3(blocks), dim3(256), 0, s, x, w, bias, spec, total);
    else hipLaunchKernelGGL(spec_tail_kernel<false>, dim3(blocks), dim3(256), 0, s, x, w, bias, spec, total);
    return hipGetLastError();
}

}  // namespace ss
