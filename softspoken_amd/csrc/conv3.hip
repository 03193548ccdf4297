// Fused ResBlock for the 32-output-channel blocks (conv8, conv9_1, spec head), bf16: one launch, h and r never
// leave the CU.
//
//   reference ResBlock (root/code/backend/pytorch_neural_nets.py:7-41, eval, BatchNorm folded):
//       y = relu( conv3x3( relu(conv3x3(x) + b1) ) + b2 + conv1x1(x) + br )
//
// Why: rocprofv3 TCC counters on the two-launch form (profiles/r01_hbm_traffic.md) show these blocks moving 3-4.3 TB/s
// of real HBM traffic (50-68 % of what the chip sustains), and more than half of all bytes are the h and r tensors
// written by launch A and read back by launch B.  Here a block owns a 16x16 output tile:
//   phase A  for each 64-byte chunk of the (virtually concatenated, up-sampled) input: 20x20 input patch in LDS;
//            h = conv3x3(x) is accumulated on the 18x18 region B needs -- 324 pixels enumerated linearly into eleven
//            32-pixel M-tiles (waves 0..2 take two) -- and r = conv1x1(x) on the wave's own 2x16 output M-tile;
//            after the last chunk relu(h + b1) is written as bf16 into an LDS patch image (0 outside the picture:
//            conv2's zero padding), r stays in accumulator registers;
//   phase B  the usual 18-step 3x3 loop reads that image; accumulators start from r; epilogue as in conv2.hip
//            (bias, ReLU, staged 16-byte stores, optional FLAT conv_flatten partial sums).
// Bytes per 16x16 tile: 20x20xCin in (+27 % over 18x18) and 16x16x32 out, instead of in + 2 x (h + r) + out.
// Cost: phase A does 352/256 = 1.375 x the MFMAs of an un-fused A and is unevenly spread (2,2,2,1,1,1,1,1 M-tiles).
#include "kernels.h"

namespace ss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr int kPixPitch = 80;     // as conv.hip / conv2.hip
static constexpr int kRowPitch = 1664;
static constexpr int kXP = 20;           // input patch side (2-pixel halo)
static constexpr int kHP = 18;           // h patch side (1-pixel halo)

__device__ __forceinline__ void mma3(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void wave_lds_sync3() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_barrier3() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// BRESA: launch A's whole weight bank (nch x 10 taps x 2 KB) is resident in LDS; otherwise one chunk is streamed per stage
template <bool BRESA, bool FLAT>
__global__ __launch_bounds__(512) void resblock32_fused_kernel(ConvArgs a, int total_tiles, int lds_wa_bytes) {
    constexpr int KC = 32, ES = 2, NW = 8, NTHR = 512;
    constexpr int kTap = 2048;                            // bytes per tap (2 sub-steps x 64 lanes x 16 B), NT = 1
    constexpr int kX = kXP * kRowPitch, kH = kHP * kRowPitch;
    constexpr int NPX = kXP * kXP * 4;                    // 16-byte pieces of the input patch
    constexpr int AIT = (NPX + NTHR - 1) / NTHR;
    constexpr int NPW = 10 * kTap / 16;                   // pieces of one streamed weight chunk
    constexpr int BIT = BRESA ? 1 : (NPW + NTHR - 1) / NTHR;
    constexpr int PPP = 4, OUTP = 80;                     // result staging: 64 B of channels + pad per pixel
    constexpr int SROWS = FLAT ? 32 : 16;
    constexpr int NFS = 2;                                // FLAT: weight fragments per mel row (bf16)
    static_assert(NW * SROWS * OUTP <= kX, "result staging reuses the input patch area");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, m = lane & 31;
    const int py = (m >> 1) & 1, px = (m & 1) | ((m >> 2) << 1);   // lane's pixel in the wave's 2x16 output M-tile
    char* sX = smem;                                      // [20][1664] input patch of the current chunk
    char* sH = smem + kX;                                 // [18][1664] relu(h) of the current tile, bf16
    char* sWB = sH + kH;                                  // launch-B weights, 9 taps, resident
    char* sWA = sWB + 9 * kTap;                           // launch-A weights: all chunks (BRESA) or the current one
    char* sO = smem + wave * (SROWS * OUTP);              // result staging: aliases sX between two barriers

    const int H = a.H, W = a.W;
    const int nch = (a.C0 + a.C1) / KC;

    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, gper = gridDim.x >> 3;
    const int per = (total_tiles + 7) >> 3;
    auto tile_at = [&](int it) -> int {
        const int idx = local + it * gper;
        const int t = xcd * per + idx;
        return (idx < per && t < total_tiles) ? t : -1;
    };
    struct Tile { int n, y0, x0; };
    auto decode = [&](int t) -> Tile {
        Tile d;
        d.x0 = (t % a.tiles_x) * 16; t /= a.tiles_x;
        d.y0 = (t % a.tiles_y) * 16;
        d.n = t / a.tiles_y;
        return d;
    };

    u32x4 ra[AIT];
    u32x4 rb[BIT];
    auto issue_loads = [&](const Tile& d, int ci) {
        const int ch = ci * KC;
        const char* src; int Cs, up, c0;
        if (ch < a.C0) { src = (const char*)a.src0; Cs = a.C0; up = 0; c0 = ch; }
        else { src = (const char*)a.src1; Cs = a.C1; up = 1; c0 = ch - a.C0; }
        const int Hs = H >> up, Ws = W >> up;
        const char* base = src + ((size_t)d.n * Hs * Ws * Cs + c0) * ES;
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int p = tid + NTHR * it;
            const int part = p & 3, pix = p >> 2;
            const int pyy = pix / kXP, pxx = pix - pyy * kXP;
            const int Y = d.y0 - 2 + pyy, X = d.x0 - 2 + pxx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (p < NPX && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W) {
                const int off = (((Y >> up) * Ws + (X >> up)) * Cs) * ES + part * 16;
                v = *(const u32x4*)(base + off);
            }
            ra[it] = v;
        }
        if constexpr (!BRESA) {
            const char* wsrc = (const char*)a.wpk + (size_t)ci * 10 * kTap;
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int p = tid + NTHR * it;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (p < NPW) v = *(const u32x4*)(wsrc + p * 16);
                rb[it] = v;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int p = tid + NTHR * it;
            const int part = p & 3, pix = p >> 2;
            const int pyy = pix / kXP, pxx = pix - pyy * kXP;
            if (p < NPX) *(u32x4*)(sX + pyy * kRowPitch + pxx * kPixPitch + part * 16) = ra[it];
        }
        if constexpr (!BRESA) {
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int p = tid + NTHR * it;
                if (p < NPW) *(u32x4*)(sWA + p * 16) = rb[it];
            }
        }
    };

    int it_tile = 0;
    int tile = tile_at(0);
    if (tile < 0) return;
    Tile cur = decode(tile);

    for (int p = tid; p < 9 * kTap / 16; p += NTHR) *(u32x4*)(sWB + p * 16) = *(const u32x4*)((const char*)a.wpk_b + (size_t)p * 16);
    if constexpr (BRESA)
        for (int p = tid; p < lds_wa_bytes / 16; p += NTHR) *(u32x4*)(sWA + p * 16) = *(const u32x4*)((const char*)a.wpk + (size_t)p * 16);
    issue_loads(cur, 0);
    commit();
    __syncthreads();

    // phase-A M-tiles of this wave: k0 = wave, k1 = wave + 8 (exists for waves 0..2); pixel q = 32 k + m of the 18x18 region
    const bool two = wave < 3;
    int abase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int qq = 32 * (wave + 8 * t) + m;
        if (qq >= kHP * kHP) qq = 0;                      // padding rows of the last M-tile: any valid address, result discarded
        const int qy = qq / kHP, qx = qq - qy * kHP;
        abase[t] = qy * kRowPitch + qx * kPixPitch + hh * 16;
    }
    const int rbase = (2 * wave + py + 2) * kRowPitch + (px + 2) * kPixPitch + hh * 16;   // centre pixel of the wave's output M-tile in sX
    const int hbase = (2 * wave + py) * kRowPitch + px * kPixPitch + hh * 16;             // phase B: tap (0,0) of that pixel in sH
    const int boff0 = lane * 16;
    const float bias_a = a.bias_a[m], bias_b = a.bias[m];

    f32x16 accA[2], racc;
    int ci = 0;

    while (true) {
        int ci_n = ci + 1, tile_n = tile;
        Tile nxt = cur;
        if (ci_n == nch) {
            ci_n = 0;
            tile_n = tile_at(++it_tile);
            if (tile_n >= 0) nxt = decode(tile_n);
        }
        const bool has_next = tile_n >= 0;
        if (has_next) issue_loads(nxt, ci_n);
        const bool last = ci == nch - 1;

        u32x4 fb[FLAT ? 2 : 1][FLAT ? NFS : 1];
        if constexpr (FLAT) {
            if (last) {
#pragma unroll
                for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                    for (int f = 0; f < NFS; ++f)
                        fb[yy][f] = *(const u32x4*)((const char*)a.flat_w + ((cur.y0 + 2 * wave + yy) * NFS + f) * 1024 + lane * 16);
            }
        }
        if (ci == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { accA[0][r] = 0.f; accA[1][r] = 0.f; racc[r] = 0.f; }
        }
        // ---- phase A: 18 steps of the 3x3 on this wave's h M-tiles + 2 steps of the 1x1 on its output M-tile ----
        {
            const char* wbase = sWA + boff0 + (BRESA ? ci * 10 * kTap : 0);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const u32x4 xa = *(const u32x4*)(sX + rbase + sub * 32);
                const u32x4 wr = *(const u32x4*)(wbase + 9 * kTap + sub * 1024);
                mma3(racc, xa, wr);
            }
            if (two) {
                u32x4 af[2][2], bf[2];
                auto ld = [&](int st, u32x4 (&fa)[2], u32x4& fbw) {
                    const int tap = st >> 1, sub = st & 1, dy = tap / 3, dx = tap % 3;
                    fa[0] = *(const u32x4*)(sX + abase[0] + dy * kRowPitch + dx * kPixPitch + sub * 32);
                    fa[1] = *(const u32x4*)(sX + abase[1] + dy * kRowPitch + dx * kPixPitch + sub * 32);
                    fbw = *(const u32x4*)(wbase + tap * kTap + sub * 1024);
                };
                ld(0, af[0], bf[0]);
#pragma unroll
                for (int st = 0; st < 18; ++st) {
                    if (st + 1 < 18) ld(st + 1, af[(st + 1) & 1], bf[(st + 1) & 1]);
                    mma3(accA[0], af[st & 1][0], bf[st & 1]);
                    mma3(accA[1], af[st & 1][1], bf[st & 1]);
                }
            } else {
                u32x4 af[2], bf[2];
                auto ld = [&](int st, u32x4& fa, u32x4& fbw) {
                    const int tap = st >> 1, sub = st & 1, dy = tap / 3, dx = tap % 3;
                    fa = *(const u32x4*)(sX + abase[0] + dy * kRowPitch + dx * kPixPitch + sub * 32);
                    fbw = *(const u32x4*)(wbase + tap * kTap + sub * 1024);
                };
                ld(0, af[0], bf[0]);
#pragma unroll
                for (int st = 0; st < 18; ++st) {
                    if (st + 1 < 18) ld(st + 1, af[(st + 1) & 1], bf[(st + 1) & 1]);
                    mma3(accA[0], af[st & 1], bf[st & 1]);
                }
            }
        }

        if (last) {
            // ---- relu(h + b1) -> LDS patch image (bf16); 0 outside the picture (zero padding of conv2's input) ----
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 0 || two) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int q = 32 * (wave + 8 * t) + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (q < kHP * kHP) {
                            const int qy = q / kHP, qx = q - qy * kHP;
                            const int Y = cur.y0 - 1 + qy, X = cur.x0 - 1 + qx;
                            const float v = ((unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W) ? fmaxf(accA[t][r] + bias_a, 0.f) : 0.f;
                            *(__bf16*)(sH + qy * kRowPitch + qx * kPixPitch + m * ES) = (__bf16)v;
                        }
                    }
                }
            }
            lds_barrier3();                               // h image complete; every wave is done reading the input patch
            // ---- phase B: conv3x3 over the h image, accumulators start from the residual projection ----
            f32x16 acc = racc;
            {
                u32x4 af[2], bf[2];
                auto ld = [&](int st, u32x4& fa, u32x4& fbw) {
                    const int tap = st >> 1, sub = st & 1, dy = tap / 3, dx = tap % 3;
                    fa = *(const u32x4*)(sH + hbase + dy * kRowPitch + dx * kPixPitch + sub * 32);
                    fbw = *(const u32x4*)(sWB + boff0 + tap * kTap + sub * 1024);
                };
                ld(0, af[0], bf[0]);
#pragma unroll
                for (int st = 0; st < 18; ++st) {
                    if (st + 1 < 18) ld(st + 1, af[(st + 1) & 1], bf[(st + 1) & 1]);
                    mma3(acc, af[st & 1], bf[st & 1]);
                }
            }
            // ---- epilogue: bias (b2 + br) + ReLU, staged 16-byte stores, optional FLAT (staging aliases the input patch) ----
            const int Yb = cur.y0 + 2 * wave;
            f32x16 flat_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) flat_acc[r] = 0.f;
            constexpr int PASSES = 32 / SROWS, RPP = SROWS * PPP / 64;
#pragma unroll
            for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
                for (int rr = 0; rr < SROWS / 2; ++rr) {
                    const int r = pass * (SROWS / 2) + rr;
                    const int lrow = (r & 3) + 8 * (r >> 2) + 4 * hh - pass * SROWS;
                    *(__bf16*)(sO + lrow * OUTP + m * ES) = (__bf16)fmaxf(acc[r] + bias_b, 0.f);
                }
                wave_lds_sync3();
                if constexpr (FLAT) {
                    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int f = 0; f < NFS; ++f) {
                        const u32x4 av = *(const u32x4*)(sO + m * OUTP + f * 32 + hh * 16);
                        mma3(flat_acc, py == 0 ? av : zero4, fb[0][f]);
                        mma3(flat_acc, py == 1 ? av : zero4, fb[1][f]);
                    }
                }
                if (!FLAT || a.store_out) {
#pragma unroll
                    for (int k = 0; k < RPP; ++k) {
                        const int piece = lane + 64 * k;
                        const int lrow = piece / PPP, part = piece - lrow * PPP;
                        const int mrow = lrow + pass * SROWS;
                        const int Y = Yb + ((mrow >> 1) & 1), X = cur.x0 + ((mrow & 1) | ((mrow >> 2) << 1));
                        const u32x4 v16 = *(const u32x4*)(sO + lrow * OUTP + part * 16);
                        *(u32x4*)((char*)a.out + (((size_t)cur.n * H + Y) * W + X) * 32 * ES + part * 16) = v16;
                    }
                }
                wave_lds_sync3();
            }
            if constexpr (FLAT) {
                if (m < 4) {
                    const int grp = (cur.y0 + 2 * wave) / 2;
                    float* dst = a.flat_part + (((size_t)cur.n * (H / 2) + grp) * 4 + m) * W + cur.x0 + 2 * hh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        dst[4 * q] = flat_acc[4 * q] + flat_acc[4 * q + 2];
                        dst[4 * q + 1] = flat_acc[4 * q + 1] + flat_acc[4 * q + 3];
                    }
                }
            }
        }
        lds_barrier3();                                   // input patch (and staging) free for the next stage
        if (!has_next) break;
        commit();
        lds_barrier3();
        tile = tile_n; cur = nxt; ci = ci_n;
    }
}

template <bool BRESA, bool FLAT>
static hipError_t launch_fused_t(const ConvArgs& a, int total, int lds_wa, size_t lds, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)resblock32_fused_kernel<BRESA, FLAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((resblock32_fused_kernel<BRESA, FLAT>), dim3(grid), dim3(512), lds, s, a, total, lds_wa);
    return hipGetLastError();
}

// a.src0/src1 + C0/C1: block input (virtual concat); a.wpk: launch-A pack (10 taps per chunk), a.wpk_b: launch-B pack (9 taps);
// a.bias_a = b1, a.bias = b2 + br; a.out (and/or FLAT partials); Cout must be 32, H and W multiples of 16, bf16 only.
hipError_t launch_resblock32_fused(const ConvArgs& a_in, int num_cus, hipStream_t s) {
    ConvArgs a = a_in;
    if (a.Cout != 32 || a.H % 16 || a.W % 16 || a.C0 % 32 || a.C1 % 32 || !a.wpk || !a.wpk_b || !a.bias_a || !a.bias) return hipErrorInvalidValue;
    a.tiles_y = a.H / 16; a.tiles_x = a.W / 16;
    const long total_l = (long)a.N * a.tiles_y * a.tiles_x;
    if (total_l <= 0 || total_l > 0x7fffffff) return hipErrorInvalidValue;
    const int total = (int)total_l;
    const int nch = (a.C0 + a.C1) / 32;
    const size_t fixed = (size_t)(kXP + kHP) * kRowPitch + 9 * 2048;
    const bool bresa = fixed + (size_t)nch * 10 * 2048 <= 160 * 1024;
    const int lds_wa = bresa ? nch * 10 * 2048 : 10 * 2048;
    const size_t lds = fixed + lds_wa;
    int grid = num_cus;                                   // one 8-wave block per CU
    if (grid > total) grid = total;
    grid = (grid + 7) / 8 * 8;
    if (a.flat_part) return bresa ? launch_fused_t<true, true>(a, total, lds_wa, lds, grid, s) : launch_fused_t<false, true>(a, total, lds_wa, lds, grid, s);
    return bresa ? launch_fused_t<true, false>(a, total, lds_wa, lds, grid, s) : launch_fused_t<false, false>(a, total, lds_wa, lds, grid, s);
}

}  // namespace ss
