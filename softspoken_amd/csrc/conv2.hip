// conv3x3 implicit GEMM, second structure: persistent blocks with register prefetch.
//
// Same GEMM view, LDS image and fragment packing as conv.hip (which stays as the simple reference
// structure); what changes is how latency is hidden.  rocprofv3 counters on the first structure showed the
// matrix pipe 12 % busy, LDS 14 % busy and 61 % of wave time in waits: each block sat on its own
// global -> LDS staging.  Here a block walks a list of (tile, K-chunk) stages; while the MFMAs of stage s
// run from LDS, the global loads of stage s+1 (the next chunk, or the first chunk of the block's next tile)
// are already in flight into registers and are written to LDS after the compute.  Weights of layers whose
// whole folded filter bank fits (<= 72 KB per block) stay resident in LDS for the kernel's lifetime.  Results
// leave through a per-wave LDS staging tile as 16-byte pieces (whole 64..384-byte pixel rows), and the
// block -> tile map gives each XCD a contiguous range of tiles so halos and weights hit that XCD's L2.
//
// Reference: root/code/backend/pytorch_neural_nets.py:7-41,142-197 (see conv.hip for the op-level mapping).
#include "kernels.h"
#include <cstdlib>

namespace ss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr int kPixPitch = 80;     // as conv.hip
static constexpr int kRowPitch = 1664;
static constexpr int kPatch = 18;

template <bool BF16>
__device__ __forceinline__ void mma2(f32x16& acc, const u32x4& a, const u32x4& b) {
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

// LDS ops of one wave execute in issue order, so a wave's own write -> read needs no hardware wait; the asm
// statement only pins the compiler's order (and drains lgkmcnt, which is cheap).  It must NOT wait on vmcnt:
// the next stage's prefetch loads and this tile's output stores are meant to stay in flight.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// workgroup barrier that orders LDS only (a __syncthreads() would also emit vmcnt(0))
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FIRST: the 3x3 input is not read from memory but produced on the fly from the single-channel feature map
//        (conv1_1.conv1 = Conv2d(1,32,3)+BN+ReLU, K = 9, VALU) straight into the LDS patch image; the 1 -> 32
//        1x1 residual reads the same staged features.  Removes the h1 tensor and the conv_first launch.
// FLAT:  the epilogue also reduces conv_flatten's (128,1) kernel over this wave's rows and 32 channels into
//        per-row-group partial sums (fixed order, no atomics) -> the mask head needs no c9 tensor.
template <bool BF16, int NT, int MTW, bool BRES, bool FIRST, bool FLAT>
__global__ __launch_bounds__(256) void conv3x3_v2_kernel(ConvArgs a, int total_tiles, int lds_b_bytes) {
    constexpr int KC = BF16 ? 32 : 16;
    constexpr int ES = BF16 ? 2 : 4;
    constexpr int kTapBytes = 2 * NT * 1024;
    constexpr int PR = 8 * MTW + 2;                       // patch rows
    constexpr int kA = PR * kRowPitch;
    constexpr int NPA = PR * kPatch * 4;                  // 16-byte pieces of one patch
    constexpr int AIT = (NPA + 255) / 256;
    constexpr int NPB = 9 * kTapBytes / 16;               // pieces of a 9-tap weight chunk
    constexpr int BIT = BRES ? 1 : (NPB + 255) / 256;
    constexpr int PPP = 32 * NT * ES / 16;                // 16-byte pieces per output pixel
    constexpr int OUTP = 32 * NT * ES + 16;               // staging pitch per pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, m = lane & 31;
    const int py = (m >> 1) & 1, px = (m & 1) | ((m >> 2) << 1);   // m = (x&1) | (y<<1) | ((x>>1)<<2)
    char* sA = smem;
    char* sB = smem + kA;
    char* sO = sB + lds_b_bytes + wave * (32 * OUTP);
    float* sF = (float*)(sB + lds_b_bytes + 4 * (32 * OUTP));     // FIRST: [PR+2][20] feature patch, then [9][32] weights + [32] bias
    float* sW = sF + (PR + 2) * 20;

    const int H = a.H, W = a.W;
    const int ngroups = a.Cout / (32 * NT);
    const int nmain = (a.C0 + a.C1) / KC, nres = (a.R0 + a.R1) / KC, nch = nmain + nres;
    const int all_taps = nmain * 9 + nres;

    // block -> tiles: XCD x (blockIdx & 7, round-robin dispatch) owns the contiguous range [x*per, (x+1)*per)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, gper = gridDim.x >> 3;
    const int per = (total_tiles + 7) >> 3;
    auto tile_at = [&](int it) -> int {
        const int idx = local + it * gper;
        const int t = xcd * per + idx;
        return (idx < per && t < total_tiles) ? t : -1;
    };
    struct Tile { int n, y0, x0, g; };
    auto decode = [&](int t) -> Tile {
        Tile d;
        d.g = t % ngroups; t /= ngroups;
        d.x0 = (t % a.tiles_x) * 16; t /= a.tiles_x;
        d.y0 = (t % a.tiles_y) * (8 * MTW);
        d.n = t / a.tiles_y;
        return d;
    };

    u32x4 ra[AIT];
    u32x4 rb[BIT];
    constexpr int NF = ((PR + 2) * 20 + 255) / 256;       // FIRST: feature values per thread
    float rf[NF];

    auto issue_loads = [&](const Tile& d, int ci) {
        if constexpr (FIRST) {
            if (ci == 0) {
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    const int idx = tid + 256 * k;
                    const int fy = idx / 20, fx = idx - fy * 20;
                    const int Y = d.y0 - 2 + fy, X = d.x0 - 2 + fx;
                    rf[k] = (idx < (PR + 2) * 20 && Y >= 0 && Y < H && X >= 0 && X < W) ? a.rank1_src[((size_t)d.n * H + Y) * W + X] : 0.f;
                }
            }
            return;
        }
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
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int p = tid + 256 * it;
            const int part = p & 3, pix = p >> 2;
            const int pyy = pix / kPatch, pxx = pix - pyy * kPatch;
            const int Y = d.y0 - 1 + pyy, X = d.x0 - 1 + pxx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (p < NPA && Y >= 0 && Y < H && X >= 0 && X < W) {
                const int Ys = up ? (Y >> 1) : Y, Xs = up ? (X >> 1) : X;
                const size_t e = (((size_t)d.n * Hs + Ys) * Ws + Xs) * Cs + c0;
                v = *(const u32x4*)(src + e * ES + part * 16);
            }
            ra[it] = v;
        }
        if constexpr (!BRES) {
            const int npieces = (is_res ? 1 : 9) * (kTapBytes / 16);
            const char* wsrc = (const char*)a.wpk + ((size_t)d.g * all_taps + (is_res ? nmain * 9 + (ci - nmain) : ci * 9)) * kTapBytes;
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int p = tid + 256 * it;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (p < npieces) v = *(const u32x4*)(wsrc + (size_t)p * 16);
                rb[it] = v;
            }
        }
    };
    auto commit = [&](const Tile& d, int ci) {
        if constexpr (FIRST) {
            if (ci == 0) {
#pragma unroll
                for (int k = 0; k < NF; ++k) { const int idx = tid + 256 * k; if (idx < (PR + 2) * 20) sF[idx] = rf[k]; }
                lds_barrier();
            }
            // h1 = relu(conv3x3(feat) + b) for the patch pixels inside the image, 0 outside (conv2's zero padding)
            constexpr int CPP = 16 / ES;                  // channels per 16-byte piece
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                const int p = tid + 256 * it;
                const int part = p & 3, pix = p >> 2;
                const int pyy = pix / kPatch, pxx = pix - pyy * kPatch;
                if (p < NPA) {
                    const int Y = d.y0 - 1 + pyy, X = d.x0 - 1 + pxx;
                    u32x4 outv = {0u, 0u, 0u, 0u};
                    if (Y >= 0 && Y < H && X >= 0 && X < W) {
                        float f9[9];
#pragma unroll
                        for (int t = 0; t < 9; ++t) f9[t] = sF[(pyy + t / 3) * 20 + pxx + t % 3];
                        const int ch0 = ci * KC + part * CPP;
                        float o[CPP];
#pragma unroll
                        for (int e = 0; e < CPP; ++e) {
                            float acc1 = sW[288 + ch0 + e];
#pragma unroll
                            for (int t = 0; t < 9; ++t) acc1 = fmaf(sW[t * 32 + ch0 + e], f9[t], acc1);
                            o[e] = fmaxf(acc1, 0.f);
                        }
                        if constexpr (BF16) {
                            bf16x8 hv;
#pragma unroll
                            for (int e = 0; e < 8; ++e) hv[e] = (__bf16)o[e];
                            outv = __builtin_bit_cast(u32x4, hv);
                        } else {
                            outv = __builtin_bit_cast(u32x4, f32x4{o[0], o[1], o[2], o[3]});
                        }
                    }
                    *(u32x4*)(sA + pyy * kRowPitch + pxx * kPixPitch + part * 16) = outv;
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int p = tid + 256 * it;
            const int part = p & 3, pix = p >> 2;
            const int pyy = pix / kPatch, pxx = pix - pyy * kPatch;
            if (p < NPA) *(u32x4*)(sA + pyy * kRowPitch + pxx * kPixPitch + part * 16) = ra[it];
        }
        if constexpr (!BRES) {
            const int npieces = (ci >= nmain ? 1 : 9) * (kTapBytes / 16);
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int p = tid + 256 * it;
                if (p < npieces) *(u32x4*)(sB + p * 16) = rb[it];
            }
        }
    };

    int it_tile = 0;
    int tile = tile_at(0);
    if (tile < 0) return;                                 // whole block idle (block-uniform)
    Tile cur = decode(tile);

    if constexpr (BRES) {                                 // the layer's whole filter bank, once
        const char* wsrc = (const char*)a.wpk;
        for (int p = tid; p < lds_b_bytes / 16; p += 256) *(u32x4*)(sB + p * 16) = *(const u32x4*)(wsrc + (size_t)p * 16);
    }
    if constexpr (FIRST) {
        for (int i = tid; i < 320; i += 256) sW[i] = i < 288 ? a.first_w[i] : a.first_b[i - 288];
        __syncthreads();
    }
    issue_loads(cur, 0);
    commit(cur, 0);
    __syncthreads();

    f32x16 acc[MTW][NT];
    const int aoff0 = (2 * MTW * wave + py) * kRowPitch + px * kPixPitch + (BF16 ? hh * 16 : hh * 32);
    const int boff0 = lane * 16;
    int ci = 0;

    while (true) {
        // ---- which stage comes next (block-uniform) ----
        int ci_n = ci + 1, tile_n = tile;
        Tile nxt = cur;
        if (ci_n == nch) {
            ci_n = 0;
            tile_n = tile_at(++it_tile);
            if (tile_n >= 0) nxt = decode(tile_n);
        }
        const bool has_next = tile_n >= 0;
        if (has_next) issue_loads(nxt, ci_n);             // global loads in flight during the MFMAs below

        constexpr int NFS = (32 / KC) * 2;                // flatten weight fragments per mel row
        u32x4 fb[FLAT ? MTW : 1][2][FLAT ? NFS : 1];
        if constexpr (FLAT) {
            if (ci == nch - 1) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                        for (int f = 0; f < NFS; ++f)
                            fb[mt][yy][f] = *(const u32x4*)((const char*)a.flat_w + ((size_t)(cur.y0 + 2 * MTW * wave + 2 * mt + yy) * NFS + f) * 1024 + lane * 16);
            }
        }
        float r1v[MTW][16];
        if (!FIRST && a.rank1_src && ci == nch - 1) {     // 1 -> Cout 1x1 residual input: issued before the MFMAs, used after
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int Y = cur.y0 + 2 * MTW * wave + 2 * mt + ((r >> 1) & 1), X = cur.x0 + (r & 1) + 2 * hh + 4 * (r >> 2);
                    r1v[mt][r] = a.rank1_src[((size_t)cur.n * H + Y) * W + X];
                }
        }
        if (ci == 0) {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        // ---- MFMA on the staged chunk ----
        const bool is_res = ci >= nmain;
        const char* bbase = sB + boff0 + (BRES ? (is_res ? nmain * 9 + (ci - nmain) : ci * 9) * kTapBytes : 0);
        if (!is_res) {
            // 18 steps (9 taps x 2 sub-steps); fragments of step s+1 are requested before the MFMAs of step s
            u32x4 af[2][MTW], bfr[2][NT];
            auto load_frags = [&](int st, u32x4 (&fa)[MTW], u32x4 (&fb)[NT]) {
                const int tap = st >> 1, sub = st & 1, dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
                    fa[mt] = *(const u32x4*)(sA + aoff0 + (2 * mt + dy) * kRowPitch + dx * kPixPitch + (BF16 ? sub * 32 : sub * 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fb[nt] = *(const u32x4*)(bbase + tap * kTapBytes + (sub * NT + nt) * 1024);
            };
            load_frags(0, af[0], bfr[0]);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + 1 < 18) load_frags(st + 1, af[(st + 1) & 1], bfr[(st + 1) & 1]);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma2<BF16>(acc[mt][nt], af[st & 1][mt], bfr[st & 1][nt]);
            }
        } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                u32x4 af[MTW], bfr[NT];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
                    af[mt] = *(const u32x4*)(sA + aoff0 + (2 * mt + 1) * kRowPitch + kPixPitch + (BF16 ? sub * 32 : sub * 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bfr[nt] = *(const u32x4*)(bbase + (sub * NT + nt) * 1024);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma2<BF16>(acc[mt][nt], af[mt], bfr[nt]);
            }
        }

        // ---- last chunk of the tile: bias (+ rank-1) + ReLU, staged 16-byte stores, optional 2x2 max-pool ----
        if (ci == nch - 1) {
            const int co0 = cur.g * 32 * NT;
            f32x16 flat_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) flat_acc[r] = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const int Yb = cur.y0 + 2 * MTW * wave + 2 * mt;
                float pooled[NT][4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int co = co0 + nt * 32 + m;
                    const float b = a.bias[co];
                    const float r1w = a.rank1_src ? a.rank1_w[co] : 0.f;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // C/D row of the 32x32 MFMA = pixel index in the M-tile
                        const int mrow = (r & 3) + 8 * (r >> 2) + 4 * hh;
                        const int Y = Yb + ((r >> 1) & 1), X = cur.x0 + (r & 1) + 2 * hh + 4 * (r >> 2);
                        float t = acc[mt][nt][r] + b;
                        if constexpr (FIRST) t += r1w * sF[(2 * MTW * wave + 2 * mt + ((r >> 1) & 1) + 2) * 20 + (r & 1) + 2 * hh + 4 * (r >> 2) + 2];
                        else if (a.rank1_src) t += r1w * r1v[mt][r];
                        if (a.relu) t = fmaxf(t, 0.f);
                        v[r] = t;
                        char* dst = sO + mrow * OUTP + (nt * 32 + m) * ES;
                        if constexpr (BF16) *(__bf16*)dst = (__bf16)t; else *(float*)dst = t;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) pooled[nt][q] = fmaxf(fmaxf(v[4 * q], v[4 * q + 1]), fmaxf(v[4 * q + 2], v[4 * q + 3]));
                }
                wave_lds_sync();
                if constexpr (FLAT) {
                    // conv_flatten as a GEMM over channels with per-mel-row weights: the staged tile is read back as the
                    // A operand (row = this lane's pixel), rows of the other parity are zeroed so that one MFMA applies
                    // row Yb's weights and the next one row Yb+1's; all of a wave's rows accumulate into one C tile.
                    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int f = 0; f < NFS; ++f) {
                        const int cif = f >> 1, sub = f & 1;
                        const u32x4 av = *(const u32x4*)(sO + m * OUTP + cif * 64 + (BF16 ? sub * 32 + hh * 16 : hh * 32 + sub * 16));
                        mma2<BF16>(flat_acc, py == 0 ? av : zero4, fb[mt][0][f]);
                        mma2<BF16>(flat_acc, py == 1 ? av : zero4, fb[mt][1][f]);
                    }
                }
                if (!FLAT || a.store_out)
#pragma unroll
                for (int it = 0; it < PPP / 2; ++it) {
                    const int piece = lane + 64 * it;
                    const int mrow = piece / PPP, part = piece - mrow * PPP;
                    const int Y = Yb + ((mrow >> 1) & 1), X = cur.x0 + ((mrow & 1) | ((mrow >> 2) << 1));
                    const u32x4 v16 = *(const u32x4*)(sO + mrow * OUTP + part * 16);
                    *(u32x4*)((char*)a.out + ((((size_t)cur.n * H + Y) * W + X) * a.Cout + co0) * ES + part * 16) = v16;   // tiles divide H: always inside
                }
                wave_lds_sync();
                if (a.pool_out) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            char* dst = sO + (hh + 2 * q) * OUTP + (nt * 32 + m) * ES;
                            if constexpr (BF16) *(__bf16*)dst = (__bf16)pooled[nt][q]; else *(float*)dst = pooled[nt][q];
                        }
                    wave_lds_sync();
                    const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
                    for (int it = 0; it < (8 * PPP + 63) / 64; ++it) {
                        const int piece = lane + 64 * it;
                        const int pp = piece / PPP, part = piece - pp * PPP;
                        if (piece < 8 * PPP) {
                            const u32x4 v16 = *(const u32x4*)(sO + pp * OUTP + part * 16);
                            *(u32x4*)((char*)a.pool_out + ((((size_t)cur.n * Hp + (Yb >> 1)) * Wp + (cur.x0 >> 1) + pp) * a.Cout + co0) * ES + part * 16) = v16;
                        }
                    }
                    wave_lds_sync();
                }
            }
            if constexpr (FLAT) {
                // C tile: column = lane&31 = flatten output c (4 real), row -> (y = (r>>1)&1, x = (r&1) + 2 hh + 4 (r>>2)).
                // partial[n][row group][c][x] = sum over the wave's rows; the mask head adds the groups in order.
                if (m < 4) {
                    const int grp = (cur.y0 + 2 * MTW * wave) / (2 * MTW);
                    float* dst = a.flat_part + (((size_t)cur.n * (H / (2 * MTW)) + grp) * 4 + m) * W + cur.x0 + 2 * hh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        dst[4 * q] = flat_acc[4 * q] + flat_acc[4 * q + 2];
                        dst[4 * q + 1] = flat_acc[4 * q + 1] + flat_acc[4 * q + 3];
                    }
                }
            }
        }
        lds_barrier();                                    // every wave is done reading this stage's LDS image
        if (!has_next) break;
        commit(nxt, ci_n);                                // (the compiler waits for exactly the prefetch loads it writes)
        lds_barrier();
        tile = tile_n; cur = nxt; ci = ci_n;
    }
}

template <bool BF16, int NT, int MTW, bool BRES, bool FIRST, bool FLAT>
static hipError_t launch_v2_t(const ConvArgs& a, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_v2_kernel<BF16, NT, MTW, BRES, FIRST, FLAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3x3_v2_kernel<BF16, NT, MTW, BRES, FIRST, FLAT>), dim3(grid), dim3(256), lds, s, a, total, lds_b);
    return hipGetLastError();
}

template <bool BF16, int NT>
static hipError_t launch_v2_nt(const ConvArgs& a, int MTW, bool bres, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    if constexpr (NT == 1) {
        if (a.first_w) return launch_v2_t<BF16, 1, 2, true, true, false>(a, total, lds_b, lds, grid, s);
        if (a.flat_part) return launch_v2_t<BF16, 1, 2, true, false, true>(a, total, lds_b, lds, grid, s);
    }
    if (MTW == 2) return bres ? launch_v2_t<BF16, NT, 2, true, false, false>(a, total, lds_b, lds, grid, s)
                              : launch_v2_t<BF16, NT, 2, false, false, false>(a, total, lds_b, lds, grid, s);
    return bres ? launch_v2_t<BF16, NT, 1, true, false, false>(a, total, lds_b, lds, grid, s)
                : launch_v2_t<BF16, NT, 1, false, false, false>(a, total, lds_b, lds, grid, s);
}

hipError_t launch_conv3x3_v2(const ConvArgs& a_in, bool bf16, int NT, int num_cus, hipStream_t s) {
    ConvArgs a = a_in;
    if (a.W % 16 != 0 || a.H % 8 != 0 || a.Cout % (32 * NT) != 0 || NT < 1 || NT > 3) return hipErrorInvalidValue;
    const int kc = bf16 ? 32 : 16, es = bf16 ? 2 : 4;
    if (a.C0 % kc || a.C1 % kc || a.R0 % kc || a.R1 % kc) return hipErrorInvalidValue;
    const int MTW = (a.H % 16 == 0) ? 2 : 1;             // (a 32-row tile, 4 M-tiles per wave, measured slower: 1 block/CU)
    a.tiles_y = a.H / (8 * MTW); a.tiles_x = a.W / 16;
    const int ngroups = a.Cout / (32 * NT);
    const long total_l = (long)a.N * a.tiles_y * a.tiles_x * ngroups;
    if (total_l <= 0 || total_l > 0x7fffffff) return hipErrorInvalidValue;
    const int total = (int)total_l;
    const int tap_bytes = 2 * NT * 1024;
    const int all_taps = ((a.C0 + a.C1) / kc) * 9 + (a.R0 + a.R1) / kc;
    const bool bres = ngroups == 1 && (size_t)all_taps * tap_bytes <= 72 * 1024;
    const int lds_b = bres ? all_taps * tap_bytes : 9 * tap_bytes;
    if ((a.first_w || a.flat_part) && !(NT == 1 && MTW == 2 && bres)) return hipErrorInvalidValue;
    const size_t lds = (size_t)(8 * MTW + 2) * kRowPitch + lds_b + (size_t)4 * 32 * (32 * NT * es + 16) +
                       (a.first_w ? (size_t)((8 * MTW + 4) * 20 + 320) * 4 : 0);
    int bpc = (int)((160 * 1024) / lds);
    if (bpc < 1) return hipErrorInvalidValue;
    if (bpc > 3) bpc = 3;
    int grid = num_cus * bpc;
    if (grid > total) grid = total;
    grid = (grid + 7) / 8 * 8;                            // the tile map needs a multiple of 8 blocks (idle ones return at once)
    if (bf16) {
        switch (NT) {
            case 1: return launch_v2_nt<true, 1>(a, MTW, bres, total, lds_b, lds, grid, s);
            case 2: return launch_v2_nt<true, 2>(a, MTW, bres, total, lds_b, lds, grid, s);
            case 3: return launch_v2_nt<true, 3>(a, MTW, bres, total, lds_b, lds, grid, s);
        }
    } else {
        switch (NT) {
            case 1: return launch_v2_nt<false, 1>(a, MTW, bres, total, lds_b, lds, grid, s);
            case 2: return launch_v2_nt<false, 2>(a, MTW, bres, total, lds_b, lds, grid, s);
            case 3: return launch_v2_nt<false, 3>(a, MTW, bres, total, lds_b, lds, grid, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace ss
