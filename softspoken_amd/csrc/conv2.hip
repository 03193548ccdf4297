// conv3x3 implicit GEMM, second structure: persistent blocks, register prefetch, fused ResBlock plumbing.
//
// Same GEMM view, LDS image and fragment packing as conv.hip (which stays as the simple first structure,
// SOFTSPOKEN_CONV=1); what changes is how a ResBlock is cut into launches and how latency and issue slots are spent.
//
//   reference ResBlock (root/code/backend/pytorch_neural_nets.py:7-41, eval, BatchNorm folded):
//       idt = conv1x1(x) ; h = relu(conv3x3(x)) ; y = relu(conv3x3(h) + idt)
//   launch A (RES):  h = relu(conv3x3(x) + b1)   and   r = conv1x1(x) + br     -- the 1x1 reuses the centre-tap
//                    fragments A already has in LDS: two extra MFMA steps per K chunk, no extra staging
//   launch B:        y = relu(conv3x3(h) + b2 + r)  (+ maxpool output, + FLAT)  -- r is added in the epilogue
//   (the first structure ran the 1x1 inside B as extra K chunks: a whole 18x18 patch stage for 2-4 MFMAs per wave.)
//
// What rocprofv3 said about the first structure and what this one does about it (profiles/r01_pmc_conv.md):
//   * 61 % of wave time waiting on the block's own global -> LDS staging  -> a block walks (tile, K-chunk) stages and
//     the global loads of stage s+1 (next chunk, or first chunk of its next tile) are in flight in registers during
//     the MFMAs of stage s; barriers wait on lgkmcnt only, so prefetch loads and output stores stay in flight;
//   * with every load, MFMA and store removed the kernel still took half its time: it is instruction-issue bound on
//     its non-MFMA code -> the 16x16 tile is worked by 8 waves of one M-tile each in bf16 (twice the waves per SIMD for
//     the same LDS image), offsets are 32-bit, bounds tests unsigned;
//   * weights of layers whose whole folded bank is <= 72 KB stay resident in LDS for the kernel's lifetime;
//   * results leave through an LDS staging tile (aliasing the patch between two barriers) as 16-byte pieces;
//   * the block -> tile map gives each XCD a contiguous tile range (halo and weight reuse in that XCD's L2).
// FIRST: the 3x3 input is produced on the fly from the single-channel feature map (conv1_1.conv1 = Conv2d(1,32,3)+BN+
//        ReLU, K = 9, VALU) straight into the LDS patch image; the 1 -> 32 1x1 residual reads the same staged features.
// FLAT:  the epilogue also applies conv_flatten's (128,1) kernel on MFMA to the staged tile (per-mel-row weights: the two rows
//        of an M-tile in different output columns of one product) and writes per-row-group partial sums, added in fixed order by
//        the mask head: no c9 tensor.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

namespace ss {

// development switches exist in the dev build only (engine.h has the same helper for the host units)
// timing-only ablation bits of ConvArgs::dbg (results are wrong with them set): they exist in the dev build only
#ifdef SS_DEVBUILD
#define SS_ABL(x) (x)
#else
#define SS_ABL(x) 0
#endif
#ifdef SS_DEVBUILD
static int dev_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
static constexpr int dev_env(const char*, int dflt) { return dflt; }
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr int kPixPitch = 80;     // as conv.hip: 64 B of channels + 16 B pad per patch pixel
static constexpr int kRowPitch = 1664;   // 104 x 16 B per patch row, == 8 (mod 16) slots: conflict-free ds_read_b128
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

template <bool BF16, int NT, int MTW, int NW, bool BRES, bool RES, bool FIRST, bool FLAT>
__global__ __launch_bounds__(64 * NW) void conv3x3_v2_kernel(ConvArgs a, int total_tiles, int lds_b_bytes) {
    constexpr int KC = BF16 ? 32 : 16;                    // channels per 64-byte chunk
    constexpr int ES = BF16 ? 2 : 4;
    constexpr int kTapBytes = 2 * NT * 1024;
    constexpr int TAPS = RES ? 10 : 9;                    // packed taps per K chunk (tap 9 = the 1x1 residual projection)
    constexpr int NTHR = 64 * NW;
    constexpr int TH = 2 * MTW * NW;                      // tile rows
    constexpr int PR = TH + 2;                            // patch rows
    constexpr int kA = PR * kRowPitch;
    constexpr int NPA = PR * kPatch * 4;                  // 16-byte pieces of one patch
    constexpr int AIT = (NPA + NTHR - 1) / NTHR;
    constexpr int NPB = TAPS * kTapBytes / 16;            // pieces of one weight chunk
    constexpr int BIT = BRES ? 1 : (NPB + NTHR - 1) / NTHR;
    constexpr int PPP = 32 * ES / 16;                     // 16-byte pieces per pixel of one 32-channel tile
    constexpr int OUTP = 32 * ES + 16;                    // staging pitch per pixel (one 32-channel tile at a time)
    constexpr int SROWS = FLAT ? 32 : 16;                 // pixel rows staged per pass (FLAT reads the whole M-tile back)
    constexpr int NFS = (32 / KC) * 2;                    // FLAT: weight fragments per mel row
    // fp32 (not FLAT, whose epilogue reads the staged tile back as a matrix operand): results leave straight from the accumulator registers --
    // a register is one pixel x 32 channels per half-wave = 128 contiguous bytes per store; the residual comes in the same way
    constexpr bool DIRECT = !BF16 && !FLAT;
    static_assert(NW * SROWS * OUTP <= kA, "result staging reuses the patch area");
    static_assert(!(RES && (FIRST || FLAT)), "RES is the A launch; FIRST / FLAT belong to B launches");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, m = lane & 31;
    const int py = (m >> 1) & 1, px = (m & 1) | ((m >> 2) << 1);   // lane's pixel in a 2x16 M-tile: m = (x&1) | (y<<1) | ((x>>1)<<2)
    char* sA = smem;
    char* sB = smem + kA;
    char* sO = smem + wave * (SROWS * OUTP);              // result staging: aliases the patch (used only between two barriers)
    float* sF = (float*)(sB + lds_b_bytes);               // FIRST: [PR+2][20] feature patch, then [9][32] weights + [32] bias
    float* sW = sF + (PR + 2) * 20;

    const int H = a.H, W = a.W;
    const int ngroups = a.Cout / (32 * NT);
    const int nch = (a.C0 + a.C1) / KC;                   // K chunks (the residual is not a chunk in this structure)
    const int all_taps = nch * TAPS;

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
        d.y0 = (t % a.tiles_y) * TH;
        d.n = t / a.tiles_y;
        return d;
    };

    // this thread's patch pieces: geometry is tile-independent
    u32x4 ra[AIT];
    u32x4 rb[BIT];
    constexpr int NF = ((PR + 2) * 20 + NTHR - 1) / NTHR; // FIRST: feature values per thread
    float rf[NF];
    // FIRST: this thread always produces the same 16-byte channel group (NTHR % 4 == 0), so its 9 x CPP folded weights and
    // CPP biases are loaded once into registers (the LDS copy cost 72 reads per produced piece: 63 % LDS-busy in rocprof)
    constexpr int CPPF = 16 / ES;
    float fw[FIRST ? 9 : 1][FIRST ? CPPF : 1], fbias[FIRST ? CPPF : 1];


    auto issue_loads = [&](const Tile& d, int ci) {
        if constexpr (FIRST) {
            if (ci == 0) {
                const float* fbase = a.rank1_src + (size_t)d.n * H * W;
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    const int idx = tid + NTHR * k;
                    const int fy = idx / 20, fx = idx - fy * 20;
                    const int Y = d.y0 - 2 + fy, X = d.x0 - 2 + fx;
                    rf[k] = (idx < (PR + 2) * 20 && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W) ? fbase[Y * W + X] : 0.f;
                }
            }
            return;
        }
        const int ch = ci * KC;
        const char* src; int Cs, up, c0;
        if (ch < a.C0) { src = (const char*)a.src0; Cs = a.C0; up = 0; c0 = ch; }
        else { src = (const char*)a.src1; Cs = a.C1; up = 1; c0 = ch - a.C0; }
        const int Hs = H >> up, Ws = W >> up;
        const char* base = src + ((size_t)d.n * Hs * Ws * Cs + c0) * ES;      // block-uniform; per-lane offsets are 32-bit
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int p = tid + NTHR * it;
            const int part = p & 3, pix = p >> 2;
            const int pyy = pix / kPatch, pxx = pix - pyy * kPatch;
            const int Y = d.y0 - 1 + pyy, X = d.x0 - 1 + pxx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (p < NPA && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W && !(SS_ABL(a.dbg) & 2)) {
                const int off = (((Y >> up) * Ws + (X >> up)) * Cs) * ES + part * 16;
                v = *(const u32x4*)(base + off);
            }
            ra[it] = v;
        }
        if constexpr (!BRES) {
            const char* wsrc = (const char*)a.wpk + ((size_t)d.g * all_taps + ci * TAPS) * kTapBytes;
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int p = tid + NTHR * it;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (p < NPB) v = *(const u32x4*)(wsrc + p * 16);
                rb[it] = v;
            }
        }
    };
    auto commit = [&](const Tile& d, int ci) {
        if constexpr (FIRST) {
            if (ci == 0) {
#pragma unroll
                for (int k = 0; k < NF; ++k) { const int idx = tid + NTHR * k; if (idx < (PR + 2) * 20) sF[idx] = rf[k]; }
                lds_barrier();
            }
            // h1 = relu(conv3x3(feat) + b) for the patch pixels inside the image, 0 outside (conv2's zero padding)
            constexpr int CPP = 16 / ES;                  // channels per 16-byte piece
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                const int p = tid + NTHR * it;
                const int part = p & 3, pix = p >> 2;
                const int pyy = pix / kPatch, pxx = pix - pyy * kPatch;
                if (p < NPA) {
                    const int Y = d.y0 - 1 + pyy, X = d.x0 - 1 + pxx;
                    u32x4 outv = {0u, 0u, 0u, 0u};
                    if ((unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W) {
                        float f9[9];
#pragma unroll
                        for (int t = 0; t < 9; ++t) f9[t] = sF[(pyy + t / 3) * 20 + pxx + t % 3];
                        float o[CPP];
                        if constexpr (BF16) {             // one K chunk: the channel group is fixed -> register weights
#pragma unroll
                            for (int e = 0; e < CPP; ++e) {
                                float acc1 = fbias[e];
#pragma unroll
                                for (int t = 0; t < 9; ++t) acc1 = fmaf(fw[t][e], f9[t], acc1);
                                o[e] = fmaxf(acc1, 0.f);
                            }
                        } else {                          // fp32: two K chunks (channels 0-15, 16-31): weights from LDS
                            const int ch0 = ci * KC + part * CPP;
#pragma unroll
                            for (int e = 0; e < CPP; ++e) {
                                float acc1 = sW[288 + ch0 + e];
#pragma unroll
                                for (int t = 0; t < 9; ++t) acc1 = fmaf(sW[t * 32 + ch0 + e], f9[t], acc1);
                                o[e] = fmaxf(acc1, 0.f);
                            }
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
            const int p = tid + NTHR * it;
            const int part = p & 3, pix = p >> 2;
            const int pyy = pix / kPatch, pxx = pix - pyy * kPatch;
            if (p < NPA && !(SS_ABL(a.dbg) & 8)) *(u32x4*)(sA + pyy * kRowPitch + pxx * kPixPitch + part * 16) = ra[it];
        }
        if constexpr (!BRES) {
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int p = tid + NTHR * it;
                if (p < NPB) *(u32x4*)(sB + p * 16) = rb[it];
            }
        }
    };

    int it_tile = 0;
    int tile = tile_at(0);
    if (tile < 0) return;                                 // whole block idle (block-uniform)
    Tile cur = decode(tile);

    if constexpr (BRES) {                                 // the layer's whole filter bank, once
        const char* wsrc = (const char*)a.wpk;
        for (int p = tid; p < lds_b_bytes / 16; p += NTHR) *(u32x4*)(sB + p * 16) = *(const u32x4*)(wsrc + (size_t)p * 16);
    }
    if constexpr (FIRST) {
        for (int i = tid; i < 320; i += NTHR) sW[i] = i < 288 ? a.first_w[i] : a.first_b[i - 288];
        __syncthreads();
        if constexpr (BF16) {
#pragma unroll
            for (int e = 0; e < CPPF; ++e) {
                fbias[e] = sW[288 + (tid & 3) * CPPF + e];
#pragma unroll
                for (int t = 0; t < 9; ++t) fw[t][e] = sW[t * 32 + (tid & 3) * CPPF + e];
            }
        }
    }
    issue_loads(cur, 0);
    commit(cur, 0);
    __syncthreads();

    f32x16 acc[MTW][NT];
    f32x16 racc[RES ? MTW : 1][RES ? NT : 1];
    const int aoff0 = (2 * MTW * wave + py) * kRowPitch + px * kPixPitch + (BF16 ? hh * 16 : hh * 32);
    const int boff0 = lane * 16;
    int ci = 0;
    // Timing perturbation for the tests (-DSS_DEVBUILD builds; ConvArgs::dbg bit 10, pattern in bits 11-12), as in conv4.hip.
    int jit_n = 0;
    auto jitter = [&](int site) {
#ifdef SS_DEVBUILD
        if (a.dbg & 1024) {
            const int pat = (a.dbg >> 11) & 3;
            const bool z = pat == 0 ? ((wave + site + jit_n) & 3) == 0 : pat == 1 ? wave == 0 : pat == 2 ? wave != 0 : (wave & 1) != 0;
            if (z) __builtin_amdgcn_s_sleep(32);
        }
#else
        (void)site;
#endif
    };

    while (true) {
        ++jit_n; jitter(0);
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
        const bool last = ci == nch - 1;

        // operands of the epilogue that come from memory are requested now and used after the MFMAs
        u32x4 fb[FLAT ? MTW : 1][FLAT ? NFS : 1];
        if constexpr (FLAT) {
            if (last) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int f = 0; f < NFS; ++f)
                        fb[mt][f] = *(const u32x4*)((const char*)a.flat_w + (((cur.y0 >> 1) + MTW * wave + mt) * NFS + f) * 1024 + lane * 16);
            }
        }
        // per-channel epilogue constants: requested now so that their L2 round trip overlaps the MFMAs
        float bias_v[NT], rbias_v[RES ? NT : 1], r1w_v[FIRST ? NT : 1];
        if (last) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = cur.g * 32 * NT + nt * 32 + m;
                bias_v[nt] = a.bias[co];
                if constexpr (RES) rbias_v[nt] = a.res_bias[co];
                if constexpr (FIRST) r1w_v[nt] = a.rank1_w[co];
            }
        }
        // B launches: the residual tile r comes in as it goes out, as 16-byte pieces (requested now, used after the MFMAs)
        constexpr int RPIECES = 32 * PPP / 64;            // pieces per lane for one M-tile x 32 channels
        u32x4 radd[DIRECT ? 1 : MTW][DIRECT ? 1 : NT][RPIECES];
        float rdir[DIRECT ? MTW : 1][DIRECT ? NT : 1][16];
        // C/D row (r & 3) + 8 (r >> 2) + 4 hh of an M-tile is pixel (y = (r >> 1) & 1, x = (r & 1) + 2 hh + 4 (r >> 2))
        const size_t lane_out = ((size_t)(cur.x0 + 2 * hh) * a.Cout + cur.g * 32 * NT + m) * 4;   // DIRECT: this lane's byte offset inside a tile row, register 0
        if constexpr (DIRECT) {
            if (!FIRST && !RES && a.res_in && last) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int Y = cur.y0 + 2 * MTW * wave + 2 * mt + ((r >> 1) & 1);
                            rdir[mt][nt][r] = *(const float*)((const char*)a.res_in + ((size_t)cur.n * H + Y) * W * a.Cout * 4 + lane_out + (size_t)(((r & 1) + 4 * (r >> 2)) * a.Cout + nt * 32) * 4);
                        }
            }
        } else
        if (!FIRST && !RES && a.res_in && last) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int it = 0; it < RPIECES; ++it) {
                        const int piece = lane + 64 * it;
                        const int mrow = piece / PPP, part = piece - mrow * PPP;
                        const int Y = cur.y0 + 2 * MTW * wave + 2 * mt + ((mrow >> 1) & 1), X = cur.x0 + ((mrow & 1) | ((mrow >> 2) << 1));
                        radd[mt][nt][it] = *(const u32x4*)((const char*)a.res_in + ((((size_t)cur.n * H + Y) * W + X) * a.Cout + cur.g * 32 * NT + nt * 32) * ES + part * 16);
                    }
        }
        if (ci == 0) {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if constexpr (RES) racc[i][j][r] = 0.f; }
        }
        // ---- MFMA on the staged chunk: 18 steps (9 taps x 2 sub-steps), fragments of step s+1 requested before the
        //      MFMAs of step s; RES: the centre tap's A fragments (steps 8, 9) also feed the 1x1 projection ----
        if (!(SS_ABL(a.dbg) & 4)) {
            const char* bbase = sB + boff0 + (BRES ? ci * TAPS * kTapBytes : 0);
            // fragments are requested PD-1 steps ahead of the MFMAs that use them (LDS latency under load is several MFMAs long)
            constexpr int PD = (MTW * NT <= 2) ? 4 : 2;
            u32x4 af[PD][MTW], bfr[PD][NT];
            u32x4 rfr[RES ? 2 : 1][RES ? NT : 1];
            if constexpr (RES) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) rfr[sub][nt] = *(const u32x4*)(bbase + 9 * kTapBytes + (sub * NT + nt) * 1024);
            }
            auto load_frags = [&](int st, u32x4 (&fa)[MTW], u32x4 (&fbb)[NT]) {
                const int tap = st >> 1, sub = st & 1, dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
                    fa[mt] = *(const u32x4*)(sA + aoff0 + (2 * mt + dy) * kRowPitch + dx * kPixPitch + (BF16 ? sub * 32 : sub * 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fbb[nt] = *(const u32x4*)(bbase + tap * kTapBytes + (sub * NT + nt) * 1024);
            };
#pragma unroll
            for (int st = 0; st < PD - 1; ++st) load_frags(st, af[st], bfr[st]);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + PD - 1 < 18) load_frags(st + PD - 1, af[(st + PD - 1) % PD], bfr[(st + PD - 1) % PD]);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        mma2<BF16>(acc[mt][nt], af[st % PD][mt], bfr[st % PD][nt]);
                        if constexpr (RES) { if (st == 8 || st == 9) mma2<BF16>(racc[mt][nt], af[st % PD][mt], rfr[st & 1][nt]); }
                    }
            }
        }

        // ---- last chunk of the tile: bias (+ residual) + ReLU, staged 16-byte stores, optional 2x2 max-pool / FLAT ----
        jitter(1);
        if constexpr (DIRECT) {
          if (last) {
            jitter(2);
            const bool add_r = !FIRST && !RES && a.res_in;
            const size_t cb = (size_t)a.Cout * 4;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const int Yb = cur.y0 + 2 * MTW * wave + 2 * mt;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float b = bias_v[nt];
                    const float r1w = FIRST ? r1w_v[FIRST ? nt : 0] : 0.f;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float t = acc[mt][nt][r] + b;
                        if constexpr (FIRST) t += r1w * sF[(2 * MTW * wave + 2 * mt + ((r >> 1) & 1) + 2) * 20 + (r & 1) + 2 * hh + 4 * (r >> 2) + 2];
                        if (add_r) t += rdir[mt][nt][r];
                        if (a.relu) t = fmaxf(t, 0.f);
                        v[r] = t;
                    }
                    if (!(SS_ABL(a.dbg) & 1)) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const size_t off = ((size_t)cur.n * H + Yb + ((r >> 1) & 1)) * W * cb + lane_out + (size_t)((r & 1) + 4 * (r >> 2)) * cb + nt * 128;
                            *(float*)((char*)a.out + off) = v[r];
                            if constexpr (RES) *(float*)((char*)a.res_out + off) = racc[mt][nt][r] + rbias_v[RES ? nt : 0];
                        }
                        if (a.pool_out) {                 // registers 4 q .. 4 q + 3 are one 2 x 2 quad: pooled pixel 2 q + hh of the tile's 8
                            const int Hp = H >> 1, Wp = W >> 1;
                            char* pp = (char*)a.pool_out + ((((size_t)cur.n * Hp + (Yb >> 1)) * Wp + (cur.x0 >> 1) + hh) * a.Cout + cur.g * 32 * NT + nt * 32 + m) * 4;
#pragma unroll
                            for (int q = 0; q < 4; ++q) *(float*)(pp + (size_t)(2 * q) * cb) = fmaxf(fmaxf(v[4 * q], v[4 * q + 1]), fmaxf(v[4 * q + 2], v[4 * q + 3]));
                        }
                    }
                }
            }
          }
        } else
        if (last) {
            lds_barrier();                                // all MFMA reads of the patch are done: its area becomes result staging
            jitter(2);
            const int co0 = cur.g * 32 * NT;
            const bool add_r = !FIRST && !RES && a.res_in;
            f32x16 flat_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) flat_acc[r] = 0.f;
            constexpr int PASSES = 32 / SROWS, RPP = SROWS * PPP / 64;        // staging passes per M-tile, pieces per lane per pass
            // C/D row of the 32x32 MFMA = pixel index in the M-tile: (r&3) + 8*(r>>2) + 4*hh, so registers 0..7 hold rows
            // 0..15 and registers 8..15 rows 16..31: a pass stages SROWS rows = registers [pass*SROWS/2, (pass+1)*SROWS/2)
            auto lrow_of = [&](int r, int pass) { return (r & 3) + 8 * (r >> 2) + 4 * hh - pass * SROWS; };
            auto put = [&](int lrow, float t) {
                char* dst = sO + lrow * OUTP + m * ES;
                if constexpr (BF16) *(__bf16*)dst = (__bf16)t; else *(float*)dst = t;
            };
            auto get = [&](int lrow) -> float {
                const char* src = sO + lrow * OUTP + m * ES;
                if constexpr (BF16) return (float)*(const __bf16*)src; else return *(const float*)src;
            };
            auto store_pass = [&](char* dst_tensor, int pass, int Yb, int nt) {   // staged rows -> 16-byte pieces in memory
                if (dst_tensor && !(SS_ABL(a.dbg) & 1)) {
#pragma unroll
                    for (int k = 0; k < RPP; ++k) {
                        const int piece = lane + 64 * k;
                        const int lrow = piece / PPP, part = piece - lrow * PPP;
                        const int mrow = lrow + pass * SROWS;
                        const int Y = Yb + ((mrow >> 1) & 1), X = cur.x0 + ((mrow & 1) | ((mrow >> 2) << 1));
                        const u32x4 v16 = *(const u32x4*)(sO + lrow * OUTP + part * 16);
                        *(u32x4*)(dst_tensor + ((((size_t)cur.n * H + Y) * W + X) * a.Cout + co0 + nt * 32) * ES + part * 16) = v16;   // tiles divide H, W
                    }
                }
            };
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const int Yb = cur.y0 + 2 * MTW * wave + 2 * mt;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float b = bias_v[nt];
                    const float r1w = FIRST ? r1w_v[FIRST ? nt : 0] : 0.f;
                    float v[16];
#pragma unroll
                    for (int pass = 0; pass < PASSES; ++pass) {
                        float addp[SROWS / 2];
                        if (add_r) {                      // bounce the residual pieces through the staging tile
#pragma unroll
                            for (int k = 0; k < RPP; ++k) {
                                const int piece = lane + 64 * k;
                                const int lrow = piece / PPP, part = piece - lrow * PPP;
                                *(u32x4*)(sO + lrow * OUTP + part * 16) = radd[mt][nt][pass * RPP + k];
                            }
                            wave_lds_sync();
#pragma unroll
                            for (int rr = 0; rr < SROWS / 2; ++rr) addp[rr] = get(lrow_of(pass * (SROWS / 2) + rr, pass));
                            wave_lds_sync();
                        }
#pragma unroll
                        for (int rr = 0; rr < SROWS / 2; ++rr) {
                            const int r = pass * (SROWS / 2) + rr;
                            float t = acc[mt][nt][r] + b;
                            if constexpr (FIRST) t += r1w * sF[(2 * MTW * wave + 2 * mt + ((r >> 1) & 1) + 2) * 20 + (r & 1) + 2 * hh + 4 * (r >> 2) + 2];
                            if (add_r) t += addp[rr];
                            if (a.relu) t = fmaxf(t, 0.f);
                            v[r] = t;
                            put(lrow_of(r, pass), t);
                        }
                        wave_lds_sync();
                        if constexpr (FLAT) {
                            // conv_flatten as a GEMM over channels with per-mel-row weights: the staged tile is read back as
                            // the A operand (row = this lane's pixel); the weight operand carries row Yb's filter in columns 0..3
                            // and row Yb + 1's in columns 4..7, so one product serves both rows of the M-tile and a pixel's sums
                            // are the columns of its own row (picked below); a wave's rows accumulate into one C tile.
#pragma unroll
                            for (int f = 0; f < NFS; ++f) {
                                const int cif = f >> 1, sub = f & 1;
                                const u32x4 av = *(const u32x4*)(sO + m * OUTP + cif * 64 + (BF16 ? sub * 32 + hh * 16 : hh * 32 + sub * 16));
                                mma2<BF16>(flat_acc, av, fb[mt][f]);
                            }
                        }
                        store_pass((!FLAT || a.store_out) ? (char*)a.out : nullptr, pass, Yb, nt);
                        wave_lds_sync();
                    }
                    if constexpr (RES) {                  // the residual projection leaves un-activated, with its own bias
                        const float rb2 = rbias_v[RES ? nt : 0];
#pragma unroll
                        for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
                            for (int rr = 0; rr < SROWS / 2; ++rr) {
                                const int r = pass * (SROWS / 2) + rr;
                                put(lrow_of(r, pass), racc[mt][nt][r] + rb2);
                            }
                            wave_lds_sync();
                            store_pass((char*)a.res_out, pass, Yb, nt);
                            wave_lds_sync();
                        }
                    }
                    if (a.pool_out) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) put(hh + 2 * q, fmaxf(fmaxf(v[4 * q], v[4 * q + 1]), fmaxf(v[4 * q + 2], v[4 * q + 3])));
                        wave_lds_sync();
                        const int Hp = H >> 1, Wp = W >> 1;
                        if (lane < 8 * PPP) {             // 8 pooled pixels of this M-tile x PPP pieces
                            const int pp = lane / PPP, part = lane - pp * PPP;
                            const u32x4 v16 = *(const u32x4*)(sO + pp * OUTP + part * 16);
                            *(u32x4*)((char*)a.pool_out + ((((size_t)cur.n * Hp + (Yb >> 1)) * Wp + (cur.x0 >> 1) + pp) * a.Cout + co0 + nt * 32) * ES + part * 16) = v16;
                        }
                        wave_lds_sync();
                    }
                }
            }
            if constexpr (FLAT) {
                // C tile: column = lane&31 = flatten output c (4 real), row -> (y = (r>>1)&1, x = (r&1) + 2 hh + 4 (r>>2)).
                // partial[n][row group][c][x] = sum over the wave's rows; the mask head adds the groups in order.
                // registers 4 q + 2, 4 q + 3 are the y = 1 pixels: their sums are columns 4..7, i.e. four lanes up (row_shl:4)
                float up[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (through named floats: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this compiler)
                    const float y1a = flat_acc[4 * q + 2], y1b = flat_acc[4 * q + 3];
                    up[2 * q] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y1a), 0x104, 0xf, 0xf, false));
                    up[2 * q + 1] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y1b), 0x104, 0xf, 0xf, false));
                }
                if (m < 4) {
                    const int grp = (cur.y0 + 2 * MTW * wave) / (2 * MTW);
                    float* dst = a.flat_part + (((size_t)cur.n * (H / (2 * MTW)) + grp) * 4 + m) * W + cur.x0 + 2 * hh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        dst[4 * q] = flat_acc[4 * q] + up[2 * q];
                        dst[4 * q + 1] = flat_acc[4 * q + 1] + up[2 * q + 1];
                    }
                }
            }
        }
        jitter(3);
        lds_barrier();                                    // every wave is done with this stage's LDS image (and staging)
        if (!has_next) break;
        jitter(4);
        commit(nxt, ci_n);                                // (the compiler waits for exactly the prefetch loads it writes)
        jitter(5);
        lds_barrier();
        tile = tile_n; cur = nxt; ci = ci_n;
    }
}

template <bool BF16, int NT, int MTW, int NW, bool BRES, bool RES, bool FIRST, bool FLAT>
static hipError_t launch_v2_t(const ConvArgs& a, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};              // (per device: kernels.h allow_full_lds)
    if (hipError_t e = allow_full_lds((const void*)conv3x3_v2_kernel<BF16, NT, MTW, NW, BRES, RES, FIRST, FLAT>, attr_done)) return e;
    hipLaunchKernelGGL((conv3x3_v2_kernel<BF16, NT, MTW, NW, BRES, RES, FIRST, FLAT>), dim3(grid), dim3(64 * NW), lds, s, a, total, lds_b);
    return hipGetLastError();
}

// WITH_RES = false: the geometry is never given an A launch (r_out) -- its RES instantiations are not built.  That is the case for
// 4 waves x 2 M-tiles with three channel tiles: two accumulator sets x 3 x 2 = 192 accumulator registers, 96-152 bytes of scratch per lane
// (VERDICT r03 item 5); the product sends those launches to 8 waves x 1 M-tile (below), only the dev build's switches reach the other form.
template <bool BF16, int NT, int MTW, int NW, bool WITH_RES = true>
static hipError_t launch_v2_geo(const ConvArgs& a, bool bres, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    if constexpr (NT == 1 && MTW * NW == 8) {             // FIRST / FLAT exist for the 32-channel full-resolution layers only
        if (a.first_w) return launch_v2_t<BF16, 1, MTW, NW, true, false, true, false>(a, total, lds_b, lds, grid, s);
        if (a.flat_part) return launch_v2_t<BF16, 1, MTW, NW, true, false, false, true>(a, total, lds_b, lds, grid, s);
    }
    if constexpr (WITH_RES) {
        if (a.res_out) return bres ? launch_v2_t<BF16, NT, MTW, NW, true, true, false, false>(a, total, lds_b, lds, grid, s)
                                   : launch_v2_t<BF16, NT, MTW, NW, false, true, false, false>(a, total, lds_b, lds, grid, s);
    } else if (a.res_out) return hipErrorInvalidValue;
    return bres ? launch_v2_t<BF16, NT, MTW, NW, true, false, false, false>(a, total, lds_b, lds, grid, s)
                : launch_v2_t<BF16, NT, MTW, NW, false, false, false, false>(a, total, lds_b, lds, grid, s);
}

static bool v2_fp32_nt3_a_as_8_waves() { static const int e = dev_env("SOFTSPOKEN_FP32_A8", 1); return e != 0; }

// geometry: 16-row tiles as 8 waves x 1 M-tile in bf16 (4 x 2 in fp32 or with SOFTSPOKEN_NW=4), 8-row tiles (H == 8) as 4 x 1
template <bool BF16, int NT>
static hipError_t launch_v2_nt(const ConvArgs& a, int th, int nw, bool bres, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    if (th == 8) return launch_v2_geo<BF16, NT, 1, 4>(a, bres, total, lds_b, lds, grid, s);
    if constexpr (BF16) {
        if (nw == 8) return launch_v2_geo<true, NT, 1, 8>(a, bres, total, lds_b, lds, grid, s);
    }
    if (nw != 4) return hipErrorInvalidValue;
    if constexpr (!BF16 && NT == 3) {
        // fp32 A launches of the 96-channel blocks: two accumulator sets (h and the projection) x 3 channel tiles x 2 M-tiles = 192 registers
        // spilled at the 256-register cap (96-152 bytes of scratch per lane); as 8 waves x 1 M-tile they need 96
        if (a.res_out && v2_fp32_nt3_a_as_8_waves()) return launch_v2_geo<false, 3, 1, 8>(a, bres, total, lds_b, lds, grid, s);
    }
#ifdef SS_DEVBUILD
    return launch_v2_geo<BF16, NT, 2, 4>(a, bres, total, lds_b, lds, grid, s);
#else
    // (product: three channel tiles never reach this geometry with an A launch -- fp32 went to 8 x 1 above, bf16 runs 8 waves)
    return launch_v2_geo<BF16, NT, 2, 4, NT != 3>(a, bres, total, lds_b, lds, grid, s);
#endif
}


// bf16: 8 waves (issue-bound overhead code wants the waves); fp32: 4 waves (MFMA-bound, and the fp32 staging tile is 2x)
static int waves_per_block_16(bool bf16) {
    static const int nw_env = dev_env("SOFTSPOKEN_NW", 8);
    return (bf16 && nw_env != 4) ? 8 : 4;
}

// row groups of the FLAT partial sums for a 128-row image: one per wave of a 16-row tile
int conv_v2_flat_groups(bool bf16) { return 128 / (16 / waves_per_block_16(bf16)); }

struct V2Choice { bool ok; int th, nw, total, lds_b, grid; bool bres; size_t lds; };

static V2Choice choose_v2(ConvArgs& a, bool bf16, int NT, int num_cus) {
    V2Choice c{};
    if (a.W % 16 != 0 || a.H % 8 != 0 || a.Cout % (32 * NT) != 0 || NT < 1 || NT > 3) return c;
    const int kc = bf16 ? 32 : 16;
    if (a.C0 % kc || a.C1 % kc || a.R0 || a.R1) return c;                         // residual chunks belong to the first structure
    // tile rows: 16 (8 at the 8x16 level).  A 32-row tile (two M-tiles per wave, B fragments shared) measured 5-15 % slower.
    c.th = (a.H % 16 == 0) ? 16 : 8;
    c.nw = c.th == 16 ? waves_per_block_16(bf16) : 4;
    a.tiles_y = a.H / c.th; a.tiles_x = a.W / 16;
    const int ngroups = a.Cout / (32 * NT);
    const long total_l = (long)a.N * a.tiles_y * a.tiles_x * ngroups;
    if (total_l <= 0 || total_l > 0x7fffffff) return c;
    c.total = (int)total_l;
    const int tap_bytes = 2 * NT * 1024;
    const int taps = a.res_out ? 10 : 9;
    const int all_taps = ((a.C0 + a.C1) / kc) * taps;
    c.bres = ngroups == 1 && (size_t)all_taps * tap_bytes <= 72 * 1024;          // (streaming instead measured the same)
    c.lds_b = c.bres ? all_taps * tap_bytes : taps * tap_bytes;
    if ((a.first_w || a.flat_part) && !(NT == 1 && c.th == 16 && c.bres)) return c;
    c.lds = (size_t)(c.th + 2) * kRowPitch + c.lds_b + (a.first_w ? (size_t)((c.th + 4) * 20 + 320) * 4 : 0);
    int bpc = (int)((160 * 1024) / c.lds);
    if (bpc < 1) return c;
    if (bpc > 3) bpc = 3;
    c.grid = num_cus * bpc;
    if (c.grid > c.total) c.grid = c.total;
    c.grid = (c.grid + 7) / 8 * 8;                        // the tile map needs a multiple of 8 blocks (idle ones return at once)
    c.ok = true;
    return c;
}

// template arguments of the instantiation a launch will use, as rocprofv3 prints them:
// conv3x3_v2_kernel<BF16, NT, MTW, NW, BRES, RES, FIRST, FLAT>
const char* conv_v2_variant(const ConvArgs& a_in, bool bf16, int NT, int num_cus) {
    static thread_local char buf[96];
    ConvArgs a = a_in;
    const V2Choice c = choose_v2(a, bf16, NT, num_cus);
    if (!c.ok) return "conv3x3_v2_kernel<invalid>";
    const bool a8 = !bf16 && NT == 3 && c.th == 16 && a.res_out && v2_fp32_nt3_a_as_8_waves();
    const int mtw = c.th == 8 ? 1 : ((c.nw == 8 || a8) ? 1 : 2);
    const bool first = NT == 1 && c.th == 16 && a.first_w, flat = NT == 1 && c.th == 16 && !first && a.flat_part;
    const bool res = !first && !flat && a.res_out;
    const bool bres = (first || flat) ? true : c.bres;
    auto tf = [](bool b) { return b ? "true" : "false"; };
    snprintf(buf, sizeof buf, "conv3x3_v2_kernel<%s, %d, %d, %d, %s, %s, %s, %s>", tf(bf16), NT, mtw, a8 ? 8 : c.nw, tf(bres), tf(res), tf(first), tf(flat));
    return buf;
}

hipError_t launch_conv3x3_v2(const ConvArgs& a_in, bool bf16, int NT, int num_cus, hipStream_t s) {
    ConvArgs a = a_in;
    const V2Choice ch = choose_v2(a, bf16, NT, num_cus);
    if (!ch.ok) return hipErrorInvalidValue;
    const int th = ch.th, nw = ch.nw, total = ch.total, lds_b = ch.lds_b, grid = ch.grid;
    const bool bres = ch.bres;
    const size_t lds = ch.lds;
    if (bf16) {
        switch (NT) {
            case 1: return launch_v2_nt<true, 1>(a, th, nw, bres, total, lds_b, lds, grid, s);
            case 2: return launch_v2_nt<true, 2>(a, th, nw, bres, total, lds_b, lds, grid, s);
            case 3: return launch_v2_nt<true, 3>(a, th, nw, bres, total, lds_b, lds, grid, s);
        }
    } else {
        switch (NT) {
            case 1: return launch_v2_nt<false, 1>(a, th, nw, bres, total, lds_b, lds, grid, s);
            case 2: return launch_v2_nt<false, 2>(a, th, nw, bres, total, lds_b, lds, grid, s);
            case 3: return launch_v2_nt<false, 3>(a, th, nw, bres, total, lds_b, lds, grid, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace ss
