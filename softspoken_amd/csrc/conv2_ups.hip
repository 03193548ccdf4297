// fp32 mode: the A launch of a decoder ResBlock with its nearest-upsampled input half computed at low resolution.
//
//   reference: SpecUNet_2D.forward, root/code/backend/pytorch_neural_nets.py:171-181 -- x = cat[skip, Upsample(2, nearest)(below)],
//   then ResBlock (pytorch_neural_nets.py:7-41): h = relu(BN(conv3x3(x))), r = BN(conv1x1(x)).
//
// The upsampled half puts the SAME low-resolution pixel under two of a 3x3's three rows (and columns): for an output pixel of
// parity class (py, px) = (Y & 1, X & 1) the nine taps on those channels collapse, exactly, into four taps on the low-resolution
// tensor whose weights are sums of the original ones (weights.hip pack_conv_v2_ups; conv4_ups.hip is the same idea for f16x2).
// 9 C0 + 4 C1 multiply-adds per output value instead of 9 (C0 + C1): conv6.A / conv7.A / conv8.A / conv9_1.A issue 28 % fewer fp32
// matrix instructions, the mode's whole pass 10 % fewer.  fp32 on v_mfma_f32_32x32x2_f32 is matrix-bound (conv2.hip's launches
// sit at 100-108 TFLOP/s of a 157 TFLOP/s peak), so the products removed are time removed.
//
// Launch table -- the product's one form, conv3x3_ups32_kernel<NT = 1, MTW = 2>; Cout / 32 output-channel groups are separate tiles:
//   block    256 threads = 4 waves over one 16 x 16 output tile x 32 output channels; wave w works the 64 pixels of parity class w
//            (py = w >> 1, px = w & 1) as two M-tiles of 32 (low-resolution rows 0-3 and 4-7 of the tile, all 8 columns): an M-tile
//            holds ONE class, because an MFMA shares its weight operand between all of its pixels, and the two M-tiles of a wave share
//            every weight fragment.  Lane l: pixel row (l & 31) >> 3, column l & 7, K half l >> 5.
//   stages   per tile C0 / 16 skip chunks (18 x 18 patch of the skip tensor, 9 taps + the 1x1 projection's) then C1 / 16
//            upsampled chunks (10 x 10 patch of the low-resolution tensor, 4 classes x 4 pre-summed taps + the projection's);
//            the next stage's patch and weights are in flight in registers during this stage's matrix instructions (conv2.hip's
//            pipeline: barriers wait on LDS only); staging rounds are whole patch rows so that tile, chunk and round move scalar bases.
//   LDS      patch 18 x 1664 B = 29 952 B + weights of one stage 17 x 2 KB = 64.8 KB -> 2 blocks per CU, 256 registers per wave.
//   grid     persistent: 2 x num_cus blocks, rounded to the 8 XCDs; each XCD owns a contiguous tile range.
//   results  h = relu(. + b1) -> out, r = conv1x1(x) + br -> res_out: an accumulator register is one pixel x 32 channels per half-wave,
//            stored as it is (128 contiguous bytes per half-wave and store): no transposition through LDS.
//   measured (MI355X, 1005 windows, per launch): conv9_1.A 11.93 -> 9.43 ms, conv8.A 5.78 -> 4.54, conv7.A 4.13 -> 3.32, conv6.A 1.95 -> 1.65
//            against conv2.hip's nine-tap form; 106 TFLOP/s issued on the fp32 matrix instruction (157 at 2.4 GHz).
//   development build only: 8 waves x 1 M-tile (one block per CU: 9.50 -> 10.2 ms on conv9_1.A), NT = 2 / 3 (spill: 4-229 registers).
#include "kernels.h"

namespace ss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kPixPitch = 80;            // conv2.hip's patch image: 64 B of channels + 16 B pad per pixel, 1664 B per row
constexpr int kRowPitch = 1664;
constexpr int kPatchRows = 18;
constexpr int kA = kPatchRows * kRowPitch;

__device__ __forceinline__ void mma8(f32x16& acc, const u32x4& a, const u32x4& b) {   // K = 8: four v_mfma_f32_32x32x2_f32
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace

template <int NT, int MTW>
__global__ __launch_bounds__(512 / MTW) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_ups32_kernel(ConvArgs a, int total_tiles) {
    constexpr int NW = 8 / MTW, NTHR = 64 * NW;           // 8 M-tiles of 32 pixels: 4 parity classes x 2 halves of the tile's rows
    constexpr int kTapBytes = 2 * NT * 1024;
    constexpr int kTapsSkip = 10, kTapsUps = 17;
    constexpr int NPB_S = kTapsSkip * kTapBytes / 16, NPB_U = kTapsUps * kTapBytes / 16;
    constexpr int BIT = (NPB_U + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the class is a wave's)
    const int hh = lane >> 5, m = lane & 31;
    // M-tile MTW wave + mt: class (MTW wave + mt) >> 1, half (MTW wave + mt) & 1 -- with two M-tiles per wave they are the two halves of one class
    const int cls = (MTW * wave) >> 1, half0 = (MTW * wave) & 1, py = cls >> 1, px = cls & 1;
    const int li = 4 * half0 + (m >> 3), lj = m & 7;      // this lane's pixel (of its first M-tile), low-resolution coordinates inside the tile
    char* sA = smem;
    char* sB = smem + kA;

    const int H = a.H, W = a.W, C0 = a.C0, C1 = a.C1;
    const int n0 = C0 / 16, nch = n0 + C1 / 16;
    const int Hs = H >> 1, Ws = W >> 1;
    const int group_taps = n0 * 10 + (nch - n0) * 17;     // tap blocks of one output-channel group's bank

    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, gper = gridDim.x >> 3;
    const int per = (total_tiles + 7) >> 3;
    auto tile_at = [&](int it) -> int {
        const int idx = local + it * gper;
        const int t = xcd * per + idx;
        return (idx < per && t < total_tiles) ? t : -1;
    };
    const int ngroups = a.Cout / (32 * NT);
    struct Tile { int n, y0, x0, g; };
    auto decode = [&](int t) -> Tile {
        Tile d;
        d.g = t % ngroups; t /= ngroups;
        d.x0 = (t % a.tiles_x) * 16; t /= a.tiles_x;
        d.y0 = (t % a.tiles_y) * 16;
        d.n = t / a.tiles_y;
        return d;
    };

    // Staging work of a thread, tile-independent: a round of the skip patch is RPS whole patch rows of 72 pieces (18 pixels x 4), a
    // round of the low-resolution patch RPU rows of 40; the tile, the chunk and the round only move SCALAR bases (few live registers:
    // with per-piece address arithmetic the compiler kept ~90 loop-invariant registers and the two-blocks-per-CU form spilled)
    constexpr int RPS = NTHR / 72, SIT = (kPatchRows + RPS - 1) / RPS;
    constexpr int RPU = NTHR / 40, UIT = (10 + RPU - 1) / RPU;
    // All global traffic goes through buffer descriptors: a thread keeps ONE 32-bit offset per kind of piece, whatever moves with the
    // tile, the chunk or the round is the instruction's scalar offset, and a piece outside the image gets an offset beyond the
    // descriptor's range (the hardware returns zeros: no branch, no zero-fill).  As flat pointers the compiler kept 64-bit per-piece
    // addresses alive across the stage, spilled prefetch registers around them and waited on vmcnt(0) in the middle of the prefetch.
    // A patch's descriptor starts one row and one pixel BEFORE its window so that no scalar offset is negative.
    // (The per-thread geometry is recomputed where it is used -- a dozen vector instructions per stage beside ~5000 cycles of matrix
    // work -- instead of living in registers across the stage: `fresh` keeps the compiler from hoisting it.)
    constexpr int kRange = 0x40000000, kOutside = 0x7fff0000;
    auto rsrc_of = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, kRange, 0x00020000); };
    auto fresh = [](int v) { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return v + z; };
    struct Geo { int r, x, lds, glb; };
    auto geo_skip = [&](int t) { Geo g; g.r = t / 72; const int c = t - g.r * 72; g.x = c >> 2;
                                 g.lds = g.r * kRowPitch + g.x * kPixPitch + (c & 3) * 16; g.glb = (g.r * W + g.x) * C0 * 4 + (c & 3) * 16; return g; };
    auto geo_ups = [&](int t) { Geo g; g.r = t / 40; const int c = t - g.r * 40; g.x = c >> 2;
                                g.lds = g.r * kRowPitch + g.x * kPixPitch + (c & 3) * 16; g.glb = (g.r * Ws + g.x) * C1 * 4 + (c & 3) * 16; return g; };
    u32x4 ra[SIT], ru[UIT];                               // prefetched pieces of a skip / of a low-resolution patch (never both)
    u32x4 rb[BIT];
    const __amdgpu_buffer_rsrc_t rs_w = rsrc_of(a.wpk);
    auto issue_loads = [&](const Tile& d, int ci) {
        const bool ups = ci >= n0;
        if (!ups) {
            const __amdgpu_buffer_rsrc_t rs = rsrc_of((const char*)a.src0 + ((int64_t)d.n * H * W - (W + 1)) * C0 * 4);
            const int soff = (d.y0 * W + d.x0) * C0 * 4 + ci * 64;
            const Geo g = geo_skip(fresh(tid));
            const bool xok = g.r < RPS && (unsigned)(d.x0 - 1 + g.x) < (unsigned)W;
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                const int row = it * RPS + g.r;
                const bool ok = xok && row < kPatchRows && (unsigned)(d.y0 - 1 + row) < (unsigned)H;
                ra[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? g.glb : kOutside, soff + it * RPS * W * C0 * 4, 0);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = rsrc_of((const char*)a.src1 + ((int64_t)d.n * Hs * Ws - (Ws + 1)) * C1 * 4);
            const int soff = ((d.y0 >> 1) * Ws + (d.x0 >> 1)) * C1 * 4 + (ci - n0) * 64;
            const Geo g = geo_ups(fresh(tid));
            const bool xok = g.r < RPU && (unsigned)((d.x0 >> 1) - 1 + g.x) < (unsigned)Ws;
#pragma unroll
            for (int it = 0; it < UIT; ++it) {
                const int row = it * RPU + g.r;
                const bool ok = xok && row < 10 && (unsigned)((d.y0 >> 1) - 1 + row) < (unsigned)Hs;
                ru[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? g.glb : kOutside, soff + it * RPU * Ws * C1 * 4, 0);
            }
        }
        const int wsoff = (d.g * group_taps + (ups ? n0 * kTapsSkip + (ci - n0) * kTapsUps : ci * kTapsSkip)) * kTapBytes;
        const int np = ups ? NPB_U : NPB_S;
#pragma unroll
        for (int it = 0; it < BIT; ++it)
            rb[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, tid + NTHR * it < np ? tid * 16 : kOutside, wsoff + it * NTHR * 16, 0);
    };
    auto commit = [&](int ci) {
        const bool ups = ci >= n0;
        if (!ups) {
            const Geo g = geo_skip(fresh(tid));
#pragma unroll
            for (int it = 0; it < SIT; ++it)
                if (g.r < RPS && it * RPS + g.r < kPatchRows) *(u32x4*)(sA + it * RPS * kRowPitch + g.lds) = ra[it];
        } else {
            const Geo g = geo_ups(fresh(tid));
#pragma unroll
            for (int it = 0; it < UIT; ++it)
                if (g.r < RPU && it * RPU + g.r < 10) *(u32x4*)(sA + it * RPU * kRowPitch + g.lds) = ru[it];
        }
        const int np = ups ? NPB_U : NPB_S;
#pragma unroll
        for (int it = 0; it < BIT; ++it)
            if (tid + NTHR * it < np) *(u32x4*)(sB + it * NTHR * 16 + tid * 16) = rb[it];
    };

    int it_tile = 0;
    int tile = tile_at(0);
    if (tile < 0) return;                                 // whole block idle (block-uniform)
    Tile cur = decode(tile);
    issue_loads(cur, 0);
    commit(0);
    __syncthreads();

    f32x16 acc[MTW][NT], racc[MTW][NT];
    // skip chunks: the lane's pixel at full resolution (2 li + py, 2 lj + px), patch origin (-1, -1);
    // upsampled chunks: tap (ty, tx) of class (py, px) reads low-resolution pixel (li + py - 1 + ty, lj + px - 1 + tx), patch origin (-1, -1);
    // a wave's second M-tile lies four low-resolution rows further down
    const int aoff_s = (2 * li + py) * kRowPitch + (2 * lj + px) * kPixPitch + hh * 32;
    const int aoff_u = (li + py) * kRowPitch + (lj + px) * kPixPitch + hh * 32;
    const int boff = lane * 16;
    const int tc = (1 - py) * 2 + (1 - px);               // the tap of the 2 x 2 that lies under the output pixel itself
    int ci = 0;
    constexpr int PD = MTW * NT == 1 ? 4 : 2;             // fragments are requested PD - 1 steps ahead
    // Timing perturbation for the tests (-DSS_DEVBUILD builds; ConvArgs::dbg bit 10, pattern in bits 11-12), as in conv2.hip / conv4.hip:
    // some waves sleep at the synchronisation points, so that a missing barrier shows up as different bits
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
        float bias_v[NT], rbias_v[NT];
        if (last) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bias_v[nt] = a.bias[(cur.g * NT + nt) * 32 + m]; rbias_v[nt] = a.res_bias[(cur.g * NT + nt) * 32 + m]; }
        }
        if (ci == 0) {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; racc[i][j][r] = 0.f; }
        }
        u32x4 af[PD][MTW], bfr[PD][NT];
        if (ci < n0) {
            // ---- skip chunk: 9 taps x 2 sub-steps; the centre tap's pixel fragments also feed the 1x1 projection ----
            auto load_frags = [&](int st, u32x4 (&fa)[MTW], u32x4 (&fb)[NT]) {
                const int tap = st >> 1, sub = st & 1, dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) fa[mt] = *(const u32x4*)(sA + aoff_s + (8 * mt + dy) * kRowPitch + dx * kPixPitch + sub * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fb[nt] = *(const u32x4*)(sB + boff + tap * kTapBytes + (sub * NT + nt) * 1024);
            };
#pragma unroll
            for (int st = 0; st < PD - 1; ++st) load_frags(st, af[st], bfr[st]);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + PD - 1 < 18) load_frags(st + PD - 1, af[(st + PD - 1) % PD], bfr[(st + PD - 1) % PD]);
                u32x4 rfr[NT];                           // the projection's fragments: wanted for two of the 18 steps, not kept
                if (st == 8 || st == 9) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) rfr[nt] = *(const u32x4*)(sB + boff + 9 * kTapBytes + ((st & 1) * NT + nt) * 1024);
                }
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma8(acc[mt][nt], af[st % PD][mt], bfr[st % PD][nt]);
                if (st == 8 || st == 9) {
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) mma8(racc[mt][nt], af[st % PD][mt], rfr[nt]);
                }
                __builtin_amdgcn_sched_barrier(0);       // (the fragment ring above IS the schedule: nothing is hoisted over a step)
            }
        } else {
            // ---- upsampled chunk: this wave's class, 4 pre-summed taps x 2 sub-steps; the projection reads tap tc's pixel ----
            const char* bcls = sB + boff + cls * 4 * kTapBytes;
            auto load_frags = [&](int st, u32x4 (&fa)[MTW], u32x4 (&fb)[NT]) {
                const int tap = st >> 1, sub = st & 1, ty = tap >> 1, tx = tap & 1;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) fa[mt] = *(const u32x4*)(sA + aoff_u + (4 * mt + ty) * kRowPitch + tx * kPixPitch + sub * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fb[nt] = *(const u32x4*)(bcls + tap * kTapBytes + (sub * NT + nt) * 1024);
            };
#pragma unroll
            for (int st = 0; st < PD - 1; ++st) load_frags(st, af[st], bfr[st]);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (st + PD - 1 < 8) load_frags(st + PD - 1, af[(st + PD - 1) % PD], bfr[(st + PD - 1) % PD]);
                const bool centre = (st >> 1) == tc;      // wave-uniform
                u32x4 rfr[NT];
                if (centre) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) rfr[nt] = *(const u32x4*)(sB + boff + 16 * kTapBytes + ((st & 1) * NT + nt) * 1024);
                }
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma8(acc[mt][nt], af[st % PD][mt], bfr[st % PD][nt]);
                if (centre) {
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) mma8(racc[mt][nt], af[st % PD][mt], rfr[nt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if (last) {
            // C/D of the 32x32 tile: column = lane & 31 = output channel, row (r & 3) + 8 (r >> 2) + 4 hh = pixel of the M-tile, i.e.
            // low-resolution row r >> 2, column (r & 3) + 4 hh.  A register goes out as it is: 32 lanes x 4 bytes = one pixel's 128
            // contiguous bytes per half-wave and store -- whole lines, no transposition through LDS, no barrier before the epilogue.
            // (descriptor per window; the lane's offset is tile-independent, rows / pixels / tiles move the scalar offset)
            const int cb = a.Cout * 4;
            const __amdgpu_buffer_rsrc_t rs_o = rsrc_of((char*)a.out + (int64_t)cur.n * H * W * cb), rs_r = rsrc_of((char*)a.res_out + (int64_t)cur.n * H * W * cb);
            const int lane_off = (8 * hh + px) * cb + m * 4;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int so = ((cur.y0 + 8 * (half0 + mt) + py) * W + cur.x0) * cb + (cur.g * NT + nt) * 128;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float t = acc[mt][nt][r] + bias_v[nt];
                        if (a.relu) t = fmaxf(t, 0.f);
                        const int sr = so + ((r >> 2) * W + (r & 3)) * 2 * cb;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, t), rs_o, lane_off, sr, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, racc[mt][nt][r] + rbias_v[nt]), rs_r, lane_off, sr, 0);
                    }
                }
        }
        jitter(1);
        lds_barrier();                                    // every wave is done with this stage's LDS image
        if (!has_next) break;
        jitter(2);
        commit(ci_n);
        jitter(3);
        lds_barrier();
        tile = tile_n; cur = nxt; ci = ci_n;
    }
}

namespace {

struct Ups32Choice { bool ok; int NT, MTW, total, grid; size_t lds; };

Ups32Choice choose_ups32(ConvArgs& a, int NT, int MTW, int num_cus) {
    Ups32Choice c{};
    if (NT < 1 || NT > 3 || a.Cout % (32 * NT) || (MTW != 1 && MTW != 2) || (MTW == 2 && NT == 3)) return c;
    c.MTW = MTW;
    if (!a.src0 || !a.src1 || !a.wpk || !a.out || !a.res_out || !a.bias || !a.res_bias) return c;
    if (a.first_w || a.flat_part || a.pool_out || a.res_in || a.rank1_src || a.R0 || a.R1) return c;
    if (a.H % 16 || a.W % 16 || a.C0 < 16 || a.C1 < 16 || a.C0 % 16 || a.C1 % 16) return c;
    c.NT = NT;
    a.tiles_y = a.H / 16; a.tiles_x = a.W / 16;
    const long total_l = (long)a.N * a.tiles_y * a.tiles_x * (a.Cout / (32 * NT));
    if (total_l <= 0 || total_l > 0x7fffffff) return c;
    // per-lane byte offsets inside one window's tensors are 32-bit
    if ((long)a.H * a.W * a.C0 * 4 > 0x7fffffffL || (long)a.H * a.W * a.C1 * 4 > 0x7fffffffL) return c;
    c.total = (int)total_l;
    c.lds = (size_t)kA + (size_t)17 * 2 * c.NT * 1024;
    int bpc = (int)((160 * 1024) / c.lds);
    if (bpc < 1) return c;
    if (bpc > 2) bpc = 2;
    if (MTW == 1) bpc = 1;                                // (8-wave blocks with ~190 registers: one per CU)
#ifndef SS_DEVBUILD
    if (NT != 1 || MTW != 2) return c;                    // (the product has the one form)
#endif
    c.grid = num_cus * bpc;
    if (c.grid > c.total) c.grid = c.total;
    c.grid = (c.grid + 7) / 8 * 8;                        // the tile map needs a multiple of 8 blocks (idle ones return at once)
    c.ok = true;
    return c;
}

template <int NT, int MTW>
hipError_t launch_ups32_t(const ConvArgs& a, const Ups32Choice& ch, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = allow_full_lds((const void*)conv3x3_ups32_kernel<NT, MTW>, attr_done)) return e;
    hipLaunchKernelGGL((conv3x3_ups32_kernel<NT, MTW>), dim3(ch.grid), dim3(512 / MTW), ch.lds, s, a, ch.total);
    return hipGetLastError();
}

}  // namespace

bool conv_ups32_supports(const ConvArgs& a_in, int NT, int MTW, int num_cus) {
    ConvArgs a = a_in;
    return choose_ups32(a, NT, MTW, num_cus).ok;
}

size_t conv_ups32_weight_bytes(int C0, int C1, int Cout) {      // (the same for every NT: [group][chunk][tap][sub-step][tile][lane][4])
    return (size_t)((C0 / 16) * 10 + (C1 / 16) * 17) * 2 * (Cout / 32) * 1024;
}

const char* conv_ups32_variant(int NT, int MTW) {
    static const char* const names[2][3] = {{"conv3x3_ups32_kernel<1, 1>", "conv3x3_ups32_kernel<2, 1>", "conv3x3_ups32_kernel<3, 1>"},
                                            {"conv3x3_ups32_kernel<1, 2>", "conv3x3_ups32_kernel<2, 2>", "conv3x3_ups32_kernel<invalid>"}};
    return (NT >= 1 && NT <= 3 && (MTW == 1 || MTW == 2)) ? names[MTW - 1][NT - 1] : "conv3x3_ups32_kernel<invalid>";
}

hipError_t launch_conv3x3_ups32(const ConvArgs& a_in, int NT, int MTW, int num_cus, hipStream_t s) {
    ConvArgs a = a_in;
    const Ups32Choice ch = choose_ups32(a, NT, MTW, num_cus);
    if (!ch.ok) return hipErrorInvalidValue;
    // The product runs one 32-channel tile per block and two M-tiles per wave (wider blocks run as Cout / 32 groups of tiles): with
    // more tiles the prefetch registers of the 34 KB-per-tile upsampled stage do not fit beside the accumulators (4-229 registers
    // spilled).  The other forms exist in the development build for the comparison (SOFTSPOKEN_UPS32_NT / SOFTSPOKEN_UPS32_MTW).
    if (ch.NT == 1 && MTW == 2) return launch_ups32_t<1, 2>(a, ch, s);
#ifdef SS_DEVBUILD
    if (ch.NT == 1 && MTW == 1) return launch_ups32_t<1, 1>(a, ch, s);
    if (ch.NT == 2 && MTW == 1) return launch_ups32_t<2, 1>(a, ch, s);
    if (ch.NT == 2 && MTW == 2) return launch_ups32_t<2, 2>(a, ch, s);
    if (ch.NT == 3 && MTW == 1) return launch_ups32_t<3, 1>(a, ch, s);
#endif
    return hipErrorInvalidValue;
}

}  // namespace ss
