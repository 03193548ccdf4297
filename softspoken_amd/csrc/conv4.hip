// conv3x3 implicit GEMM, third structure (bf16): the second structure's stage loop with its vector-instruction count cut.
//
// Why: the second structure (conv2.hip) spends 15 vector + 4-5 LDS instructions per MFMA (ISA count of its stage loop),
// and on CDNA4 an MFMA 32x32x16 leaves room for about six other vector-issue slots (MI355X_MICROARCH.md, "vector-
// instruction ISSUE cost"): the SIMDs' issue ports, not the matrix pipes, HBM or LDS, set its 30 % MFMA utilisation.
// Where those instructions were and what replaces them:
//   * epilogue (11 of the 15): the 32x32 accumulator tile had rows = pixels, cols = output channels, so one lane held 16
//     pixels of ONE channel: every value was biased, optionally residual-added, relu'd behind runtime selects, converted
//     alone and written to an LDS staging tile as 2 bytes, to come back as 16-byte pieces.  Here the MFMA operands are
//     swapped (weights are the A operand, pixels the B operand): a lane holds 4 consecutive channels x 4 groups of ONE
//     pixel.  Bias enters as the accumulator's initial value (C operand, read from LDS), two values convert per
//     v_cvt_pk_bf16_f32, ReLU is one v_pk_max_i16 per pair (bf16 sign = int16 sign), v_permlane32_swap joins the two
//     half-waves' channel groups into 16-byte runs, and results go from registers to memory: no staging tile, no
//     epilogue barrier.  The residual tile and the 2x2 max-pool use the same layout (pool: quad DPP moves; the lane ->
//     pixel map keeps each 2x2 window in one quad).
//   * patch loads (3 of the 15): per-piece coordinates, bounds tests and 64-bit addresses every stage.  Here a thread's
//     piece geometry is computed once (pixel index relative to the patch origin, LDS offset, 5 edge flag bits); per stage a
//     piece costs v_mad_u32_u24 (offset), v_and + v_cmp (flags against the tile's edge mask, a scalar) and v_cndmask:
//     out-of-image pieces read offset 0, a 256-byte zero header the engine keeps in front of every activation tensor.
// LDS image, fragment packing, XCD tile map, register prefetch of the next stage and LDS-only barriers are conv2.hip's.
// fp32 stays on conv2.hip.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace ss {

// development switches exist in the dev build only (engine.h has the same helper for the host units)
#ifdef SS_DEVBUILD
static int dev_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
static constexpr int dev_env(const char*, int dflt) { return dflt; }
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

static constexpr int kPixPitch = 80;     // as conv2.hip
static constexpr int kRowPitch = 1664;
static constexpr int kPatch = 18;
// Waves may raise their priority for the MFMA loop (ConvArgs::dbg bit 5, SOFTSPOKEN_PRIO).  A same-box A/B of the bit shows no
// difference (30.6 vs 30.7 k audio-s/s, twice); builds with and without the instruction differed by +-5 % per launch in both
// directions, i.e. what moves is the compiler's schedule around it, not the hardware arbitration.
static constexpr int kMfmaPrio = 2;
static constexpr int kHdr = 256;         // zero bytes in front of every activation tensor (engine.hip ensure_workspace)

__device__ __forceinline__ void lds_barrier4() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ uint32_t relu_pk(uint32_t v) {            // bf16 pair: negative <=> int16 negative
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), s16x2{0, 0}));
}
__device__ __forceinline__ uint32_t max_pk(uint32_t a, uint32_t b) { // valid for non-negative bf16 (after ReLU)
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
// f16x2 mode: a value is the sum of two f16 halves.  hi = f16(v) (round to nearest), lo = f16(v - hi): v - hi is exact in fp32, so
// the pair carries ~22 significant bits; small low halves are f16 subnormals, which the matrix instruction keeps.
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
__device__ __forceinline__ f32x2 unpack_f16(uint32_t v) { return __builtin_convertvector(__builtin_bit_cast(f16x2, v), f32x2); }
// the largest high half seen so far, per 16-bit lane (the values are >= 0 behind the ReLU: as unsigned integers they order like the
// values, infinity and NaN on top): one instruction per pair; the test for "all exponent bits set" happens once, on the maximum
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// lo = f16(x - hi) of a pair whose high halves are packed in `hi`: the mixed-precision FMA reads the f16 half and the fp32 value, subtracts in
// fp32 (exactly: hi is x rounded) and rounds to f16 into one half of the destination -- two instructions for the pair instead of two
// conversions back, two subtractions and a pack; bit for bit the same (tools/probes/fma_mix_split.hip)
__device__ __forceinline__ uint32_t split_lo(uint32_t hi, float x0, float x1) {
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
    return l;
}
// one 32x32x16 product on 16-bit operands: bf16 (throughput mode) or f16 (f16x2 mode)
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// lanes 32..63 of x <-> lanes 0..31 of y
__device__ __forceinline__ void half_swap(uint32_t& x, uint32_t& y) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0]; y = r[1];
}

// accumulator tile of one (M-tile, 32 output channels): rows = channels, cols = pixels.  Register r of lane (m, hh) holds
// channel (r&3) + 8*(r>>2) + 4*hh of pixel m.  P[g][h] = channels 8g + 4hh + 2h + {0,1} as a bf16 pair.
struct Packed { uint32_t p[4][2]; };

// (g, hh) pairs -> 16-byte runs: after the swaps a lane holds channels [8*hh, 8*hh+8) in lo and [16 + 8*hh, 16 + 8*hh + 8) in hi
__device__ __forceinline__ void to_runs(Packed& k, u32x4& lo, u32x4& hi) {
    half_swap(k.p[0][0], k.p[1][0]); half_swap(k.p[0][1], k.p[1][1]);
    half_swap(k.p[2][0], k.p[3][0]); half_swap(k.p[2][1], k.p[3][1]);
    lo = u32x4{k.p[0][0], k.p[0][1], k.p[1][0], k.p[1][1]};
    hi = u32x4{k.p[2][0], k.p[2][1], k.p[3][0], k.p[3][1]};
}
__device__ __forceinline__ void from_runs(const u32x4& lo, const u32x4& hi, Packed& k) {   // the swap is its own inverse
    k.p[0][0] = lo[0]; k.p[0][1] = lo[1]; k.p[1][0] = lo[2]; k.p[1][1] = lo[3];
    k.p[2][0] = hi[0]; k.p[2][1] = hi[1]; k.p[3][0] = hi[2]; k.p[3][1] = hi[3];
    half_swap(k.p[0][0], k.p[1][0]); half_swap(k.p[0][1], k.p[1][1]);
    half_swap(k.p[2][0], k.p[3][0]); half_swap(k.p[2][1], k.p[3][1]);
}

// FIRST (conv1_1): the 3x3 input h1 = relu(conv3x3(features) + b) is produced on the fly into the LDS patch, by MFMA: the
// 18x18 patch pixels are enumerated linearly into 32-pixel M-tiles (waves 0..2 take two), the B operand of a lane is its
// pixel's feature neighbourhood as bf16 (half-wave 0: rows dy = 0, 1; half-wave 1: row dy = 2 and a row that meets zero
// weights), the A operand the folded 1 -> 32 filter bank (4 registers).  The block's 1 -> 32 residual projection is one
// more MFMA on the output M-tile (feature and weight each split into two bf16 so that the product keeps ~16 bits).
// FLAT (conv9_1.B): conv_flatten's (128,1) kernel rides on the packed result registers: they ARE an MFMA B operand (k =
// this lane's 8 channels of a 16-channel step, the flatten filter bank is packed in that channel order per mel row), so
// four MFMAs give both rows' 4 x 32-pixel products; the lane keeps the product of its own row, the two rows are added
// across the quad, the eight waves' sums meet in LDS and one 16-row group sum per tile goes to flat_part.
//
// SPLIT (f16x2 mode): activations are two f16 planes per tensor (x = xh + xl, the low plane a.lo_delta bytes behind the high one),
// weights two fragment banks per K chunk (w = wh + wl), and a term w x is three products, wh xh + wh xl + wl xh (the fourth,
// wl xl, is below fp32 resolution), all into the one fp32 accumulator.  A K chunk is two stages of the bf16 form:
//   part 0: patch = xl, 18 steps against bank wh          part 1: patch = xh, 36 steps: against wh, then against wl
// Both banks of a chunk sit in LDS together (staged with part 0's patch when they are not resident), no patch is fetched twice,
// and a chunk costs two pairs of barriers (a first version with three 18-step stages per chunk spent a third more on them).
// Results leave as (hi, lo) pairs; residual, pool and conv_flatten work on the fp32 values.
// RANK1 (SPLIT, conv1_1.B): the block's 1 -> 32 projection of the fp32 feature is a rank-1 term added in fp32 in the epilogue.
//
// DUO: two of the 8-wave blocks above as the two halves of one 16-wave workgroup, run in anti-phase: while half X multiplies
// (matrix pipe, LDS reads), half Y commits its next patch, issues its loads and runs its epilogue (vector memory, LDS writes, VALU),
// and the workgroup's barriers are the phase boundaries.  Two independent blocks on a CU meet in the same phase as often as not
// (stamps: a stage took 8000 cycles with two blocks per CU, 6700 alone); here the alternation is exact.  Each half owns a patch
// (and, when the weights are streamed, a bank buffer) and walks its own tiles; resident weights are shared by the two halves.
// Every thread of the workgroup executes the same number of barriers: a half that runs out of stages keeps the beat.
template <int NT, int NW, bool BRES, bool RES, bool RADD, bool POOL, int RP, bool FIRST, bool FLAT, bool PF2, bool SPLIT = false, bool RANK1 = false,
          int NH = 1>
// (registers: two blocks per CU want 128; the f16x2 A form of the 8 x 16 level -- 4 waves, streamed banks, two accumulator sets -- is held to
// two blocks per CU by its 57 KB of LDS anyway, i.e. two waves per SIMD: at 128 registers it spilled 35 of them)
__global__ __launch_bounds__(64 * NW * NH) __attribute__((amdgpu_waves_per_eu((NT <= 2 && !(SPLIT && RES && !BRES && NW == 4 && NH == 1)) ? 4 : 2)))
void conv3x3_v4_kernel(ConvArgs a, int total_tiles, int lds_b_bytes) {
    constexpr int KC = 32;
    constexpr bool DUO = NH > 1;                          // NH tiles per workgroup (2 x 8 waves or 4 x 4 waves), one beat apart
    static_assert(!DUO || (NT == 1 && NW * NH == 16 && RP == 0 && !FIRST && !FLAT && !PF2), "DUO: the plain A / B launches");
    // RING (DUO with streamed banks, four tiles): the workgroup's tiles walk the same (group, chunk) sequence a beat apart, so a
    // chunk's two banks are staged ONCE per workgroup into a two-slot ring: chunk k is read during beats 4k .. 4k+5 (tile q's two
    // stages of it multiply at beats 4k+q and 4k+2+q), its slot is free again from beat 4k+6, chunk k+2's high bank lands there at
    // beat 4k+7 (tile 0's off-phase) and its low bank at beat 4k+8 (tile 1's), each requested one off-phase earlier.
    constexpr bool RING = DUO && !BRES;
    static_assert(!RING || (SPLIT && NH == 4), "RING: f16x2, four 4-wave tiles");
    static_assert(!SPLIT || !PF2, "SPLIT: single-stage prefetch");
    static_assert(!RANK1 || (SPLIT && !RES && !RADD), "RANK1: conv1_1.B in f16x2 mode");
    static_assert(!(FIRST && SPLIT) || RANK1, "FIRST in f16x2 mode: the block's residual is the fp32 rank-1 term");
    static_assert(!FLAT || (NT == 1 && NW == 8 && BRES && (RADD || RP) && !POOL && !FIRST), "FLAT: conv9_1.B");
    static_assert(!PF2 || (BRES && !FIRST), "two-stage prefetch: resident-weight launches");
    static_assert(RP == 0 || (!RES && !RADD && !FIRST), "RP: a B launch that computes the block's projection itself (no r tensor)");
    static_assert(!FIRST || (NT == 1 && BRES && !RES && !RADD), "FIRST: conv1_1.B, one 32 -> 32 chunk, rank-1 residual");
    constexpr int kTapBytes = 2 * NT * 1024;
    constexpr int TAPS = RES ? 10 : 9;
    constexpr int NTHR = 64 * NW;
    constexpr int TH = 2 * NW;                            // tile rows: one 2x16 M-tile per wave
    constexpr int PR = TH + 2;
    constexpr int kA = PR * kRowPitch;
    constexpr int NPA = PR * kPatch * 4;
    constexpr int AIT = (NPA + NTHR - 1) / NTHR;
    constexpr int NPB = (SPLIT ? 2 : 1) * TAPS * kTapBytes / 16;
    constexpr int BIT = BRES ? 1 : (DUO ? (NPB / 2 + NTHR - 1) / NTHR : (NPB + NTHR - 1) / NTHR);   // (RING: one bank per loader tile)
    static_assert(!(RES && (RADD || POOL)), "RES is the A launch; RADD / POOL belong to B launches");
    static_assert(AIT <= 4, "edge flags are packed 8 bits per piece");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int half = DUO ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTHR) : 0;
    const int tid = DUO ? (int)threadIdx.x % NTHR : (int)threadIdx.x, lane = tid & 63;   // (DUO: thread, wave within the half)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, m = lane & 31;
    const int py = (m >> 1) & 1, px = (m & 1) | ((m >> 2) << 1);   // m = (x&1) | (y<<1) | ((x>>1)<<2): a quad of lanes = a 2x2 window
    char* sA = smem + half * kA;
    const int sb_bytes = RING ? 2 * lds_b_bytes : lds_b_bytes;         // RING: two chunk slots, shared like resident banks
    char* sB = smem + NH * kA;
    const int proj_steps = RP ? (a.C0x + a.C1x) / 16 : 0;            // RP: 16-channel K steps of the block's 1x1 projection
    // one output-channel group: the projection weights live in LDS; several groups (the instantiated cases: streamed weights with
    // NT <= 2, i.e. conv4_1, conv_bottleneck, encoder_out): the tile's group is read from memory with the stage's other loads
    constexpr bool proj_lds = RP > 0 && (BRES || NT == 3 || SPLIT);     // (SPLIT: every channel group's fragments, both banks)
    const char* sProj = smem + NH * kA + sb_bytes;                          // RP: [step][NT][64 lanes][16 B] projection weights (A operand)
    const int proj_tiles = SPLIT ? a.Cout / 32 : NT;                        // 32-channel tiles of the projection held in LDS
    const int proj_bank = proj_steps * proj_tiles * 1024;                   // (SPLIT: a bank of high halves, then one of low halves)
    const float* sBias = (const float*)(sProj + (proj_lds ? (SPLIT ? 2 : 1) * proj_bank : 0));   // [Cout] bias, RES: + [Cout] projection bias
    constexpr int FW = 20, FROWS = PR + 3;                           // FIRST: feature patch (2-pixel halo) + one spare row
    float* sFb = (float*)sBias + 32;                                 // FIRST: [32] first-conv bias, then the feature patch
    float* sF = sFb + 32;
    float* sFlat = (float*)sBias + 32;                               // FLAT: [NW][2 rows][4][16] per-wave, per-row flatten sums of the current tile

    const int H = a.H, W = a.W, Cout = a.Cout;
    const int ngroups = Cout / (32 * NT);
    const int nch_r = (a.C0 + a.C1) / KC;                          // K chunks of 32 input channels
    const int nch = SPLIT ? 2 * nch_r : nch_r;                        // stages per tile
    const int all_taps = nch_r * TAPS;

    const int xcd = blockIdx.x & 7, local = (int)(blockIdx.x >> 3) * NH + half, gper = (int)(gridDim.x >> 3) * NH;
    const int per = (total_tiles + 7) >> 3;
    auto tile_of = [&](int loc, int it) -> int {          // tile `it` of the (half-)block with index `loc` on this XCD
        if constexpr (RING) {                             // every tile of the workgroup walks a position's channel groups in the same order
            const int total_pos = total_tiles / ngroups, per_pos = (total_pos + 7) >> 3;
            const int idx = loc + (it / ngroups) * gper;
            const int pos = xcd * per_pos + idx;
            return (idx < per_pos && pos < total_pos) ? pos * ngroups + it % ngroups : -1;
        }
        const int idx = loc + it * gper;
        const int t = xcd * per + idx;
        return (idx < per && t < total_tiles) ? t : -1;
    };
    auto tile_at = [&](int it) -> int { return tile_of(local, it); };
    struct Tile { int n, y0, x0, g; };
    auto decode = [&](int t) -> Tile {
        Tile d;
        d.g = t % ngroups; t /= ngroups;
        d.x0 = (t % a.tiles_x) * 16; t /= a.tiles_x;
        d.y0 = (t % a.tiles_y) * TH;
        d.n = t / a.tiles_y;
        return d;
    };

    // ---- this thread's patch pieces: geometry fixed for the kernel's lifetime ----
    uint32_t pix_full[AIT], pix_half[AIT], lds_off[AIT], flags = 0;
    const uint32_t part16 = (tid & 3) * 16;              // NTHR % 4 == 0: a thread always moves the same 16-byte part of a pixel
    {
        const int Wh = W >> 1;
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int p = tid + NTHR * it;
            const int pix = p >> 2;
            const int pyy = pix / kPatch, pxx = pix - pyy * kPatch;
            lds_off[it] = pyy * kRowPitch + pxx * kPixPitch + part16;
            pix_full[it] = pyy * W + pxx;                                   // from the patch origin (y0-1, x0-1)
            pix_half[it] = ((pyy + 1) >> 1) * Wh + ((pxx + 1) >> 1);       // nearest-upsampled source, from (y0/2-1, x0/2-1)
            const uint32_t f = (pyy == 0 ? 1u : 0u) | (pyy == PR - 1 ? 2u : 0u) | (pxx == 0 ? 4u : 0u) | (pxx == kPatch - 1 ? 8u : 0u) |
                               (p >= NPA ? 16u : 0u);
            flags |= f << (8 * it);
        }
    }
    // PF2: patch pieces are requested TWO stages ahead (register sets ra0 / ra1 alternate): ~40 KB per block in flight
    // instead of ~20 KB.  Used where resident weights already hold a CU to two blocks, so that the 12 extra registers cost
    // no occupancy (measured: conv9_1.A 397 -> 361 us; on the three-block and streamed launches it lost a block and time).
    u32x4 ra0[AIT], ra1[AIT];
    u32x4 rb[BIT];
    float rf = 0.f;                                       // FIRST: this thread's feature value of the next tile
    static_assert(!FIRST || (PR + 2) * FW <= NTHR, "one feature value per thread");

    auto issue_patch = [&](const Tile& d, int ci, u32x4 (&ra)[AIT]) {
        if constexpr (FIRST) {
            if constexpr (SPLIT) { if (ci & 1) return; }      // a tile's second stage works from the features already in LDS
            const int fy = tid / FW, fx = tid - fy * FW;
            const int Y = d.y0 - 2 + fy, X = d.x0 - 2 + fx;
            rf = (tid < (PR + 2) * FW && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W)
                     ? a.rank1_src[((size_t)d.n * H + Y) * W + X] : 0.f;
            return;
        }
        const int ch = (SPLIT ? ci >> 1 : ci) * KC;
        const int64_t plane = (SPLIT && !(ci & 1)) ? a.lo_delta : 0;      // part 0 multiplies the low halves
        const char* base; uint32_t cs2, toff; bool up;
        if (ch < a.C0) {
            base = (const char*)a.src0 - kHdr + plane; cs2 = 2u * a.C0; up = false;
            toff = kHdr + ((((uint32_t)d.n * H + d.y0 - 1) * W + d.x0 - 1) * a.C0 + ch) * 2u;            // mod 2^32; valid pieces land >= kHdr
        } else {
            base = (const char*)a.src1 - kHdr + plane; cs2 = 2u * a.C1; up = true;
            toff = kHdr + ((((uint32_t)d.n * (H >> 1) + (d.y0 >> 1) - 1) * (W >> 1) + (d.x0 >> 1) - 1) * a.C1 + (ch - a.C0)) * 2u;
        }
        // edge mask of the tile against the piece flags: a piece is outside the image when its halo side is an image border
        uint32_t tm = (d.y0 == 0 ? 1u : 0u) | (d.y0 + TH == H ? 2u : 0u) | (d.x0 == 0 ? 4u : 0u) | (d.x0 + 16 == W ? 8u : 0u) | 16u;
        const uint32_t tp = toff + part16;
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const uint32_t pix = up ? pix_half[it] : pix_full[it];
            uint32_t off = __umul24(pix, cs2) + tp;
            if (flags & (tm << (8 * it))) off = 0;        // the zero header
            ra[it] = *(const u32x4*)(base + off);
        }
    };
    auto issue_weights = [&](const Tile& d, int ci) {
        if constexpr (!BRES && !RING) {
            if constexpr (SPLIT) { if (ci & 1) return; }          // part 1 works on the banks part 0 staged
            // SPLIT: per chunk the bank of high halves, then the bank of low halves: both staged together (NPB covers the two)
            const char* wsrc = SPLIT ? (const char*)a.wpk + ((size_t)d.g * nch_r + (ci >> 1)) * 2 * (TAPS * kTapBytes)
                                     : (const char*)a.wpk + ((size_t)d.g * all_taps + ci * TAPS) * kTapBytes;
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int p = tid + NTHR * it;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (p < NPB) v = *(const u32x4*)(wsrc + p * 16);
                rb[it] = v;
            }
        }
    };
    auto commit = [&](u32x4 (&ra)[AIT], bool patch, bool bank) {    // (SPLIT: what the stage being committed needs; else both)
        if constexpr (FIRST) return;
        if (patch) {
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                if ((it + 1) * NTHR <= NPA || !(flags & (16u << (8 * it)))) *(u32x4*)(sA + lds_off[it]) = ra[it];
            }
        }
        if constexpr (!BRES && !RING) {
            if (bank) {
#pragma unroll
                for (int it = 0; it < BIT; ++it) {
                    const int p = tid + NTHR * it;
                    if (p < NPB) *(u32x4*)(sB + p * 16) = rb[it];
                }
            }
        }
    };
    // RING: the loader duty of tiles 0 (high banks) and 1 (low banks) in the off-phase of their stage slot s: even s requests the
    // bank of chunk s / 2 + 1, odd s writes it into its slot.  Chunk k of the walk is (group, chunk) number k mod (groups x chunks),
    // which is also its place in memory.  n_chunks = chunks of the longest walk (tile 0's).
    constexpr int NPH = NPB / 2;                          // 16-byte pieces of one bank
    auto ring_duty = [&](int s, int n_chunks) {
        if constexpr (RING) {
            if (half > 1) return;
            const int k = (s >> 1) + 1;
            if (k >= n_chunks) return;
            if (!(s & 1)) {
                const char* wsrc = (const char*)a.wpk + ((size_t)(k % (ngroups * nch_r)) * 2 + half) * (TAPS * kTapBytes);
#pragma unroll
                for (int it = 0; it < BIT; ++it) {
                    const int p = tid + NTHR * it;
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (p < NPH) v = *(const u32x4*)(wsrc + p * 16);
                    rb[it] = v;
                }
            } else {
                char* dst = sB + (k & 1) * lds_b_bytes + half * (TAPS * kTapBytes);
#pragma unroll
                for (int it = 0; it < BIT; ++it) {
                    const int p = tid + NTHR * it;
                    if (p < NPH) *(u32x4*)(dst + p * 16) = rb[it];
                }
            }
        }
    };

    // SPLIT + RP ("projection in B" in f16x2): the K steps [c RP, c RP + RP) of the projection belong to chunk c.  Their pixel
    // fragments (both planes, straight from the block input) and, when the projection banks are not in LDS, the weight fragments are
    // requested in the off-phase of the chunk's part-0 stage and multiplied at the head of its part-1 stage, before the fragment
    // reads of the 3x3 loop start: they never share registers with a multiply loop.  Three products per step, as everywhere.
    // Several steps per chunk (conv9_1: four): the first RPH of them are requested in part 0's off-phase and multiplied at the head of
    // part 1, where the others are requested into the same registers and multiplied behind the 3x3 loop -- half of the fragments
    // are live at a time, and none across an epilogue.
    constexpr int RPH = RP >= 2 ? RP / 2 : RP;
    u32x4 pxh[(SPLIT && RP) ? RPH : 1], pxl[(SPLIT && RP) ? RPH : 1];
    auto issue_proj = [&](const Tile& d, int chunk, auto k0c, auto k1c) {
        constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
        if constexpr (SPLIT && RP > 0 && K1 > K0) {
            const uint32_t t_full = kHdr + ((((uint32_t)d.n * H + d.y0 + 2 * wave) * W + d.x0) * a.C0x) * 2u;
            const uint32_t t_half = kHdr + ((((uint32_t)d.n * (H >> 1) + (d.y0 >> 1) + wave) * (W >> 1) + (d.x0 >> 1)) * a.C1x) * 2u;
            // (an opaque zero per call: left to itself the compiler forms every lane address of this block once, ahead of the stage loop,
            // and pays for the 64-bit loop invariants with spills)
            uint32_t zl = 0;
            asm volatile("" : "+v"(zl));
            const uint32_t xf_o = (uint32_t)((py * W + px) * a.C0x + hh * 8) * 2u + zl, xh_o = (uint32_t)((px >> 1) * a.C1x + hh * 8) * 2u + zl;
#pragma unroll
            for (int k = K0; k < K1; ++k) {
                const int sidx = chunk * RP + k, ch = sidx * 16;                   // block-uniform
                const char* base; uint32_t off;
                if (sidx >= proj_steps) { base = (const char*)a.xp0 - kHdr; off = 0; }     // (a step past the end multiplies the zero header)
                else if (ch < a.C0x) { base = (const char*)a.xp0 - kHdr; off = t_full + xf_o + ch * 2u; }
                else { base = (const char*)a.xp1 - kHdr; off = t_half + xh_o + (ch - a.C0x) * 2u; }
                pxh[k - K0] = *(const u32x4*)(base + off);
                pxl[k - K0] = *(const u32x4*)(base + a.lo_delta + off);
            }
        }
    };

    int it_tile = 0;
    struct Stage { int ci; Tile d; };
    auto next_stage = [&](const Stage& s0, Stage& n) -> bool {     // block-uniform; walks (tile, chunk) in order
        n = s0; n.ci = s0.ci + 1;
        if (n.ci == nch) {
            n.ci = 0;
            const int t = tile_at(++it_tile);
            if (t < 0) return false;
            n.d = decode(t);
        }
        return true;
    };
    int my_stages = 0, max_stages = 0;                    // DUO: stages of this half / of the longer half (the workgroup's beat count)
    if constexpr (DUO) {
        const int l0 = (int)(blockIdx.x >> 3) * NH;      // (the group with the lowest index has the most tiles)
        int n0 = 0, nmine = 0;
        while (tile_of(l0, n0) >= 0) ++n0;
        while (tile_of(l0 + half, nmine) >= 0) ++nmine;
        my_stages = nmine * nch;
        max_stages = n0 * nch;
        if (max_stages == 0) return;                      // whole workgroup idle
    } else {
        if (tile_at(0) < 0) return;                       // whole block idle (block-uniform)
    }
    // (DUO: a half without tiles runs the prologue on tile 0 -- valid addresses, results unused -- and then only keeps the beat)
    Stage cs{0, decode((DUO && my_stages == 0) ? 0 : tile_at(0))}, n1 = cs, n2 = cs;

    if constexpr (BRES || RING) {                         // resident banks / RING: chunk 0 of the walk into slot 0
        const char* wsrc = (const char*)a.wpk;
        for (int p = tid; p < lds_b_bytes / 16; p += NTHR) *(u32x4*)(sB + p * 16) = *(const u32x4*)(wsrc + (size_t)p * 16);
    }
    for (int i = tid; i < Cout * (RES ? 2 : 1); i += NTHR)
        ((float*)sBias)[i] = i < Cout ? a.bias[i] : a.res_bias[i - Cout];
    if constexpr (proj_lds)
        for (int p = tid; p < (SPLIT ? 2 : 1) * proj_steps * proj_tiles * 64; p += NTHR) *(u32x4*)((char*)sProj + p * 16) = *(const u32x4*)((const char*)a.proj_w + (size_t)p * 16);
    // ---- FIRST: constants of the producer ----
    u32x4 wfirst = {0u, 0u, 0u, 0u}, wr1 = {0u, 0u, 0u, 0u}, wfirst_lo = {0u, 0u, 0u, 0u};
    constexpr int NMT = (PR * kPatch + 31) / 32;          // M-tiles of the patch; wave w produces w and w + NW
    int pf_off[2], pa_off[2]; uint32_t pflags = 0;
    if constexpr (FIRST) {
        for (int i = tid; i < 32; i += NTHR) sFb[i] = a.first_b[i];
        for (int i = tid; i < FROWS * FW; i += NTHR) sF[i] = 0.f;
        const float* w9 = a.first_w;                      // [9][32], tap-major
        auto wv = [&](int t) { return w9[t * 32 + m]; };
        if constexpr (SPLIT) {                            // f16x2: the filter bank as two f16 halves (wr1 is unused: RANK1 adds the residual in fp32)
            auto hi2 = [&](float x, float y) { return pack_f16(x, y); };
            auto lo2 = [&](float x, float y) { const f32x2 b = unpack_f16(pack_f16(x, y)); return pack_f16(x - b[0], y - b[1]); };
            if (hh == 0) {
                wfirst = u32x4{hi2(wv(0), wv(1)), hi2(wv(2), 0.f), hi2(wv(3), wv(4)), hi2(wv(5), 0.f)};
                wfirst_lo = u32x4{lo2(wv(0), wv(1)), lo2(wv(2), 0.f), lo2(wv(3), wv(4)), lo2(wv(5), 0.f)};
            } else {
                wfirst = u32x4{hi2(wv(6), wv(7)), hi2(wv(8), 0.f), 0u, 0u};
                wfirst_lo = u32x4{lo2(wv(6), wv(7)), lo2(wv(8), 0.f), 0u, 0u};
            }
        } else if (hh == 0) {
            wfirst = u32x4{pack_bf16(wv(0), wv(1)), pack_bf16(wv(2), 0.f), pack_bf16(wv(3), wv(4)), pack_bf16(wv(5), 0.f)};
            const float w = a.rank1_w[m];
            const float whi = (float)(__bf16)w, wlo = w - whi;
            wr1 = u32x4{pack_bf16(whi, whi), pack_bf16(wlo, 0.f), 0u, 0u};
        } else {
            wfirst = u32x4{pack_bf16(wv(6), wv(7)), pack_bf16(wv(8), 0.f), 0u, 0u};
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = 32 * (wave + NW * t) + m;
            const bool in = q < PR * kPatch;
            const int qq = in ? q : 0;
            const int qy = qq / kPatch, qx = qq - qy * kPatch;
            pf_off[t] = (qy + 2 * hh) * FW + qx;
            pa_off[t] = qy * kRowPitch + qx * kPixPitch + hh * 16;
            const uint32_t f = (qy == 0 ? 1u : 0u) | (qy == PR - 1 ? 2u : 0u) | (qx == 0 ? 4u : 0u) | (qx == kPatch - 1 ? 8u : 0u) | (in ? 0u : 16u);
            pflags |= f << (8 * t);
        }
    }
    // h1 patch of tile d from the feature patch in sF (all waves; the caller puts barriers around it)
    // (f16x2: part 0 computes h1, writes its low halves and keeps the high halves' runs in hk_lo / hk_hi; part 1 only writes those)
    u32x4 hk_lo[(FIRST && SPLIT) ? 2 : 1], hk_hi[(FIRST && SPLIT) ? 2 : 1];
    auto produce = [&](const Tile& d, int part) {
        const uint32_t tm = (d.y0 == 0 ? 1u : 0u) | (d.y0 + TH == H ? 2u : 0u) | (d.x0 == 0 ? 4u : 0u) | (d.x0 + 16 == W ? 8u : 0u) | 16u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (wave + NW * t < NMT) {                    // wave-uniform
                if constexpr (SPLIT) {
                    if (part) {                           // the high halves computed with the low ones, one stage ago
                        if (!(pflags & (16u << (8 * t)))) {
                            *(u32x4*)(sA + pa_off[t]) = hk_lo[t];
                            *(u32x4*)(sA + pa_off[t] + 32) = hk_hi[t];
                        }
                        continue;
                    }
                }
                const float* fp = sF + pf_off[t];
                f32x16 h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *(const f32x4*)(sFb + 8 * g + 4 * hh);
                    h[4 * g] = b4[0]; h[4 * g + 1] = b4[1]; h[4 * g + 2] = b4[2]; h[4 * g + 3] = b4[3];
                }
                const uint32_t keep = (pflags & (tm << (8 * t))) ? 0u : 0xffffffffu;   // 0 outside the picture: conv2's zero padding
                Packed k, kh;
                if constexpr (SPLIT) {                    // features and filters as f16 halves, three products; h1 leaves as its two halves
                    const float fv[6] = {fp[0], fp[1], fp[2], fp[FW], fp[FW + 1], fp[FW + 2]};
                    uint32_t bh[4], bl[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x0 = fv[(i >> 1) * 3 + (i & 1) * 2], x1 = (i & 1) ? 0.f : fv[(i >> 1) * 3 + 1];
                        bh[i] = pack_f16(x0, x1);
                        bl[i] = split_lo(bh[i], x0, x1);
                    }
                    const u32x4 boph = {bh[0], bh[1], bh[2], bh[3]}, bopl = {bl[0], bl[1], bl[2], bl[3]};
                    h = mfma16<true>(wfirst, bopl, h);
                    h = mfma16<true>(wfirst_lo, boph, h);
                    h = mfma16<true>(wfirst, boph, h);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int hq = 0; hq < 2; ++hq) {
                            const float x0 = fmaxf(h[4 * g + 2 * hq], 0.f), x1 = fmaxf(h[4 * g + 2 * hq + 1], 0.f);
                            const uint32_t ph = pack_f16(x0, x1);
                            k.p[g][hq] = split_lo(ph, x0, x1) & keep;
                            kh.p[g][hq] = ph & keep;
                        }
                    to_runs(kh, hk_lo[t], hk_hi[t]);
                } else {
                    const u32x4 bop = {pack_bf16(fp[0], fp[1]), pack_bf16(fp[2], 0.f), pack_bf16(fp[FW], fp[FW + 1]), pack_bf16(fp[FW + 2], 0.f)};
                    h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfirst), __builtin_bit_cast(bf16x8, bop), h, 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int hq = 0; hq < 2; ++hq) k.p[g][hq] = relu_pk(pack_bf16(h[4 * g + 2 * hq], h[4 * g + 2 * hq + 1])) & keep;
                }
                u32x4 lo, hi;
                to_runs(k, lo, hi);
                if (!(pflags & (16u << (8 * t)))) {
                    *(u32x4*)(sA + pa_off[t]) = lo;
                    *(u32x4*)(sA + pa_off[t] + 32) = hi;
                }
            }
        }
    };

    issue_patch(cs.d, 0, ra0);
    issue_weights(cs.d, 0);
    commit(ra0, true, true);
    if constexpr (FIRST) {
        __syncthreads();                                  // sF zero fill, sFb
        if (tid < (PR + 2) * FW) sF[tid] = rf;
        __syncthreads();
        produce(cs.d, 0);                                 // (f16x2: stage 0 multiplies the low halves)
    }
    // lookahead: n1 / n2 / n3 = the stages after the current one (block-uniform); their patches are in flight in ra0 / ra1
    Stage n3 = cs;
    bool ok1 = next_stage(cs, n1), ok2 = false, ok3 = false;
    if (ok1) { issue_patch(n1.d, n1.ci, ra0); issue_weights(n1.d, n1.ci); }      // (the commits above have consumed ra0 / rb)
    if constexpr (PF2) { ok2 = ok1 && next_stage(n1, n2); if (ok2) issue_patch(n2.d, n2.ci, ra1); }
    __syncthreads();

    f32x16 acc[NT];
    f32x16 racc[RES ? NT : 1];
    const int aoff0 = (2 * wave + py) * kRowPitch + px * kPixPitch + hh * 16;
    const int boff0 = lane * 16;
    // where this lane's 16-byte runs of its pixel go, relative to the wave's M-tile origin (row y0 + 2 wave, column x0)
    const uint32_t st_off = (uint32_t)((py * W + px) * Cout + hh * 8) * 2u;
    // f16x2 r tensors: fp32 accumulator fragments, one 2 x 16-pixel x 32-channel M-tile after the other
    // ([n][H/2][W/16][Cout/32]); an M-tile's 4 KB sit half in each plane's slot of the tensor: [lane][registers 0..7] in the
    // plane of high halves, [lane][registers 8..15] in the other (each slot is 2 KB per M-tile, as for an activation tensor)
    auto r_mtile = [&](const Tile& t, uint32_t co) -> uint32_t {     // byte offset of this wave's M-tile (channel tile co / 32) in a slot
        return (((((uint32_t)t.n * (H >> 1) + (t.y0 >> 1) + wave) * (W >> 4) + (t.x0 >> 4)) * (uint32_t)(Cout >> 5)) + (co >> 5)) * 2048u;
    };
    const uint32_t pl_off = (uint32_t)((m >> 2) * Cout + hh * 8) * 2u;     // pooled pixel (m >> 2) of the M-tile's 1x8 pooled row
    // RP: this lane's pixel in the block input x (full-resolution source / nearest-upsampled half-resolution source), as byte
    // offsets from the wave's M-tile origin; + 32 s bytes for K step s, + 16 hh for the lane's half of the 16 channels
    const uint32_t xf_off = (uint32_t)((py * W + px) * a.C0x + hh * 8) * 2u;
    const uint32_t xh_off = (uint32_t)((px >> 1) * a.C1x + hh * 8) * 2u;

    // One stage.  Order: MFMAs of stage k | barrier | commit of stage k+1 (its loads were issued a whole stage or two ago),
    // loads of stage k+2 (PF2: k+3) into the registers just freed | barrier | epilogue of stage k.  The epilogue's stores are
    // then followed by a full MFMA phase before anything waits on vmcnt again: on gfx9 loads and stores share that counter and
    // may retire out of order with each other, so every wait for a load is a wait for all earlier stores as well.
    bool flat_pending = false; Tile flat_tile = cs.d;
    // Timing perturbation for the tests (library built with -DSS_DEVBUILD only -- the sleeps are scheduling barriers and cost 2-5 % --;
    // ConvArgs::dbg bit 10, pattern in bits 11-12): chosen waves sleep ~1 us at the stage's synchronisation points.  Results must
    // not change; a missing barrier shows up as a changed bit.
    int jit_n = 0;
    int stage_no = 0;                                     // RING: this tile's stage count = its slot in the workgroup's beat
#ifdef SS_DEVBUILD
    // stamps (ConvArgs::stamps, dev build: SOFTSPOKEN_STAMP_LAYER): shader-clock time between the stage's synchronisation points,
    // summed per wave: [0] MFMA phase, [1] wait at barrier 1, [2] commit + issue, [3] wait at barrier 2, [4] epilogue + loop turn
    uint32_t st_prev = 0, st_sum[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};    // [5], [6]: of the commit + issue segment, the wait for the loads / the LDS writes
#endif
    auto jitter = [&](int site) {
#ifdef SS_DEVBUILD
        if (a.stamps) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime();
            const int seg = site == 0 ? 4 : site == 1 ? 0 : site == 2 ? 1 : site == 4 ? 2 : site == 5 ? 3 : site == 6 ? 4 : -1;
            if (seg >= 0 && (site != 0 || jit_n > 1)) st_sum[seg] += t - st_prev;
            if (seg >= 0) st_prev = t;
        }
        if (a.dbg & 1024) {
            const int pat = (a.dbg >> 11) & 3;
            const bool z = pat == 0 ? ((wave + site + jit_n) & 3) == 0 : pat == 1 ? wave == 0 : pat == 2 ? wave != 0 : (wave & 1) != 0;
            if (z) __builtin_amdgcn_s_sleep(32);
        }
#else
        (void)site;
#endif
    };
    auto flat_reduce = [&](const Tile& t) {               // 4 channels x 16 columns: a tile's 16-row group sum, waves in fixed order
        if (tid < 64) {
            float sgrp = 0.f;
#pragma unroll
            for (int w = 0; w < 2 * NW; ++w) sgrp += sFlat[w * 64 + tid];
            a.flat_part[(((size_t)t.n * a.tiles_y + t.y0 / TH) * 4 + (tid >> 4)) * W + t.x0 + (tid & 15)] = sgrp;
        }
    };
    auto stage = [&](auto part_c, u32x4 (&ra_a)[AIT]) -> bool {   // ra_a: holds stage k+1's patch, then receives the newest stage's
        // SPLIT: which of a chunk's two stages this is (compile time: a tile's stages come in pairs, so the call sites below
        // alternate 0, 1); the next stage always needs its patch committed, its banks only when it is a part 0
        constexpr int PART = SPLIT ? decltype(part_c)::value : 0;
        constexpr int NEXT = SPLIT ? 1 - PART : 0;
        constexpr bool kCommitPatch = true, kCommitBank = !SPLIT || NEXT == 0;
        constexpr int NSTEP = (SPLIT && PART == 1) ? 36 : 18;             // part 1: the 18 (tap, sub-step) pairs against wh, then against wl
        const Tile cur = cs.d;
        const int ci = cs.ci;
        const bool last = ci == nch - 1;
        ++jit_n; jitter(0);
        const uint32_t co0 = cur.g * 32 * NT;
        const uint32_t o_tile = ((((uint32_t)cur.n * H + cur.y0 + 2 * wave) * W + cur.x0) * Cout + co0) * 2u;   // wave-uniform

        // B launches: the residual runs of this lane's pixel, requested now and used after the MFMAs
        u32x4 rlo[RADD ? NT : 1], rhi[RADD ? NT : 1];
        u32x4 rlo2[(RADD && SPLIT) ? NT : 1], rhi2[(RADD && SPLIT) ? NT : 1];   // SPLIT: the same runs of the low plane
        float r1f = 0.f;                                  // RANK1: this lane's feature value
        if constexpr (RANK1) {
            const uint32_t pixel = last ? ((uint32_t)cur.n * H + cur.y0 + 2 * wave + py) * W + cur.x0 + px : 0u;
            r1f = a.rank1_src[pixel];
        }
        if constexpr (RADD) {
            // Unconditional, and added to the accumulators unconditionally below: a stage that is not the tile's last reads the
            // tensor's zero header.  (Under `if (last)` the compiler copied the registers right behind the loads, i.e. waited
            // for them before the MFMAs; loaded but unused on one path it drained vmcnt, stores included, at the loop top.)
            if constexpr (SPLIT) {
                // f16x2: the A launch left r as fp32 accumulator fragments (see its epilogue): four 16-byte loads per 32-channel tile
                // that are added to the accumulator as they are -- no split, no channel shuffle, no conversion on either side
                const char* rq = (const char*)a.res_in - kHdr + (last ? kHdr + r_mtile(cur, co0) + lane * 32u : 0u);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const char* rn = rq + (last ? nt * 2048 : 0);      // (not last: every read stays inside the zero header)
                    rlo[nt] = *(const u32x4*)(rn); rhi[nt] = *(const u32x4*)(rn + 16);
                    rlo2[nt] = *(const u32x4*)(rn + a.lo_delta); rhi2[nt] = *(const u32x4*)(rn + a.lo_delta + 16);
                }
            } else {
                const char* rp = (const char*)a.res_in - kHdr + (last ? kHdr + o_tile + st_off : 0u);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { rlo[nt] = *(const u32x4*)(rp + nt * 64); rhi[nt] = *(const u32x4*)(rp + nt * 64 + 32); }
            }
        }
        // FLAT: filter fragments of this wave's mel row pair x two channel steps: rows 0..3 of the fragment are the 4 flatten
        // channels with the upper mel row's weights, rows 4..7 with the lower row's, so ONE product serves both rows of the M-tile
        u32x4 fw[FLAT ? 2 : 1];
        u32x4 fw2[(FLAT && SPLIT) ? 2 : 1];               // SPLIT: their low halves (second bank of flat_w4)
        auto load_flat_w = [&]() {
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                fw[sx] = *(const u32x4*)((const char*)a.flat_w4 + (((cur.y0 >> 1) + wave) * 2 + sx) * 1024 + lane * 16);
                if constexpr (SPLIT) fw2[sx] = *(const u32x4*)((const char*)a.flat_w4 + 64 * 2 * 1024 + (((cur.y0 >> 1) + wave) * 2 + sx) * 1024 + lane * 16);
            }
        };
        if constexpr (FLAT && !(SPLIT && RP > 0)) load_flat_w();     // (with the projection's fragments still live: behind its products, below)
        // RP: the K steps [ci * RP, ci * RP + RP) of the projection ride on this stage (steps past the end read the zero header, so
        // every stage issues the same loads and MFMAs: nothing is predicated, see the residual loads above)
        u32x4 xf[RP ? RP : 1];
        if constexpr (RP > 0 && !SPLIT) {
            const uint32_t t_full = kHdr + ((((uint32_t)cur.n * H + cur.y0 + 2 * wave) * W + cur.x0) * a.C0x) * 2u;
            const uint32_t t_half = kHdr + ((((uint32_t)cur.n * (H >> 1) + (cur.y0 >> 1) + wave) * (W >> 1) + (cur.x0 >> 1)) * a.C1x) * 2u;
#pragma unroll
            for (int k = 0; k < RP; ++k) {
                const int sidx = ci * RP + k, ch = sidx * 16;                       // block-uniform
                const char* base; uint32_t off;
                if (sidx >= proj_steps) { base = (const char*)a.xp0 - kHdr; off = 0; }
                else if (ch < a.C0x) { base = (const char*)a.xp0 - kHdr; off = t_full + xf_off + ch * 2u; }
                else { base = (const char*)a.xp1 - kHdr; off = t_half + xh_off + (ch - a.C0x) * 2u; }
                xf[k] = *(const u32x4*)(base + off);
            }
        }
        u32x4 wpg[(RP && !proj_lds) ? RP : 1][NT];        // RP, several groups: this tile's projection weights, [step][all 32-channel tiles][lane]
        if constexpr (RP > 0 && !proj_lds && !SPLIT) {
            {
                const int tiles = Cout / 32;
#pragma unroll
                for (int k = 0; k < RP; ++k) {
                    const int sidx = ci * RP + k < proj_steps ? ci * RP + k : 0;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        wpg[k][nt] = *(const u32x4*)((const char*)a.proj_w + ((size_t)(sidx * tiles + cur.g * NT + nt) * 64 + lane) * 16);
                }
            }
        }
        if (ci == 0) {                                    // accumulators start from the bias (the MFMA's C operand)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *(const f32x4*)(sBias + co0 + nt * 32 + 8 * g + 4 * hh);
                    acc[nt][4 * g] = b4[0]; acc[nt][4 * g + 1] = b4[1]; acc[nt][4 * g + 2] = b4[2]; acc[nt][4 * g + 3] = b4[3];
                    if constexpr (RES) {
                        const f32x4 r4 = *(const f32x4*)(sBias + Cout + co0 + nt * 32 + 8 * g + 4 * hh);
                        racc[nt][4 * g] = r4[0]; racc[nt][4 * g + 1] = r4[1]; racc[nt][4 * g + 2] = r4[2]; racc[nt][4 * g + 3] = r4[3];
                    }
                }
        }
        auto proj_products = [&](auto k0c, auto k1c) {     // + conv1x1(x): steps [K0, K1) of this chunk, their fragments in slots 0 ..
            constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
#pragma unroll
            for (int k = K0; k < K1; ++k) {
                const int sidx = (ci >> 1) * RP + k < proj_steps ? (ci >> 1) * RP + k : 0;   // (a step past the end multiplies zeros)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int wt = (sidx * proj_tiles + (int)(co0 >> 5) + nt) * 1024 + lane * 16;
                    const u32x4 wh = *(const u32x4*)(sProj + wt), wl = *(const u32x4*)(sProj + proj_bank + wt);
                    acc[nt] = mfma16<true>(wh, pxl[k - K0], acc[nt]);
                    acc[nt] = mfma16<true>(wl, pxh[k - K0], acc[nt]);
                    acc[nt] = mfma16<true>(wh, pxh[k - K0], acc[nt]);
                }
            }
        };
        using KZ_ = std::integral_constant<int, 0>; using KH_ = std::integral_constant<int, RPH>; using KR_ = std::integral_constant<int, RP>;
        if constexpr (SPLIT && RP > 0 && PART == 1) {     // the steps requested in part 0's off-phase; then the request for the others
            proj_products(KZ_{}, KH_{});
            __builtin_amdgcn_sched_barrier(0);            // (their operands are dead before anything else is requested)
            if constexpr (RP > RPH) issue_proj(cur, ci >> 1, KH_{}, KR_{});
        }
        if constexpr (FLAT && SPLIT && RP > 0) load_flat_w();
        {
            // (resident banks of several channel groups -- DUO only -- lie as in memory: [group][chunk][bank])
            const char* bbase = sB + boff0 + (BRES ? ((DUO ? cur.g * nch : 0) + (SPLIT ? (ci >> 1) * 2 : ci)) * TAPS * kTapBytes
                                                   : RING ? ((stage_no >> 1) & 1) * lds_b_bytes : 0);
            constexpr int PD = (NT == 1) ? 4 : 2;        // fragment prefetch depth (NT = 2 at depth 4 spills under its 128-register cap)
            u32x4 af[PD], bfr[PD][NT];
            u32x4 rfr[RES ? 2 : 1][RES ? NT : 1];
            if (a.dbg & 32) __builtin_amdgcn_s_setprio(kMfmaPrio);
            // part 1 of SPLIT, NT = 1: a pixel fragment is read once and multiplied against both banks (1.5 KB of LDS reads per
            // product instead of 2: the LDS array is as busy as the matrix pipe in these stages)
            constexpr bool kMerged = SPLIT && NSTEP == 36 && NT == 1;
            if constexpr (kMerged) {
                constexpr int PM = 3;
                u32x4 pf[PM], wh_[PM], wl_[PM], rh_[RES ? 2 : 1], rl_[RES ? 2 : 1];
                const char* bb0 = bbase; const char* bb1 = bbase + TAPS * kTapBytes;
                auto load3 = [&](int st, int slot) {
                    const int tap = st >> 1, sub = st & 1, dy = tap / 3, dx = tap % 3;
                    pf[slot] = *(const u32x4*)(sA + aoff0 + dy * kRowPitch + dx * kPixPitch + sub * 32);
                    wh_[slot] = *(const u32x4*)(bb0 + tap * kTapBytes + sub * 1024);
                    wl_[slot] = *(const u32x4*)(bb1 + tap * kTapBytes + sub * 1024);
                };
#pragma unroll
                for (int st = 0; st < PM - 1; ++st) load3(st, st);
#pragma unroll
                for (int st = 0; st < 18; ++st) {
                    if (st + PM - 1 < 18) load3(st + PM - 1, (st + PM - 1) % PM);
                    if constexpr (RES) {
                        if (st == 6) {
#pragma unroll
                            for (int sub = 0; sub < 2; ++sub) {
                                rh_[sub] = *(const u32x4*)(bb0 + 9 * kTapBytes + sub * 1024);
                                rl_[sub] = *(const u32x4*)(bb1 + 9 * kTapBytes + sub * 1024);
                            }
                        }
                    }
                    const u32x4 pixv = pf[st % PM];
                    acc[0] = mfma16<true>(wh_[st % PM], pixv, acc[0]);
                    acc[0] = mfma16<true>(wl_[st % PM], pixv, acc[0]);
                    if constexpr (RES) {
                        if (st == 8 || st == 9) {
                            racc[0] = mfma16<true>(rh_[st & 1], pixv, racc[0]);
                            racc[0] = mfma16<true>(rl_[st & 1], pixv, racc[0]);
                        }
                    }
                }
            } else
#pragma unroll
            for (int bank = 0; bank < NSTEP / 18; ++bank) {          // part 1 of SPLIT: the 18 steps against wh, then against wl
                const char* bb = bbase + bank * TAPS * kTapBytes;
                auto load_frags = [&](int st, u32x4& fa, u32x4 (&fbb)[NT]) {
                    const int tap = st >> 1, sub = st & 1, dy = tap / 3, dx = tap % 3;
                    fa = *(const u32x4*)(sA + aoff0 + dy * kRowPitch + dx * kPixPitch + sub * 32);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) fbb[nt] = *(const u32x4*)(bb + tap * kTapBytes + (sub * NT + nt) * 1024);
                };
                if (bank) __builtin_amdgcn_sched_barrier(0);        // (one round's fragment reads stay out of the other's: no registers for both)
#pragma unroll
                for (int st = 0; st < PD - 1; ++st) load_frags(st, af[st], bfr[st]);
#pragma unroll
                for (int st = 0; st < 18; ++st) {
                    if (st + PD - 1 < 18) load_frags(st + PD - 1, af[(st + PD - 1) % PD], bfr[(st + PD - 1) % PD]);
                    if constexpr (RES) {                  // the 1x1 projection's fragments: requested two steps before their use
                        if (st == 6) {
#pragma unroll
                            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) rfr[sub][nt] = *(const u32x4*)(bb + 9 * kTapBytes + (sub * NT + nt) * 1024);
                        }
                    }
                    const u32x4 pixv = af[st % PD];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {     // weights are the A operand (rows = channels), pixels the B operand
                        acc[nt] = mfma16<SPLIT>(bfr[st % PD][nt], pixv, acc[nt]);
                        if constexpr (RES) {
                            if (st == 8 || st == 9) racc[nt] = mfma16<SPLIT>(rfr[st & 1][nt], pixv, racc[nt]);
                        }
                    }
                }
            }
        }

        if constexpr (SPLIT && RP > RPH && PART == 1) proj_products(KH_{}, KR_{});      // (requested at the head of this stage)
        if constexpr (RP > 0 && !SPLIT) {                 // + conv1x1(x): pixel fragments straight from memory, weights from LDS
#pragma unroll
            for (int k = 0; k < RP; ++k) {
                const int sidx = ci * RP + k < proj_steps ? ci * RP + k : 0;       // (a step past the end multiplies zeros)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    u32x4 wp;
                    if constexpr (proj_lds) wp = *(const u32x4*)(sProj + (sidx * NT + nt) * 1024 + lane * 16); else wp = wpg[k][nt];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wp), __builtin_bit_cast(bf16x8, xf[k]), acc[nt], 0, 0, 0);
                }
            }
        }
        if (a.dbg & 32) __builtin_amdgcn_s_setprio(0);
        if constexpr (FIRST && !SPLIT) {                  // + conv1x1(features): hi/lo split keeps the rank-1 term near fp32
            const float f = sF[(2 * wave + py + 2) * FW + px + 2];
            const float fhi = (float)(__bf16)f, flo = f - fhi;
            u32x4 bop = {0u, 0u, 0u, 0u};
            if (hh == 0) { bop[0] = pack_bf16(fhi, flo); bop[1] = pack_bf16(fhi, 0.f); }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr1), __builtin_bit_cast(bf16x8, bop), acc[0], 0, 0, 0);
        }
        auto epilogue = [&]() {
            __builtin_amdgcn_sched_barrier(0);                // keep the consumers of this stage's global loads behind the MFMAs
            if constexpr (SPLIT) {
                // f16x2: everything below works on the fp32 accumulators; values leave as (hi, lo) pairs of f16 runs, one per plane
#ifdef SS_DEVBUILD
                uint32_t te0 = 0;
                if (a.stamps) te0 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
                uint32_t ovf = 0;                             // the largest high halves seen (infinity, NaN: all exponent bits set)
                auto split_store = [&](const float (&v)[16], char* dst) {
                    Packed kh, kl;
    #pragma unroll
                    for (int g = 0; g < 4; ++g)
    #pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float x0 = v[4 * g + 2 * h], x1 = v[4 * g + 2 * h + 1];
                            kh.p[g][h] = pack_f16(x0, x1);
                            ovf = pk_max_u16(ovf, kh.p[g][h]);
                            kl.p[g][h] = split_lo(kh.p[g][h], x0, x1);
                        }
                    return std::pair<Packed, Packed>(kh, kl);
                };
                if constexpr (RADD) {
    #pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        // (whole vectors are cast: __builtin_bit_cast on an element expression of an ext_vector reads element 0)
                        const f32x4 f0 = __builtin_bit_cast(f32x4, rlo[nt]), f1 = __builtin_bit_cast(f32x4, rhi[nt]);
                        const f32x4 f2 = __builtin_bit_cast(f32x4, rlo2[nt]), f3 = __builtin_bit_cast(f32x4, rhi2[nt]);
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[nt][e] += f0[e];
                            acc[nt][4 + e] += f1[e];
                            acc[nt][8 + e] += f2[e];
                            acc[nt][12 + e] += f3[e];
                        }
                    }
                }
#ifdef SS_DEVBUILD
                if (a.stamps) { const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime(); st_sum[10] += t - te0; te0 = t; }
#endif
                if (last) {
#ifdef SS_DEVBUILD
                    char* op = (char*)a.out + ((a.dbg & 128) ? ((o_tile + st_off) & 0x3ffffu) : (o_tile + st_off));   // timing-only: dbg bit 7 folds the stores into 256 KB
#else
                    char* op = (char*)a.out + (o_tile + st_off);
#endif
    #pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v[16];
                        if constexpr (RANK1) {                    // + conv1x1(feature): one fp32 multiply-add per channel
    #pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 w4 = *(const f32x4*)(a.rank1_w + co0 + nt * 32 + 8 * g + 4 * hh);
    #pragma unroll
                                for (int e = 0; e < 4; ++e) acc[nt][4 * g + e] = fmaf(r1f, w4[e], acc[nt][4 * g + e]);
                            }
                        }
    #pragma unroll
                        for (int r = 0; r < 16; ++r) {                // ReLU as an integer max: negative floats (and -0) are negative integers;
                            const float av = acc[nt][r];              // (a copy: __builtin_bit_cast on an ext_vector element reads element 0)
                            const int b = __builtin_bit_cast(int, av);   // one instruction (fmaxf: a NaN-quieting v_max first)
                            v[r] = __builtin_bit_cast(float, b > 0 ? b : 0);
                        }
                        const auto kk = split_store(v, op);
                        Packed kh = kk.first, kl = kk.second;
                        if constexpr (FLAT) {
                            f32x16 dd;
    #pragma unroll
                            for (int r = 0; r < 16; ++r) dd[r] = 0.f;
    #pragma unroll
                            for (int sx = 0; sx < 2; ++sx) {
                                const u32x4 bh = {kh.p[2 * sx][0], kh.p[2 * sx][1], kh.p[2 * sx + 1][0], kh.p[2 * sx + 1][1]};
                                const u32x4 bl = {kl.p[2 * sx][0], kl.p[2 * sx][1], kl.p[2 * sx + 1][0], kl.p[2 * sx + 1][1]};
                                dd = mfma16<true>(fw[sx], bl, dd); dd = mfma16<true>(fw2[sx], bh, dd); dd = mfma16<true>(fw[sx], bh, dd);
                            }
                            // registers 0..3 = product rows 0..3 (upper row's weights) in half-wave 0, rows 4..7 (lower row's) in half-wave 1:
                            // a pixel's own value is where the half-wave equals its row
    #pragma unroll
                            for (int c4 = 0; c4 < 4; ++c4)
                                if (hh == py) sFlat[((wave * 2 + hh) * 4 + c4) * 16 + px] = dd[c4];
                        }
                        u32x4 lo, hi;
#ifdef SS_DEVBUILD
                        const bool st_ok = !(a.dbg & 64);             // timing-only: dbg bit 6 drops the f16x2 epilogue's activation stores
#else
                        constexpr bool st_ok = true;
#endif
#ifdef SS_DEVBUILD
                        if ((a.dbg & 256) && (!FLAT || a.store_out)) {   // timing-only: each store instruction covers whole 64-byte pixels of one row
                            char* o1 = (char*)a.out + o_tile + (uint32_t)(px * Cout) * 2u + (hh + 2 * py) * 16;
                            char* o2 = o1 + (uint32_t)(W * Cout) * 2u;
                            to_runs(kh, lo, hi);
                            *(u32x4*)(o1 + nt * 64) = lo;
                            *(u32x4*)(o2 + nt * 64) = hi;
                            to_runs(kl, lo, hi);
                            *(u32x4*)(o1 + a.lo_delta + nt * 64) = lo;
                            *(u32x4*)(o2 + a.lo_delta + nt * 64) = hi;
                        } else
#endif
                        if ((!FLAT || a.store_out) && st_ok) {
                            to_runs(kh, lo, hi);
                            *(u32x4*)(op + nt * 64) = lo;
                            *(u32x4*)(op + nt * 64 + 32) = hi;
                            to_runs(kl, lo, hi);
                            *(u32x4*)(op + a.lo_delta + nt * 64) = lo;
                            *(u32x4*)(op + a.lo_delta + nt * 64 + 32) = hi;
                        }
                        if constexpr (RES) {                      // the residual projection leaves un-activated (its bias came in through C)
                            // as fp32 accumulator fragments (r_mtile above): launch B adds them to its accumulator as they are
                            char* rq = (char*)a.res_out + r_mtile(cur, co0) + nt * 2048 + lane * 32u;
                            const f32x16& rv = racc[nt];
                            *(f32x4*)(rq) = f32x4{rv[0], rv[1], rv[2], rv[3]};
                            *(f32x4*)(rq + 16) = f32x4{rv[4], rv[5], rv[6], rv[7]};
                            *(f32x4*)(rq + a.lo_delta) = f32x4{rv[8], rv[9], rv[10], rv[11]};
                            *(f32x4*)(rq + a.lo_delta + 16) = f32x4{rv[12], rv[13], rv[14], rv[15]};
                        }
                        if constexpr (POOL) {                     // 2x2 max over the quad in fp32, then split
                            float pv[16];
    #pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                // the values are >= 0 (after the ReLU): their bit patterns order like the values, and an integer max
                                // needs no NaN quieting first and takes the lane permutation as a modifier -- 2 instructions, not 6
                                int x = __builtin_bit_cast(int, v[r]);
                                x = max(x, __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true));
                                x = max(x, __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true));
                                pv[r] = __builtin_bit_cast(float, x);
                            }
                            const uint32_t p_tile = ((((uint32_t)cur.n * (H >> 1) + (cur.y0 >> 1) + wave) * (W >> 1) + (cur.x0 >> 1)) * Cout + co0) * 2u;
                            char* pp = (char*)a.pool_out + (p_tile + pl_off);
#ifdef SS_DEVBUILD
                            if (a.dbg & 128) pp = (char*)a.pool_out + ((p_tile + pl_off) & 0x3ffffu);
#endif
                            const auto pk2 = split_store(pv, pp);
                            Packed ph = pk2.first, pl = pk2.second;
                            to_runs(ph, lo, hi);
                            const u32x4 pvh = (m & 1) ? hi : lo;
                            to_runs(pl, lo, hi);
                            const u32x4 pvl = (m & 1) ? hi : lo;
                            if ((m & 3) < 2 && st_ok) {
                                *(u32x4*)(pp + nt * 64 + (m & 1) * 32) = pvh;
                                *(u32x4*)(pp + a.lo_delta + nt * 64 + (m & 1) * 32) = pvl;
                            }
                        }
                    }
                }
                if (last && (((ovf & 0x7fff7fffu) + 0x04000400u) & 0x80008000u)) atomicOr(a.range_flag, 1);       // (rare: the engine turns it into SS_ERR_RANGE)
#ifdef SS_DEVBUILD
                if (a.stamps) { const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime(); st_sum[11] += t - te0; }
#endif
                return;
            }
            if constexpr (RADD) {
    #pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    Packed rk;
                    from_runs(rlo[nt], rhi[nt], rk);
    #pragma unroll
                    for (int g = 0; g < 4; ++g)
    #pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            acc[nt][4 * g + 2 * h] += __builtin_bit_cast(float, rk.p[g][h] << 16);
                            acc[nt][4 * g + 2 * h + 1] += __builtin_bit_cast(float, rk.p[g][h] & 0xffff0000u);
                        }
                }
            }
            if (last) {                                       // registers -> memory: no staging, no barrier of its own
                char* op = (char*)a.out + (o_tile + st_off);
    #pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    Packed k;
    #pragma unroll
                    for (int g = 0; g < 4; ++g)
    #pragma unroll
                        for (int h = 0; h < 2; ++h) k.p[g][h] = relu_pk(pack_bf16(acc[nt][4 * g + 2 * h], acc[nt][4 * g + 2 * h + 1]));
                    if constexpr (FLAT) {
                        f32x16 dd;
    #pragma unroll
                        for (int r = 0; r < 16; ++r) dd[r] = 0.f;
    #pragma unroll
                        for (int sx = 0; sx < 2; ++sx) {
                            const bf16x8 bop = __builtin_bit_cast(bf16x8, u32x4{k.p[2 * sx][0], k.p[2 * sx][1], k.p[2 * sx + 1][0], k.p[2 * sx + 1][1]});
                            dd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[sx]), bop, dd, 0, 0, 0);
                        }
                        // rows 0..3 of the product (registers 0..3 of half-wave 0) = the 4 flatten channels with the upper mel row's
                        // weights, rows 4..7 (half-wave 1) with the lower row's: a pixel's own value is where the half-wave equals its row
    #pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4)
                            if (hh == py) sFlat[((wave * 2 + hh) * 4 + c4) * 16 + px] = dd[c4];
                    }
                    Packed kp;
                    if constexpr (POOL) {                     // 2x2 max over the quad (lanes 4q .. 4q+3), before the channel shuffle
    #pragma unroll
                        for (int g = 0; g < 4; ++g)
    #pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                uint32_t v = k.p[g][h];
                                v = max_pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                                v = max_pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
                                kp.p[g][h] = v;
                            }
                    }
                    u32x4 lo, hi;
                    if (!FLAT || a.store_out) {               // c9 itself is only needed when the spec head runs
                        to_runs(k, lo, hi);
                        *(u32x4*)(op + nt * 64) = lo;
                        *(u32x4*)(op + nt * 64 + 32) = hi;
                    }
                    if constexpr (RES) {                      // the residual projection leaves un-activated (its bias came in through C)
                        Packed kr;
    #pragma unroll
                        for (int g = 0; g < 4; ++g)
    #pragma unroll
                            for (int h = 0; h < 2; ++h) kr.p[g][h] = pack_bf16(racc[nt][4 * g + 2 * h], racc[nt][4 * g + 2 * h + 1]);
                        to_runs(kr, lo, hi);
                        char* rp = (char*)a.res_out + (o_tile + st_off);
                        *(u32x4*)(rp + nt * 64) = lo;
                        *(u32x4*)(rp + nt * 64 + 32) = hi;
                    }
                    if constexpr (POOL) {
                        to_runs(kp, lo, hi);
                        // the four lanes of a quad hold the same pooled pixel: lane 0 writes its first run, lane 1 the second, so that
                        // one instruction (both half-waves) stores whole 64-byte pixels
                        const u32x4 pv = (m & 1) ? hi : lo;
                        if ((m & 3) < 2) {
                            const uint32_t p_tile = ((((uint32_t)cur.n * (H >> 1) + (cur.y0 >> 1) + wave) * (W >> 1) + (cur.x0 >> 1)) * Cout + co0) * 2u;
                            char* pp = (char*)a.pool_out + (p_tile + pl_off);
                            *(u32x4*)(pp + nt * 64 + (m & 1) * 32) = pv;
                        }
                    }
                }
            }
        };
        // PF2 launches (two stages of loads in flight already) ran faster with the epilogue ahead of the commit; the others with
        // it behind (their next loads go out a barrier and an epilogue earlier, and the stores get an MFMA phase to drain)
        constexpr bool EPI_EARLY = PF2;
        jitter(1);
        if constexpr (EPI_EARLY) epilogue();
        lds_barrier4();                                   // every wave is done reading this stage's LDS image
        jitter(2);
#ifdef SS_DEVBUILD
        uint32_t tw_i = 0;
#endif
        if constexpr (DUO) {
            // the half's off-phase: commit, loads, epilogue -- while the other half multiplies.  No barrier depends on ok1 / ok2.
            if (ok1) {
                commit(ra_a, kCommitPatch, kCommitBank);
                ok2 = next_stage(n1, n2);
                if (ok2) { issue_patch(n2.d, n2.ci, ra_a); issue_weights(n2.d, n2.ci); }
            }
            ring_duty(stage_no, max_stages >> 1);
            ++stage_no;
            jitter(4);
            epilogue();
            jitter(6);
            lds_barrier4();
            jitter(5);
            cs = n1; n1 = n2;
            const bool more = ok1;
            ok1 = ok1 && ok2;
            return more;
        }
        if constexpr (FLAT) { if (flat_pending) flat_reduce(flat_tile); }
        if (ok1) {
            if constexpr (FIRST) {
                if constexpr (!SPLIT || NEXT == 0) {      // a new tile's features (f16x2: its first stage)
                    if (tid < (PR + 2) * FW) sF[tid] = rf;
                    lds_barrier4();
                }
                jitter(3);
                produce(n1.d, NEXT);
            } else {
#ifdef SS_DEVBUILD
                uint32_t tw0 = 0;
                if (a.stamps) { tw0 = (uint32_t)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime(); st_sum[5] += t - tw0; tw0 = t; }
#endif
                commit(ra_a, kCommitPatch, kCommitBank);
#ifdef SS_DEVBUILD
                if (a.stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime(); st_sum[6] += t - tw0; tw_i = t; }
#endif
            }
            if constexpr (PF2) {
                ok3 = ok2 && next_stage(n2, n3);
                if (ok3) issue_patch(n3.d, n3.ci, ra_a);
            } else {
                ok2 = next_stage(n1, n2);
#ifdef SS_DEVBUILD
                if (a.stamps) { const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime(); st_sum[7] += t - tw_i; tw_i = t; }
#endif
                if (ok2) issue_patch(n2.d, n2.ci, ra_a);
#ifdef SS_DEVBUILD
                if (a.stamps) { const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime(); st_sum[8] += t - tw_i; tw_i = t;
                                const uint32_t t2 = (uint32_t)__builtin_amdgcn_s_memtime(); st_sum[9] += t2 - t; }
#endif
            }
            if (ok2) issue_weights(n2.d, n2.ci);
            if constexpr (SPLIT && RP > 0 && PART == 0) issue_proj(cur, ci >> 1, KZ_{}, KH_{});
            jitter(4);
            lds_barrier4();
        } else if constexpr (FLAT) {
            lds_barrier4();                               // the block's last stage: the previous tile's sums are read (flat_reduce
        }                                                 // above) before this tile's epilogue overwrites them
        jitter(5);
        if constexpr (!EPI_EARLY) epilogue();
        if constexpr (FLAT) { flat_pending = last; flat_tile = cur; }
        if (!ok1) {
            if constexpr (FLAT) { lds_barrier4(); if (flat_pending) flat_reduce(flat_tile); }
            return false;
        }
        cs = n1; n1 = n2; ok1 = ok2;
        if constexpr (PF2) { n2 = n3; ok2 = ok3; }
        return true;
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    if constexpr (DUO) {
        // beats: X = M B (C I E) B M ..., Y = B M B (C I E) B ...: Y starts one barrier late, X ends one barrier late
        for (int i = 0; i < half; ++i) lds_barrier4();
        for (int s = 0; s < max_stages; s += SPLIT ? 2 : 1) {
            if (s < my_stages) {
                stage(P0{}, ra0);
                if constexpr (SPLIT) stage(P1{}, ra0);
            } else {                                      // out of stages: keep the beat (and a loader tile its duty)
                lds_barrier4(); ring_duty(s, max_stages >> 1); lds_barrier4();
                if constexpr (SPLIT) { lds_barrier4(); ring_duty(s + 1, max_stages >> 1); lds_barrier4(); }
            }
        }
        for (int i = half; i < NH - 1; ++i) lds_barrier4();
    } else if constexpr (SPLIT) {
        while (stage(P0{}, ra0) && stage(P1{}, ra0)) {}
    } else if constexpr (PF2) {
        while (stage(P0{}, ra0) && stage(P0{}, ra1)) {}
    } else {
        while (stage(P0{}, ra0)) {}
    }
#ifdef SS_DEVBUILD
    if (a.stamps && lane == 0) {
        uint32_t* p = (uint32_t*)a.stamps + (((size_t)blockIdx.x * NH + half) * NW + wave) * 16;
        for (int i = 0; i < 5; ++i) p[i] = st_sum[i];
        p[5] = (uint32_t)jit_n; p[6] = st_sum[5]; p[7] = st_sum[6]; p[8] = st_sum[7]; p[9] = st_sum[8]; p[10] = st_sum[9]; p[11] = st_sum[10]; p[12] = st_sum[11];
    }
#endif
}

template <int NT, int NW, bool BRES, bool RES, bool RADD, bool POOL, int RP, bool FIRST, bool FLAT, bool PF2, bool SPLIT = false, bool RANK1 = false,
          int NH = 1>
static hipError_t launch_v4_k(const ConvArgs& a, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};              // (per device: kernels.h allow_full_lds)
    if (hipError_t e = allow_full_lds((const void*)conv3x3_v4_kernel<NT, NW, BRES, RES, RADD, POOL, RP, FIRST, FLAT, PF2, SPLIT, RANK1, NH>, attr_done)) return e;
    hipLaunchKernelGGL((conv3x3_v4_kernel<NT, NW, BRES, RES, RADD, POOL, RP, FIRST, FLAT, PF2, SPLIT, RANK1, NH>), dim3(grid), dim3(64 * NW * NH),
                       lds, s, a, total, lds_b);
    return hipGetLastError();
}

// DUO forms (f16x2, resident banks shared by the two halves): the plain A (RES) and B (RADD, + POOL) launches with 8-wave tiles
template <int NW, int NH, bool BRES>
static hipError_t launch_v4_duo(const ConvArgs& a, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    if constexpr (BRES && NH == 4) {
        if (a.plain) return launch_v4_k<1, NW, true, false, false, false, 0, false, false, false, true, false, NH>(a, total, lds_b, lds, grid, s);
    }
    if (a.res_out) return launch_v4_k<1, NW, BRES, true, false, false, 0, false, false, false, true, false, NH>(a, total, lds_b, lds, grid, s);
    if (a.pool_out) return launch_v4_k<1, NW, BRES, false, true, true, 0, false, false, false, true, false, NH>(a, total, lds_b, lds, grid, s);
    return launch_v4_k<1, NW, BRES, false, true, false, 0, false, false, false, true, false, NH>(a, total, lds_b, lds, grid, s);
}

// f16x2 launches: A = RES (h and r out), B = RADD (+ POOL), conv9_1.B = RADD + FLAT, conv1_1.B = RANK1 + POOL
template <int NT, int NW>
static hipError_t launch_v4_split(const ConvArgs& a, bool bres, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    if constexpr (NT == 1 && NW == 8) {
        if (a.rank1_src && a.first_w) return launch_v4_k<1, 8, true, false, false, true, 0, true, false, false, true, true>(a, total, lds_b, lds, grid, s);
        if (a.rank1_src) return launch_v4_k<1, 8, true, false, false, true, 0, false, false, false, true, true>(a, total, lds_b, lds, grid, s);
        if (a.flat_part && a.proj_w) return launch_v4_k<1, 8, true, false, false, false, 4, false, true, false, true>(a, total, lds_b, lds, grid, s);
        if (a.flat_part) return launch_v4_k<1, 8, true, false, true, false, 0, false, true, false, true>(a, total, lds_b, lds, grid, s);
        if (a.proj_w) return bres ? hipErrorInvalidValue : launch_v4_k<1, 8, false, false, false, true, 1, false, false, false, true>(a, total, lds_b, lds, grid, s);
    }
    if (a.res_out) return bres ? launch_v4_k<NT, NW, true, true, false, false, 0, false, false, false, true>(a, total, lds_b, lds, grid, s)
                               : launch_v4_k<NT, NW, false, true, false, false, 0, false, false, false, true>(a, total, lds_b, lds, grid, s);
    if (a.pool_out) return bres ? launch_v4_k<NT, NW, true, false, true, true, 0, false, false, false, true>(a, total, lds_b, lds, grid, s)
                                : launch_v4_k<NT, NW, false, false, true, true, 0, false, false, false, true>(a, total, lds_b, lds, grid, s);
    return bres ? launch_v4_k<NT, NW, true, false, true, false, 0, false, false, false, true>(a, total, lds_b, lds, grid, s)
                : launch_v4_k<NT, NW, false, false, true, false, 0, false, false, false, true>(a, total, lds_b, lds, grid, s);
}

// two-stage prefetch where the launch is LDS-limited to two blocks per CU anyway (resident weights) and NT <= 2 keeps it under 128 registers
static bool v4_pf2(bool bres, size_t lds, int NT, bool first, bool flat) {
    static const int env = dev_env("SOFTSPOKEN_PF2", 1);
    return env && bres && !first && !flat && NT <= 2 && lds * 3 > 160 * 1024;   // (callers exclude NT = 2 A launches: they would spill)
}

template <int NT, int NW, bool BRES, bool RES, bool RADD, bool POOL, int RP = 0, bool FIRST = false, bool FLAT = false>
static hipError_t launch_v4_t(const ConvArgs& a, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    if constexpr (BRES && !FIRST && !FLAT && RP == 0 && NT <= 2 && NW == 8 && !(RES && NT == 2)) {
        if (v4_pf2(true, lds, NT, false, false)) return launch_v4_k<NT, NW, BRES, RES, RADD, POOL, RP, FIRST, FLAT, true>(a, total, lds_b, lds, grid, s);
    }
    return launch_v4_k<NT, NW, BRES, RES, RADD, POOL, RP, FIRST, FLAT, false>(a, total, lds_b, lds, grid, s);
}

// K steps of the projection a stage of a "projection in B" launch carries: ceil(steps / chunks), one of 1, 2, 4
static int v4_rp(const ConvArgs& a) {
    if (!a.proj_w) return 0;
    const int steps = (a.C0x + a.C1x) / 16, nch = (a.C0 + a.C1) / 32;
    const int per = (steps + nch - 1) / nch;
    return per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : per <= 6 ? 6 : -1;
}

template <int NT, int NW>
static hipError_t launch_v4_kind(const ConvArgs& a, bool bres, int total, int lds_b, size_t lds, int grid, hipStream_t s) {
    const int rp = v4_rp(a);
    if constexpr (NT == 1 && NW == 8) {
        if (a.first_w) return launch_v4_t<1, 8, true, false, false, true, 0, true>(a, total, lds_b, lds, grid, s);
        if (a.flat_part) return rp == 4 ? launch_v4_t<1, 8, true, false, false, false, 4, false, true>(a, total, lds_b, lds, grid, s)
                                        : launch_v4_t<1, 8, true, false, true, false, 0, false, true>(a, total, lds_b, lds, grid, s);
    }
    if (rp) {                                             // "projection in B" launches: the instantiations the network needs
        if constexpr (NT == 2 && NW == 8) { if (bres && a.pool_out && rp == 1) return launch_v4_t<2, 8, true, false, false, true, 1>(a, total, lds_b, lds, grid, s); }
        if constexpr (NT == 3 && NW == 8) { if (!bres && a.pool_out && rp == 2) return launch_v4_t<3, 8, false, false, false, true, 2>(a, total, lds_b, lds, grid, s); }
        if constexpr (NT == 2 && NW == 8) { if (!bres && a.pool_out && rp == 2) return launch_v4_t<2, 8, false, false, false, true, 2>(a, total, lds_b, lds, grid, s); }
        if constexpr (NT == 1 && NW == 4) { if (!bres && !a.pool_out && rp == 2) return launch_v4_t<1, 4, false, false, false, false, 2>(a, total, lds_b, lds, grid, s); }
        if constexpr (NT == 2 && NW == 8) { if (bres && !a.pool_out && rp == 6) return launch_v4_t<2, 8, true, false, false, false, 6>(a, total, lds_b, lds, grid, s); }
        return hipErrorInvalidValue;
    }
    if (a.plain) return bres ? launch_v4_t<NT, NW, true, false, false, false>(a, total, lds_b, lds, grid, s)
                             : launch_v4_t<NT, NW, false, false, false, false>(a, total, lds_b, lds, grid, s);
    if (a.res_out) return bres ? launch_v4_t<NT, NW, true, true, false, false>(a, total, lds_b, lds, grid, s)
                               : launch_v4_t<NT, NW, false, true, false, false>(a, total, lds_b, lds, grid, s);
    if (a.pool_out) return bres ? launch_v4_t<NT, NW, true, false, true, true>(a, total, lds_b, lds, grid, s)
                                : launch_v4_t<NT, NW, false, false, true, true>(a, total, lds_b, lds, grid, s);
    return bres ? launch_v4_t<NT, NW, true, false, true, false>(a, total, lds_b, lds, grid, s)
                : launch_v4_t<NT, NW, false, false, true, false>(a, total, lds_b, lds, grid, s);
}

#ifndef SS_RPROJ_DEFAULT
#define SS_RPROJ_DEFAULT 2
#endif
#ifndef SS_DUO_DEFAULT
#define SS_DUO_DEFAULT 4
#endif
struct V4Choice { bool ok; int nw, total, lds_b, grid; bool bres; size_t lds; int duo; };

static V4Choice choose_v4(ConvArgs& a, int NT, int num_cus, int prec) {
    V4Choice c{};
    if (prec != 1 && prec != 2) return c;
    const bool split = prec == 2;
    if (!a.relu || a.R0 || a.R1) return c;
    const bool first = a.first_w != nullptr, flat = a.flat_part != nullptr, proj = a.proj_w != nullptr;
    const bool rank1 = split && a.rank1_src != nullptr;
    if (split) {      // forms of the f16x2 mode: A with the r tensor, B adding it (+ pool, + flatten), conv1_1.B with the rank-1 residual
        if (a.lo_delta <= 0) return c;
        // "projection in B" (no r tensor): conv9_1.B (flatten form, four K steps on its one chunk) and conv2_1.B (two groups, pool, one
        // step per chunk); their A launches are `plain` and exist in the four-tile resident form only (checked below)
        // (conv9_1 in this form: A 4020 -> 3440 us per 1005 windows, its flatten B launch 2375 -> 2740 us with the four steps in two
        // halves around part 1's loop -- with all eight fragments in flight at once it spilled 100 bytes and took 3640 us;
        // SOFTSPOKEN_RPROJ in the dev build: 0 = no block, 1 = conv2_1 only, 2 = conv2_1 and conv9_1)
        static const int rproj_env = dev_env("SOFTSPOKEN_RPROJ", SS_RPROJ_DEFAULT);
        if (proj && !(NT == 1 && a.H % 16 == 0 && ((flat && v4_rp(a) == 4 && rproj_env == 2) || (!flat && a.pool_out && a.Cout == 64 && v4_rp(a) == 1)))) return c;
        if (first && !rank1) return c;
        if (rank1 && !(NT == 1 && a.rank1_w && a.pool_out && !a.res_out && !a.res_in && !flat && a.C0 == 32 && a.C1 == 0 && a.H % 16 == 0)) return c;
        if (NT == 2 || !(NT == 1 || (a.H % 16 == 0))) return c;                                  // instantiated: NT = 1 (8- and 4-wave tiles), NT = 3 (8-wave)
    }
    const int rp = v4_rp(a);
    if (flat && !(NT == 1 && a.Cout == 32 && a.C0 == 32 && a.C1 == 0 && a.H % 16 == 0 && a.flat_w4 && (a.res_in || rp == 4) && !a.pool_out && !first)) return c;
    if (first) {                                                                                  // conv1_1.B: features in, c1 + p1 out
        if (!(NT == 1 && a.Cout == 32 && a.C0 == 32 && a.C1 == 0 && a.H % 16 == 0 && a.first_b && a.rank1_src && a.rank1_w && a.pool_out &&
              !a.res_out && !a.res_in && !proj && !a.plain)) return c;
    } else if (proj) {                                                                            // B launch that computes the projection itself
        if (a.rank1_src || a.res_out || a.res_in || a.plain || rp < 0 || !a.xp0 || a.C0x % 16 || a.C1x % 16 || (a.C1x && !a.xp1)) return c;
        if (a.C0 != a.Cout || a.C1 != 0) return c;
        if ((double)a.N * a.H * a.W * std::max(a.C0x, a.C1x) * 2.0 + kHdr >= 4294967296.0) return c;
    } else if (a.plain) {                                                                         // A launch without the projection
        if (a.rank1_src || a.res_out || a.res_in || a.pool_out) return c;
    } else if (!rank1) {
        if (a.rank1_src) return c;
        if (!(a.res_out || a.res_in) || (a.res_out && (a.res_in || a.pool_out))) return c;        // A launch or B launch of a ResBlock
    }
    if (a.W % 16 != 0 || a.H % 8 != 0 || a.Cout % (32 * NT) != 0 || NT < 1 || NT > 3) return c;
    if (a.C0 % 32 || a.C1 % 32 || (a.C1 && ((a.H | a.W) & 1))) return c;
    if ((double)a.N * a.H * a.W * std::max(a.Cout, std::max(a.C0, a.C1)) * 2.0 + kHdr >= 4294967296.0) return c;   // 32-bit byte offsets
    c.nw = (a.H % 16 == 0) ? 8 : 4;
    const int th = 2 * c.nw;
    a.tiles_y = a.H / th; a.tiles_x = a.W / 16;
    const int ngroups = a.Cout / (32 * NT);
    const long total_l = (long)a.N * a.tiles_y * a.tiles_x * ngroups;
    if (total_l <= 0 || total_l > 0x7fffffff) return c;
    c.total = (int)total_l;
    const int tap_bytes = 2 * NT * 1024;
    const int taps = a.res_out ? 10 : 9;
    const int all_taps = ((a.C0 + a.C1) / 32) * taps;
    static const int bres_kb = dev_env("SOFTSPOKEN_BRES_KB", 72);
    const int banks = split ? 2 : 1;                                                              // f16x2: high and low halves of the weights
    c.bres = ngroups == 1 && (size_t)all_taps * tap_bytes * banks <= (size_t)((first || flat || rank1) ? 72 : bres_kb) * 1024;
    c.lds_b = c.bres ? all_taps * tap_bytes * banks : taps * tap_bytes * banks;
    if ((first || flat || rank1) && !c.bres) return c;
    if (proj && !flat && !((NT == 2 && c.nw == 8 && c.bres && a.pool_out && rp == 1) ||                       // conv2_1
                           (NT == 3 && c.nw == 8 && !c.bres && ngroups == 1 && a.pool_out && rp == 2) ||      // conv3_1
                           (NT == 2 && c.nw == 8 && !c.bres && ngroups > 1 && a.pool_out && rp == 2) ||       // conv4_1
                           (NT == 1 && c.nw == 4 && !c.bres && ngroups > 1 && !a.pool_out && rp == 2) ||      // conv_bottleneck, encoder_out
                           (split && NT == 1 && c.nw == 8 && !c.bres && ngroups > 1 && a.pool_out && rp == 1) ||      // conv2_1 in f16x2
                           (NT == 2 && c.nw == 8 && c.bres && !a.pool_out && rp == 6)))                       // conv7: its A launch gains more
                                                                                                              // (473 -> 349 us) than B loses (165 -> 222);
                                                                                                              // conv8 in this form: -46 / +144 us, not taken
        return c;                                                                                 // instantiated forms
    // DUO (conv3x3_v4_kernel): several tiles per 16-wave workgroup, a beat apart; one workgroup per CU.  2 x 8 waves or 4 x 4 waves
    // (SOFTSPOKEN_DUO in the dev build: 0, 2, 4)
    static const int duo_env = dev_env("SOFTSPOKEN_DUO", SS_DUO_DEFAULT);
    // (the four-tile form's tiles are 8 rows: it also takes the 8 x 16 level -- conv_bottleneck.A / encoder_out.A over the bank ring, one
    // whole picture per tile -- where the independent 4-wave blocks ran at 190 TFLOP/s)
    static const int duo8_env = dev_env("SOFTSPOKEN_DUO_H8", 1);
    if ((duo_env == 2 || duo_env == 4) && split && NT == 1 && (c.nw == 8 || (duo_env == 4 && duo8_env)) && !first && !flat && !proj && !rank1) {
        const int nh = duo_env, thd = 32 / nh;           // tile rows: 16 (8 waves) or 8 (4 waves)
        const size_t fixed = nh * (size_t)(thd + 2) * kRowPitch + (size_t)a.Cout * 4 * (a.res_out ? 2 : 1);
        const size_t chunk_b = (size_t)taps * tap_bytes * banks;
        const size_t all_b = (size_t)ngroups * all_taps * tap_bytes * banks;    // every channel group's banks (conv2_1.A: 2 x 40 KB)
        const bool bres = fixed + all_b <= 160 * 1024;
        static const int ring_env = dev_env("SOFTSPOKEN_RING", 1);
        // streamed banks: a two-slot ring shared by the four tiles.  A launches: conv7.A 1440 -> 1370 us, conv8.A 2040 -> 1960 us,
        // conv4_1.A 415 -> 390 us per 1005 windows.  B launches (residual loads, the long epilogue): the large ones lost 6-7 % in this
        // form in round 2; from the 32 x 64 level down it wins (round 3, same box: conv_bottleneck.B / encoder_out.B 186 -> 128 us,
        // conv4_1.B 515 -> 500, conv7.B 564 -> 556; the 96-channel blocks' B launches as three groups: conv3_1.B 1233 -> 1197, conv6.B 306 -> 289)
        const bool ring = !bres && nh == 4 && ring_env && (a.res_out != nullptr || ring_env == 2 || (a.H <= 32 && a.res_in != nullptr));
        const size_t lds = fixed + (bres ? all_b : 2 * chunk_b);
        // measured (tools/ab_layers.sh, f16x2, 1005 windows, alternating runs on one box): with shared resident banks conv9_1.A
        // 4820 -> 4440 us as 2 x 8 waves and -> 4020 us as 4 x 4 waves (its 80 KB of banks fit beside the patches but not twice
        // beside one), conv8.B 650 -> 607 us (4 x 4); with streamed banks the 2 x 8 form lost 2-8 % (a half's bank commit sits in
        // the other half's multiply phase); conv9_1.B (FLAT) as 4 x 4: 2765 -> 2946 us, not taken (its epilogue is the long
        // phase, and the 8-row tiles read 11 % more halo)
        if (((bres && (!a.plain || nh == 4)) || (ring && !a.plain)) && lds <= 160 * 1024) {
            c.duo = nh; c.bres = bres; c.lds_b = (int)(bres ? all_b : chunk_b); c.lds = lds;
            c.nw = 16 / nh;
            a.tiles_y = a.H / thd;
            c.total = (int)((long)a.N * a.tiles_y * a.tiles_x * ngroups);
            c.grid = (num_cus + 7) / 8 * 8;
            if (c.grid * nh > c.total) c.grid = ((c.total + nh - 1) / nh + 7) / 8 * 8;
            c.ok = true;
            return c;
        }
    }
    if (split && a.plain) return c;                       // (f16x2 plain A launches: the four-tile resident form above or none)
    c.lds = (size_t)(th + 2) * kRowPitch + c.lds_b + (size_t)a.Cout * 4 * (a.res_out ? 2 : 1) + (first ? (size_t)(32 + (th + 5) * 20) * 4 : 0) +
            (flat ? (size_t)c.nw * 2 * 64 * 4 : 0) + (proj && (ngroups == 1 || split) ? (size_t)((a.C0x + a.C1x) / 16) * (split ? a.Cout / 32 : NT) * 1024 * banks : 0);
    int bpc = (int)((160 * 1024) / c.lds);
    if (bpc < 1) return c;
    if (bpc > 3) bpc = 3;
    { static const int cap = dev_env("SOFTSPOKEN_BPC", 3); if (bpc > cap) bpc = cap; }      // (dev build: fewer blocks per CU)
    c.grid = num_cus * bpc;
    if (c.grid > c.total) c.grid = c.total;
    c.grid = (c.grid + 7) / 8 * 8;
    c.ok = true;
    return c;
}

int conv_v4_flat_groups() { return 128 / 16; }

bool conv_v4_supports(const ConvArgs& a_in, int NT, int num_cus, int prec) {
    ConvArgs a = a_in;
    return choose_v4(a, NT, num_cus, prec).ok;
}

// conv3x3_v4_kernel<NT, NW, BRES, RES, RADD, POOL, RP, FIRST, FLAT, PF2, SPLIT, RANK1, NH> as rocprofv3 prints it
const char* conv_v4_variant(const ConvArgs& a_in, int NT, int num_cus, int prec) {
    static thread_local char buf[112];
    ConvArgs a = a_in;
    const V4Choice c = choose_v4(a, NT, num_cus, prec);
    if (!c.ok) return "conv3x3_v4_kernel<invalid>";
    auto tf = [](bool b) { return b ? "true" : "false"; };
    const bool res = a.res_out != nullptr, first = a.first_w != nullptr, flat = a.flat_part != nullptr;
    const int rp = v4_rp(a);
    const bool split = prec == 2, rank1 = split && a.rank1_src != nullptr;
    const bool radd = !res && !first && !a.plain && rp == 0 && !rank1;
    const bool pf2 = !split && rp == 0 && c.nw == 8 && !(res && NT == 2) && v4_pf2(c.bres, c.lds, NT, first, flat);
    if (split)
        snprintf(buf, sizeof buf, "conv3x3_v4_kernel<%d, %d, %s, %s, %s, %s, %d, %s, %s, false, true, %s, %d>", NT, c.nw, tf(c.bres), tf(res), tf(radd),
                 tf(!res && a.pool_out), rp, tf(first), tf(flat), tf(rank1), c.duo ? c.duo : 1);
    else
        snprintf(buf, sizeof buf, "conv3x3_v4_kernel<%d, %d, %s, %s, %s, %s, %d, %s, %s, %s, false, false, 1>", NT, c.nw, tf(c.bres), tf(res), tf(radd),
                 tf(!res && a.pool_out), rp, tf(first), tf(flat), tf(pf2));
    return buf;
}

hipError_t launch_conv3x3_v4(const ConvArgs& a_in, int NT, int num_cus, int prec, hipStream_t s) {
    ConvArgs a = a_in;
    const V4Choice c = choose_v4(a, NT, num_cus, prec);
    if (!c.ok) return hipErrorInvalidValue;
    if (prec == 2) {
        if (c.duo == 2) return launch_v4_duo<8, 2, true>(a, c.total, c.lds_b, c.lds, c.grid, s);
        if (c.duo == 4) return c.bres ? launch_v4_duo<4, 4, true>(a, c.total, c.lds_b, c.lds, c.grid, s)
                                      : launch_v4_duo<4, 4, false>(a, c.total, c.lds_b, c.lds, c.grid, s);
        if (c.nw == 8) {
            switch (NT) {
                case 1: return launch_v4_split<1, 8>(a, c.bres, c.total, c.lds_b, c.lds, c.grid, s);
                case 3: return launch_v4_split<3, 8>(a, c.bres, c.total, c.lds_b, c.lds, c.grid, s);
            }
        } else if (NT == 1) return launch_v4_split<1, 4>(a, c.bres, c.total, c.lds_b, c.lds, c.grid, s);
        return hipErrorInvalidValue;
    }
    if (c.nw == 8) {
        switch (NT) {
            case 1: return launch_v4_kind<1, 8>(a, c.bres, c.total, c.lds_b, c.lds, c.grid, s);
            case 2: return launch_v4_kind<2, 8>(a, c.bres, c.total, c.lds_b, c.lds, c.grid, s);
            case 3: return launch_v4_kind<3, 8>(a, c.bres, c.total, c.lds_b, c.lds, c.grid, s);
        }
    } else {
        switch (NT) {
            case 1: return launch_v4_kind<1, 4>(a, c.bres, c.total, c.lds_b, c.lds, c.grid, s);
            case 2: return launch_v4_kind<2, 4>(a, c.bres, c.total, c.lds_b, c.lds, c.grid, s);
            case 3: return launch_v4_kind<3, 4>(a, c.bres, c.total, c.lds_b, c.lds, c.grid, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace ss
