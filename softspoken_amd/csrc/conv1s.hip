// conv1_1 (ResBlock(1, 32) at 128 x 256: pytorch_neural_nets.py:7-41, 107, 156) as a ROW-STREAMING kernel, f16x2 mode.
//
// Why a structure of its own.  In conv4.hip this launch (FIRST + RANK1 + POOL, 8-wave blocks, 16 x 16-pixel tiles) is the one big launch
// that is bound by neither roof: 2.4 TB/s of HBM, the matrix pipe ~40 % busy, 7.2 vector instructions per matrix instruction
// (What's weak 4).  Its input is one channel, so everything a tile needs from memory is 20 x 20 floats -- and all of the
// stage machinery (patch image in LDS, two barriers per stage, the h1 patch produced by MFMA into LDS and read back as fragments) serves
// a 32-channel tensor, h1, that never needs to leave the registers it is born in:
//
//   * A wave owns a STRIP of 32 columns and walks down its rows.  An M-tile is one row of the strip: lane (m, hh) = column x0 + m,
//     channel half hh.  h1 of a row comes out of three MFMAs (feature neighbourhood x folded 1 -> 32 filter, hi / lo halves) as a
//     32-channel x 32-column accumulator tile; ReLU'd, split into f16 halves and packed pairwise, registers 8 s .. 8 s + 7 ARE the
//     B operand of the second conv's K step s (MI355X guide, "an accumulator tile as the next MFMA's operand") -- with the K order
//     permuted, which the weight banks are packed for (weights.hip pack_conv_stream).  The wave keeps h1 of three rows (48 registers).
//   * The 3 x 3 needs h1 at columns x - 1, x, x + 1.  Instead of shifting operands, the taps of one dx accumulate into their own
//     tile, P_dx[x'] = sum_dy W[dy][dx] h1[y + dy][x'], and out[x] = P_-1[x - 1] + P_0[x] + P_+1[x + 1]: two additions per value with a
//     wavefront DPP shift on one source (v_add_f32_dpp wave_shr:1 / wave_shl:1; tools/probes/dpp_wave_shift.hip shows gfx950 executes
//     them).  Lanes 0 and 31 of a tile have no neighbour and are not stored: a strip yields 30 columns, nine strips cover a row
//     (12.5 % of the products are spent on the overlap; in exchange there is NO LDS traffic but the weight fragments, and NO barrier
//     after the prologue: waves never wait for each other).
//   * Same products as conv4.hip's form (three f16 products per term, fp32 accumulate); the block's 1 -> 32 residual (rank 1) and both
//     biases ride in spare K slots of the matrix products; ReLU, both planes of c1 and of pool1 = maxpool2x2(c1) written from
//     registers (the row pair's first row waits in LDS words of its own).
// Work unit = (window, band of `rows` rows, strip); a wave's units are independent.  LDS: the second conv's banks (36 KB), per wave a
// feature patch (5 KB) and the pooling row (4 KB).  Two waves per SIMD.
//
// Two forms, one idea.  conv1_stream_kernel (below, first): one 32-column tile per strip row on v_mfma_f32_32x32x16_f16 -- the round's
// first version, kept in the DEVELOPMENT build (SOFTSPOKEN_C1S_FORM=32).  conv1_stream16_kernel (further down): two interleaved
// 16-pixel tiles on v_mfma_f32_16x16x32_f16 -- the product's form; its header says what changes.  Launch table (both): grid = one
// 512-thread block per CU (8 waves, 2 per SIMD, amdgpu_waves_per_eu(2, 2)), a wave takes units blockIdx * 8 + wave, + 8 * grid, ...;
// TRACK = per-value f16 range test (false when weights.hip proved the range: ConvPlan::s1_range_proven); dynamic LDS 109 / 95 KB.
#include "kernels.h"
#include <algorithm>
#include <type_traits>

namespace ss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr int kH = 128, kW = 256, kC = 32;
static constexpr int kStrips = 9, kStripCols = 30;       // valid columns per strip: lanes 1 .. 30 of the 32-column tile
static constexpr int kBank = 9 * 2 * 1024;               // one plane of the second conv's weights: [tap][K step][lane][16 B]
static constexpr int kS1Waves = 8;                       // two waves per SIMD (twelve at 168 registers measured the same within the noise, and sat on the edge of spilling)
static constexpr int kFPitch = 36;                       // floats per row of a wave's feature patch: columns x0 - 1 .. x0 + 32 (34) + 2 spare
static constexpr int kMaxRows = 32;                      // most rows of a work unit (the patch holds rows y0 - 2 .. y0 + rows + 1)
static constexpr int kFPatch = (kMaxRows + 5) * kFPitch; // floats per wave (+ one row that the last, unused look-ahead of a unit reads)

__device__ __forceinline__ uint32_t s1_pack(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
// lo = f16(x - hi) for a pair whose high halves are packed in `hi` (conv4.hip split_lo: exact subtraction in fp32, one rounding)
__device__ __forceinline__ uint32_t s1_split_lo(uint32_t hi, float x0, float x1) {
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
    return l;
}
__device__ __forceinline__ uint32_t s1_pk_max_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x16 s1_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void s1_half_swap(uint32_t& x, uint32_t& y) {     // lanes 32..63 of x <-> lanes 0..31 of y
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0]; y = r[1];
}
// lane i <- lane i - 1 / lane i + 1 over the whole wavefront (lane 0 / 63: zero).  Lanes 0 and 32 (31 and 63) of the result belong to
// tile columns without a neighbour in this strip; they are never stored.
__device__ __forceinline__ float s1_from_left(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float s1_from_right(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float s1_relu(float x) {      // integer max: negative floats (and -0) are negative integers (conv4.hip's form)
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// h1 of one strip row as the second conv's B operands: [K step][plane: 0 = high halves, 1 = low halves]
struct S1Row { u32x4 f[2][2]; };

// TRACK: test every stored high half for "beyond the f16 range" as conv4.hip does.  Not needed (and 16 vector instructions per row
// saved) when the host has PROVEN the range from the weights: a feature is sqrt(log10(mel + 1)) <= sqrt(log10(FLT_MAX)) = 6.21 or not
// finite -- the latter is caught where the features are loaded --, so |h1| and |c1| have bounds that weights.hip computes
// (ConvPlan::s1_range_proven).
template <bool TRACK>
__global__ __launch_bounds__(64 * kS1Waves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv1_stream_kernel(ConvArgs a, int rows_per_unit, int total_units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;                                      // [plane][tap][K step][lane][16 B]
    char* sK = smem + 2 * kBank;                          // [3][lane][16 B]: (unused), the first conv's bank of low halves, the rank-1 / bias operand
    char* sPrev = sK + 3 * 1024;                          // [wave][4][lane][16 B]: the even row of a row pair (ReLU'd fp32 values), for the pooling
    float* sF = (float*)(sPrev + kS1Waves * 4096);        // [wave][row][kFPitch]: the unit's features, zero outside the picture

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, m = lane & 31;
    for (int p = tid; p < 2 * kBank / 16; p += 64 * kS1Waves) *(u32x4*)(sW + p * 16) = *(const u32x4*)((const char*)a.wpk + (size_t)p * 16);
    // first conv's filter bank as the A operand of ONE K = 16 step: half-wave 0 holds taps (dy = -1: dx -1, 0, +1, pad; dy = 0: ..., pad),
    // half-wave 1 (dy = +1: ..., pad; zeros); as two f16 halves
    // The two pad slots of half-wave 0 carry the first conv's bias as an f16 pair (b_hi, b_lo) against features of 1.0: the bias comes
    // out of the products (22 bits of it, as of every weight) and the accumulator starts from the inline constant 0.
    // wr: the block's 1 -> 32 residual (rank 1) and the second conv's bias the same way -- K slots (wr_hi, wr_hi, wr_lo, b_hi, b_lo)
    // against (f_hi, f_lo, f_hi, 1, 1): ONE product per row instead of 16 multiply-adds and 16 bias moves.
    u32x4 wf_hi, wf_lo, wr;
    {
        const float* w9 = a.first_w;                      // [9][32], tap-major
        auto wv = [&](int t) { return w9[t * 32 + m]; };
        auto hi2 = [&](float x, float y) { return s1_pack(x, y); };
        auto lo2 = [&](float x, float y) { return s1_split_lo(s1_pack(x, y), x, y); };
        auto lo1 = [&](float x) { return x - (float)(_Float16)x; };
        const float b1 = a.first_b[m], b2 = a.bias[m], r1 = a.rank1_w[m];
        if (hh == 0) {
            wf_hi = u32x4{hi2(wv(0), wv(1)), hi2(wv(2), b1), hi2(wv(3), wv(4)), hi2(wv(5), lo1(b1))};
            wf_lo = u32x4{lo2(wv(0), wv(1)), lo2(wv(2), 0.f) & 0xffffu, lo2(wv(3), wv(4)), lo2(wv(5), 0.f) & 0xffffu};
            wr = u32x4{hi2(r1, r1), hi2(lo1(r1), b2), hi2(lo1(b2), 0.f), 0u};
        } else {
            wf_hi = u32x4{hi2(wv(6), wv(7)), hi2(wv(8), 0.f), 0u, 0u};
            wf_lo = u32x4{lo2(wv(6), wv(7)), lo2(wv(8), 0.f), 0u, 0u};
            wr = u32x4{0u, 0u, 0u, 0u};
        }
    }
    if (tid < 64) { *(u32x4*)(sK + 1024 + lane * 16) = wf_lo; *(u32x4*)(sK + 2048 + lane * 16) = wr; }   // (read back once per row)
    const uint32_t kOneHi = 0x3c000000u;                 // f16 pair (0, 1.0)
    const char* sKl = sK + lane * 16;
    __syncthreads();                                      // the only barrier: from here on the waves share nothing but read-only LDS

    // (Measured without effect on this kernel, each within the +-3 % between two runs: three waves per SIMD (168 registers, twelve waves per
    // CU) instead of two, starting the SIMD's waves a third of a row apart, a token that lets one wave of a SIMD at a time into its block
    // of products, s_setprio around that block, non-temporal stores for c1 (1931-1958 against 1937-1980 us): tools/experiments/README.md.  Timing-only
    // ablations of the dev build, per 1005 windows: everything 2037 us | without the stores 1635 | without the second conv's products
    // 1735 | without the pooled rows 1724 | without stores and products 821: the stores (5.4 GB: 3.3 TB/s of pure writes at this speed),
    // the matrix pipe (58 products per row: 1.09 ms at 100 %) and vector issue (~250 instructions per row) each fill 55-65 % of the
    // kernel's time and overlap imperfectly.)
    const char* wl_base = sW + lane * 16;
#ifdef SS_DEVBUILD
    // stamps (dev build, SOFTSPOKEN_STAMP_LAYER=conv1_1.B): shader-clock time per row, summed per wave: [0] products, [1] h1 production,
    // [2] combine + split + stores, [3] pooled row, [4] the fence in front of the products (one stamp's own cost); [5] rows
    uint32_t st_sum[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u}, st_prev = 0;       // [6]: a unit's prologue (patch fill, the first two h1 rows)
    auto stamp = [&](int seg) {
        if (a.stamps) { const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime(); if (seg >= 0) st_sum[seg] += t - st_prev; st_prev = t; }
    };
#else
    auto stamp = [&](int) {};
#endif
    uint32_t ovf = 0;                                     // the largest high halves stored (conv4.hip: all exponent bits set = infinity / NaN)

    for (int unit = (int)blockIdx.x * kS1Waves + wave; unit < total_units; unit += (int)gridDim.x * kS1Waves) {
        stamp(-1);
        const int bands = kH / rows_per_unit;
        const int s = unit % kStrips, b = (unit / kStrips) % bands, n = unit / (kStrips * bands);
        const int x0 = kStripCols * s - 1, y0 = b * rows_per_unit;
        const int x = x0 + m;                             // this lane's column
        const bool col_in = (unsigned)x < (unsigned)kW;
        const uint32_t keep = col_in ? 0xffffffffu : 0u;  // h1 outside the picture is the second conv's zero padding: only the first and the
        const bool edge_strip = s == 0 || s == kStrips - 1;  // last strip have such columns
        const bool st_lane = m >= 1 && m <= kStripCols && x < kW;          // lanes whose results are stored
        const bool pl_lane = st_lane && (m & 1) && m < kStripCols;         // ... and the left lane of a pooled pair (x even)
        // ---- the unit's feature patch: rows y0 - 2 .. y0 + rows + 1, columns x0 - 1 .. x0 + 32, zero outside the picture (the first conv's
        // zero padding), into this wave's LDS region: all loads in flight together, ONE wait per unit.  (Feature loads inside the row loop
        // would make every row wait on vmcnt, which on gfx9 also counts the row's stores: a store's round trip per row.) ----
        float* pf = sF + wave * kFPatch;
        {
            const float* fn = a.rank1_src + (size_t)n * kH * kW;
            const int nrow = rows_per_unit + 4;
            const int c = lane < 34 ? lane : 33, gx = x0 - 1 + c;
            const bool cok = lane < 34 && (unsigned)gx < (unsigned)kW;
            const int gxc = min(max(gx, 0), kW - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this wave's reads of the previous unit's patch have returned)
            bool fbad = false;
            // (all of the unit's loads in flight together -- the registers of the row loop are free here --, one wait, then the writes)
            constexpr int NR = kMaxRows + 4;
            float tv[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int gy = y0 - 2 + j;
                const bool ok = cok && (unsigned)gy < (unsigned)kH && j < nrow;
                const float t = fn[(size_t)min(max(gy, 0), kH - 1) * kW + gxc];
                // a feature that is not finite (a NaN or infinite sample in a float WAV) is reported here: behind the matrix products a
                // NaN may carry either sign, and the integer ReLU below turns a negative one into 0
                fbad |= (__builtin_bit_cast(uint32_t, t) & 0x7f800000u) == 0x7f800000u;
                tv[j] = ok ? t : 0.f;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j)
                if (lane < 34 && j < nrow) pf[j * kFPitch + c] = tv[j];
            if (fbad) ovf = 0x7c007c00u;
        }
        // this lane's reads: row (r - 1 + hh) .. of the patch for h1 row r sit at patch row (r - y0 + 1 + hh) ..; columns m, m + 1, m + 2
        const float* pl = pf + m;

        S1Row H0, H1, H2;                                 // h1 of rows y - 1, y, y + 1 of the output row y being computed, in rotating roles
        char* prevp = sPrev + wave * 4096 + lane * 16;    // the even row of a row pair waits here (16 registers otherwise)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) { H0.f[q][pl] = u32x4{0u, 0u, 0u, 0u}; H1.f[q][pl] = u32x4{0u, 0u, 0u, 0u}; H2.f[q][pl] = u32x4{0u, 0u, 0u, 0u}; }

        // ---- produce h1 of row r into Hn (the loads of the row two steps on are requested first) ----
        // features of the row about to be produced, requested a whole row ahead (behind the previous row's products)
        f32x4 fa = {0.f, 0.f, 0.f, 0.f}, fb = fa;
        float fcen = 0.f;                                 // f(y, x) of the output row that goes with it (rank-1 term)
        auto request = [&](int r) {                       // for produce(r) and the output row r - 1
            const float* q0 = pl + (r - y0 + 1 + 2 * hh) * kFPitch;
            fa = f32x4{q0[0], q0[1], q0[2], 0.f};
            const float* q1 = q0 + (hh ? 0 : kFPitch);
            fb = f32x4{q1[0], q1[1], q1[2], 0.f};
            fcen = pl[(r - y0 + 1) * kFPitch + 1];
        };
        auto produce = [&](int r, S1Row& Hn, const u32x4& kwl) {
            if ((unsigned)r < (unsigned)kH) {             // (wave-uniform)
                // half-wave 0: rows r - 1 (patch row r - y0 + 1), r, and 1.0 against the bias slots; half-wave 1: row r + 1, and zeros
                // against the zero half of the filter bank
                const f32x4 oa = fa;
                const f32x4 ob = hh ? f32x4{0.f, 0.f, 0.f, 0.f} : fb;
                uint32_t bh[4], bl[4];
                bh[0] = s1_pack(oa[0], oa[1]); bl[0] = s1_split_lo(bh[0], oa[0], oa[1]);
                bh[1] = s1_pack(oa[2], 0.f);   bl[1] = s1_split_lo(bh[1], oa[2], 0.f);
                bh[2] = s1_pack(ob[0], ob[1]); bl[2] = s1_split_lo(bh[2], ob[0], ob[1]);
                bh[3] = s1_pack(ob[2], 0.f);   bl[3] = s1_split_lo(bh[3], ob[2], 0.f);
                if (!hh) { bh[1] |= kOneHi; bh[3] |= kOneHi; }      // (bl: the low half of 1.0 is 0)
                const u32x4 boph = {bh[0], bh[1], bh[2], bh[3]}, bopl = {bl[0], bl[1], bl[2], bl[3]};
                f32x16 h;
#pragma unroll
                for (int i = 0; i < 16; ++i) h[i] = 0.f;
                h = s1_mfma(wf_hi, bopl, h);
                h = s1_mfma(kwl, boph, h);
                h = s1_mfma(wf_hi, boph, h);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) {
                        const float v0 = s1_relu(h[4 * g + 2 * hq]), v1 = s1_relu(h[4 * g + 2 * hq + 1]);
                        const uint32_t ph = s1_pack(v0, v1);
                        if constexpr (TRACK) ovf = s1_pk_max_u16(ovf, ph);     // (h1 beyond the f16 range: its high half is infinity)
                        Hn.f[g >> 1][0][2 * (g & 1) + hq] = ph;
                        Hn.f[g >> 1][1][2 * (g & 1) + hq] = s1_split_lo(ph, v0, v1);
                    }
                if (edge_strip) {                         // (wave-uniform)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                            for (int e = 0; e < 4; ++e) Hn.f[q][pl][e] &= keep;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) Hn.f[q][pl] = u32x4{0u, 0u, 0u, 0u};
            }
        };

        // one output row: h1 of row y + 1 into Hn, then the row's products against Ha (row y - 1), Hb (y), Hn (y + 1)
        auto row_step = [&](int y, const S1Row& Ha, const S1Row& Hb, S1Row& Hn) {
            const float fc = fcen;                        // f(y, x) (requested with row y + 1's features)
            stamp(-1);
            // the row's first weight fragments and the rank-1 / bias operand are requested before h1 is produced: their LDS round trip
            // (200-400 cycles with twelve waves reading) runs behind that work instead of in front of the first product
            constexpr int PD = 2, RS = PD + 1;
            u32x4 wh[RS], wl[RS];
            auto rd = [&](int g) {
                const int dy = g / 6, q = (g / 3) & 1, dx = g % 3;
                const int off = ((dy * 3 + dx) * 2 + q) * 1024;
                wh[g % RS] = *(const u32x4*)(wl_base + off);
                wl[g % RS] = *(const u32x4*)(wl_base + kBank + off);
            };
            const u32x4 wrq = *(const u32x4*)(sKl + 2048), kwl = *(const u32x4*)(sKl + 1024);
#pragma unroll
            for (int g = 0; g < PD; ++g) rd(g);
            __builtin_amdgcn_sched_barrier(0);
            produce(y + 1, Hn, kwl);
            request(y + 2);                               // (lands behind this row's products)
            stamp(1);
            // ---- second conv: P_dx = sum over dy, K steps of W[dy][dx] x h1[y + dy] (three products per term) ----
            f32x16 pm, p0, pp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { pm[i] = 0.f; p0[i] = 0.f; pp[i] = 0.f; }
            {   // rank-1 residual + bias: (f_hi, f_lo, f_hi, 1, 1) against (wr_hi, wr_hi, wr_lo, b_hi, b_lo), half-wave 0's K slots
                const uint32_t fh = s1_pack(fc, fc);      // (f_hi, f_hi)
                const uint32_t fl = s1_split_lo(fh, fc, fc);
                u32x4 bop = {(fh & 0xffffu) | (fl << 16), (fh & 0xffffu) | kOneHi, 0x3c00u, 0u};
                if (hh) bop = u32x4{0u, 0u, 0u, 0u};
                p0 = s1_mfma(wrq, bop, p0);
            }
            // 18 groups g = (dy, K step, dx) of three products; a group's two weight fragments are requested three groups ahead (the LDS
            // round trip is ~130 cycles loaded, a group's products 96).  The fences keep that order: left to itself the scheduler sinks
            // every read to just in front of its products and waits lgkmcnt(0) there.
            __builtin_amdgcn_sched_barrier(0);
            stamp(4);
            {
#ifdef SS_DEVBUILD
                if (!(a.relu & 32))                       // timing-only: no second-conv products
#endif
#pragma unroll
                for (int g = 0; g < 18; ++g) {
                    const int dy = g / 6, q = (g / 3) & 1, dx = g % 3;
                    const S1Row& Hr = dy == 0 ? Ha : (dy == 1 ? Hb : Hn);
                    const u32x4 xh = Hr.f[q][0], xl = Hr.f[q][1];
                    if (g + PD < 18) rd(g + PD);
                    __builtin_amdgcn_sched_barrier(0);
                    f32x16& acc = dx == 0 ? pm : (dx == 1 ? p0 : pp);
                    acc = s1_mfma(wh[g % RS], xl, acc);
                    acc = s1_mfma(wl[g % RS], xh, acc);
                    acc = s1_mfma(wh[g % RS], xh, acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            stamp(0);
            // ---- out[x] = P_-1[x - 1] + P_0[x] + P_+1[x + 1] + f(y, x) wr, ReLU, both planes, pooled pair rows ----
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float t = p0[i] + s1_from_left(pm[i]);
                t = t + s1_from_right(pp[i]);
                v[i] = s1_relu(t);
            }
            auto store_rows = [&](const float (&val)[16], char* dst, bool on, auto track_c) {
                constexpr bool track = TRACK && decltype(track_c)::value;
                uint32_t kh[8], kl[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    kh[i] = s1_pack(val[2 * i], val[2 * i + 1]);
                    // (every lane's value enters the range test, the unstored edge lanes' too: they are sums of the same size)
                    if (track) ovf = s1_pk_max_u16(ovf, kh[i]);
                    kl[i] = s1_split_lo(kh[i], val[2 * i], val[2 * i + 1]);
                }
                // (g, hh) pairs -> 16-byte runs (conv4.hip to_runs): pair index 2 g + hq; after the swaps a lane holds channels
                // [8 hh, 8 hh + 8) in its first run and [16 + 8 hh, 16 + 8 hh + 8) in its second
                s1_half_swap(kh[0], kh[2]); s1_half_swap(kh[1], kh[3]); s1_half_swap(kh[4], kh[6]); s1_half_swap(kh[5], kh[7]);
                s1_half_swap(kl[0], kl[2]); s1_half_swap(kl[1], kl[3]); s1_half_swap(kl[4], kl[6]); s1_half_swap(kl[5], kl[7]);
#ifdef SS_DEVBUILD
                if (a.relu & 16) on = false;              // timing-only: no stores
#endif
                if (on) {
                    *(u32x4*)(dst) = u32x4{kh[0], kh[1], kh[2], kh[3]};
                    *(u32x4*)(dst + 32) = u32x4{kh[4], kh[5], kh[6], kh[7]};
                    *(u32x4*)(dst + a.lo_delta) = u32x4{kl[0], kl[1], kl[2], kl[3]};
                    *(u32x4*)(dst + a.lo_delta + 32) = u32x4{kl[4], kl[5], kl[6], kl[7]};
                }
            };
            f32x4 prv[4];
            if (y & 1) {                                  // the pair's first row comes back from LDS behind the split and the stores below
#pragma unroll
                for (int g = 0; g < 4; ++g) prv[g] = *(const f32x4*)(prevp + g * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            store_rows(v, (char*)a.out + (((size_t)n * kH + y) * kW + (size_t)max(x, 0)) * (kC * 2) + hh * 16, st_lane, std::true_type{});
            stamp(2);
#ifdef SS_DEVBUILD
            if (a.relu & 64) return;                      // timing-only: no pooled rows
#endif
            if (y & 1) {                                  // (wave-uniform) second row of a pair: 2 x 2 maximum, values are >= 0
                float pv[16], prev[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    prev[4 * g] = prv[g][0]; prev[4 * g + 1] = prv[g][1]; prev[4 * g + 2] = prv[g][2]; prev[4 * g + 3] = prv[g][3];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    // (values are >= 0 behind the ReLU: their bit patterns order like the values -- an integer max needs no NaN quieting)
                    const int t = max(__builtin_bit_cast(int, prev[i]), __builtin_bit_cast(int, v[i]));
                    pv[i] = __builtin_bit_cast(float, max(t, __builtin_amdgcn_update_dpp(0, t, 0x130, 0xf, 0xf, true)));
                }
                store_rows(pv, (char*)a.pool_out + (((size_t)n * (kH / 2) + (y >> 1)) * (kW / 2) + (size_t)(max(x, 0) >> 1)) * (kC * 2) + hh * 16, pl_lane,
                           std::false_type{});           // (a pooled value is one of the values tested above)
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) *(f32x4*)(prevp + g * 1024) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            }
            stamp(3);
#ifdef SS_DEVBUILD
            ++st_sum[5];
#endif
        };
        request(y0 - 1); produce(y0 - 1, H0, *(const u32x4*)(sKl + 1024));
        request(y0);     produce(y0, H1, *(const u32x4*)(sKl + 1024));
        request(y0 + 1);
        stamp(6);
        // the three h1 rows change roles instead of places: three steps per turn (48 registers of moves per row otherwise)
        int k = 0;
        for (; k + 3 <= rows_per_unit; k += 3) {
            row_step(y0 + k, H0, H1, H2);
            row_step(y0 + k + 1, H1, H2, H0);
            row_step(y0 + k + 2, H2, H0, H1);
        }
        if (k < rows_per_unit) {
            row_step(y0 + k, H0, H1, H2);
            if (k + 1 < rows_per_unit) row_step(y0 + k + 1, H1, H2, H0);
        }
    }
#ifdef SS_DEVBUILD
    if (a.stamps && lane == 0) {
        uint32_t* p = (uint32_t*)a.stamps + ((size_t)blockIdx.x * kS1Waves + wave) * 16;
        for (int i = 0; i < 7; ++i) p[i] = st_sum[i];
    }
#endif
    if (((ovf & 0x7fff7fffu) + 0x04000400u) & 0x80008000u) atomicOr(a.range_flag, 1);       // (rare: the engine turns it into SS_ERR_RANGE)
}

// =========================================================================================================
// The same kernel on v_mfma_f32_16x16x32_f16 (round 4, late).  Measured first as a timing-only ablation (tools/experiments/
// r04_mfma_16x16x32_timing.patch): the second conv's 54 products per row as 108 of the small shape, operands unchanged: 2035 -> 1480 us per
// 1005 windows -- the small shape moves half the accumulator registers per multiply-add and costs about a quarter less energy, which under
// the power cap is time (tools/probes/mfma_valu_overlap.hip: a matrix-only loop 2.17 -> 1.65 us).  What changes with 16-pixel tiles:
//   * Layouts (tools/probes/mfma_16x16x32_layout.hip): A lane l = row l & 15, k = 8 (l >> 4) + j; B lane l = column l & 15, same k; D lane l
//     = column l & 15, rows 4 (l >> 4) + r.  Lane (g, c) = (l >> 4, l & 15).  A strip's 32 columns are TWO pixel tiles, INTERLEAVED: tile t
//     holds strip columns m = 2 c + t.  Then the 3 x 3's column neighbours of tile 1 are tile 0's values in the SAME lane and those of
//     tile 0 are tile 1's one lane over: out[0] = P0[0] + row_shr:1(P-[1]) + P+[1], out[1] = P0[1] + P-[0] + row_shl:1(P+[0]) -- as many
//     shifted additions as with one 32-column tile; and a pooled pair (x even, x + 1) is (tile 1 lane c, tile 0 lane c + 1): one DPP max.
//   * Channels: output row i = 4 g + r of channel tile u is channel 8 g + 4 u + r, in both convs.  So the first conv's two result tiles
//     of a lane ARE the next operand's eight K values in natural order (k = input channel), and a lane's results of both channel tiles
//     are channels 8 g .. 8 g + 7 of its pixel: one 16-byte store per pixel tile and plane, no half-wave swaps.
//   * First conv: K = 32 holds the 9 taps against (b_hi | b_lo) side by side: lanes g = 0 / 1 carry the high halves of rows (r - 1, r) /
//     (r + 1, and 1.0 for the bias pair), lanes g = 2 / 3 the low halves; w_hi x (b_hi + b_lo) + bias is ONE product, w_lo x b_hi a second
//     one on the same operand (its A is zero in the b_lo lanes).  8 products per row where the 32-column form issues 3 of twice the size.
// Per row and wave: 8 + 4 + 108 products of 16 matrix cycles (1920; the other form 58 of 32 = 1856), about the same vector instructions
// (operand building twice, no swaps, half the pooling).  Same numbers as the other form up to the summation order inside a product.
// =========================================================================================================
struct S16Row { u32x4 f[2][2]; };                        // [pixel tile][plane: 0 = high halves, 1 = low halves]: the second conv's B operands

__device__ __forceinline__ f32x4 s16_mfma(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int s16_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }   // lane c <- lane c - 1 of its 16 (c = 0: 0)
__device__ __forceinline__ int s16_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true); }   // lane c <- lane c + 1 (c = 15: 0)
__device__ __forceinline__ float s16_shr1f(float v) { return __int_as_float(s16_shr1(__float_as_int(v))); }
__device__ __forceinline__ float s16_shl1f(float v) { return __int_as_float(s16_shl1(__float_as_int(v))); }

static constexpr int kFPatch16 = (kMaxRows + 6) * kFPitch;   // + a row of zeros (the odd lane groups' second feature row)

template <bool TRACK>
__global__ __launch_bounds__(64 * kS1Waves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv1_stream16_kernel(ConvArgs a, int rows_per_unit, int total_units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;                                      // [plane][tap][channel tile][lane][16 B]
    char* sK = smem + 2 * kBank;                          // [4][lane][16 B]: the first conv's w_lo operand (channel tile 0, 1), the rank-1 / bias operand (0, 1)
    char* sPrev = sK + 4 * 1024;                          // [wave][2][lane][16 B]: the even row's column maxima of a row pair, for the pooling
    float* sF = (float*)(sPrev + kS1Waves * 2048);        // [wave][row][kFPitch]: the unit's features, zero outside the picture; last row: zeros

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    for (int p = tid; p < 2 * kBank / 16; p += 64 * kS1Waves) *(u32x4*)(sW + p * 16) = *(const u32x4*)((const char*)a.wpk + (size_t)p * 16);
    // first conv's filter as A operands (row i = lane & 15 of channel tile u is channel 8 (i >> 2) + 4 u + (i & 3)); K slots of lane group g:
    //   g = 0: taps of rows r - 1 and r (dx -1, 0, +1, pad each)      g = 1: taps of row r + 1, then the bias pair (b_hi, b_lo) against 1.0
    //   g = 2, 3: the same taps again, against the features' low halves (no bias)
    u32x4 wf1[2];                                         // w_hi (+ bias): stays in registers; w_lo and the rank-1 operand wait in LDS
    {
        const float* w9 = a.first_w;                      // [9][32], tap-major
        auto hi2 = [&](float x, float y) { return s1_pack(x, y); };
        auto lo2 = [&](float x, float y) { return s1_split_lo(s1_pack(x, y), x, y); };
        auto lo1 = [&](float x) { return x - (float)(_Float16)x; };
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = 8 * (c >> 2) + 4 * u + (c & 3);
            auto wv = [&](int t) { return w9[t * 32 + ch]; };
            const float b1 = a.first_b[ch], b2 = a.bias[ch], r1 = a.rank1_w[ch];
            u32x4 lo = {0u, 0u, 0u, 0u}, wr = {0u, 0u, 0u, 0u};
            if ((g & 1) == 0) {
                wf1[u] = u32x4{hi2(wv(0), wv(1)), hi2(wv(2), 0.f), hi2(wv(3), wv(4)), hi2(wv(5), 0.f)};
                if (g == 0) lo = u32x4{lo2(wv(0), wv(1)), lo2(wv(2), 0.f) & 0xffffu, lo2(wv(3), wv(4)), lo2(wv(5), 0.f) & 0xffffu};
            } else {
                wf1[u] = g == 1 ? u32x4{hi2(wv(6), wv(7)), hi2(wv(8), b1), hi2(lo1(b1), 0.f), 0u} : u32x4{hi2(wv(6), wv(7)), hi2(wv(8), 0.f), 0u, 0u};
                if (g == 1) lo = u32x4{lo2(wv(6), wv(7)), lo2(wv(8), 0.f) & 0xffffu, 0u, 0u};
            }
            // rank-1 residual + second conv's bias: K slots (wr_hi, wr_hi, wr_lo, b_hi, b_lo) against (f_hi, f_lo, f_hi, 1, 1), lane group 0 only
            if (g == 0) wr = u32x4{hi2(r1, r1), hi2(lo1(r1), b2), hi2(lo1(b2), 0.f), 0u};
            if (tid < 64) { *(u32x4*)(sK + u * 1024 + lane * 16) = lo; *(u32x4*)(sK + (2 + u) * 1024 + lane * 16) = wr; }
        }
    }
    if (lane < kFPitch) sF[wave * kFPatch16 + (kMaxRows + 5) * kFPitch + lane] = 0.f;     // this wave's row of zeros
    const uint32_t one1 = g == 1 ? 0x3c000000u : 0u, one2 = g == 1 ? 0x00003c00u : 0u;   // the 1.0s against the bias pair (K slots 3 and 4 of group 1)
    const bool lo_group = g >= 2;
    const char* sKl = sK + lane * 16;
    __syncthreads();                                      // the only barrier

    const char* wl_base = sW + lane * 16;
#ifdef SS_DEVBUILD
    uint32_t st_sum[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u}, st_prev = 0;      // (segments as in the 32-column form)
    auto stamp = [&](int seg) {
        if (a.stamps) { const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime(); if (seg >= 0) st_sum[seg] += t - st_prev; st_prev = t; }
    };
#else
    auto stamp = [&](int) {};
#endif
    uint32_t ovf = 0;

    for (int unit = (int)blockIdx.x * kS1Waves + wave; unit < total_units; unit += (int)gridDim.x * kS1Waves) {
        stamp(-1);
        const int bands = kH / rows_per_unit;
        const int s = unit % kStrips, b = (unit / kStrips) % bands, n = unit / (kStrips * bands);
        const int x0 = kStripCols * s - 1, y0 = b * rows_per_unit;
        const bool edge_strip = s == 0 || s == kStrips - 1;
        uint32_t keep[2]; bool st_lane[2];
        int xt[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int mm = 2 * c + t;                     // this lane's strip column in pixel tile t
            xt[t] = x0 + mm;
            keep[t] = (unsigned)xt[t] < (unsigned)kW ? 0xffffffffu : 0u;
            st_lane[t] = mm >= 1 && mm <= kStripCols && xt[t] < kW;
        }
        const bool pl_lane = st_lane[1] && 2 * c + 1 < kStripCols;      // tile 1's lane is the left (x even) pixel of a pooled pair
        // ---- the unit's feature patch (as in the 32-column form) ----
        float* pf = sF + wave * kFPatch16;
        {
            const float* fn = a.rank1_src + (size_t)n * kH * kW;
            const int nrow = rows_per_unit + 4;
            const int cc = lane < 34 ? lane : 33, gx = x0 - 1 + cc;
            const bool cok = lane < 34 && (unsigned)gx < (unsigned)kW;
            const int gxc = min(max(gx, 0), kW - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bool fbad = false;
            constexpr int NR = kMaxRows + 4;
            float tv[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int gy = y0 - 2 + j;
                const bool ok = cok && (unsigned)gy < (unsigned)kH && j < nrow;
                const float t = fn[(size_t)min(max(gy, 0), kH - 1) * kW + gxc];
                fbad |= (__builtin_bit_cast(uint32_t, t) & 0x7f800000u) == 0x7f800000u;
                tv[j] = ok ? t : 0.f;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j)
                if (lane < 34 && j < nrow) pf[j * kFPitch + cc] = tv[j];
            if (fbad) ovf = 0x7c007c00u;
        }
        // this lane's feature reads for h1 row r: patch row of picture row Y is Y - y0 + 2; even lane groups read rows r - 1 and r, odd
        // ones row r + 1 and the row of zeros; columns 2 c + t + (0, 1, 2) = the pixel's x - 1, x, x + 1
        const float* pcol = pf + 2 * c;
        const float* zrow = pf + (kMaxRows + 5) * kFPitch + 2 * c;

        S16Row H0, H1, H2;
        char* prevp = sPrev + wave * 2048 + lane * 16;
        float fa[2][3], fb[2][3], fcen[2];
        auto request = [&](int r) {                       // for produce(r) and the output row r - 1
            const float* q0 = pcol + (r - y0 + 1 + 2 * (g & 1)) * kFPitch;
            const float* q1 = (g & 1) ? zrow : q0 + kFPitch;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t][0] = q0[t]; fa[t][1] = q0[t + 1]; fa[t][2] = q0[t + 2];
                fb[t][0] = q1[t]; fb[t][1] = q1[t + 1]; fb[t][2] = q1[t + 2];
                fcen[t] = pcol[(r - y0 + 1) * kFPitch + t + 1];
            }
        };
        auto zero_row = [&](S16Row& Hn) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) Hn.f[t][pl] = u32x4{0u, 0u, 0u, 0u};
        };
        auto produce = [&](int r, S16Row& Hn, const u32x4 (&kwl)[2]) {
            if ((unsigned)r < (unsigned)kH) {             // (wave-uniform)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    uint32_t bh[4], bl[4];
                    bh[0] = s1_pack(fa[t][0], fa[t][1]); bl[0] = s1_split_lo(bh[0], fa[t][0], fa[t][1]);
                    bh[1] = s1_pack(fa[t][2], 0.f);      bl[1] = s1_split_lo(bh[1], fa[t][2], 0.f);
                    bh[2] = s1_pack(fb[t][0], fb[t][1]); bl[2] = s1_split_lo(bh[2], fb[t][0], fb[t][1]);
                    bh[3] = s1_pack(fb[t][2], 0.f);      bl[3] = s1_split_lo(bh[3], fb[t][2], 0.f);
                    bh[1] |= one1; bh[2] |= one2;
                    const u32x4 bop = lo_group ? u32x4{bl[0], bl[1], bl[2], bl[3]} : u32x4{bh[0], bh[1], bh[2], bh[3]};
                    uint32_t ph[2][2], pl[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        f32x4 h = {0.f, 0.f, 0.f, 0.f};
                        h = s16_mfma(kwl[u], bop, h);
                        h = s16_mfma(wf1[u], bop, h);
#pragma unroll
                        for (int hq = 0; hq < 2; ++hq) {
                            const float v0 = s1_relu(h[2 * hq]), v1 = s1_relu(h[2 * hq + 1]);
                            ph[u][hq] = s1_pack(v0, v1);
                            if constexpr (TRACK) ovf = s1_pk_max_u16(ovf, ph[u][hq]);
                            pl[u][hq] = s1_split_lo(ph[u][hq], v0, v1);
                        }
                    }
                    Hn.f[t][0] = u32x4{ph[0][0], ph[0][1], ph[1][0], ph[1][1]};
                    Hn.f[t][1] = u32x4{pl[0][0], pl[0][1], pl[1][0], pl[1][1]};
                    if (edge_strip) {                     // (wave-uniform) h1 outside the picture is the second conv's zero padding
#pragma unroll
                        for (int pq = 0; pq < 2; ++pq)
#pragma unroll
                            for (int e = 0; e < 4; ++e) Hn.f[t][pq][e] &= keep[t];
                    }
                }
            } else zero_row(Hn);
        };

        auto row_step = [&](int y, const S16Row& Ha, const S16Row& Hb, S16Row& Hn) {
            const float fc[2] = {fcen[0], fcen[1]};       // f(y, x) (requested with row y + 1's features)
            stamp(-1);
            // a group = one tap: both channel tiles' fragments (four 16-byte reads), requested one group ahead (a group's products: 192 cycles)
            u32x4 wh[2][2], wl[2][2];                     // [ring slot][channel tile]
            auto rd = [&](int tap) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    wh[tap & 1][u] = *(const u32x4*)(wl_base + (tap * 2 + u) * 1024);
                    wl[tap & 1][u] = *(const u32x4*)(wl_base + kBank + (tap * 2 + u) * 1024);
                }
            };
            const u32x4 kwl[2] = {*(const u32x4*)(sKl), *(const u32x4*)(sKl + 1024)};
            const u32x4 wrq[2] = {*(const u32x4*)(sKl + 2048), *(const u32x4*)(sKl + 3072)};
            rd(0);
            __builtin_amdgcn_sched_barrier(0);
            produce(y + 1, Hn, kwl);
            request(y + 2);
            stamp(1);
            // ---- second conv: P[dx][t][u] = sum over dy of W[dy][dx] x h1[y + dy] (three products per term) ----
            f32x4 P[3][2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {                 // rank-1 residual + bias start P[dx = 0]; the others start at 0
                const uint32_t fh = s1_pack(fc[t], fc[t]);
                const uint32_t fl = s1_split_lo(fh, fc[t], fc[t]);
                u32x4 bop = {(fh & 0xffffu) | (fl << 16), (fh & 0xffffu) | 0x3c000000u, 0x3c00u, 0u};
                if (g != 0) bop = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    P[0][t][u] = f32x4{0.f, 0.f, 0.f, 0.f}; P[2][t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    P[1][t][u] = s16_mfma(wrq[u], bop, f32x4{0.f, 0.f, 0.f, 0.f});
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            stamp(4);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
                const S16Row& Hr = dy == 0 ? Ha : (dy == 1 ? Hb : Hn);
                if (tap + 1 < 9) rd(tap + 1);
                __builtin_amdgcn_sched_barrier(0);
                // the four result tiles (pixel tile x channel tile) take turns: a product's accumulator was last written four products earlier
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            P[dx][t][u] = s16_mfma(pr == 1 ? wl[tap & 1][u] : wh[tap & 1][u], Hr.f[t][pr == 0 ? 1 : 0], P[dx][t][u]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            stamp(0);
            // ---- out[x] = P_-1[x - 1] + P_0[x] + P_+1[x + 1], ReLU: tile 0's neighbours are tile 1's lanes c - 1 / c, tile 1's are tile 0's c / c + 1 ----
            float v[2][2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[0][u][r] = s1_relu(P[1][0][u][r] + s16_shr1f(P[0][1][u][r]) + P[2][1][u][r]);
                    v[1][u][r] = s1_relu(P[1][1][u][r] + P[0][0][u][r] + s16_shl1f(P[2][0][u][r]));
                }
            // a lane's eight values of a pixel tile are channels 8 g .. 8 g + 7 of its pixel: one 16-byte run per plane
            auto store_px = [&](const float (&val)[2][4], char* dst, bool on, auto track_c) {
                constexpr bool track = TRACK && decltype(track_c)::value;
                uint32_t kh[4], kl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e0 = val[i >> 1][2 * (i & 1)], e1 = val[i >> 1][2 * (i & 1) + 1];
                    kh[i] = s1_pack(e0, e1);
                    if (track) ovf = s1_pk_max_u16(ovf, kh[i]);
                    kl[i] = s1_split_lo(kh[i], e0, e1);
                }
                if (on) {
                    *(u32x4*)(dst) = u32x4{kh[0], kh[1], kh[2], kh[3]};
                    *(u32x4*)(dst + a.lo_delta) = u32x4{kl[0], kl[1], kl[2], kl[3]};
                }
            };
            f32x4 prv[2];
            if (y & 1) { prv[0] = *(const f32x4*)(prevp); prv[1] = *(const f32x4*)(prevp + 1024); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                store_px(v[t], (char*)a.out + (((size_t)n * kH + y) * kW + (size_t)max(xt[t], 0)) * (kC * 2) + g * 16, st_lane[t], std::true_type{});
            stamp(2);
            // ---- pooling: the pair (x even, x + 1) is (tile 1 lane c, tile 0 lane c + 1); values are >= 0: integer maxima ----
            float cm[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cm[u][r] = __int_as_float(max(__float_as_int(v[1][u][r]), s16_shl1(__float_as_int(v[0][u][r]))));
            if (y & 1) {                                  // (wave-uniform) second row of a pair
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cm[u][r] = __int_as_float(max(__float_as_int(cm[u][r]), __float_as_int(prv[u][r])));
                store_px(cm, (char*)a.pool_out + (((size_t)n * (kH / 2) + (y >> 1)) * (kW / 2) + (size_t)(max(xt[1], 0) >> 1)) * (kC * 2) + g * 16, pl_lane,
                         std::false_type{});              // (a pooled value is one of the values tested above)
            } else {
                *(f32x4*)(prevp) = f32x4{cm[0][0], cm[0][1], cm[0][2], cm[0][3]};
                *(f32x4*)(prevp + 1024) = f32x4{cm[1][0], cm[1][1], cm[1][2], cm[1][3]};
            }
            stamp(3);
#ifdef SS_DEVBUILD
            ++st_sum[5];
#endif
        };
        {
            const u32x4 kwl0[2] = {*(const u32x4*)(sKl), *(const u32x4*)(sKl + 1024)};
            request(y0 - 1); produce(y0 - 1, H0, kwl0);
            request(y0);     produce(y0, H1, kwl0);
            request(y0 + 1);
        }
        stamp(6);
        int k = 0;
        for (; k + 3 <= rows_per_unit; k += 3) {
            row_step(y0 + k, H0, H1, H2);
            row_step(y0 + k + 1, H1, H2, H0);
            row_step(y0 + k + 2, H2, H0, H1);
        }
        if (k < rows_per_unit) {
            row_step(y0 + k, H0, H1, H2);
            if (k + 1 < rows_per_unit) row_step(y0 + k + 1, H1, H2, H0);
        }
    }
#ifdef SS_DEVBUILD
    if (a.stamps && lane == 0) {
        uint32_t* p = (uint32_t*)a.stamps + ((size_t)blockIdx.x * kS1Waves + wave) * 16;
        for (int i = 0; i < 7; ++i) p[i] = st_sum[i];
    }
#endif
    if (((ovf & 0x7fff7fffu) + 0x04000400u) & 0x80008000u) atomicOr(a.range_flag, 1);
}

bool conv1_stream_supports(const ConvArgs& a) {
    return a.H == kH && a.W == kW && a.Cout == kC && a.first_w && a.first_b && a.rank1_src && a.rank1_w && a.wpk && a.out && a.pool_out &&
           a.lo_delta != 0 && a.range_flag && a.N > 0;
}
const char* conv1_stream_variant(const ConvArgs& a, int form) {
    if (form == 16) return a.plain ? "conv1_stream16_kernel<false>" : "conv1_stream16_kernel<true>";
    return a.plain ? "conv1_stream_kernel<false>" : "conv1_stream_kernel<true>";
}
size_t conv1_stream_weight_bytes() { return 2 * (size_t)kBank; }

hipError_t launch_conv1_stream(const ConvArgs& a, int form, int rows_per_unit, int num_cus, hipStream_t s) {
    if (!conv1_stream_supports(a) || rows_per_unit < 2 || (rows_per_unit & 1) || kH % rows_per_unit || rows_per_unit > kMaxRows) return hipErrorInvalidValue;
    const int64_t total = (int64_t)a.N * (kH / rows_per_unit) * kStrips;
    if (total >= (int64_t)1 << 30) return hipErrorInvalidValue;
    const int cus = num_cus > 0 ? num_cus : 256;
    const int grid = (int)std::min<int64_t>(cus, (total + kS1Waves - 1) / kS1Waves);
    static std::atomic<uint64_t> attr_done{0}, attr_done2{0}, attr_done3{0}, attr_done4{0};
    if (form == 16) {                                     // 16-pixel tiles (wpk: pack_conv_stream16's banks)
        const size_t lds = 2 * (size_t)kBank + 4 * 1024 + (size_t)kS1Waves * 2048 + (size_t)kS1Waves * kFPatch16 * sizeof(float);
        if (hipError_t e = allow_full_lds((const void*)conv1_stream16_kernel<true>, attr_done3)) return e;
        if (hipError_t e = allow_full_lds((const void*)conv1_stream16_kernel<false>, attr_done4)) return e;
        if (a.plain) hipLaunchKernelGGL(conv1_stream16_kernel<false>, dim3(grid), dim3(64 * kS1Waves), lds, s, a, rows_per_unit, (int)total);
        else hipLaunchKernelGGL(conv1_stream16_kernel<true>, dim3(grid), dim3(64 * kS1Waves), lds, s, a, rows_per_unit, (int)total);
        return hipGetLastError();
    }
#ifdef SS_DEVBUILD
    // the 32-column form (the round's first version: one tile per strip row on v_mfma_f32_32x32x16_f16) stays in the development build for the comparison
    const size_t lds = 2 * (size_t)kBank + 3 * 1024 + (size_t)kS1Waves * 4096 + (size_t)kS1Waves * kFPatch * sizeof(float);
    if (hipError_t e = allow_full_lds((const void*)conv1_stream_kernel<true>, attr_done)) return e;
    if (hipError_t e = allow_full_lds((const void*)conv1_stream_kernel<false>, attr_done2)) return e;
    if (a.plain) hipLaunchKernelGGL(conv1_stream_kernel<false>, dim3(grid), dim3(64 * kS1Waves), lds, s, a, rows_per_unit, (int)total);   // (plain: range proven on the host)
    else hipLaunchKernelGGL(conv1_stream_kernel<true>, dim3(grid), dim3(64 * kS1Waves), lds, s, a, rows_per_unit, (int)total);
    return hipGetLastError();
#else
    (void)attr_done; (void)attr_done2;
    return hipErrorInvalidValue;
#endif
}

}  // namespace ss
