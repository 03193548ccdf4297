// Front-end kernels for gfx950: PCM decode + mixdown, polyphase resample, fused STFT -> power -> mel ->
// sqrt(log10(x+1)), and the overlap-averaging of per-window logits.
//
// Reference: root/code/backend/voice_activity.py:32-69 (load_audio), root/code/backend/pytorch_neural_nets.py:92-99,
// 144-153 (torchaudio MelSpectrogram(n_fft=2048, win_length=512, hop=256, n_mels=128, f_max=8000) then
// sqrt(log10(.+1)) and [:, :, :256]), root/code/frontend/NNDetector.py:153-190 (averaging).
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace ss {

// timing-only ablation bits of FrontendTables::dbg (results are wrong with them set).  Only the dev build's host code can set
// them (engine.hip: dev_env); the kernel tests them at run time in both builds on purpose: with the tests folded away the compiler
// schedules the frame loop differently and spills (256 VGPRs + 796 bytes of scratch per lane, 3.6 x slower on the GPU).
#define SS_FEDBG(tb) ((tb).dbg)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// =========================================================================================================
// Fused mel front-end.
//
// One frame = 512 samples x[256t-256 .. 256t+255] of the window (reflect at t = 0), Hann-weighted,
// zero-padded to 2048, |rFFT|^2, 128 triangular mel filters over bins 0..743, sqrt(log10(mel + 1)).
// The 2048-point real FFT is a 1024-point complex FFT of the 256 packed samples z[n] = x[2n] + i x[2n+1];
// only the first quarter of its input is non-zero, so with k = 4m + r it is four 256-point FFTs of
// z[n] * W1024^(n r) (the window and this pre-twiddle are one table).  A wave does one frame at a time:
// lane = (r, n0) holds the 16 points n = 16 n1 + n0 and runs a radix-16 FFT in registers, one LDS
// transpose, a second radix-16 -> Z[4(m0 + 16 m1) + r].  The real-FFT untangling, power, mel reduction
// and log scaling follow from LDS.  A block = 4 waves = 32 consecutive frames of one window; the
// [128 mel][32 frame] tile is staged in LDS and written as 64-byte row segments.
// Nothing but the input samples and the feature tile touches HBM (no 1025 x 259 spectrogram).
// =========================================================================================================

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// forward radix-4 butterfly, W4 = -i
__device__ __forceinline__ void radix4(float2 x0, float2 x1, float2 x2, float2 x3, float2& y0, float2& y1, float2& y2, float2& y3) {
    const float2 s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
    y0 = cadd(s02, s13);
    y2 = csub(s02, s13);
    y1 = make_float2(d02.x + d13.y, d02.y - d13.x);   // d02 - i d13
    y3 = make_float2(d02.x - d13.y, d02.y + d13.x);   // d02 + i d13
}

// 16-point forward DFT in registers, natural order in and out (4 x 4 Cooley-Tukey).
__device__ __forceinline__ void fft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    float2 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) radix4(v[j], v[j + 4], v[j + 8], v[j + 12], t[j][0], t[j][1], t[j][2], t[j][3]);
    // twiddles W16^(j b)
    t[1][1] = cmul(t[1][1], make_float2(C1, -S1));
    t[1][2] = cmul(t[1][2], make_float2(R2, -R2));
    t[1][3] = cmul(t[1][3], make_float2(S1, -C1));
    t[2][1] = cmul(t[2][1], make_float2(R2, -R2));
    t[2][2] = make_float2(t[2][2].y, -t[2][2].x);                 // W16^4 = -i
    t[2][3] = cmul(t[2][3], make_float2(-R2, -R2));
    t[3][1] = cmul(t[3][1], make_float2(S1, -C1));
    t[3][2] = cmul(t[3][2], make_float2(-R2, -R2));
    t[3][3] = cmul(t[3][3], make_float2(-C1, S1));               // W16^9
#pragma unroll
    for (int b = 0; b < 4; ++b) radix4(t[0][b], t[1][b], t[2][b], t[3][b], v[b], v[b + 4], v[b + 8], v[b + 12]);
}

// The same in packed fp32: a complex number is a 64-bit register pair, and an add, a subtract, a multiplication by -i (operand
// halves swapped, one negated) and each half of a complex product are single v_pk_*_f32 instructions -- the front-end kernel is
// bound by vector-instruction issue, and this form has 0.6x the instructions of the scalar one (the compiler folds the swaps and
// signs into op_sel / neg modifiers; built without the SLP vectoriser, which paired scalars with v_mov instead).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 rot_mi(f2 a) { return f2{a.y, -a.x}; }                  // a * (-i)
// a * b with b given as the table entry T = (b.x, b.x, -b.y, b.y): a * T.xy + swap(a) * T.zw -- the swap is an op_sel of the
// multiply, so a complex product is two packed instructions (a broadcast of a's high half or a negated half would each cost a move)
__device__ __forceinline__ f2 cmulT(f2 a, f32x4 T) { return a * f2{T[0], T[1]} + f2{a.y, a.x} * f2{T[2], T[3]}; }
__device__ __forceinline__ f2 cmulc(f2 a, float c, float sn) { return a * f2{c, c} + f2{a.y, a.x} * f2{-sn, sn}; }   // a * (c + i sn)
__device__ __forceinline__ void radix4v(f2 x0, f2 x1, f2 x2, f2 x3, f2& y0, f2& y1, f2& y2, f2& y3) {
    const f2 s02 = x0 + x2, d02 = x0 - x2, s13 = x1 + x3, d13 = x1 - x3;
    y0 = s02 + s13;
    y2 = s02 - s13;
    y1 = d02 + rot_mi(d13);                           // d02 - i d13
    y3 = d02 - rot_mi(d13);                           // d02 + i d13
}
__device__ __forceinline__ void fft16v(f2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    f2 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) radix4v(v[j], v[j + 4], v[j + 8], v[j + 12], t[j][0], t[j][1], t[j][2], t[j][3]);
    t[1][1] = cmulc(t[1][1], C1, -S1);
    t[1][2] = cmulc(t[1][2], R2, -R2);
    t[1][3] = cmulc(t[1][3], S1, -C1);
    t[2][1] = cmulc(t[2][1], R2, -R2);
    t[2][2] = rot_mi(t[2][2]);                        // W16^4 = -i
    t[2][3] = cmulc(t[2][3], -R2, -R2);
    t[3][1] = cmulc(t[3][1], S1, -C1);
    t[3][2] = cmulc(t[3][2], -R2, -R2);
    t[3][3] = cmulc(t[3][3], -C1, S1);                // W16^9
#pragma unroll
    for (int b = 0; b < 4; ++b) radix4v(t[0][b], t[1][b], t[2][b], t[3][b], v[b], v[b + 4], v[b + 8], v[b + 12]);
}

static constexpr int kTrRow = 18;           // float2 per transpose row (first structure, stft512): 16 + 2 pad (144 B) -> conflict-free b128 reads

// Wave-private LDS buffers are ordered by the LDS's in-order execution; this only pins the compiler (and must not wait
// on vmcnt: the next frame's sample loads are in flight).
__device__ __forceinline__ void fe_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

#ifdef SS_DEVBUILD      // the first structure: dev build only (SOFTSPOKEN_FEDBG=256), for A/B runs against the second
static constexpr int kFeWaves = 8;          // waves per block: each walks its own (window, 4-frame group) units.  8 x 9.2 KB of per-wave
                                            // buffers + 43 KB of shared tables = 117 KB of LDS.  12 waves fit (154 KB; 168 registers with the
                                            // tables read from LDS): 525 vs 510 us per 1024 windows with the pre-twiddles in registers here


// Persistent blocks (one per CU): the window x pre-twiddle table, the inter-pass twiddles, W2048^k and the mel weights
// are staged in LDS once per block (30 KB); a block then walks (window, 32-frame group) items.  Constants live in LDS,
// not registers, so that a wave needs ~110 VGPRs and two waves share a SIMD; the FFT buffers are wave-private, so the
// only block barriers are the two around the output tile.
__global__ __launch_bounds__(64 * kFeWaves) void frontend_v1_kernel(const float* __restrict__ arena, const int64_t* __restrict__ win_off,
                                                                 int n_windows, FrontendTables tb, float* __restrict__ feat) {
    __shared__ float4 s_pt[4 * 256];                  // (w0 c, w1 c | -w1 s, w0 s): z[n] * W1024^(n r) = (x, y) * first pair + (y, x) * second pair
    __shared__ float4 s_tw[16 * 16];                  // W256^(n0 m0), [m0][n0], as (re, re, -im, im): see cmulT
    __shared__ float4 s_wk[768];                      // exp(-2 pi i k / 2048), k < 768, likewise
    __shared__ float s_mw[64 * kMelPitch];            // per lane: kMelLo + kMelHi zero-padded mel weights
    __shared__ float2 s_tr[kFeWaves][64 * kTrRow];    // per-wave transpose / Z buffer (1152 float2 >= 1024)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 64 * kFeWaves) { const float4 p = tb.pretw[i]; s_pt[i] = make_float4(p.x, p.w, -p.y, p.z); }   // from (w0 c, w1 s, w0 s, w1 c)
    for (int i = tid; i < 256; i += 64 * kFeWaves) { const float2 w = tb.w2048[(8 * (i & 15) * (i >> 4)) & 2047]; s_tw[i] = make_float4(w.x, w.x, -w.y, w.y); }
    for (int i = tid; i < 768; i += 64 * kFeWaves) { const float2 w = tb.w2048[i]; s_wk[i] = make_float4(w.x, w.x, -w.y, w.y); }
    for (int i = tid; i < 64 * kMelPitch; i += 64 * kFeWaves) s_mw[i] = tb.mel_wp[i];
    __syncthreads();

    const int r = lane >> 4, q = lane & 15;           // pass 1: q = n0; pass 2: q = m0
    // the two mel filters of this lane (a long one and a short one: balanced)
    const int j1 = lane, j2 = 127 - lane;
    const int st1 = tb.mel_start[j1], st2 = tb.mel_start[j2];
    f2* tr = (f2*)s_tr[wave];
    float* pw = (float*)s_tr[wave];                   // the power spectrum takes the Z buffer's place once every lane has read its bins
    const int zw = 16 * r + ((q + 4 * r) & 15);       // this lane's slot in a 64-entry row of the Z buffer
    // window x pre-twiddle of this lane's 16 points: the same for every frame, kept in registers (the kernel is bound by LDS traffic:
    // 16 KB of the ~100 KB a frame moves through the LDS were these reads)
    f32x4 ptr[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) ptr[n1] = ((const f32x4*)s_pt)[r * 256 + q + 16 * n1];
    const f32x4* twl = (const f32x4*)s_tw + q;        // + 16 m0
    const f32x4* wk = (const f32x4*)s_wk;

    // samples of frame t of a window: z[16 n1 + q] = (x[i], x[i+1]), i = 32 n1 + 2 q, at window index i - 256 + 256 t
    auto load_samples = [&](const float* x, int t, f2 (&sm)[16]) {
        if (t > 0) {                                  // uniform base + one 32-bit lane offset + immediates: no 64-bit address math per load
            const char* xb = (const char*)x + (uint32_t)(256 * (t - 1) + 2 * q) * 4u;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) sm[n1] = *(const f2*)(xb + 128 * n1);
        } else {                                      // center=True, pad_mode='reflect': x[-k] = x[k]
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int i = 32 * n1 + 2 * q;
                const int a0 = i - 256, a1 = i - 255;
                sm[n1] = f2{x[a0 < 0 ? -a0 : a0], x[a1 < 0 ? -a1 : a1]};
            }
        }
    };

    // work unit = 4 consecutive frames of one window (64 units per window); the waves of a block share nothing but the tables
    const int64_t n_units = (int64_t)n_windows * 64;
    for (int64_t unit = (int64_t)blockIdx.x * kFeWaves + wave; unit < n_units; unit += (int64_t)gridDim.x * kFeWaves) {
        const int n = (int)(unit >> 6), f0 = (int)(unit & 63) * 4;
        const float* x = arena + win_off[n];
        f2 sm[16];
        load_samples(x, f0, sm);
        float o1[4], o2[4];                           // this wave's four frames of the lane's two mel rows
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            f2 v[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const f32x4 pt = ptr[n1];
                v[n1] = sm[n1] * f2{pt[0], pt[1]} + f2{sm[n1].y, sm[n1].x} * f2{pt[2], pt[3]};
            }
            if (!(SS_FEDBG(tb) & 1)) fft16v(v);                 // over n1 -> index m0
#pragma unroll
            for (int m0 = 0; m0 < 16; ++m0) v[m0] = cmulT(v[m0], twl[16 * m0]);
#pragma unroll
            for (int m0 = 0; m0 < 16; ++m0) tr[(r * 16 + m0) * kTrRow + q] = v[m0];
            fe_wave_sync();
            {
                const f32x4* row = (const f32x4*)(tr + (r * 16 + q) * kTrRow);   // 144-byte rows: 16-byte aligned
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const f32x4 w4 = row[p];
                    v[2 * p] = f2{w4[0], w4[1]};
                    v[2 * p + 1] = f2{w4[2], w4[3]};
                }
            }
            if (!(SS_FEDBG(tb) & 1)) fft16v(v);                 // over n0 -> index m1 ; v[m1] = Z[4 (q + 16 m1) + r]
            fe_wave_sync();
            // Z buffer: k = 4 (m0 + 16 m1) + r sits at 64 m1 + 16 r + ((m0 + 4 r) & 15).  (Without the rotation by 4 r the untangle's
            // reads below, lane -> (r, m0) = (lane & 3, lane >> 2), put r = 0 and r = 2 on the same banks: 2-way conflicts.)
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) tr[64 * m1 + zw] = v[m1];
            // the next frame's samples fly during the untangle and the mel sums (requested here, not before the FFT: their 32
            // registers would sit on top of the FFT's 64 and push a 12-wave block over its 168)
            if (f < 3 && !(SS_FEDBG(tb) & 16)) load_samples(x, f0 + f + 1, sm);
            fe_wave_sync();
            // real-FFT untangle + power for bins k < 768 (bins above 743 carry no mel weight).  k and 1024 - k come out of one
            // butterfly: X[k] = (a - i W^k d) / 2, X[1024 - k] = conj(a + i W^k d) / 2 with a = Z[k] + conj Z[1024-k], d = Z[k] - conj Z[1024-k];
            // so k = 64 i + lane covers 0..511 and the bins 513..767 ride along with 257..511 (512 pairs with itself).
            if (!(SS_FEDBG(tb) & 2)) {
                auto zat = [&](int kk) { return tr[64 * (kk >> 6) + 16 * (kk & 3) + (((kk >> 2) + 4 * (kk & 3)) & 15)]; };
                // with u = a - i W^k d:  |X[k]|^2 = |u|^2 / 4,  |X[1024 - k]|^2 = |a + i W^k d|^2 / 4.  The powers wait in registers until
                // every lane has read its Z bins, then overwrite the buffer (mel taps past bin 767 read the zeros written behind them)
                float pk[8], pc[4], p512 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 64 * i + lane;
                    const int kc = (1024 - k) & 1023;
                    const f2 zk = zat(k), zz = zat(kc);
                    const f2 zc = f2{zz.x, -zz.y};
                    const f2 a = zk + zc, d = zk - zc;
                    const f2 rw = rot_mi(cmulT(d, wk[k]));                // -i W^k d
                    const f2 u = a + rw, uu = u * u;
                    pk[i] = 0.25f * (uu.x + uu.y);
                    if (i >= 4) {                         // partner bin 1024 - k in 513..768 (768 itself, from k = 256, is not needed)
                        const f2 w = a - rw, ww = w * w;
                        pc[i - 4] = 0.25f * (ww.x + ww.y);
                    }
                }
                if (lane == 0) {                          // k = 512
                    const f2 zk = zat(512);
                    const f2 zc = f2{zk.x, -zk.y};
                    const f2 a = zk + zc, d = zk - zc;
                    const f2 u = a + rot_mi(cmulT(d, wk[512])), uu = u * u;
                    p512 = 0.25f * (uu.x + uu.y);
                }
                fe_wave_sync();
#pragma unroll
                for (int i = 0; i < 8; ++i) pw[64 * i + lane] = pk[i];
#pragma unroll
                for (int i = 4; i < 8; ++i) if (i > 4 || lane > 0) pw[1024 - (64 * i + lane)] = pc[i - 4];
                if (lane == 0) pw[512] = p512;
                if (lane < kMelHi) pw[768 + lane] = 0.f;
            }
            fe_wave_sync();
            {
                // the lane's two filters with fixed trip counts (weights zero-padded, the spectrum followed by zeros): no selects,
                // every LDS address is a per-lane base plus an immediate; the sums run in ascending bin order as before
                float m1s = 0.f, m2s = 0.f;
                if (!(SS_FEDBG(tb) & 4)) {
                    const float* p1 = pw + st1; const float* p2 = pw + st2;
                    const float* w1 = s_mw + lane * kMelPitch; const float* w2 = w1 + kMelLo;
#pragma unroll
                    for (int b = 0; b < kMelLo; ++b) m1s = fmaf(w1[b], p1[b], m1s);
#pragma unroll
                    for (int b = 0; b < kMelHi; ++b) m2s = fmaf(w2[b], p2[b], m2s);
                }
                // exactly as written in the reference: float32 log10(x + 1), then sqrt (no log1p, no fp64)
                if (SS_FEDBG(tb) & 8) { o1[f] = m1s; o2[f] = m2s; }
                else { o1[f] = sqrtf(log10f(m1s + 1.0f)); o2[f] = sqrtf(log10f(m2s + 1.0f)); }
            }
            fe_wave_sync();                               // pw / tr are rewritten by the next frame
        }
        // the wave's 4 consecutive frames of each mel row leave as one 16-byte store per lane and row: no output tile in LDS and
        // no block barrier anywhere in the loop (the waves of a block only share the read-only tables)
        float* dst = feat + (size_t)n * 128 * 256 + f0;
        *(f32x4*)(dst + j1 * 256) = f32x4{o1[0], o1[1], o1[2], o1[3]};
        *(f32x4*)(dst + j2 * 256) = f32x4{o2[0], o2[1], o2[2], o2[3]};
    }
}


#endif  // SS_DEVBUILD

// =========================================================================================================
// Fused mel front-end (second structure; the first one, round 1's, is kept in the dev build for A/B runs).
//
// What the first structure's counters showed (profiles/r01_pmc_conv.md): ~82 KB through the LDS per frame, a quarter of the LDS
// cycles lost to bank conflicts (mel sums), five LDS round trips per frame, 64 registers of window x pre-twiddle per lane.
// This structure removes those:
//   * 16 lanes own a frame (a wave: 4 consecutive frames) and the four quarter-spectra r = 0..3 (k = 4 m + r) are separate passes.
//     The pre-twiddle W1024^(n r) then splits into a compile-time part W64^(n1 r) and a part that merges with the inter-pass
//     twiddle into ONE table entry W1024^(n0 (4 m0 + r)).
//   * The real-FFT untangle's partner of bin k = 4 m + r is bin 1024 - k = 4 (255 - m) + (4 - r): lane 15 - m0, register
//     15 - m1 of pass 4 - r -- a DPP row_mirror, no spectrum buffer in LDS (r = 0 pairs m with 256 - m: mirror, then rotate by
//     one lane; lane 0 pairs inside itself).
//   * One 16 x 16 transpose per pass through an XOR-swizzled, unpadded 8 KB image; passes run two at a time (1 with 3, 0 with 2),
//     each through its own image, so that a wave has two independent dependency chains.  The power spectrum goes over the images
//     in two halves (frames 0-1, then 2-3), padded by 4 words per 64 bins so that filters starting 16 or 32 bins apart do not
//     share banks, and is read in 8-byte pairs; mel weights sit in registers for a half's two frames.
// ~30 KB through the LDS per frame, no bank conflicts in the mel sums, 3 LDS round trips per unit of four frames.
// Arithmetic: float32 throughout, log10f(x + 1.0f) then sqrtf as written in the reference.  Measured (MI355X, 1005 windows): the
// same ~510 us as the first structure -- the kernel is bound by vector-instruction issue, not by the LDS any more: ~1150 packed
// (4 cycles each: gfx950 issues v_pk_*_f32 at the scalar FLOP rate) + ~1280 other vector instructions per unit = ~7200 SIMD
// cycles per unit, i.e. a floor of ~190 us per 1005 windows at 2.4 GHz (26 % of the 8 TB/s roof for 395 672 B per window); at two
// waves per SIMD (256 registers; 12 waves spill) 37 % of that floor is reached.  DESIGN.md section 5 has the accounting.
// =========================================================================================================
static constexpr int kFe2Waves = 8;

// a * (s.x + i s.y) with the twiddle in one register pair: the broadcasts, the swap and the sign ride on op_sel / neg_lo
__device__ __forceinline__ f2 cmul2(f2 a, f2 s) {
#ifdef SS_FE_PLAIN_CMUL
    return a * f2{s.x, s.x} + f2{a.y, a.x} * f2{-s.y, s.y};
#endif
    f2 t, o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(s));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(o) : "v"(a), "v"(s), "v"(t));
    return o;
}
__device__ __forceinline__ float dpp_mirror(float v) {   // lane L of a 16-lane row <- lane 15 - L
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_ror1(float v) {     // lane L of a 16-lane row <- lane (L - 1) mod 16
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
}
__device__ __forceinline__ f2 mirror2(f2 v) { return f2{dpp_mirror(v.x), dpp_mirror(v.y)}; }

template <int R>
__device__ __forceinline__ void fe2_pretwiddle(const f2 (&xw)[16], f2 (&v)[16]) {   // v[n1] = xw[n1] * W64^(n1 R)
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        constexpr float kTwoPiOver64 = 0.09817477042468103f;
        const int e = (n1 * R) & 63;
        if (e == 0) v[n1] = xw[n1];
        else if (e == 16) v[n1] = rot_mi(xw[n1]);                          // W64^16 = -i
        else if (e == 32) v[n1] = f2{-xw[n1].x, -xw[n1].y};
        else if (e == 48) v[n1] = f2{-xw[n1].y, xw[n1].x};                 // +i
        else v[n1] = cmulc(xw[n1], __builtin_cosf(kTwoPiOver64 * (float)e), -__builtin_sinf(kTwoPiOver64 * (float)e));
    }
}

__global__ __launch_bounds__(64 * kFe2Waves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void frontend_kernel(const float* __restrict__ arena, const int64_t* __restrict__ win_off, int n_windows, FrontendTables tb,
                     float* __restrict__ feat) {
    __shared__ f2 s_tw[4 * 256];                       // [r][n0][m0]
    __shared__ f2 s_wk[3 * 256];                       // [r][m1][m0], r = 0..2 (bins 4 m + 3 come out of r = 1's butterflies)
    __shared__ f2 s_win[256];
    __shared__ __attribute__((aligned(16))) float s_mw[64 * kMelRow];
    __shared__ __attribute__((aligned(16))) char s_tr[kFe2Waves][2 * 8192];   // two transpose images per wave: two passes in flight

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 64 * kFe2Waves) { const float2 a = tb.twt[i]; s_tw[i] = f2{a.x, a.y}; }
    for (int i = tid; i < 768; i += 64 * kFe2Waves) { const float2 b = tb.wkt[i]; s_wk[i] = f2{b.x, b.y}; }
    for (int i = tid; i < 256; i += 64 * kFe2Waves) { const float2 a = tb.win2[i]; s_win[i] = f2{a.x, a.y}; }
    for (int i = tid; i < 64 * kMelRow; i += 64 * kFe2Waves) s_mw[i] = tb.mel_wq[i];
    __syncthreads();

    const int f = lane >> 4, q = lane & 15;            // frame of the unit, lane of the frame (n0 in pass A, m0 afterwards)
    char* tr = s_tr[wave];
    // transpose image: row (16 f + m0) of 128 bytes = 8 chunks of two complex numbers, chunk index XOR (row >> 1): the 16-byte row
    // reads of every lane group and the 8-byte column writes are conflict-free without padding
    // chunk j of this lane's column (writes, j = m0 >> 1) / row (reads) sits at base ^ (j << 4): bits 4..6 of the bases hold q >> 1
    const int wr_b0 = (f * 16) * 128 + (q & 1) * 8 + ((q >> 1) << 4);       // ^ ((m0 >> 1) << 4), + m0 * 128
    const int rd_b0 = (f * 16 + q) * 128 + ((q >> 1) << 4);                // ^ (p << 4)
    const int pw_wr = ((f & 1) * kPwWords + 4 * q) * 4;   // + 272 m1: this lane's four bins 64 m1 + 4 q + (0..3) of its frame
    const int p0n = tb.mel_p0[2 * lane] * 4, p0w = tb.mel_p0[2 * lane + 1] * 4;
    const int j1 = lane, j2 = 127 - lane;

    const int64_t n_units = (int64_t)n_windows * 64;
    for (int64_t unit = (int64_t)blockIdx.x * kFe2Waves + wave; unit < n_units; unit += (int64_t)gridDim.x * kFe2Waves) {
        const int n = (int)(unit >> 6), f0 = (int)(unit & 63) * 4;
        const float* x = arena + win_off[n];
        const int t = f0 + f;
        // Table reads are per-lane and the same in every unit: left to itself the compiler hoists all of them out of the unit loop
        // (~250 registers of loop invariants, i.e. spills).  An opaque lane offset per unit keeps each read next to its use.
        int lq = q * 8, zero = 0;
        asm volatile("" : "+v"(lq), "+v"(zero));
        const int wr_base = wr_b0 + zero, rd_base = rd_b0 + zero;
        const char* twp = (const char*)s_tw + lq;          // + ((R * 16 + n0) * 16) * 8
        const char* wkp = (const char*)s_wk + lq;          // + ((r * 16 + m1) * 16) * 8
        const char* winp = (const char*)s_win + lq;        // + (16 n1) * 8
        // ---- samples z[16 n1 + q] = (x[i], x[i + 1]), i = 32 n1 + 2 q - 256 + 256 t, times the window ----
        auto load_samples = [&](f2 (&sm)[16]) {
            if (f0 > 0) {                                  // wave-uniform: no frame of the unit touches the reflected edge
                const char* xb = (const char*)x + (uint32_t)(256 * (t - 1) + 2 * q) * 4u;
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) sm[n1] = *(const f2*)(xb + 128 * n1);
            } else {                                       // center=True, pad_mode='reflect': x[-k] = x[k], frame 0 (lanes f == 0) only
                // frame 0's first half reads the pair (x[-a0], x[-a0 - 1]) = the 8 bytes at &x[-a0 - 1], swapped (a0 = 32 n1 + 2 q - 256
                // is even and negative there); everything else is the plain pair at &x[a0]
                const bool refl = t == 0;
                const char* xb = (const char*)x + (2 * q - 256 + 256 * t) * 4;
                const char* xr = (const char*)x + (255 - 2 * q) * 4;
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) {
                    const f2 pv = *(const f2*)(refl ? xr - 128 * n1 : xb + 128 * n1);
                    sm[n1] = refl ? f2{pv.y, pv.x} : pv;
                }
#pragma unroll
                for (int n1 = 8; n1 < 16; ++n1) sm[n1] = *(const f2*)(xb + 128 * n1);
            }
        };
        f2 xw[16];
        load_samples(xw);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) xw[n1] = xw[n1] * *(const f2*)(winp + 128 * n1);

        // |X[64 m1 + 4 q + r]|^2 of this lane's frame, as register pairs: pa[m1] = (r = 0, r = 2), pb[m1] = (r = 1, r = 3) -- the two
        // pass groups each fill their own pairs; the power-spectrum buffer keeps that order inside every group of four bins
        f2 pa[12], pb[12];

        // Two passes at a time (r = 1 with 3, then 0 with 2), each through its own transpose image: two independent dependency
        // chains per wave, so that one chain's LDS round trip or butterfly latency is filled with the other's instructions.
        // Afterwards va / vb hold Z_RA / Z_RB [q + 16 m1] in register m1.
        auto pass_pair = [&](auto ra, auto rb, f2 (&va)[16], f2 (&vb)[16]) {
            constexpr int RA = decltype(ra)::value, RB = decltype(rb)::value;
            fe2_pretwiddle<RA>(xw, va);
            fe2_pretwiddle<RB>(xw, vb);
            fft16v(va);                                    // over n1 -> m0
            fft16v(vb);
#pragma unroll
            for (int m0 = 0; m0 < 16; ++m0) {
                *(f2*)(tr + (wr_base ^ ((m0 >> 1) << 4)) + m0 * 128) = va[m0];
                *(f2*)(tr + 8192 + (wr_base ^ ((m0 >> 1) << 4)) + m0 * 128) = vb[m0];
            }
            fe_wave_sync();
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const f32x4 a4 = *(const f32x4*)(tr + (rd_base ^ (p << 4)));
                const f32x4 b4 = *(const f32x4*)(tr + 8192 + (rd_base ^ (p << 4)));
                va[2 * p] = f2{a4[0], a4[1]}; va[2 * p + 1] = f2{a4[2], a4[3]};
                vb[2 * p] = f2{b4[0], b4[1]}; vb[2 * p + 1] = f2{b4[2], b4[3]};
            }
            fe_wave_sync();                                // (the next pair writes the images again)
#pragma unroll
            for (int n0 = 1; n0 < 16; ++n0) {              // W1024^(n0 (4 m0 + R)); n0 = 0: 1
                va[n0] = cmul2(va[n0], *(const f2*)(twp + (RA * 16 + n0) * 128));
                vb[n0] = cmul2(vb[n0], *(const f2*)(twp + (RB * 16 + n0) * 128));
            }
            fft16v(va);                                    // over n0 -> m1
            fft16v(vb);
        };
        // butterfly of the real-FFT untangle for bin k (own value zk = Z[k], zz = Z[1024 - k]) -> (|X[k]|^2, |X[1024 - k]|^2)
        auto bfly = [&](f2 zk, f2 zz, f2 w) -> f2 {
            const f2 zc = f2{zz.x, -zz.y};
            const f2 a = zk + zc, d = zk - zc;
            const f2 rw = rot_mi(cmul2(d, w));             // -i W2048^k d
            const f2 u = a + rw, uu = u * u;
            const f2 o = a - rw, oo = o * o;
            return f2{0.25f * (uu.x + uu.y), 0.25f * (oo.x + oo.y)};
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        {
            f2 z1[16], z3[16];
            pass_pair(I1{}, I3{}, z1, z3);
            // r = 1 with r = 3: bin 4 m + 1 pairs with 4 (255 - m) + 3: lane 15 - m0, register 15 - m1 of the other pass
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) {
                const f2 pp = bfly(z1[m1], mirror2(z3[15 - m1]), *(const f2*)(wkp + (1 * 16 + m1) * 128));
                if (m1 < 12) pb[m1].x = pp.x;
                if (m1 >= 4) pb[15 - m1].y = dpp_mirror(pp.y);
            }
        }
        {
            f2 z0[16], z2[16];
            pass_pair(I0{}, I2{}, z0, z2);
            // r = 0: bin 4 m pairs with 4 (256 - m): lane (16 - m0) mod 16, register 15 - m1 -- lane 0 pairs inside itself, register (16 - m1) mod 16
#pragma unroll
            for (int m1 = 0; m1 < 12; ++m1) {
                const f2 src = z0[15 - m1];
                f2 zz = f2{dpp_ror1(dpp_mirror(src.x)), dpp_ror1(dpp_mirror(src.y))};     // lane L <- lane (16 - L) mod 16
                if (q == 0) zz = z0[(16 - m1) & 15];
                pa[m1].x = bfly(z0[m1], zz, *(const f2*)(wkp + (0 * 16 + m1) * 128)).x;
            }
            // r = 2: bin 4 m + 2 pairs with 4 (255 - m) + 2: lane 15 - m0, register 15 - m1 of the same pass; registers 0..7 do the work
#pragma unroll
            for (int m1 = 0; m1 < 8; ++m1) {
                const f2 pp = bfly(z2[m1], mirror2(z2[15 - m1]), *(const f2*)(wkp + (2 * 16 + m1) * 128));
                pa[m1].y = pp.x;
                if (m1 >= 4) pa[15 - m1].y = dpp_mirror(pp.y);   // bins 512 < k < 768 belong to the mirrored lane's registers 8..11
            }
        }
#ifdef SS_DEVBUILD
        if (tb.dbg & 512) {                                // tools/fe_spectrum_check.py: 128 bins of the power spectrum instead of the mel rows
            const int sel = (tb.dbg >> 10) & 7;
#pragma unroll
            for (int m1 = 0; m1 < 12; ++m1)
                if ((m1 >> 1) == sel) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) feat[(size_t)n * 32768 + (size_t)(64 * m1 + 4 * q + r - 128 * sel) * 256 + t] = (r & 1) ? pb[m1][r >> 1] : pa[m1][r >> 1];
                }
            continue;
        }
#endif
        // ---- mel: the unit's four frames in two halves through the (now free) transpose buffer ----
        // Round 1 of a half: the narrow filters of its two frames (14 weights in registers); round 2: the wide ones (36 weights).  The
        // fences keep the compiler from issuing all of a half's reads (and both weight sets) up front: it has no registers for that.
        float o1[4], o2[4];
        int lrow = lane * kMelRow * 4;
        asm volatile("" : "+v"(lrow));
        const char* wrow = (const char*)s_mw + lrow;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            fe_wave_sync();
            if ((f >> 1) == half) {
#pragma unroll
                for (int m1 = 0; m1 < 12; ++m1) { *(f2*)(tr + pw_wr + 272 * m1) = pa[m1]; *(f2*)(tr + pw_wr + 272 * m1 + 8) = pb[m1]; }
                if (q == 15) {                             // the pad words must hold finite values: zero weights meet them
#pragma unroll
                    for (int m1 = 0; m1 < 12; ++m1) *(f32x4*)(tr + pw_wr + 272 * m1 + 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            fe_wave_sync();
            __builtin_amdgcn_sched_barrier(0);
            {
                f32x4 w4[(2 * kMelPairsLo + 3) / 4];
#pragma unroll
                for (int i = 0; i < (2 * kMelPairsLo + 3) / 4; ++i) w4[i] = *(const f32x4*)(wrow + 16 * i);
#pragma unroll
                for (int ff = 0; ff < 2; ++ff) {
                    const char* pb = tr + ff * (kPwWords * 4) + p0n;
                    float ms = 0.f;
#pragma unroll
                    for (int i = 0; i < kMelPairsLo; ++i) {
                        const f2 pv = *(const f2*)(pb + 8 * i);
                        ms = fmaf(w4[(2 * i) / 4][(2 * i) % 4], pv.x, ms);
                        ms = fmaf(w4[(2 * i + 1) / 4][(2 * i + 1) % 4], pv.y, ms);
                    }
                    o1[2 * half + ff] = ms;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // the wide filter's weights start at float 2 kMelPairsLo = 14 of the row: 16-byte reads from float 12 on
                constexpr int kFirst = (2 * kMelPairsLo) / 4 * 4, kCnt = (2 * kMelPairsLo + 2 * kMelPairsHi - kFirst + 3) / 4;
                f32x4 w4[kCnt];
#pragma unroll
                for (int i = 0; i < kCnt; ++i) w4[i] = *(const f32x4*)(wrow + 4 * kFirst + 16 * i);
#pragma unroll
                for (int ff = 0; ff < 2; ++ff) {
                    const char* pb = tr + ff * (kPwWords * 4) + p0w;
                    float ms = 0.f;
#pragma unroll
                    for (int i = 0; i < kMelPairsHi; ++i) {
                        const f2 pv = *(const f2*)(pb + 8 * i);
                        constexpr int o = 2 * kMelPairsLo - kFirst;
                        ms = fmaf(w4[(o + 2 * i) / 4][(o + 2 * i) % 4], pv.x, ms);
                        ms = fmaf(w4[(o + 2 * i + 1) / 4][(o + 2 * i + 1) % 4], pv.y, ms);
                    }
                    o2[2 * half + ff] = ms;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // as written in the reference: float32 log10(x + 1), then sqrt (no log1p, no fp64).  x + 1 >= 1, so neither the logarithm's
        // argument nor the root's can be subnormal: the hardware's log2 (1 ulp) times log10(2) and its square root (1 ulp) are within
        // 3e-7 relative of the exact value -- 2e-6 of the largest feature, against the 1e-5 the features are held to -- and cost 5
        // instructions per value where log10f + sqrtf with their subnormal paths cost 27 (7 % of the unit's vector instructions)
        auto sl = [](float m) { return __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(m + 1.0f) * 0.30102999566398120f); };
#pragma unroll
        for (int i = 0; i < 4; ++i) { o1[i] = sl(o1[i]); o2[i] = sl(o2[i]); }
        fe_wave_sync();                                    // the next unit's first transpose rewrites the buffer
        // four consecutive frames of each mel row leave as one 16-byte store per lane and row
        float* dst = feat + (size_t)n * 128 * 256 + f0;
        *(f32x4*)(dst + j1 * 256) = f32x4{o1[0], o1[1], o1[2], o1[3]};
        *(f32x4*)(dst + j2 * 256) = f32x4{o2[0], o2[1], o2[2], o2[3]};
    }
}

// =========================================================================================================
// Review-screen spectrogram: voice_activity.py:148-154 wav_to_spec = |librosa.stft(data, n_fft=512,
// win_length=512, hop_length=256)| -> [257][1 + n/256], centred frames, zero padding at both ends, periodic Hann.
// A 512-point real FFT is a 256-point complex FFT of the packed samples = the two radix-16 passes of the front-end
// without its four-way split: a 16-lane group does one frame, a wave four consecutive frames.
// =========================================================================================================
__global__ __launch_bounds__(256) void stft512_mag_kernel(const float* __restrict__ x, int64_t n, int64_t n_frames, float* __restrict__ out) {
    __shared__ float s_win[512];
    __shared__ float2 s_tw[256];                      // W256^(n0 m0), [m0][n0]
    __shared__ float2 s_wk[257];                      // exp(-2 pi i k / 512)
    __shared__ float2 s_tr[4][64 * kTrRow];           // per wave: transpose rows, then Z[frame r][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 512; i += 256) s_win[i] = 0.5f - 0.5f * cospif((float)i / 256.0f);        // 0.5 - 0.5 cos(2 pi i / 512)
    for (int i = tid; i < 256; i += 256) { float sn, cs; sincospif(-(float)((i & 15) * (i >> 4)) / 128.0f, &sn, &cs); s_tw[i] = make_float2(cs, sn); }
    for (int i = tid; i < 257; i += 256) { float sn, cs; sincospif(-(float)i / 256.0f, &sn, &cs); s_wk[i] = make_float2(cs, sn); }
    __syncthreads();
    const int r = lane >> 4, q = lane & 15;
    float2* tr = s_tr[wave];
    for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * 4; t0 < n_frames; t0 += (int64_t)gridDim.x * 16) {
        const int64_t t = t0 + r;                     // this 16-lane group's frame (may be past the end: computed, not stored)
        float2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int i = 32 * n1 + 2 * q;            // sample pair (i, i + 1) of the frame
            const int64_t a0 = 256 * t - 256 + i;
            const float x0 = (a0 >= 0 && a0 < n) ? x[a0] : 0.f, x1 = (a0 + 1 >= 0 && a0 + 1 < n) ? x[a0 + 1] : 0.f;
            v[n1] = make_float2(x0 * s_win[i], x1 * s_win[i + 1]);
        }
        fft16(v);
#pragma unroll
        for (int m0 = 0; m0 < 16; ++m0) v[m0] = cmul(v[m0], s_tw[16 * m0 + q]);
#pragma unroll
        for (int m0 = 0; m0 < 16; ++m0) tr[(r * 16 + m0) * kTrRow + q] = v[m0];
        fe_wave_sync();
        {
            const f32x4* row = (const f32x4*)(tr + (r * 16 + q) * kTrRow);
#pragma unroll
            for (int p = 0; p < 8; ++p) { const f32x4 w4 = row[p]; v[2 * p] = make_float2(w4[0], w4[1]); v[2 * p + 1] = make_float2(w4[2], w4[3]); }
        }
        fft16(v);                                     // v[m1] = Z[q + 16 m1] of frame r
        fe_wave_sync();
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) tr[r * 256 + 16 * m1 + q] = v[m1];
        fe_wave_sync();
        // untangle: X[k] = (Z[k] + conj Z[256-k]) / 2 - i W512^k (Z[k] - conj Z[256-k]) / 2, k = 0..256; lane -> bins lane + 64 i
#pragma unroll
        for (int fr = 0; fr < 4; ++fr) {
            if (t0 + fr >= n_frames) break;
            const float2* Z = tr + fr * 256;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int k = 64 * i + lane;
                if (k > 256) break;
                const float2 zk = Z[k & 255], zz = Z[(256 - k) & 255];
                const float2 zc = make_float2(zz.x, -zz.y);
                const float2 a = cadd(zk, zc), d = csub(zk, zc);
                const float2 wd = cmul(s_wk[k], d);
                const float xr = 0.5f * (a.x + wd.y), xi = 0.5f * (a.y - wd.x);
                out[(int64_t)k * n_frames + t0 + fr] = sqrtf(xr * xr + xi * xi);
            }
        }
        fe_wave_sync();
    }
}

hipError_t launch_stft512_mag(const float* x, int64_t n, int64_t n_frames, float* out, int num_cus, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    const int64_t groups = (n_frames + 15) / 16;
    const unsigned grid = (unsigned)std::min<int64_t>(groups, (int64_t)(num_cus > 0 ? num_cus : 256) * 4);
    hipLaunchKernelGGL(stft512_mag_kernel, dim3(grid), dim3(256), 0, s, x, n, n_frames, out);
    return hipGetLastError();
}

hipError_t launch_frontend(const float* arena, const int64_t* win_off, int n, const FrontendTables& t, float* feat, int num_cus,
                           hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int grid = num_cus > 0 ? num_cus : 256;
#ifdef SS_DEVBUILD
    if (t.dbg & 256) {                                    // (engine.hip: SOFTSPOKEN_FEDBG) the first structure, for A/B runs
        if ((int64_t)grid * kFeWaves > (int64_t)n * 64) grid = (int)(((int64_t)n * 64 + kFeWaves - 1) / kFeWaves);
        hipLaunchKernelGGL(frontend_v1_kernel, dim3(grid), dim3(64 * kFeWaves), 0, s, arena, win_off, n, t, feat);
        return hipGetLastError();
    }
#endif
    if ((int64_t)grid * kFe2Waves > (int64_t)n * 64) grid = (int)(((int64_t)n * 64 + kFe2Waves - 1) / kFe2Waves);
    hipLaunchKernelGGL(frontend_kernel, dim3(grid), dim3(64 * kFe2Waves), 0, s, arena, win_off, n, t, feat);
    return hipGetLastError();
}

// =========================================================================================================
// PCM -> float32 mono.  libsndfile's float conversion (x / 2^(bits-1); unsigned 8-bit is offset by 128),
// then librosa.to_mono == mean over channels in float32 (voice_activity.py:37-38, 61-62).
// =========================================================================================================
__device__ __forceinline__ float decode_sample(const unsigned char* p, int format, int64_t idx) {
    switch (format) {
        case 1: return ((float)p[idx] - 128.0f) / 128.0f;
        case 2: return (float)((const short*)p)[idx] / 32768.0f;
        case 3: {
            const unsigned char* b = p + idx * 3;
            int v = (int)b[0] | ((int)b[1] << 8) | ((int)b[2] << 16);
            if (v & 0x800000) v -= 0x1000000;
            return (float)v / 8388608.0f;
        }
        case 4: return (float)((double)((const int*)p)[idx] / 2147483648.0);
        case 5: return ((const float*)p)[idx];
        case 6: return (float)((const double*)p)[idx];
        // AIFF / AIFF-C: big-endian samples, 8-bit ones signed; the same float conversion
        case 7: return (float)(signed char)p[idx] / 128.0f;
        case 8: { const unsigned char* b = p + idx * 2; return (float)(short)((unsigned)b[0] << 8 | b[1]) / 32768.0f; }
        case 9: {
            const unsigned char* b = p + idx * 3;
            int v = (int)b[2] | ((int)b[1] << 8) | ((int)b[0] << 16);
            if (v & 0x800000) v -= 0x1000000;
            return (float)v / 8388608.0f;
        }
        case 10: return (float)((double)(int)__builtin_bswap32(((const uint32_t*)p)[idx]) / 2147483648.0);
        case 11: return __builtin_bit_cast(float, __builtin_bswap32(((const uint32_t*)p)[idx]));
        default: return (float)__builtin_bit_cast(double, __builtin_bswap64(((const uint64_t*)p)[idx]));
    }
}

// =========================================================================================================
// Polyphase Kaiser-windowed-sinc resampler to 22 050 Hz (stands where librosa.resample -> soxr_hq stands,
// voice_activity.py:65-67).  out[m] = sum_j taps[(m M) mod L][j] * in[(m M) div L + j - half + 1].
// float32 multiply then add in tap order (no FMA contraction) so the CPU oracle can match it bit for bit.
// =========================================================================================================
// One launch covers every file of a batch (files share format / rate / channels); a single file is a batch of one.
__global__ __launch_bounds__(256) void decode_mono_batch_kernel(const unsigned char* __restrict__ pcm, int format, int channels,
                                                                const BatchFile* __restrict__ files, float* __restrict__ mono) {
    const BatchFile f = files[blockIdx.y];
    const int bps = format == 1 ? 1 : format == 2 ? 2 : format == 3 ? 3 : format == 6 ? 8 : 4;
    const unsigned char* base = pcm + f.pcm_off;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < f.frames; i += (int64_t)gridDim.x * 256) {
        float acc = decode_sample(base, format, i * channels);
        for (int c = 1; c < channels; ++c) acc = __fadd_rn(acc, decode_sample(base, format, i * channels + c));
        mono[f.mono_off + i] = channels > 1 ? __fdiv_rn(acc, (float)channels) : acc;
    }
    (void)bps;
}

__global__ __launch_bounds__(256) void resample_batch_kernel(const float* __restrict__ mono, const BatchFile* __restrict__ files, int L,
                                                             int M, int half, const float* __restrict__ taps, float* __restrict__ arena) {
#pragma clang fp contract(off)
    const BatchFile f = files[blockIdx.y];
    const float* in = mono + f.mono_off;
    float* out = arena + f.out_off;
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < f.n_out; m += (int64_t)gridDim.x * 256) {
        const int64_t pos = m * M;
        const int64_t base = pos / L;
        const int phase = (int)(pos - base * L);
        const float* tp = taps + (size_t)phase * (2 * half);
        float acc = 0.f;
        for (int j = 0; j < 2 * half; ++j) {
            const int64_t idx = base + j - half + 1;
            const float sv = (idx >= 0 && idx < f.frames) ? in[idx] : 0.f;
            { const float pr = tp[j] * sv; acc = acc + pr; }   // two roundings, as the oracle
        }
        out[m] = acc;
    }
}

// Same arithmetic with the whole polyphase table staged in LDS (rows padded to an odd pitch: lanes hold different
// phases of the same tap index, which would otherwise all hit one bank).  Used when L * (2 half + 1) floats fit.
// A block owns kResOut consecutive output samples of one file; 32-bit index math (positions relative to the block).
static constexpr int kResOut = 4096;
static constexpr int kResThreads = 1024;     // 4 waves per SIMD; the 115 KB table allows one block per CU
// Persistent: a block stages the table once and then walks (file, 4096-output chunk) items -- staging it per chunk was ~45 %
// of the kernel (4352 blocks x 115 KB for the C2 job).
__global__ __launch_bounds__(kResThreads) void resample_batch_lds_kernel(const float* __restrict__ mono, const BatchFile* __restrict__ files,
                                                                         int n_files, int chunks_per_file, int L, int M, int half, int span,
                                                                         const float* __restrict__ taps, float* __restrict__ arena) {
#pragma clang fp contract(off)
    extern __shared__ float s_taps[];
    const int nt = 2 * half, pitch = nt | 1;
    float* s_x = s_taps + L * pitch;                      // span > 0: the input samples an item touches, zero outside the file
    for (int p = threadIdx.x / 64; p < L; p += kResThreads / 64)              // one phase (row) per wave per pass: no division per element
        for (int j = threadIdx.x & 63; j < nt; j += 64) s_taps[p * pitch + j] = taps[p * nt + j];
    __syncthreads();
    const int n_items = n_files * chunks_per_file;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int fi = item / chunks_per_file, chunk = item - fi * chunks_per_file;
        const BatchFile f = files[fi];
        const int64_t m0 = (int64_t)chunk * kResOut;
        if (m0 >= f.n_out) continue;                      // block-uniform
        const float* in = mono + f.mono_off;
        float* out = arena + f.out_off;
        const int64_t pos0 = m0 * M;
        const int64_t base0 = pos0 / L;
        const int ph0 = (int)(pos0 - base0 * L);              // position of output m0 is base0 + ph0 / L
        const int n_here = (int)((f.n_out - m0) < kResOut ? (f.n_out - m0) : kResOut);
        if (span > 0) {
            // stage x[i_lo .. i_lo + span): every tap of every output of the item reads LDS, and the file's ends need no test
            const int64_t i_lo = base0 - half + 1;
            for (int i = threadIdx.x; i < span; i += kResThreads) {
                const int64_t idx = i_lo + i;
                s_x[i] = (idx >= 0 && idx < f.frames) ? in[idx] : 0.f;
            }
            __syncthreads();
            for (int k = threadIdx.x; k < n_here; k += kResThreads) {
                const int rel = ph0 + k * M;                  // < L + 4096 * M, fits 32 bits for every audio rate
                const int db = rel / L;
                const float* tp = s_taps + (rel - db * L) * pitch;
                const float* xp = s_x + db;
                float acc = 0.f;
#pragma unroll 4
                for (int j = 0; j < nt; ++j) { const float pr = tp[j] * xp[j]; acc = acc + pr; }   // two roundings, as the oracle
                out[m0 + k] = acc;
            }
            __syncthreads();                              // s_x is rewritten by the next item
            continue;
        }
        for (int k = threadIdx.x; k < n_here; k += kResThreads) {
            const int rel = ph0 + k * M;
            const int db = rel / L;
            const int phase = rel - db * L;
            const int64_t base = base0 + db;
            const float* tp = s_taps + phase * pitch;
            float acc = 0.f;
            const int64_t i0 = base - half + 1;
            if (i0 >= 0 && i0 + nt <= f.frames) {
                for (int j = 0; j < nt; ++j) { const float pr = tp[j] * in[i0 + j]; acc = acc + pr; }
            } else {
                for (int j = 0; j < nt; ++j) {
                    const int64_t idx = i0 + j;
                    const float sv = (idx >= 0 && idx < f.frames) ? in[idx] : 0.f;
                    const float pr = tp[j] * sv; acc = acc + pr;
                }
            }
            out[m0 + k] = acc;
        }
    }
}

// Third form (round 3): decode + mixdown fused into the staging, four outputs per thread.
//   * Outputs m and m + L have the same phase (tap row) and inputs M apart, so a thread owns ONE phase p and four consecutive
//     "rows" r (outputs m0 + r L + p): a tap is read once for four multiply-adds.
//   * The input tile lies in LDS as X[u][v], u = i' mod M, v = i' div M (i' = input index from the tile's first sample): the four
//     rows' samples for tap j -- i' = r M + b(p) + j -- are FOUR CONSECUTIVE WORDS X[(b + j) mod M][(b + j) div M + r .. + 3], one
//     16-byte read.  b + j wraps past M at most once (the launch requires 2 half <= M); behind the wrap the row index is one higher,
//     which would make the read misaligned, so the rows u < 2 half exist a second time shifted by one (X1[u][v] = X[u][v + 1]).
//   * The tile is filled straight from the PCM: decode (x / 2^(bits-1)), channel mean in float32 -- decode_mono_batch_kernel's
//     arithmetic --, zero outside the file: no mono tensor is written or read (C5: 2 x 4 bytes per 48 kHz frame less HBM traffic
//     and one launch less).
// Arithmetic per output is unchanged -- float32 multiply then add, tap order, two roundings -- so both oracles still match bit for bit.
// LDS reads per multiply-add: 1/4 tap word + 1 sample word (second form: 1 + 1, and four-byte reads at half the LDS rate).
static constexpr int kRes3Threads = 1024;
static constexpr int kRes3Stage = 9;                      // most tile samples a thread stages per item (the launch checks the span)
// FAST: 16-bit PCM, one or two channels (a frame is 2 or 4 bytes): the NEXT item's frames are requested as raw words before the
// current item's multiply loop and decoded behind it, so their latency hides behind the arithmetic.  Other formats decode at the
// request (all of a thread's requests go out together, but the tile waits for them).
// Which lane takes which phase (round 4).  A thread reads X[b(p) + j][4 g ..] as one 16-byte word: the 16 lanes of an LDS service group
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} of each half-wave: MI355X guide, LDS) are conflict-free exactly when their rows start on 16
// different banks-of-four, and with an odd number of 16-byte slots per row that is: 16 different values of b(p) mod 16.  With phases in lane
// order the lanes of a group span ~35 rows whose residues collide (2.3-way conflicts on every read, 49 % of the LDS cycles:
// profiles/r03_pmc.md).  So the phases are dealt to the service groups by residue class: phase p, the k-th of its class b(p) mod 16,
// goes to position (b(p) mod 16) of service group k.  A 4-row group then takes `lpg` = 32 ceil(max class size / 2) lanes instead of L
// (48 k -> 22.05 k: 160 for 147 phases), a few of them idle; the tap rows lie in lane order, so their reads stay conflict-free too.
// lpg = L: phases in lane order (the geometry falls back to it when the dealt form does not fit).
__device__ __forceinline__ int res3_slot(int k, int c) {  // lane (within a group's lanes) of position c in service group k
    const int a = k & 1 ? (c < 8 ? 4 + c : (c < 12 ? 8 + c : 16 + c)) : (c < 4 ? c : (c < 8 ? 8 + c : 12 + c));
    return 32 * (k >> 1) + a;
}
template <bool FAST>
__global__ __launch_bounds__(kRes3Threads) void resample_fused_kernel(const unsigned char* __restrict__ pcm, int format, int channels,
                                                                      const BatchFile* __restrict__ files, int n_files, int items_per_file,
                                                                      int L, int M, int half, int groups, int pitch_v, int lpg,
                                                                      const float* __restrict__ taps, float* __restrict__ arena) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float s_res3[];
    const int nt = 2 * half, pitch_t = nt | 1;
    const int R = 4 * groups;                             // rows per item: outputs [m0, m0 + L R), m0 a multiple of L
    float* s_x = s_res3;                                  // [M][pitch_v]
    float* s_x1 = s_x + (size_t)M * pitch_v;              // [nt][pitch_v]: rows u < nt shifted by one
    float* s_t = s_x1 + (size_t)nt * pitch_v;             // [lpg][pitch_t]: row s = the taps of the phase in lane s (taps row (p M) mod L)
    int* s_perm = (int*)(s_t + (size_t)lpg * pitch_t);    // [lpg]: phase of lane s of a group, or -1
    const int tid = threadIdx.x;
    int* s_cls = s_perm + lpg;                            // [L]: residue class of phase ph
    for (int s = tid; s < lpg; s += kRes3Threads) s_perm[s] = lpg == L ? s : -1;
    if (tid < L) s_cls[tid] = (int)(((int64_t)tid * M) / L) & 15;
    __syncthreads();
    if (lpg != L && tid < L) {                            // phase `tid`, the k-th of its class in ascending order, takes service group k
        const int c = s_cls[tid];
        int k = 0;
        for (int ph = 0; ph < tid; ++ph) k += s_cls[ph] == c;
        s_perm[res3_slot(k, c)] = tid;
    }
    __syncthreads();
    for (int s = tid / 64; s < lpg; s += kRes3Threads / 64) {
        const int ph = s_perm[s];
        if (ph < 0) continue;
        const int q = (int)(((int64_t)ph * M) % L);
        for (int j = tid & 63; j < nt; j += 64) s_t[s * pitch_t + j] = taps[(size_t)q * nt + j];
    }
    const int g = tid / lpg, slot = tid - g * lpg;
    const int p_of = g < groups ? s_perm[slot] : -1;
    const bool worker = p_of >= 0;
    const int p = worker ? p_of : 0;
    const int b = (int)(((int64_t)p * M) / L);            // input offset of phase p inside a row
    const int jx = M - b < nt ? M - b : nt;               // taps j >= jx lie behind the wrap of b + j past M
    const float* tp = s_t + (worker ? slot : 0) * pitch_t;
    const float* xa = s_x + (size_t)b * pitch_v + 4 * g;                       // + j pitch_v:  X[b + j][4 g ..]
    const float* xb = s_x1 + ((size_t)b * pitch_v + 4 * g) - (size_t)M * pitch_v;   // + j pitch_v:  X1[b + j - M][4 g ..]  (j >= jx)
    const int span = R * M + M + nt;                      // input samples an item touches (rounded up to whole rows)
    const int n_items = n_files * items_per_file;
    // this thread's places in the tile: fixed for the kernel's lifetime
    int off_x[kRes3Stage], off_x1[kRes3Stage];
#pragma unroll
    for (int k = 0; k < kRes3Stage; ++k) {
        const int ip = tid + k * kRes3Threads;
        const int vv = ip / M, u = ip - vv * M;
        off_x[k] = (ip < span && vv < pitch_v) ? u * pitch_v + vv : -1;
        off_x1[k] = (ip < span && u < nt && vv >= 1 && vv - 1 < pitch_v) ? u * pitch_v + vv - 1 : -1;
    }
    uint32_t raw[kRes3Stage];
    float val[kRes3Stage];
    auto valid_item = [&](int item, BatchFile& f, int& it) -> bool {
        if (item >= n_items) return false;
        const int fi = item / items_per_file;
        it = item - fi * items_per_file;
        f = files[fi];
        return (int64_t)it * L * R < f.n_out;
    };
    auto fetch = [&](const BatchFile& f, int it) {
        const int64_t i_lo = (int64_t)it * R * M - half + 1;                  // input index of the tile's first sample
        const unsigned char* base = pcm + f.pcm_off;
#pragma unroll
        for (int k = 0; k < kRes3Stage; ++k) {
            const int64_t idx = i_lo + tid + k * kRes3Threads;
            const bool in = off_x[k] >= 0 && idx >= 0 && idx < f.frames;
            if constexpr (FAST) {
                raw[k] = 0u;
                if (in) raw[k] = channels == 2 ? ((const uint32_t*)base)[idx] : (uint32_t)((const unsigned short*)base)[idx];
            } else {
                float v = 0.f;
                if (in) {
                    float acc = decode_sample(base, format, idx * channels);
                    for (int c = 1; c < channels; ++c) acc = __fadd_rn(acc, decode_sample(base, format, idx * channels + c));
                    v = channels > 1 ? __fdiv_rn(acc, (float)channels) : acc;
                }
                val[k] = v;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < kRes3Stage; ++k) {
            float v;
            if constexpr (FAST) {                         // decode_mono_batch_kernel's arithmetic: x / 32768, float32 channel mean
                const float s0 = (float)(short)(raw[k] & 0xffffu) / 32768.0f;
                if (channels == 2) { const float s1 = (float)(short)(raw[k] >> 16) / 32768.0f; v = __fdiv_rn(__fadd_rn(s0, s1), 2.0f); }
                else v = s0;
            } else v = val[k];
            if (off_x[k] >= 0) s_x[off_x[k]] = v;
            if (off_x1[k] >= 0) s_x1[off_x1[k]] = v;
        }
    };
    // walk this block's items; items past a file's end are skipped (block-uniform)
    int item = blockIdx.x;
    BatchFile f{}, fn{};
    int it = 0, itn = 0;
    while (item < n_items && !valid_item(item, f, it)) item += gridDim.x;
    if (item >= n_items) return;
    fetch(f, it);
    for (;;) {
        __syncthreads();                                  // the previous item's reads (and, first time, the table) are done / visible
        store_tile();
        __syncthreads();
        int next = item + gridDim.x;
        while (next < n_items && !valid_item(next, fn, itn)) next += gridDim.x;
        const bool more = next < n_items;
        if (more) fetch(fn, itn);                         // in flight during the multiply loop
        if (worker) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            auto tap = [&](const float4& x4, float t) {                        // two roundings per term, as the oracle
                { const float pr = t * x4.x; a0 = a0 + pr; }
                { const float pr = t * x4.y; a1 = a1 + pr; }
                { const float pr = t * x4.z; a2 = a2 + pr; }
                { const float pr = t * x4.w; a3 = a3 + pr; }
            };
            // (requesting a group of four taps' reads ahead of the arithmetic, ping-pong, measured slower: 1.73 against 1.55 ms on the C5
            // job -- four waves per SIMD already cover the LDS latency; what the loop waits for is the LDS array itself, 16-byte reads with
            // 2-way bank conflicts between the lanes' rows)
#pragma unroll 4
            for (int j = 0; j < nt; ++j) tap(*(const float4*)((j < jx ? xa : xb) + (size_t)j * pitch_v), tp[j]);
            float* out = arena + f.out_off;
            const int64_t m = (int64_t)it * L * R + (int64_t)(4 * g) * L + p;
            if (m < f.n_out) out[m] = a0;
            if (m + L < f.n_out) out[m + L] = a1;
            if (m + 2 * (int64_t)L < f.n_out) out[m + 2 * (int64_t)L] = a2;
            if (m + 3 * (int64_t)L < f.n_out) out[m + 3 * (int64_t)L] = a3;
        }
        if (!more) break;
        item = next; f = fn; it = itn;
    }
}

// geometry of the fused form for a rate pair, or groups = 0 when it does not apply (table or tile too large, 2 half > M, L > 1024)
struct Res3Geom { int groups, pitch_v, lpg; size_t lds; };
static Res3Geom res3_geometry(int L, int M, int half) {
    Res3Geom gm{0, 0, 0, 0};
    const int nt = 2 * half;
    if (L > kRes3Threads || nt > M) return gm;
    const int groups = kRes3Threads / L;
    const int R = 4 * groups;
    int pv = R + 2;                                       // rows v the tile needs: r + (b + j) div M <= R - 1 + 1, + the shifted copy's read
    pv = (pv + 3) & ~3;                                   // 16-byte rows
    if (((pv / 4) & 1) == 0) pv += 4;                     // an odd number of 16-byte slots per row: rows u, u + 1, ... start 16 banks-of-four apart mod 16
    if ((size_t)R * M + M + nt > (size_t)kRes3Stage * kRes3Threads) return gm;
    // phases dealt to the LDS service groups by b(p) mod 16 (see the kernel): lanes per 4-row group = 32 ceil(largest class / 2)
    int cnt[16] = {0};
    for (int ph = 0; ph < L; ++ph) ++cnt[(int)(((int64_t)ph * M) / L) % 16];
    int nsg = 0;
    for (int c = 0; c < 16; ++c) nsg = std::max(nsg, cnt[c]);
    int lpg = 32 * ((nsg + 1) / 2);
    auto lds_of = [&](int lanes) { return ((size_t)M * pv + (size_t)nt * pv + (size_t)lanes * (nt | 1) + (size_t)lanes + (size_t)L) * sizeof(float); };
    // (interpolating pairs, M <= L: neighbouring lanes read the same or the next row -- broadcasts and adjacent banks -- and measured the
    // same or a little slower dealt: 16 k -> 22.05 k 118 -> 128 us per 1005 windows; decimating pairs spread a service group over ~35 rows:
    // 48 k -> 22.05 k 1.53 -> 1.41 ms on the C5 job)
    if (M <= L || lpg * groups > kRes3Threads || lds_of(lpg) > 160 * 1024) lpg = L;     // phases in lane order
    const size_t lds = lds_of(lpg);
    if (lds > 160 * 1024) return gm;
    gm.groups = groups; gm.pitch_v = pv; gm.lpg = lpg; gm.lds = lds;
    return gm;
}

bool resample_fused_applies(int L, int M, int half) { return res3_geometry(L, M, half).groups > 0; }

hipError_t launch_resample_fused(const void* pcm, int format, int channels, const BatchFile* d_files, int n_files, int64_t max_out, int L, int M,
                                 int half, const float* taps, float* arena, int num_cus, hipStream_t s) {
    if (n_files <= 0 || max_out <= 0) return hipSuccess;
    const Res3Geom gm = res3_geometry(L, M, half);
    if (!gm.groups) return hipErrorInvalidValue;
    static std::atomic<uint64_t> attr_done{0};              // (per device: kernels.h allow_full_lds)
    if (hipError_t e = allow_full_lds((const void*)resample_fused_kernel<true>, attr_done)) return e;
    static std::atomic<uint64_t> attr_done2{0};
    if (hipError_t e = allow_full_lds((const void*)resample_fused_kernel<false>, attr_done2)) return e;
    const int64_t per_item = (int64_t)L * 4 * gm.groups;
    const int64_t ipf = (max_out + per_item - 1) / per_item;
    if (ipf * n_files >= (int64_t)1 << 30) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)std::min<int64_t>(ipf * n_files, num_cus > 0 ? num_cus : 256);
    if (format == 2 && channels <= 2)
        hipLaunchKernelGGL(resample_fused_kernel<true>, dim3(grid), dim3(kRes3Threads), gm.lds, s, (const unsigned char*)pcm, format, channels, d_files,
                           n_files, (int)ipf, L, M, half, gm.groups, gm.pitch_v, gm.lpg, taps, arena);
    else
        hipLaunchKernelGGL(resample_fused_kernel<false>, dim3(grid), dim3(kRes3Threads), gm.lds, s, (const unsigned char*)pcm, format, channels, d_files,
                           n_files, (int)ipf, L, M, half, gm.groups, gm.pitch_v, gm.lpg, taps, arena);
    return hipGetLastError();
}

hipError_t launch_decode_mono_batch(const void* pcm, int format, int channels, const BatchFile* d_files, int n_files, int64_t max_frames,
                                    float* mono, hipStream_t s) {
    if (n_files <= 0 || max_frames <= 0) return hipSuccess;
    const unsigned gx = (unsigned)std::min<int64_t>((max_frames + 255) / 256, 4096);
    hipLaunchKernelGGL(decode_mono_batch_kernel, dim3(gx, (unsigned)n_files), dim3(256), 0, s, (const unsigned char*)pcm, format,
                       channels, d_files, mono);
    return hipGetLastError();
}

// =========================================================================================================
// Silencer (silencer_ui.py:974-998): every sample decoded to float32 as the loader does, frames inside a
// reviewed interval zeroed, the rest written as 16-bit PCM -- lrintf(x * 32767) without clipping, which is
// what libsndfile does for a float buffer written to a PCM_16 WAV (soundfile's default subtype).
// `ranges` holds n_ranges disjoint, ascending [begin, end) frame pairs.
// =========================================================================================================
__global__ __launch_bounds__(256) void silence_encode_kernel(const unsigned char* __restrict__ pcm, int format, int channels,
                                                             int64_t frames, const int64_t* __restrict__ ranges, int n_ranges,
                                                             short* __restrict__ out) {
#pragma clang fp contract(off)
    const int64_t total = frames * channels;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
        short v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t idx = i + k;
            if (idx >= total) { v[k] = 0; continue; }
            const int64_t fr = idx / channels;
            int lo = 0, hi = n_ranges;                 // first range whose end is beyond fr
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (ranges[2 * mid + 1] <= fr) lo = mid + 1; else hi = mid;
            }
            const bool cut = lo < n_ranges && ranges[2 * lo] <= fr;
            const float x = decode_sample(pcm, format, idx) * 32767.0f;
            v[k] = cut ? (short)0 : (short)__float2int_rn(x);
        }
        if (i + 3 < total) {
            *(uint2*)(out + i) = make_uint2((unsigned)(unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16),
                                            (unsigned)(unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16));
        } else {
            for (int k = 0; k < 4 && i + k < total; ++k) out[i + k] = v[k];
        }
    }
}

hipError_t launch_silence_encode(const void* pcm, int format, int channels, int64_t frames, const int64_t* d_ranges, int n_ranges,
                                 short* out, hipStream_t s) {
    const int64_t total = frames * channels;
    if (total <= 0) return hipSuccess;
    const unsigned gx = (unsigned)std::min<int64_t>((total + 1023) / 1024, 8192);
    hipLaunchKernelGGL(silence_encode_kernel, dim3(gx), dim3(256), 0, s, (const unsigned char*)pcm, format, channels, frames, d_ranges,
                       n_ranges, out);
    return hipGetLastError();
}

hipError_t launch_resample_batch(const float* mono, const BatchFile* d_files, int n_files, int64_t max_out, int L, int M, int half,
                                 const float* taps, float* arena, int num_cus, hipStream_t s) {
    if (n_files <= 0 || max_out <= 0) return hipSuccess;
    const size_t lds = (size_t)L * ((2 * half) | 1) * sizeof(float);
    if (lds <= 150 * 1024 && (int64_t)kResOut * M < (int64_t)1 << 30) {
        static std::atomic<uint64_t> attr_done{0};              // (per device: kernels.h allow_full_lds)
        if (hipError_t e = allow_full_lds((const void*)resample_batch_lds_kernel, attr_done)) return e;
        const int64_t cpf = (max_out + kResOut - 1) / kResOut;
        if (cpf * n_files < (int64_t)1 << 30) {
            const unsigned grid = (unsigned)std::min<int64_t>(cpf * n_files, num_cus > 0 ? num_cus : 256);
            // input samples of one item: floor((L - 1 + (kResOut - 1) M) / L) + 2 half; staged in LDS when they fit next to the table
            int span = (int)(((int64_t)L - 1 + (int64_t)(kResOut - 1) * M) / L) + 2 * half + 1;
            size_t lds_all = lds + (size_t)span * sizeof(float);
            if (lds_all > 160 * 1024) { span = 0; lds_all = lds; }
            hipLaunchKernelGGL(resample_batch_lds_kernel, dim3(grid), dim3(kResThreads), lds_all, s, mono, d_files, n_files, (int)cpf, L, M, half, span,
                               taps, arena);
            return hipGetLastError();
        }
    }
    const unsigned gx = (unsigned)std::min<int64_t>((max_out + 255) / 256, 4096);
    hipLaunchKernelGGL(resample_batch_kernel, dim3(gx, (unsigned)n_files), dim3(256), 0, s, mono, d_files, L, M, half, taps, arena);
    return hipGetLastError();
}

// =========================================================================================================
// Overlap averaging (NNDetector.py:168-186): window i adds its 256 logits at bin start[i] = round(51.2 i);
// float64 sum in window order, divided by the count; bins never covered keep count 0 and are dropped later.
// =========================================================================================================
__global__ __launch_bounds__(256) void average_kernel(const float* __restrict__ logits, const AvgFile* __restrict__ files,
                                                      const int32_t* __restrict__ starts, double* __restrict__ avg,
                                                      int32_t* __restrict__ count) {
    const AvgFile fi = files[blockIdx.y];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= fi.n_bins) return;
    const int32_t* st = starts + fi.start_off;
    int lo = (int)((double)(j - 255) / 51.2) - 1;
    if (lo < 0) lo = 0;
    int hi = (int)((double)j / 51.2) + 1;
    if (hi > fi.W - 1) hi = fi.W - 1;
    double s = 0.0;
    int c = 0;
    for (int i = lo; i <= hi; ++i) {
        const int d = j - st[i];
        if (d >= 0 && d < 256) { s += (double)logits[(fi.logit_off + i) * 256 + d]; ++c; }
    }
    avg[fi.bin_off + j] = c ? s / (double)c : 0.0;
    count[fi.bin_off + j] = c;
}

// The host's region finding (NNDetector.py:103-143) only needs two facts per bin: covered by a window, and average > threshold (the
// same double comparison the host would make).  One 64-bit word of each per 64 bins goes back instead of 12 bytes per bin.
__global__ __launch_bounds__(256) void bin_masks_kernel(const double* __restrict__ avg, const int32_t* __restrict__ count, int64_t total_bins,
                                                        double threshold, unsigned long long* __restrict__ above,
                                                        unsigned long long* __restrict__ covered) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = j < total_bins;
    const bool cov = in && count[j] >= 1;
    const bool abv = cov && avg[j] > threshold;
    const unsigned long long mc = __ballot(cov), ma = __ballot(abv);
    if ((threadIdx.x & 63) == 0) { above[j >> 6] = ma; covered[j >> 6] = mc; }
}

hipError_t launch_bin_masks(const double* avg, const int32_t* count, int64_t total_bins, double threshold, unsigned long long* above,
                            unsigned long long* covered, hipStream_t s) {
    if (total_bins <= 0) return hipSuccess;
    hipLaunchKernelGGL(bin_masks_kernel, dim3((unsigned)((total_bins + 255) / 256)), dim3(256), 0, s, avg, count, total_bins, threshold, above, covered);
    return hipGetLastError();
}

hipError_t launch_average(const float* logits, const AvgFile* files, int n_files, const int32_t* starts, double* avg, int32_t* count,
                          int max_bins, hipStream_t s) {
    if (n_files <= 0 || max_bins <= 0) return hipSuccess;
    hipLaunchKernelGGL(average_kernel, dim3((unsigned)((max_bins + 255) / 256), (unsigned)n_files), dim3(256), 0, s, logits, files,
                       starts, avg, count);
    return hipGetLastError();
}

}  // namespace ss
