// Front-end kernels for gfx950: PCM decode + mixdown, polyphase resample, fused STFT -> power -> mel ->
// sqrt(log10(x+1)), and the overlap-averaging of per-window logits.
//
// Reference: root/code/backend/voice_activity.py:32-69 (load_audio), root/code/backend/pytorch_neural_nets.py:92-99,
// 144-153 (torchaudio MelSpectrogram(n_fft=2048, win_length=512, hop=256, n_mels=128, f_max=8000) then
// sqrt(log10(.+1)) and [:, :, :256]), root/code/frontend/NNDetector.py:153-190 (averaging).
#include "kernels.h"
#include <algorithm>

namespace ss {

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

static constexpr int kFramesPerBlock = 32;
static constexpr int kTrRow = 18;          // float2 per transpose row: 16 + 2 pad (144 B) -> conflict-free b128 reads
static constexpr int kOutPitch = 33;

__global__ __launch_bounds__(256) void frontend_kernel(const float* __restrict__ arena, const int64_t* __restrict__ win_off,
                                                       FrontendTables tb, float* __restrict__ feat) {
    __shared__ float2 s_w[2048];                      // exp(-2 pi i j / 2048)
    __shared__ float2 s_tr[4][64 * kTrRow];           // per-wave transpose / Z buffer (1152 float2 >= 1024)
    __shared__ float s_p[4][768];                     // per-wave power spectrum
    __shared__ float s_out[128 * kOutPitch];          // [mel][frame] tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x >> 3, fg = blockIdx.x & 7;
    const float* x = arena + win_off[n];

    for (int i = tid; i < 2048; i += 256) s_w[i] = tb.w2048[i];
    __syncthreads();

    const int r = lane >> 4, q = lane & 15;           // pass 1: q = n0; pass 2: q = m0
    float4 pt[16];
    float2 tw2[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) pt[n1] = tb.pretw[r * 256 + 16 * n1 + q];
#pragma unroll
    for (int m0 = 0; m0 < 16; ++m0) tw2[m0] = s_w[(8 * q * m0) & 2047];

    // the two mel filters of this lane (long + short: balanced)
    const int j1 = lane, j2 = 127 - lane;
    const int st1 = tb.mel_start[j1], cn1 = tb.mel_count[j1], of1 = tb.mel_off[j1];
    const int st2 = tb.mel_start[j2], cn2 = tb.mel_count[j2], of2 = tb.mel_off[j2];

    float2* tr = s_tr[wave];
    float* pw = s_p[wave];

    for (int f = 0; f < 8; ++f) {
        const int fl = wave * 8 + f;                  // frame inside the block's tile
        const int t = fg * kFramesPerBlock + fl;      // frame index 0..255
        float2 v[16];
        if (t > 0) {
            const float* xs = x + 256 * (t - 1) + 2 * q;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const float2 s = *(const float2*)(xs + 32 * n1);
                v[n1] = make_float2(s.x * pt[n1].x - s.y * pt[n1].y, s.x * pt[n1].z + s.y * pt[n1].w);
            }
        } else {                                      // center=True, pad_mode='reflect': x[-k] = x[k]
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int i = 32 * n1 + 2 * q;        // sample i of the frame sits at window index i - 256
                const int a0 = i - 256, a1 = i - 255;
                const float s0 = x[a0 < 0 ? -a0 : a0], s1 = x[a1 < 0 ? -a1 : a1];
                v[n1] = make_float2(s0 * pt[n1].x - s1 * pt[n1].y, s0 * pt[n1].z + s1 * pt[n1].w);
            }
        }
        fft16(v);                                     // over n1 -> index m0
#pragma unroll
        for (int m0 = 0; m0 < 16; ++m0) v[m0] = cmul(v[m0], tw2[m0]);

        __syncthreads();                              // previous frame's Z / power reads are done
#pragma unroll
        for (int m0 = 0; m0 < 16; ++m0) tr[(r * 16 + m0) * kTrRow + q] = v[m0];
        __syncthreads();
        {
            const f32x4* row = (const f32x4*)(tr + (r * 16 + q) * kTrRow);   // 144-byte rows: 16-byte aligned
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const f32x4 w4 = row[p];
                v[2 * p] = make_float2(w4[0], w4[1]);
                v[2 * p + 1] = make_float2(w4[2], w4[3]);
            }
        }
        fft16(v);                                     // over n0 -> index m1 ; v[m1] = Z[4 (q + 16 m1) + r]
        __syncthreads();
        // Z buffer, lane-linear: position of k = 4 (m0 + 16 m1) + r is 64 m1 + 16 r + m0
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) tr[64 * m1 + lane] = v[m1];
        __syncthreads();
        // real-FFT untangle + power for bins k = 64 i + lane, k < 768 (bins above 743 carry no mel weight)
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int k = 64 * i + lane;
            const int kc = (1024 - k) & 1023;
            const float2 zk = tr[64 * (k >> 6) + 16 * (k & 3) + ((k >> 2) & 15)];
            const float2 zz = tr[64 * (kc >> 6) + 16 * (kc & 3) + ((kc >> 2) & 15)];
            const float2 zc = make_float2(zz.x, -zz.y);
            const float2 a = cadd(zk, zc), d = csub(zk, zc);
            const float2 wd = cmul(s_w[k], d);
            const float xr = 0.5f * (a.x + wd.y), xi = 0.5f * (a.y - wd.x);   // X = (a - i W^k d) / 2
            pw[k] = xr * xr + xi * xi;
        }
        __syncthreads();
        {
            float m1s = 0.f, m2s = 0.f;
            for (int b = 0; b < cn1; ++b) m1s = fmaf(tb.mel_w[of1 + b], pw[st1 + b], m1s);
            for (int b = 0; b < cn2; ++b) m2s = fmaf(tb.mel_w[of2 + b], pw[st2 + b], m2s);
            // exactly as written in the reference: float32 log10(x + 1), then sqrt (no log1p, no fp64)
            s_out[j1 * kOutPitch + fl] = sqrtf(log10f(m1s + 1.0f));
            s_out[j2 * kOutPitch + fl] = sqrtf(log10f(m2s + 1.0f));
        }
    }
    __syncthreads();
    {
        const int row = tid >> 1, half = tid & 1;     // 128 rows x 2 halves of 16 frames
        float* dst = feat + ((size_t)n * 128 + row) * 256 + fg * kFramesPerBlock + half * 16;
        const float* src = s_out + row * kOutPitch + half * 16;
#pragma unroll
        for (int p = 0; p < 4; ++p) *(f32x4*)(dst + 4 * p) = f32x4{src[4 * p], src[4 * p + 1], src[4 * p + 2], src[4 * p + 3]};
    }
}

hipError_t launch_frontend(const float* arena, const int64_t* win_off, int n, const FrontendTables& t, float* feat, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(frontend_kernel, dim3(n * 8), dim3(256), 0, s, arena, win_off, t, feat);
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
        default: return (float)((const double*)p)[idx];
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
__global__ __launch_bounds__(256) void resample_batch_lds_kernel(const float* __restrict__ mono, const BatchFile* __restrict__ files,
                                                                 int L, int M, int half, const float* __restrict__ taps,
                                                                 float* __restrict__ arena) {
#pragma clang fp contract(off)
    extern __shared__ float s_taps[];
    const BatchFile f = files[blockIdx.y];
    const int64_t m0 = (int64_t)blockIdx.x * kResOut;
    if (m0 >= f.n_out) return;
    const int nt = 2 * half, pitch = nt | 1;
    for (int i = threadIdx.x; i < L * nt; i += 256) { const int p = i / nt; s_taps[p * pitch + (i - p * nt)] = taps[i]; }
    __syncthreads();
    const float* in = mono + f.mono_off;
    float* out = arena + f.out_off;
    const int64_t pos0 = m0 * M;
    const int64_t base0 = pos0 / L;
    const int ph0 = (int)(pos0 - base0 * L);              // position of output m0 is base0 + ph0 / L
    const int n_here = (int)((f.n_out - m0) < kResOut ? (f.n_out - m0) : kResOut);
    for (int k = threadIdx.x; k < n_here; k += 256) {
        const int rel = ph0 + k * M;                      // < L + 4096 * M, fits 32 bits for every audio rate
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

hipError_t launch_decode_mono_batch(const void* pcm, int format, int channels, const BatchFile* d_files, int n_files, int64_t max_frames,
                                    float* mono, hipStream_t s) {
    if (n_files <= 0 || max_frames <= 0) return hipSuccess;
    const unsigned gx = (unsigned)std::min<int64_t>((max_frames + 255) / 256, 4096);
    hipLaunchKernelGGL(decode_mono_batch_kernel, dim3(gx, (unsigned)n_files), dim3(256), 0, s, (const unsigned char*)pcm, format,
                       channels, d_files, mono);
    return hipGetLastError();
}

hipError_t launch_resample_batch(const float* mono, const BatchFile* d_files, int n_files, int64_t max_out, int L, int M, int half,
                                 const float* taps, float* arena, hipStream_t s) {
    if (n_files <= 0 || max_out <= 0) return hipSuccess;
    const size_t lds = (size_t)L * ((2 * half) | 1) * sizeof(float);
    if (lds <= 150 * 1024 && (int64_t)kResOut * M < (int64_t)1 << 30) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)resample_batch_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        const unsigned gx = (unsigned)((max_out + kResOut - 1) / kResOut);
        hipLaunchKernelGGL(resample_batch_lds_kernel, dim3(gx, (unsigned)n_files), dim3(256), lds, s, mono, d_files, L, M, half, taps, arena);
        return hipGetLastError();
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

hipError_t launch_average(const float* logits, const AvgFile* files, int n_files, const int32_t* starts, double* avg, int32_t* count,
                          int max_bins, hipStream_t s) {
    if (n_files <= 0 || max_bins <= 0) return hipSuccess;
    hipLaunchKernelGGL(average_kernel, dim3((unsigned)((max_bins + 255) / 256), (unsigned)n_files), dim3(256), 0, s, logits, files,
                       starts, avg, count);
    return hipGetLastError();
}

}  // namespace ss
