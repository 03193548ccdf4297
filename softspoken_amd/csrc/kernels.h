// Internal launch interface between the host engine (engine.cpp) and the HIP kernels.
// Not part of the C ABI (include/softspoken.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace ss {

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process with contexts on several devices
// (ss_create takes a device id) must set it on each, and contexts run on several host threads.  done: one bit per device, per kernel.
inline hipError_t allow_full_lds(const void* kernel, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// ---- implicit-GEMM 3x3 (+ fused 1x1 residual) convolution on MFMA ---------------------------------
// Activations are NHWC ([window][H][W][C]), element type float or bf16.  One launch computes
//   out = act( conv3x3(cat[src0, up2(src1)]) + conv1x1(cat[res0, up2(res1)]) + rank1 + bias )
// where every term but the 3x3 is optional, and optionally also writes maxpool2x2(out).
struct ConvArgs {
    const void* src0; const void* src1;   // 3x3 input: [N][H][W][C0] and (nullable) [N][H/2][W/2][C1], nearest-upsampled
    const void* res0; const void* res1;   // 1x1 input, same convention (R0 / R1 channels)
    const void* wpk;                      // weights in MFMA fragment order (pack_conv_weights in engine.cpp)
    const float* bias;                    // [Cout] folded BatchNorm shift(s)
    const float* rank1_src;               // (nullable) [N][H][W] fp32 single-channel input of a 1->Cout 1x1 conv
    const float* rank1_w;                 // [Cout]
    void* out;                            // [N][H][W][Cout]
    void* pool_out;                       // (nullable) [N][H/2][W/2][Cout]
    int N, H, W, C0, C1, R0, R1, Cout;
    int relu;
    int tiles_y, tiles_x;
    int dbg;                              // ablation switches for tools/ (0 in production)
    // conv2.hip fusions (null = off)
    const float* first_w; const float* first_b;   // FIRST: conv1_1.conv1 folded weights [9][32] + bias [32]; input = rank1_src
    const void* flat_w; float* flat_part;         // FLAT: conv_flatten weights as MFMA fragments per mel row pair (columns 0-3 / 4-7); partial sums [N][H/4][4][W]
    int store_out;                                // FLAT: also write `out` (needed when the spec head runs)
    const void* flat_w4;                          // FLAT in conv4.hip: [128 rows][2 steps][64 lanes][8 bf16], channel order of the packed results
    void* res_out; const float* res_bias;         // A launch (RES): r = conv1x1(x) + br -> [N][H][W][Cout], weights = tap 9 of each chunk
    const void* res_in;                           // B launch: r, added before the ReLU
    const void* wpk_b; const float* bias_a;       // fused ResBlock (conv3.hip): launch-B weights, b1 (bias = b2 + br)
    // conv4.hip, "projection in B": launch A writes h only (plain = 1); launch B computes the block's 1x1 projection itself from the
    // centre pixels of the block input x = cat[xp0 (C0x channels), up2(xp1) (C1x channels)], read straight from memory as MFMA operands
    int plain;
    const void* proj_w;                           // [(C0x + C1x) / 16 steps][NT][64 lanes][8 bf16]: projection weights, A-operand order
    const void* xp0; const void* xp1; int C0x, C1x;
    // f16x2 mode: every activation tensor is two f16 planes (high and low halves of the values); a tensor pointer names the high
    // plane and the low plane lies lo_delta bytes behind it (the same distance for every tensor of a workspace)
    int64_t lo_delta;
    int* range_flag;                              // f16x2: set to 1 when a value that is being split does not fit an f16 (overflow, NaN)
    void* stamps;                                 // dev build: [grid][waves][8] uint32 segment times of a stage (null otherwise)
};
// NT = number of 32-wide output-channel tiles per block (1..3); Cout % (32*NT) == 0.
// second structure (conv2.hip): persistent blocks, register prefetch, resident weights, staged stores
hipError_t launch_conv3x3_v2(const ConvArgs& a, bool bf16, int NT, int num_cus, hipStream_t s);
const char* conv_v2_variant(const ConvArgs& a, bool bf16, int NT, int num_cus);   // instantiation name, as rocprofv3 prints it
int conv_v2_flat_groups(bool bf16);   // row groups per window in ConvArgs::flat_part
// conv4.hip: 16-bit operands on the bf16 / f16 matrix instructions, A / B launches of a ResBlock (inputs need the engine's 256-byte
// zero header).  prec: 1 = bf16 (one plane per tensor), 2 = f16x2 (two f16 planes per tensor, three products per term)
bool conv_v4_supports(const ConvArgs& a, int NT, int num_cus, int prec);
const char* conv_v4_variant(const ConvArgs& a, int NT, int num_cus, int prec);
int conv_v4_flat_groups();           // row groups per window in ConvArgs::flat_part when conv4.hip's FLAT launch ran
hipError_t launch_conv3x3_v4(const ConvArgs& a, int NT, int num_cus, int prec, hipStream_t s);

// conv4_ups.hip (f16x2): the plain A launch of a decoder block whose upsampled input half runs at low resolution with four pre-summed
// taps per output parity class (weights: weights.hip pack_conv_split_ups, conv_ups_weight_bytes of them)
bool conv_ups_supports(const ConvArgs& a, int num_cus);
const char* conv_ups_variant();
size_t conv_ups_weight_bytes(int C0, int C1);
hipError_t launch_conv3x3_ups(const ConvArgs& a, int num_cus, hipStream_t s);
// ... and the A launch that also writes the block's 1x1 projection r (res_out), any number of 32-channel output groups, its banks
// streamed through a three-slot ring of half-chunk entries (weights: pack_conv_split_upsr; biases travel inside them)
bool conv_upsr_supports(const ConvArgs& a, int num_cus);
const char* conv_upsr_variant();
size_t conv_upsr_weight_bytes(int C0, int C1, int Cout);
hipError_t launch_conv3x3_upsr(const ConvArgs& a, int num_cus, hipStream_t s);

// conv2_ups.hip (fp32): the A launch of a decoder block (h -> out, r -> res_out) with four pre-summed taps per output parity class on
// the upsampled input half (weights: weights.hip pack_conv_v2_ups, conv_ups32_weight_bytes of them); Cout = 32, 64 or 96
// NT: 32-channel output tiles per block (the bank is packed for it); Cout / (32 NT) output-channel groups are separate tiles
// MTW: M-tiles per wave (1: 8 waves per block, 2: 4 waves, each the two halves of one parity class)
bool conv_ups32_supports(const ConvArgs& a, int NT, int MTW, int num_cus);
const char* conv_ups32_variant(int NT, int MTW);
size_t conv_ups32_weight_bytes(int C0, int C1, int Cout);
hipError_t launch_conv3x3_ups32(const ConvArgs& a, int NT, int MTW, int num_cus, hipStream_t s);

// conv1s.hip (f16x2): conv1_1 as a row-streaming kernel -- a wave owns a 32-column strip, h1 stays in registers, no LDS traffic but the
// weight fragments, no barrier after the prologue.  wpk: pack_conv_stream's banks (conv1_stream_weight_bytes); the other fields as for
// conv4.hip's FIRST + RANK1 + POOL launch (first_w / first_b, rank1_src = features, rank1_w, bias = b2 + br, out, pool_out).
// rows_per_unit: rows of a (window, band, strip) work unit (even, divides 128)
bool conv1_stream_supports(const ConvArgs& a);
const char* conv1_stream_variant(const ConvArgs& a, int form);   // (a.plain = 1: the f16 range of h1 and c1 is proven from the weights, no run-time test)
size_t conv1_stream_weight_bytes();
// form: 32 = one 32-column tile per strip row on v_mfma_f32_32x32x16_f16 (wpk: pack_conv_stream), 16 = two interleaved 16-pixel tiles on
// v_mfma_f32_16x16x32_f16 (wpk: pack_conv_stream16; same size)
hipError_t launch_conv1_stream(const ConvArgs& a, int form, int rows_per_unit, int num_cus, hipStream_t s);

// heads.hip
// ResBlock1D(4,4) + Conv1d(4,1,1) fed by the FLAT partial sums [N][n_parts][4][256]: sums them in order, adds conv_flatten's bias,
// ReLU, then the 1-D head -> logits [N][256]
// fscale: exact power of two that takes conv_flatten's partial sums back to true units (f16x2 channel normalisation; 1 otherwise)
struct Head1dWeights { float w1[4][4][3], b1[4], w2[4][4][3], wr[4][4], b2r[4], wo[4], bo, fscale; };
hipError_t launch_mask_head_parts(const float* parts, int n_parts, const float* flat_bias, const Head1dWeights& hw, float* logits,
                                  int N, hipStream_t s);
// spec head tail: Conv2d(32,2,1) + bias + ReLU: x NHWC [N][128][256][32] -> spec NCHW [N][2][128][256] fp32
// prec: 0 = fp32 activations, 1 = bf16, 2 = two f16 planes (lo_delta apart)
hipError_t launch_spec_tail(const void* x, int64_t lo_delta, const float* w /*[2][32]*/, const float* bias, float* spec, int N, int prec,
                            hipStream_t s);

// ---- front-end ----------------------------------------------------------------------------------------
static constexpr int kMelLo = 10, kMelHi = 32, kMelPitch = 43;   // taps of a lane's narrow / wide filter; odd LDS pitch
struct FrontendTables {
    const float4* pretw;    // [4][256]: window x pre-twiddle, (w0*c, w1*s, w0*s, w1*c) for z[n] * W1024^(n r)
    const float2* w2048;    // [2048]: exp(-2 pi i j / 2048)
    const int* mel_start;   // [128] first bin of filter j
    const int* mel_count;   // [128] number of bins
    const int* mel_off;     // [128] offset into mel_w
    const float* mel_w;     // packed non-zero weights, ascending bin
    int mel_nw;             // number of packed weights (<= 1536)
    const float* mel_wp;    // [64][kMelPitch]: lane l's filters l (kMelLo taps) and 127 - l (kMelHi taps), zero-padded
    int dbg;                // ablation switches for tools/ (0 in production); bit 8: the first kernel (dev build A/B runs)
    // second kernel (16 lanes per frame, passes r = 0..3)
    const float2* win2;     // [256]: (w[2n], w[2n+1])
    const float2* twt;      // [4][16 n0][16 m0]: W1024^(n0 (4 m0 + r)), the twiddle between the two radix-16 passes
    const float2* wkt;      // [4][16 m1][16 m0]: W2048^(4 (m0 + 16 m1) + r), the real-FFT untangle's twiddle
    const float* mel_wq;    // [64][kMelRow]: lane l's weights for paired reads: 2 kMelPairsLo of filter l, 2 kMelPairsHi of filter 127 - l
    const int* mel_p0;      // [64][2]: first word of those two runs in the padded power-spectrum buffer (even)
};
static constexpr int kMelPairsLo = 9, kMelPairsHi = 20, kMelRow = 60;    // 8-byte reads per narrow / wide filter; row pitch (16-byte multiple)
static constexpr int kPwWords = 816;                                      // 768 bins + 4 pad words behind every 64
// windows: arena offsets of each window's first sample.  feat: [n][128][256] fp32.
hipError_t launch_frontend(const float* arena, const int64_t* win_off, int n, const FrontendTables& t, float* feat,
                           int num_cus, hipStream_t s);

// ---- decode / mixdown / resample ----------------------------------------------------------------------
// batched: every file of a job in one launch; sr == 22050 files go mono -> arena directly (L == M == 1, half == 0)
struct BatchFile { int64_t pcm_off; int64_t frames; int64_t mono_off; int64_t out_off; int64_t n_out; };
hipError_t launch_decode_mono_batch(const void* pcm, int format, int channels, const BatchFile* d_files, int n_files,
                                    int64_t max_frames, float* mono, hipStream_t s);
// review-screen spectrogram: |STFT| with n_fft = win = 512, hop 256, centred, zero-padded -> [257][n_frames]
hipError_t launch_stft512_mag(const float* x, int64_t n, int64_t n_frames, float* out, int num_cus, hipStream_t s);
// silencer: decode -> zero [begin,end) frame ranges (disjoint, ascending) -> 16-bit PCM
hipError_t launch_silence_encode(const void* pcm, int format, int channels, int64_t frames, const int64_t* d_ranges, int n_ranges,
                                 short* out, hipStream_t s);
// decode + mixdown + polyphase resampling in one launch (no mono tensor); applies when resample_fused_applies says so
bool resample_fused_applies(int L, int M, int half);
hipError_t launch_resample_fused(const void* pcm, int format, int channels, const BatchFile* d_files, int n_files, int64_t max_out, int L, int M,
                                 int half, const float* taps, float* arena, int num_cus, hipStream_t s);
hipError_t launch_resample_batch(const float* mono, const BatchFile* d_files, int n_files, int64_t max_out, int L, int M, int half,
                                 const float* taps, float* arena, int num_cus, hipStream_t s);

// ---- overlap averaging (NNDetector.py:153-190), double accumulation ---------------------------------
struct AvgFile { int64_t logit_off; int64_t bin_off; int32_t W; int32_t n_bins; int64_t start_off; };
hipError_t launch_average(const float* logits, const AvgFile* files, int n_files, const int32_t* starts, double* avg,
                          int32_t* count, int max_bins, hipStream_t s);
// covered / above-threshold bit per bin, 64 bins per word (words = ceil(total_bins / 64), both arrays)
hipError_t launch_bin_masks(const double* avg, const int32_t* count, int64_t total_bins, double threshold, unsigned long long* above,
                            unsigned long long* covered, hipStream_t s);

}  // namespace ss
