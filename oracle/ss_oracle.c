/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not product code, never linked into libsoftspoken_hip.so.
 *
 * Plain-C restatement of the reference's "Run Voice Detector" path, written independently of the HIP
 * kernels (different FFT, direct NCHW convolutions, BatchNorm applied un-folded in the reference's
 * order).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
 *
 * Parity status: the CNN body / averaging / regions are checked against goldens produced by the
 * reference's own classes (tests/golden); the mel front-end restates torchaudio's documented algorithm
 * ("parity unpinned" at that boundary, torchaudio is not in the image); decode/resample restate
 * voice_activity.py:32-69 with the build's own resampler ("parity unpinned").
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/ss_oracle.c -o oracle/_build/libss_oracle.so -lm
 * Citations are file:line under /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SR 22050
#define WIN 66150
#define PI 3.14159265358979323846

/* ---------------------------------------------------------------------------------------------------
 * weights blob access ("SSWBLOB1", same container the library takes)
 * ------------------------------------------------------------------------------------------------- */
typedef struct { char name[96]; uint32_t dtype, ndim; int64_t shape[4]; uint64_t offset, nbytes; } entry_t;

static const float* blob_f32(const void* blob, const char* name) {
    const char* b = (const char*)blob;
    uint32_t n; memcpy(&n, b + 8, 4);
    for (uint32_t i = 0; i < n; ++i) {
        entry_t e; memcpy(&e, b + 16 + (size_t)i * sizeof(entry_t), sizeof(entry_t));
        if (strncmp(e.name, name, 96) == 0) return (const float*)(b + e.offset);
    }
    fprintf(stderr, "ss_oracle: tensor %s missing\n", name);
    abort();
}

/* ---------------------------------------------------------------------------------------------------
 * A2  decode / mono / resample  (voice_activity.py:37-38, 61-62, 65-67)
 * ------------------------------------------------------------------------------------------------- */
void so_decode_mono(const unsigned char* p, int fmt, int ch, int64_t frames, float* mono) {
    for (int64_t i = 0; i < frames; ++i) {
        float acc = 0.f;
        for (int c = 0; c < ch; ++c) {
            const int64_t k = i * ch + c;
            float v;
            switch (fmt) {
                case 1: v = ((float)p[k] - 128.0f) / 128.0f; break;
                case 2: { int16_t s; memcpy(&s, p + 2 * k, 2); v = (float)s / 32768.0f; } break;
                case 3: { int32_t s = p[3 * k] | (p[3 * k + 1] << 8) | (p[3 * k + 2] << 16); if (s & 0x800000) s -= 0x1000000; v = (float)s / 8388608.0f; } break;
                case 4: { int32_t s; memcpy(&s, p + 4 * k, 4); v = (float)((double)s / 2147483648.0); } break;
                case 5: memcpy(&v, p + 4 * k, 4); break;
                default: { double d; memcpy(&d, p + 8 * k, 8); v = (float)d; } break;
            }
            acc = c == 0 ? v : acc + v;
        }
        mono[i] = ch > 1 ? acc / (float)ch : acc;
    }
}

static double bessel_i0(double x) {
    double s = 1.0, t = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) { t *= q / ((double)k * k); s += t; if (t < s * 1e-17) break; }
    return s;
}

static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

int64_t so_resampled_length(int64_t frames, int sr) { return sr == SR ? frames : (frames * SR + sr - 1) / sr; }

/* Kaiser-windowed sinc polyphase, the build's own design (32 zero crossings, beta 12, roll-off 0.95) */
void so_resample(const float* in, int64_t n_in, int sr_in, float* out, int64_t n_out) {
    if (sr_in == SR) { memcpy(out, in, (size_t)n_out * 4); return; }
    const int g = gcd_i(sr_in, SR), L = SR / g, M = sr_in / g;
    const double scale = sr_in > SR ? (double)SR / sr_in : 1.0;
    const double fc = 0.95 * scale, beta = 12.0;
    const int half = (int)ceil(32.0 / scale);
    float* taps = (float*)malloc((size_t)L * 2 * half * 4);
    const double i0b = bessel_i0(beta);
    for (int p = 0; p < L; ++p) {
        const double frac = (double)(((int64_t)p * M) % L) / L;
        for (int j = 0; j < 2 * half; ++j) {
            const double d = (double)(j - half + 1) - frac, xx = fc * d;
            const double sinc = xx == 0.0 ? 1.0 : sin(PI * xx) / (PI * xx);
            double w = 0.0;
            if (fabs(d) <= half) { double u = 1.0 - (d / half) * (d / half); if (u < 0) u = 0; w = bessel_i0(beta * sqrt(u)) / i0b; }
            taps[(size_t)p * 2 * half + j] = (float)(fc * sinc * w);
        }
    }
#pragma omp parallel for
    for (int64_t m = 0; m < n_out; ++m) {
        const int64_t pos = m * M, base = pos / L;
        const int ph = (int)(pos - base * L);
        volatile float acc = 0.f;      /* volatile: one rounding per multiply and per add, no FMA contraction */
        for (int j = 0; j < 2 * half; ++j) {
            const int64_t idx = base + j - half + 1;
            const float sv = (idx >= 0 && idx < n_in) ? in[idx] : 0.f;
            volatile float pr = taps[(size_t)ph * 2 * half + j] * sv;
            acc = acc + pr;
        }
        out[m] = acc;
    }
    free(taps);
}

/* ---------------------------------------------------------------------------------------------------
 * A3  mel front-end (pytorch_neural_nets.py:92-99,144-153): Hann(512) frame, centred zero-pad to 2048
 * (torch.stft), reflect padding, |rFFT|^2, fb matmul, sqrt(log10(x+1)), first 256 frames.
 * Double-precision radix-2 FFT of the full 2048-sample frame: deliberately not the kernel's algorithm.
 * ------------------------------------------------------------------------------------------------- */
static void fft2048(double* re, double* im) {
    const int n = 2048;
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * PI / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * k), wi = sin(ang * k);
                const int a = i + k, b = i + k + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi; re[a] += xr; im[a] += xi;
            }
    }
}

void so_mel_features(const float* x /*[n][66150]*/, int n, const float* window /*512*/, const float* fb /*[1025][128]*/,
                     float* out /*[n][128][256]*/) {
#pragma omp parallel for collapse(2)
    for (int w = 0; w < n; ++w)
        for (int t = 0; t < 256; ++t) {
            double re[2048], im[2048];
            memset(re, 0, sizeof re); memset(im, 0, sizeof im);
            const float* xs = x + (size_t)w * WIN;
            for (int i = 0; i < 512; ++i) {
                int s = 256 * t - 256 + i;            /* centre=True: frame t is centred on sample 256 t */
                if (s < 0) s = -s;                    /* pad_mode='reflect' */
                re[768 + i] = (double)(xs[s] * window[i]);   /* window zero-padded centrally to n_fft */
            }
            fft2048(re, im);
            float mel[128];
            for (int j = 0; j < 128; ++j) mel[j] = 0.f;
            for (int k = 0; k < 1025; ++k) {
                const float mag = (float)sqrt(re[k] * re[k] + im[k] * im[k]);   /* .abs() */
                const float pw = mag * mag;                                      /* .pow(2) */
                const float* row = fb + (size_t)k * 128;
                for (int j = 0; j < 128; ++j) mel[j] += pw * row[j];
            }
            for (int j = 0; j < 128; ++j) out[((size_t)w * 128 + j) * 256 + t] = sqrtf(log10f(mel[j] + 1.0f));
        }
}

/* ---------------------------------------------------------------------------------------------------
 * A4  SpecUNet_2D body in eval mode (pytorch_neural_nets.py:7-77,142-197), NCHW float32, un-folded BN
 * ------------------------------------------------------------------------------------------------- */
static void conv2d(const float* in, int cin, int H, int W, const float* w, int cout, int k, float* out) {
    const int pad = k / 2;
#pragma omp parallel for
    for (int co = 0; co < cout; ++co) {
        float* o = out + (size_t)co * H * W;
        memset(o, 0, (size_t)H * W * 4);
        for (int ci = 0; ci < cin; ++ci) {
            const float* ip = in + (size_t)ci * H * W;
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx) {
                    const float wv = w[(((size_t)co * cin + ci) * k + ky) * k + kx];
                    const int dy = ky - pad, dx = kx - pad;
                    const int y0 = dy < 0 ? -dy : 0, y1 = dy > 0 ? H - dy : H;
                    const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? W - dx : W;
                    for (int y = y0; y < y1; ++y) {
                        float* orow = o + (size_t)y * W;
                        const float* irow = ip + (size_t)(y + dy) * W + dx;
                        for (int xx = x0; xx < x1; ++xx) orow[xx] += wv * irow[xx];
                    }
                }
        }
    }
}

static void bn_(float* x, int c, int hw, const void* blob, const char* prefix, int relu) {
    char key[160];
    snprintf(key, sizeof key, "%s.weight", prefix); const float* g = blob_f32(blob, key);
    snprintf(key, sizeof key, "%s.bias", prefix); const float* b = blob_f32(blob, key);
    snprintf(key, sizeof key, "%s.running_mean", prefix); const float* mu = blob_f32(blob, key);
    snprintf(key, sizeof key, "%s.running_var", prefix); const float* var = blob_f32(blob, key);
    for (int ch = 0; ch < c; ++ch) {
        const float inv = 1.0f / sqrtf(var[ch] + 1e-5f), alpha = g[ch] * inv, beta = b[ch] - mu[ch] * alpha;
        float* p = x + (size_t)ch * hw;
        for (int i = 0; i < hw; ++i) { float v = p[i] * alpha + beta; p[i] = (relu && v < 0.f) ? 0.f : v; }
    }
}

/* ResBlock.forward (:32-41); k1 = kernel size (3), H==1 for the 1-D block */
static float* resblock(const float* x, int cin, int cout, int H, int W, const void* blob, const char* name) {
    char key[160], pre[160];
    const size_t hw = (size_t)H * W;
    float* idt = (float*)malloc(hw * cout * 4);
    float* h = (float*)malloc(hw * cout * 4);
    float* o = (float*)malloc(hw * cout * 4);
    snprintf(key, sizeof key, "%s.residual.0.weight", name); conv2d(x, cin, H, W, blob_f32(blob, key), cout, 1, idt);
    snprintf(pre, sizeof pre, "%s.residual.1", name); bn_(idt, cout, (int)hw, blob, pre, 0);
    snprintf(key, sizeof key, "%s.conv1.0.weight", name);
    if (H == 1) {   /* Conv1d k=3 == Conv2d with a (1,3) kernel: reuse conv2d through a 3x3 kernel whose outer rows are zero */
        const float* w1 = blob_f32(blob, key);
        float* w3 = (float*)calloc((size_t)cout * cin * 9, 4);
        for (int i = 0; i < cout * cin; ++i) for (int kx = 0; kx < 3; ++kx) w3[(size_t)i * 9 + 3 + kx] = w1[(size_t)i * 3 + kx];
        conv2d(x, cin, H, W, w3, cout, 3, h); free(w3);
    } else conv2d(x, cin, H, W, blob_f32(blob, key), cout, 3, h);
    snprintf(pre, sizeof pre, "%s.conv1.1", name); bn_(h, cout, (int)hw, blob, pre, 1);
    snprintf(key, sizeof key, "%s.conv2.0.weight", name);
    if (H == 1) {
        const float* w1 = blob_f32(blob, key);
        float* w3 = (float*)calloc((size_t)cout * cout * 9, 4);
        for (int i = 0; i < cout * cout; ++i) for (int kx = 0; kx < 3; ++kx) w3[(size_t)i * 9 + 3 + kx] = w1[(size_t)i * 3 + kx];
        conv2d(h, cout, H, W, w3, cout, 3, o); free(w3);
    } else conv2d(h, cout, H, W, blob_f32(blob, key), cout, 3, o);
    snprintf(pre, sizeof pre, "%s.conv2.1", name); bn_(o, cout, (int)hw, blob, pre, 0);
    for (size_t i = 0; i < hw * cout; ++i) { float v = o[i] + idt[i]; o[i] = v < 0.f ? 0.f : v; }
    free(idt); free(h);
    return o;
}

static float* maxpool(const float* x, int c, int H, int W) {
    float* o = (float*)malloc((size_t)c * (H / 2) * (W / 2) * 4);
    for (int ch = 0; ch < c; ++ch)
        for (int y = 0; y < H / 2; ++y)
            for (int xx = 0; xx < W / 2; ++xx) {
                const float* p = x + ((size_t)ch * H + 2 * y) * W + 2 * xx;
                const float a = p[0] > p[1] ? p[0] : p[1], b = p[W] > p[W + 1] ? p[W] : p[W + 1];
                o[((size_t)ch * (H / 2) + y) * (W / 2) + xx] = a > b ? a : b;
            }
    return o;
}

/* torch.cat([skip, Upsample(nearest x2)(low)], dim=1) (:168-180) */
static float* cat_up(const float* skip, int cs, const float* low, int cl, int H, int W) {
    float* o = (float*)malloc((size_t)(cs + cl) * H * W * 4);
    memcpy(o, skip, (size_t)cs * H * W * 4);
    for (int ch = 0; ch < cl; ++ch)
        for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < W; ++xx)
                o[((size_t)(cs + ch) * H + y) * W + xx] = low[((size_t)ch * (H / 2) + y / 2) * (W / 2) + xx / 2];
    return o;
}

/* feats [n][128][256] -> mask [n][256] (+ spec [n][2][128][256] if non-NULL; flatten taps if non-NULL) */
void so_unet_forward(const void* blob, const float* feats, int n, float* mask, float* spec, float* flat_out) {
    for (int b = 0; b < n; ++b) {
        const float* x = feats + (size_t)b * 32768;
        float* c1 = resblock(x, 1, 32, 128, 256, blob, "conv1_1");
        float* p1 = maxpool(c1, 32, 128, 256);
        float* c2 = resblock(p1, 32, 64, 64, 128, blob, "conv2_1");
        float* p2 = maxpool(c2, 64, 64, 128);
        float* c3 = resblock(p2, 64, 96, 32, 64, blob, "conv3_1");
        float* p3 = maxpool(c3, 96, 32, 64);
        float* c4 = resblock(p3, 96, 128, 16, 32, blob, "conv4_1");
        float* p4 = maxpool(c4, 128, 16, 32);
        float* bo = resblock(p4, 128, 128, 8, 16, blob, "conv_bottleneck");
        float* en = resblock(bo, 128, 128, 8, 16, blob, "encoder_out");
        float* m1 = cat_up(c4, 128, en, 128, 16, 32);
        float* c6 = resblock(m1, 256, 96, 16, 32, blob, "conv6");
        float* m2 = cat_up(c3, 96, c6, 96, 32, 64);
        float* c7 = resblock(m2, 192, 64, 32, 64, blob, "conv7");
        float* m3 = cat_up(c2, 64, c7, 64, 64, 128);
        float* c8 = resblock(m3, 128, 32, 64, 128, blob, "conv8");
        float* m4 = cat_up(c1, 32, c8, 32, 128, 256);
        float* c9 = resblock(m4, 64, 32, 128, 256, blob, "conv9_1");
        if (spec) {   /* :184-185 */
            float* s = resblock(c9, 32, 32, 128, 256, blob, "spec_output_conv.0");
            const float* w = blob_f32(blob, "spec_output_conv.1.weight"); const float* bs = blob_f32(blob, "spec_output_conv.1.bias");
            for (int co = 0; co < 2; ++co)
                for (int i = 0; i < 32768; ++i) {
                    float a = 0.f;
                    for (int ci = 0; ci < 32; ++ci) a += w[co * 32 + ci] * s[(size_t)ci * 32768 + i];
                    a += bs[co];
                    spec[((size_t)b * 2 + co) * 32768 + i] = a < 0.f ? 0.f : a;
                }
            free(s);
        }
        /* conv_flatten (:188-192): kernel (128,1) */
        float flat[4 * 256];
        const float* wf = blob_f32(blob, "conv_flatten.weight"); const float* bf = blob_f32(blob, "conv_flatten.bias");
        for (int co = 0; co < 4; ++co)
            for (int t = 0; t < 256; ++t) {
                float a = 0.f;
                for (int ci = 0; ci < 32; ++ci)
                    for (int h = 0; h < 128; ++h) a += wf[((size_t)co * 32 + ci) * 128 + h] * c9[((size_t)ci * 128 + h) * 256 + t];
                a += bf[co];
                flat[co * 256 + t] = a < 0.f ? 0.f : a;
            }
        if (flat_out) memcpy(flat_out + (size_t)b * 1024, flat, sizeof flat);
        float* r = resblock(flat, 4, 4, 1, 256, blob, "mask_output_conv.0");     /* :195 */
        const float* wo = blob_f32(blob, "mask_output_conv.1.weight"); const float* bo1 = blob_f32(blob, "mask_output_conv.1.bias");
        for (int t = 0; t < 256; ++t) {
            float a = 0.f;
            for (int ci = 0; ci < 4; ++ci) a += wo[ci] * r[ci * 256 + t];
            mask[(size_t)b * 256 + t] = a + bo1[0];
        }
        free(c1); free(p1); free(c2); free(p2); free(c3); free(p3); free(c4); free(p4); free(bo); free(en);
        free(m1); free(c6); free(m2); free(c7); free(m3); free(c8); free(m4); free(c9); free(r);
    }
}

/* ---------------------------------------------------------------------------------------------------
 * A1 / A7 / A8  (NNDetector.py:55-82, 153-190, 103-143; worker.py:100)
 * ------------------------------------------------------------------------------------------------- */
int64_t so_plan_windows(double duration_s) {
    const double L = nearbyint(duration_s * 22050.0) + 6.0 * 22050.0;
    const int64_t W = (int64_t)ceil((L - 66150.0) / 13230.0);
    return W < 0 ? 0 : W;
}

/* logits [W][256]; avg/idx sized >= round(n_padded/22050*256/3); returns kept count */
int64_t so_average(const float* logits, int64_t W, int64_t n_padded, double* avg, int64_t* idx) {
    const double secs = (double)n_padded / 22050.0;
    const int64_t nb = (int64_t)nearbyint(secs * 256.0 / 3.0);
    double* s = (double*)calloc((size_t)nb, 8);
    double* c = (double*)calloc((size_t)nb, 8);
    for (int64_t i = 0; i < W; ++i) {
        const int64_t st = (int64_t)nearbyint((double)i * 0.6 / (3.0 / 256.0));
        for (int t = 0; t < 256; ++t) { s[st + t] += (double)logits[i * 256 + t]; c[st + t] += 1.0; }
    }
    int64_t k = 0;
    for (int64_t j = 0; j < nb; ++j) if (c[j] >= 1.0) { avg[k] = s[j] / c[j]; idx[k] = j; ++k; }
    free(s); free(c);
    return k;
}

static double bin_time(int64_t i) { char b[64]; snprintf(b, sizeof b, "%.4f", (double)i / (256.0 / 3.0)); return strtod(b, NULL); }

/* returns region count; out holds (start - 3, end - 3) pairs */
int64_t so_regions(const double* avg, const int64_t* idx, int64_t n, double thr, double brk, double* out, int64_t cap) {
    int64_t nr = 0; int open = 0; double st = 0, en = 0;
    double* runs = (double*)malloc((size_t)(n + 2) * 16);
    for (int64_t i = 0; i < n; ++i) {
        if (avg[i] > thr) { const double t = bin_time(idx[i]); if (!open) { st = t; open = 1; } en = t; }
        else if (open) { runs[2 * nr] = st; runs[2 * nr + 1] = en; ++nr; open = 0; }
    }
    if (open) { runs[2 * nr] = st; runs[2 * nr + 1] = en; ++nr; }
    int64_t m = 0;
    for (int64_t i = 0; i < nr; ++i) {
        if (m > 0 && runs[2 * i] - out[2 * (m - 1) + 1] <= brk) out[2 * (m - 1) + 1] = runs[2 * i + 1];
        else if (m < cap) { out[2 * m] = runs[2 * i]; out[2 * m + 1] = runs[2 * i + 1]; ++m; }
    }
    free(runs);
    for (int64_t i = 0; i < m; ++i) { out[2 * i] -= 3.0; out[2 * i + 1] -= 3.0; }
    return m;
}
