// Host engine behind the C ABI (include/softspoken.h): weights blob -> folded BatchNorm -> MFMA fragment
// packing; signal arena in HBM; per-chunk launch sequence of the U-Net; averaging + region finding.
// Reference files are cited per function (paths relative to the reference root).
#include "../../include/softspoken.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace ss;

// ------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

struct ss_ctx;
static int fail(ss_ctx* c, int code, const std::string& msg);

#define HIPCHK(c, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail((c), SS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

// ------------------------------------------------------------------------------------------------------
// weights blob ("SSWBLOB1"): header {magic[8], u32 n, u32 0}, n entries {name[96], u32 dtype (0 f32, 1 i64),
// u32 ndim, i64 shape[4], u64 offset, u64 nbytes}, then tensor data (offsets from blob start).
// ------------------------------------------------------------------------------------------------------
struct BlobEntry { char name[96]; uint32_t dtype, ndim; int64_t shape[4]; uint64_t offset, nbytes; };
static_assert(sizeof(BlobEntry) == 152, "blob entry layout");

struct Blob {
    std::map<std::string, BlobEntry> e;
    const char* base = nullptr;
    size_t size = 0;
    bool has(const std::string& k) const { return e.count(k) != 0; }
    const float* f32(const std::string& k, size_t count, std::string& err) const {
        auto it = e.find(k);
        if (it == e.end()) { err = "weights blob: missing tensor '" + k + "'"; return nullptr; }
        if (it->second.dtype != 0 || it->second.nbytes != count * 4) {
            err = "weights blob: tensor '" + k + "' has wrong dtype/size"; return nullptr;
        }
        return (const float*)(base + it->second.offset);
    }
};

static bool parse_blob(const void* p, size_t n, Blob& b, std::string& err) {
    if (n < 16 || memcmp(p, "SSWBLOB1", 8) != 0) { err = "weights blob: bad magic"; return false; }
    uint32_t cnt; memcpy(&cnt, (const char*)p + 8, 4);
    if (16 + (size_t)cnt * sizeof(BlobEntry) > n) { err = "weights blob: truncated table"; return false; }
    b.base = (const char*)p; b.size = n;
    for (uint32_t i = 0; i < cnt; ++i) {
        BlobEntry en; memcpy(&en, (const char*)p + 16 + (size_t)i * sizeof(BlobEntry), sizeof(BlobEntry));
        en.name[95] = 0;
        if (en.offset > n || en.nbytes > n - en.offset /* no sum: it could wrap */ || (en.offset & 3)) { err = std::string("weights blob: bad extent for ") + en.name; return false; }
        b.e[en.name] = en;
    }
    return true;
}

// conv weight [cout][cin][k] with BatchNorm (eval, eps 1e-5) folded in:
//   w' = w * gamma / sqrt(var + eps),  b' = beta - mean * gamma / sqrt(var + eps)     
struct Folded { int cout = 0, cin = 0, k = 0; std::vector<float> w, b; };

static bool fold_conv_bn(const Blob& bl, const std::string& conv, const std::string& bn, int cout, int cin, int k, Folded& f,
                         std::string& err) {
    const float* w = bl.f32(conv + ".weight", (size_t)cout * cin * k, err);
    const float* g = bl.f32(bn + ".weight", cout, err);
    const float* be = bl.f32(bn + ".bias", cout, err);
    const float* mu = bl.f32(bn + ".running_mean", cout, err);
    const float* var = bl.f32(bn + ".running_var", cout, err);
    if (!w || !g || !be || !mu || !var) return false;
    f.cout = cout; f.cin = cin; f.k = k;
    f.w.resize((size_t)cout * cin * k); f.b.resize(cout);
    for (int c = 0; c < cout; ++c) {
        const double sc = (double)g[c] / std::sqrt((double)var[c] + 1e-5);
        for (int i = 0; i < cin * k; ++i) f.w[(size_t)c * cin * k + i] = (float)((double)w[(size_t)c * cin * k + i] * sc);
        f.b[c] = (float)((double)be[c] - (double)mu[c] * sc);
    }
    return true;
}

static uint16_t f2bf(float x) {   // round-to-nearest-even, NaN stays NaN
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// MFMA fragment order for conv3x3_mfma_kernel (conv.hip): for output-channel group g, K chunk ci, tap t,
// sub-step s, 32-channel tile nt, lane l = (j = l&31, h = l>>5):
//   bf16: 8 values  W[cout = g*32*NT + nt*32 + j][k = ci*32 + s*16 + h*8 + e][t]
//   fp32: 4 values  W[cout][k = ci*16 + h*8 + s*4 + e][t]
// main chunks carry 9 taps, residual (1x1) chunks one.
static void pack_conv(const Folded* w3, const Folded* wr, bool bf16, int NT, std::vector<char>& out) {
    const int cout = w3 ? w3->cout : wr->cout;
    const int KC = bf16 ? 32 : 16, per = bf16 ? 8 : 4, ES = bf16 ? 2 : 4;
    const int nmain = w3 ? w3->cin / KC : 0, nres = wr ? wr->cin / KC : 0;
    const int ngroups = cout / (32 * NT);
    const size_t tap_bytes = (size_t)2 * NT * 1024;
    out.assign((size_t)ngroups * (nmain * 9 + nres) * tap_bytes, 0);
    auto put = [&](size_t byte_off, float v) {
        if (bf16) { uint16_t h = f2bf(v); memcpy(&out[byte_off], &h, 2); }
        else memcpy(&out[byte_off], &v, 4);
    };
    for (int g = 0; g < ngroups; ++g)
        for (int ci = 0; ci < nmain + nres; ++ci) {
            const bool is_res = ci >= nmain;
            const Folded* f = is_res ? wr : w3;
            const int cc = is_res ? ci - nmain : ci;
            const int ntaps = is_res ? 1 : 9;
            const size_t cbase = ((size_t)g * (nmain * 9 + nres) + (is_res ? nmain * 9 + cc : cc * 9)) * tap_bytes;
            for (int t = 0; t < ntaps; ++t)
                for (int s = 0; s < 2; ++s)
                    for (int nt = 0; nt < NT; ++nt)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < per; ++e) {
                                const int j = l & 31, h = l >> 5;
                                const int k = bf16 ? cc * 32 + s * 16 + h * 8 + e : cc * 16 + h * 8 + s * 4 + e;
                                const int co = g * 32 * NT + nt * 32 + j;
                                const float v = f->w[((size_t)co * f->cin + k) * f->k + t];
                                put(cbase + (size_t)t * tap_bytes + ((size_t)(s * NT + nt) * 64 + l) * 16 + (size_t)e * ES, v);
                            }
        }
}

// Second structure (conv2.hip): per K chunk 9 taps of the 3x3 and, for an A launch, a tenth "tap" holding the 1x1
// residual projection of the same input channels.  Same lane / sub-step layout as pack_conv.
static void pack_conv_v2(const Folded& w3, const Folded* wr, bool bf16, int NT, std::vector<char>& out) {
    const int KC = bf16 ? 32 : 16, per = bf16 ? 8 : 4, ES = bf16 ? 2 : 4;
    const int nch = w3.cin / KC, taps = wr ? 10 : 9, ngroups = w3.cout / (32 * NT);
    const size_t tap_bytes = (size_t)2 * NT * 1024;
    out.assign((size_t)ngroups * nch * taps * tap_bytes, 0);
    for (int g = 0; g < ngroups; ++g)
        for (int ci = 0; ci < nch; ++ci)
            for (int t = 0; t < taps; ++t)
                for (int s = 0; s < 2; ++s)
                    for (int nt = 0; nt < NT; ++nt)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < per; ++e) {
                                const int j = l & 31, h = l >> 5;
                                const int k = bf16 ? ci * 32 + s * 16 + h * 8 + e : ci * 16 + h * 8 + s * 4 + e;
                                const int co = g * 32 * NT + nt * 32 + j;
                                const float v = t < 9 ? w3.w[((size_t)co * w3.cin + k) * 9 + t] : wr->w[(size_t)co * wr->cin + k];
                                const size_t off = (((size_t)g * nch + ci) * taps + t) * tap_bytes + ((size_t)(s * NT + nt) * 64 + l) * 16 + (size_t)e * ES;
                                if (bf16) { uint16_t hv = f2bf(v); memcpy(&out[off], &hv, 2); } else memcpy(&out[off], &v, 4);
                            }
}

// ------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------
struct ConvPlan {          // one launch of conv3x3_mfma_kernel
    std::string name;
    void* d_w = nullptr; float* d_bias = nullptr; float* d_rank1 = nullptr;
    // second structure (conv2.hip): A = conv1 + residual projection (10 taps per chunk), B = conv2 only
    void* d_w2 = nullptr; float* d_bias2 = nullptr; float* d_res_bias = nullptr;
    // third structure, "projection in B" (conv4.hip RP): A = conv1 alone (9 taps per chunk); B = conv2 + the block's 1x1 projection
    // of its own input, weights in MFMA A-operand order per 16-channel step, bias b2 + br
    void* d_w3 = nullptr; void* d_proj = nullptr; float* d_bias3 = nullptr;
    int Cout = 0, NT = 1, C0 = 0, C1 = 0, R0 = 0, R1 = 0, H = 0, W = 0;
    bool relu = true;
};

struct FileRec {
    int64_t off = 0;        // arena offset of the padded signal
    int64_t n = 0;          // samples at 22 050 Hz (unpadded)
    int64_t n_padded = 0;
    double duration = 0;    // header duration in seconds (frames / sample_rate)
    // results of the last ss_run
    int64_t W = 0, win_base = 0;
    int64_t bin_off = 0; int n_bins = 0;                  // this file's slice of ss_ctx::h_avg / h_cnt
    std::vector<ss_region> regions;
};

struct KStat { std::string name; int64_t launches = 0; double ms = 0, flops = 0, bytes = 0; };
struct PendingEvt { int sid; hipEvent_t a, b; };

struct ss_ctx {
    int device = 0;
    uint32_t flags = 0;
    bool bf16 = false, profile = false, has_model = false;
    hipStream_t stream = nullptr;
    std::string err;
    int chunk = 1024;                                      // most windows per pass of the network (bf16: ~20 GB of activations)
    int num_cus = 256, conv_version = 2;

    // tables + weights on device
    float4* d_pretw = nullptr; float2* d_w2048 = nullptr;
    int *d_mel_start = nullptr, *d_mel_count = nullptr, *d_mel_off = nullptr; float* d_mel_w = nullptr; float* d_mel_wp = nullptr; int mel_nw = 0;
    float *d_first_w = nullptr, *d_first_b = nullptr;
    float *d_flat_w = nullptr, *d_flat_b = nullptr; void* d_flat_frag = nullptr; void* d_flat_frag4 = nullptr;
    int flat_groups = 0;                                  // row groups the last FLAT launch wrote per window
    float *d_spec_w = nullptr, *d_spec_b = nullptr;
    Head1dWeights head{};
    std::vector<ConvPlan> convs;     // in launch order; pairs (A, B) per ResBlock, conv1_1 has only B
    std::vector<void*> owned;        // device allocations to free

    // bin masks of the last run, all files (covered by a window / average above the threshold; 64 bins per word): pinned, so that
    // ss_run_begin's copies are asynchronous.  The averages themselves stay on the device until ss_get_avg asks for them.
    unsigned long long *d_above = nullptr, *d_cov = nullptr, *h_above = nullptr, *h_cov = nullptr; size_t mask_cap = 0, cov_cap = 0, hmask_cap = 0;
    // The last ENDED run: its files' window / bin bookkeeping and its two masks (the pinned buffers swap places with h_above / h_cov
    // at ss_run_end), from which the regions are found when they are first asked for.  It stays readable while the next job is added
    // and in flight -- the host half of job k can run behind the device half of job k + 1 in ONE context.
    struct ResFile { int64_t W = 0, win_base = 0, bin_off = 0; int n_bins = 0; std::vector<ss_region> regions; };
    std::vector<ResFile> res_files; bool res_valid = false, res_regions = false; double res_thr = 0, res_brk = 0;
    unsigned long long *r_above = nullptr, *r_cov = nullptr; size_t rmask_cap = 0;
    uint64_t begin_gen = 0, res_gen = 0;               // avg / logits of the ended run live in device buffers the next ss_run_begin reuses
    std::vector<double> h_avg; std::vector<int32_t> h_cnt; bool avg_on_host = false; int64_t total_bins = 0;
    // a run between ss_run_begin and ss_run_end
    bool run_pending = false; double pend_thr = 0, pend_brk = 0; std::vector<struct AvgFile> pend_af;
    double t_in = 0, t_plan = 0, t_sync = 0, t_loop = 0;
    // activation workspace for `ws_chunk` windows
    int ws_chunk = 0;
    std::map<std::string, void*> act;
    float* d_feat = nullptr; float* d_flat = nullptr; float* d_flat_part = nullptr;

    // arena
    float* d_arena = nullptr; size_t arena_cap = 0, arena_used = 0;
    std::vector<FileRec> files;
    void* d_pcm = nullptr; size_t pcm_cap = 0;
    float* d_sx = nullptr; size_t sx_cap = 0;                    // review-screen spectrogram: samples in, magnitudes out
    float* d_sm = nullptr; size_t sm_cap = 0;
    short* d_sil_out = nullptr; size_t sil_out_cap = 0;          // silencer output / frame ranges
    int64_t* d_sil_ranges = nullptr; size_t sil_ranges_cap = 0;
    float* d_mono = nullptr; size_t mono_cap = 0;
    BatchFile* d_batch = nullptr; size_t batch_cap = 0;
    std::map<std::pair<int, int>, std::pair<float*, int>> taps;   // (sr_in) -> device taps, half

    // run state
    int64_t* d_winoff = nullptr; size_t winoff_cap = 0;
    float* d_logits = nullptr; size_t logits_cap = 0;
    float* d_spec = nullptr; size_t spec_cap = 0;
    double* d_avg = nullptr; int32_t* d_count = nullptr; size_t avg_cap = 0;
    int32_t* d_starts = nullptr; size_t starts_cap = 0;
    AvgFile* d_avgfiles = nullptr; size_t avgfiles_cap = 0;
    std::vector<float> h_logits; bool logits_valid = false; int64_t total_windows = 0;
    hipEvent_t ev_run0 = nullptr, ev_run1 = nullptr; double last_run_ms = 0;

    // profiling
    std::vector<KStat> stats; std::vector<PendingEvt> pending; std::vector<hipEvent_t> evpool;
};

static int fail(ss_ctx* c, int code, const std::string& msg) {
    g_err = msg;
    if (c) c->err = msg;
    return code;
}

static int stat_id(ss_ctx* c, const std::string& name) {
    for (size_t i = 0; i < c->stats.size(); ++i) if (c->stats[i].name == name) return (int)i;
    KStat k; k.name = name; c->stats.push_back(k);
    return (int)c->stats.size() - 1;
}

struct ScopedLaunch {      // times one launch with HIP events on the context's stream when profiling
    ss_ctx* c; int sid; hipEvent_t a = nullptr, b = nullptr;
    ScopedLaunch(ss_ctx* c_, const std::string& name, double flops, double bytes) : c(c_) {
        sid = stat_id(c, name);
        c->stats[sid].launches++; c->stats[sid].flops += flops; c->stats[sid].bytes += bytes;
        if (c->profile) {
            auto get = [&]() { hipEvent_t e; if (!c->evpool.empty()) { e = c->evpool.back(); c->evpool.pop_back(); } else hipEventCreate(&e); return e; };
            a = get(); b = get();
            hipEventRecord(a, c->stream);
        }
    }
    ~ScopedLaunch() {
        if (c->profile) { hipEventRecord(b, c->stream); c->pending.push_back({sid, a, b}); }
    }
};

static void resolve_events(ss_ctx* c) {
    for (auto& p : c->pending) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->stats[p.sid].ms += ms;
        c->evpool.push_back(p.a); c->evpool.push_back(p.b);
    }
    c->pending.clear();
}

template <typename T>
static int dev_upload(ss_ctx* c, T** dst, const void* src, size_t bytes) {
    void* p = nullptr;
    HIPCHK(c, hipMalloc(&p, bytes ? bytes : 16));
    c->owned.push_back(p);
    if (bytes) HIPCHK(c, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = (T*)p;
    return SS_OK;
}

template <typename T>
static int ensure(ss_ctx* c, T** p, size_t* cap, size_t need_elems, bool keep = false) {
    if (need_elems <= *cap && *p) return SS_OK;
    size_t ncap = std::max(need_elems, *cap + *cap / 2);
    void* np = nullptr;
    HIPCHK(c, hipMalloc(&np, std::max<size_t>(ncap * sizeof(T), 256)));
    if (keep && *p && *cap) {
        HIPCHK(c, hipMemcpyAsync(np, *p, *cap * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (*p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(*p)); }
    *p = (T*)np; *cap = ncap;
    return SS_OK;
}

// ------------------------------------------------------------------------------------------------------
// build: tables, folded + packed weights, launch plan
// ------------------------------------------------------------------------------------------------------
static int build_tables(ss_ctx* c, const Blob& bl) {
    std::string err;
    const double PI = 3.14159265358979323846;
    // window: the checkpoint's torchaudio buffer when present (SURVEY.md section 7 "Hard parts"), else periodic Hann
    std::vector<float> win(512);
    if (bl.has("mel_spectrogram.spectrogram.window")) {
        const float* w = bl.f32("mel_spectrogram.spectrogram.window", 512, err);
        if (!w) return fail(c, SS_ERR_FORMAT, err);
        memcpy(win.data(), w, 2048);
    } else {
        // torch.hann_window evaluates this in float32 (arange * float(2 pi / N), cos, * -0.5, + 0.5); this
        // emulation is within 1 float32 ulp of cos of torch's table (real checkpoints carry the buffer itself)
        for (int i = 0; i < 512; ++i) {
            const float ang = (float)i * (float)(2.0 * PI / 512.0);
            win[i] = (float)std::cos((double)ang) * -0.5f + 0.5f;
        }
    }
    std::vector<float2> w2048(2048);
    for (int j = 0; j < 2048; ++j) w2048[j] = make_float2((float)std::cos(2.0 * PI * j / 2048.0), (float)-std::sin(2.0 * PI * j / 2048.0));
    std::vector<float4> pretw(4 * 256);
    for (int r = 0; r < 4; ++r)
        for (int n = 0; n < 256; ++n) {
            const double ang = -2.0 * PI * (double)(n * r) / 1024.0;
            const double cr = std::cos(ang), si = std::sin(ang);
            const double w0 = win[2 * n], w1 = win[2 * n + 1];
            pretw[r * 256 + n] = make_float4((float)(w0 * cr), (float)(w1 * si), (float)(w0 * si), (float)(w1 * cr));
        }
    // mel filterbank: the checkpoint's `fb` buffer when present; else the torchaudio recipe in float32
    std::vector<float> fb((size_t)1025 * 128);
    if (bl.has("mel_spectrogram.mel_scale.fb")) {
        const float* f = bl.f32("mel_spectrogram.mel_scale.fb", (size_t)1025 * 128, err);
        if (!f) return fail(c, SS_ERR_FORMAT, err);
        memcpy(fb.data(), f, fb.size() * 4);
    } else {
        std::vector<float> all(1025), fpts(130);
        for (int i = 0; i < 1025; ++i) all[i] = (float)(11025.0 * i / 1024.0);
        const float mmin = 0.f, mmax = (float)(2595.0 * std::log10(1.0 + 8000.0 / 700.0));
        const float step = (mmax - mmin) / 129.0f;
        for (int i = 0; i < 130; ++i) {
            const float mp = i < 65 ? mmin + step * (float)i : mmax - step * (float)(129 - i);
            fpts[i] = 700.0f * (powf(10.0f, mp / 2595.0f) - 1.0f);
        }
        for (int k = 0; k < 1025; ++k)
            for (int j = 0; j < 128; ++j) {
                const float down = (-1.0f * (fpts[j] - all[k])) / (fpts[j + 1] - fpts[j]);
                const float up = (fpts[j + 2] - all[k]) / (fpts[j + 2] - fpts[j + 1]);
                fb[(size_t)k * 128 + j] = std::max(0.0f, std::min(down, up));
            }
    }
    std::vector<int> mstart(128), mcount(128), moff(128);
    std::vector<float> mw;
    for (int j = 0; j < 128; ++j) {
        int lo = -1, hi = -1;
        for (int k = 0; k < 1025; ++k) if (fb[(size_t)k * 128 + j] != 0.f) { if (lo < 0) lo = k; hi = k; }
        if (lo < 0) { lo = 0; hi = -1; }
        if (hi >= 768) return fail(c, SS_ERR_FORMAT, "mel filterbank has weight above bin 767 (front-end kernel computes bins 0..767)");
        mstart[j] = lo; mcount[j] = hi - lo + 1; moff[j] = (int)mw.size();
        for (int k = lo; k <= hi; ++k) mw.push_back(fb[(size_t)k * 128 + j]);
    }
    int rc;
    if ((rc = dev_upload(c, &c->d_pretw, pretw.data(), pretw.size() * sizeof(float4)))) return rc;
    if ((rc = dev_upload(c, &c->d_w2048, w2048.data(), w2048.size() * sizeof(float2)))) return rc;
    if ((rc = dev_upload(c, &c->d_mel_start, mstart.data(), 512))) return rc;
    if ((rc = dev_upload(c, &c->d_mel_count, mcount.data(), 512))) return rc;
    if ((rc = dev_upload(c, &c->d_mel_off, moff.data(), 512))) return rc;
    if (mw.size() > 1536) return fail(c, SS_ERR_FORMAT, "mel filterbank has more than 1536 non-zero weights");
    // the front-end kernel gives lane l filter l (up to kMelLo taps) and filter 127 - l (up to kMelHi taps), weights zero-padded
    // to those fixed trip counts so that its loop has no per-tap selects
    std::vector<float> mwp((size_t)64 * kMelPitch, 0.f);
    for (int l = 0; l < 64; ++l) {
        const int j1 = l, j2 = 127 - l;
        if (mcount[j1] > kMelLo || mcount[j2] > kMelHi)
            return fail(c, SS_ERR_FORMAT, "mel filterbank: a filter is wider than the front-end kernel's fixed trip counts (10 / 32 taps)");
        for (int b = 0; b < mcount[j1]; ++b) mwp[(size_t)l * kMelPitch + b] = mw[moff[j1] + b];
        for (int b = 0; b < mcount[j2]; ++b) mwp[(size_t)l * kMelPitch + kMelLo + b] = mw[moff[j2] + b];
    }
    if ((rc = dev_upload(c, &c->d_mel_wp, mwp.data(), mwp.size() * 4))) return rc;
    c->mel_nw = (int)mw.size();
    if ((rc = dev_upload(c, &c->d_mel_w, mw.data(), mw.size() * 4))) return rc;
    return SS_OK;
}

// 32-channel tiles per block; the 8x16 bottom level uses NT = 1 so that 4 x more blocks exist
static int pick_nt(int cout, int H) { return H <= 8 ? 1 : (cout == 96 ? 3 : (cout >= 64 ? 2 : 1)); }

// One ResBlock (pytorch_neural_nets.py:7-41) -> launch A (conv1+BN+ReLU) and launch B (conv2+BN + residual+BN, add, ReLU).
static int build_resblock(ss_ctx* c, const Blob& bl, const std::string& name, int cin0, int cin1, int cout, int H, int W) {
    std::string err;
    const int cin = cin0 + cin1;
    Folded f1, f2, fr;
    if (!fold_conv_bn(bl, name + ".conv1.0", name + ".conv1.1", cout, cin, 9, f1, err)) return fail(c, SS_ERR_FORMAT, err);
    if (!fold_conv_bn(bl, name + ".conv2.0", name + ".conv2.1", cout, cout, 9, f2, err)) return fail(c, SS_ERR_FORMAT, err);
    if (!fold_conv_bn(bl, name + ".residual.0", name + ".residual.1", cout, cin, 1, fr, err)) return fail(c, SS_ERR_FORMAT, err);
    const int NT = pick_nt(cout, H);
    int rc;
    std::vector<char> pk;
    std::vector<float> b2r(cout);
    for (int i = 0; i < cout; ++i) b2r[i] = f2.b[i] + fr.b[i];
    if (cin == 1) {
        // conv1_1: first conv is the VALU kernel, the residual is a rank-1 term of launch B
        std::vector<float> w9((size_t)9 * 32);
        for (int co = 0; co < 32; ++co) for (int t = 0; t < 9; ++t) w9[(size_t)t * 32 + co] = f1.w[(size_t)co * 9 + t];
        if ((rc = dev_upload(c, &c->d_first_w, w9.data(), w9.size() * 4))) return rc;
        if ((rc = dev_upload(c, &c->d_first_b, f1.b.data(), 128))) return rc;
        ConvPlan B; B.name = name + ".B"; B.Cout = cout; B.NT = NT; B.C0 = cout; B.H = H; B.W = W;
        pack_conv(&f2, nullptr, c->bf16, NT, pk);
        if ((rc = dev_upload(c, (char**)&B.d_w, pk.data(), pk.size()))) return rc;
        if ((rc = dev_upload(c, &B.d_bias, b2r.data(), cout * 4))) return rc;
        if ((rc = dev_upload(c, &B.d_rank1, fr.w.data(), cout * 4))) return rc;
        pack_conv_v2(f2, nullptr, c->bf16, NT, pk);
        if ((rc = dev_upload(c, (char**)&B.d_w2, pk.data(), pk.size()))) return rc;
        B.d_bias2 = B.d_bias;                         // b2 + br (the rank-1 residual has no separate tensor)
        c->convs.push_back(B);
        return SS_OK;
    }
    ConvPlan A; A.name = name + ".A"; A.Cout = cout; A.NT = NT; A.C0 = cin0; A.C1 = cin1; A.H = H; A.W = W;
    pack_conv(&f1, nullptr, c->bf16, NT, pk);
    if ((rc = dev_upload(c, (char**)&A.d_w, pk.data(), pk.size()))) return rc;
    if ((rc = dev_upload(c, &A.d_bias, f1.b.data(), cout * 4))) return rc;
    pack_conv_v2(f1, &fr, c->bf16, NT, pk);
    if ((rc = dev_upload(c, (char**)&A.d_w2, pk.data(), pk.size()))) return rc;
    A.d_bias2 = A.d_bias;
    if ((rc = dev_upload(c, &A.d_res_bias, fr.b.data(), cout * 4))) return rc;
    c->convs.push_back(A);
    if (c->bf16) {
        pack_conv_v2(f1, nullptr, true, NT, pk);
        if ((rc = dev_upload(c, (char**)&A.d_w3, pk.data(), pk.size()))) return rc;
        c->convs.back().d_w3 = A.d_w3;
    }
    ConvPlan B; B.name = name + ".B"; B.Cout = cout; B.NT = NT; B.C0 = cout; B.R0 = cin0; B.R1 = cin1; B.H = H; B.W = W;
    if (c->bf16 && cin % 16 == 0) {
        // [step][32-channel tile][lane][slot j]: row (output channel) = 32 tile + (lane & 31), input channel = 16 step + 8 (lane >> 5) + j
        const int steps = cin / 16, tiles = cout / 32;
        std::vector<uint16_t> pj((size_t)steps * tiles * 64 * 8);
        for (int st = 0; st < steps; ++st) for (int t = 0; t < tiles; ++t) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j)
            pj[(((size_t)st * tiles + t) * 64 + l) * 8 + j] = f2bf(fr.w[(size_t)(32 * t + (l & 31)) * cin + 16 * st + 8 * (l >> 5) + j]);
        if ((rc = dev_upload(c, (char**)&B.d_proj, (const char*)pj.data(), pj.size() * 2))) return rc;
        if ((rc = dev_upload(c, &B.d_bias3, b2r.data(), cout * 4))) return rc;
    }
    pack_conv(&f2, &fr, c->bf16, NT, pk);
    if ((rc = dev_upload(c, (char**)&B.d_w, pk.data(), pk.size()))) return rc;
    if ((rc = dev_upload(c, &B.d_bias, b2r.data(), cout * 4))) return rc;
    pack_conv_v2(f2, nullptr, c->bf16, NT, pk);
    if ((rc = dev_upload(c, (char**)&B.d_w2, pk.data(), pk.size()))) return rc;
    if ((rc = dev_upload(c, &B.d_bias2, f2.b.data(), cout * 4))) return rc;
    c->convs.push_back(B);
    return SS_OK;
}

static int build_model(ss_ctx* c, const Blob& bl) {
    int rc;
    // launch order == pytorch_neural_nets.py:156-181
    struct RB { const char* n; int c0, c1, co, H, W; };
    const RB rbs[] = {{"conv1_1", 1, 0, 32, 128, 256},  {"conv2_1", 32, 0, 64, 64, 128},      {"conv3_1", 64, 0, 96, 32, 64},
                      {"conv4_1", 96, 0, 128, 16, 32},  {"conv_bottleneck", 128, 0, 128, 8, 16}, {"encoder_out", 128, 0, 128, 8, 16},
                      {"conv6", 128, 128, 96, 16, 32},  {"conv7", 96, 96, 64, 32, 64},        {"conv8", 64, 64, 32, 64, 128},
                      {"conv9_1", 32, 32, 32, 128, 256}, {"spec_output_conv.0", 32, 0, 32, 128, 256}};
    for (const RB& r : rbs)
        if ((rc = build_resblock(c, bl, r.n, r.c0, r.c1, r.co, r.H, r.W))) return rc;
    std::string err;
    // conv_flatten (pytorch_neural_nets.py:133): weight (4, 32, 128, 1) -> [h][ci][c]
    const float* wf = bl.f32("conv_flatten.weight", 4 * 32 * 128, err);
    const float* bf = bl.f32("conv_flatten.bias", 4, err);
    if (!wf || !bf) return fail(c, SS_ERR_FORMAT, err);
    std::vector<float> wfl((size_t)128 * 32 * 4);
    for (int co = 0; co < 4; ++co) for (int ci = 0; ci < 32; ++ci) for (int h = 0; h < 128; ++h)
        wfl[((size_t)h * 32 + ci) * 4 + co] = wf[((size_t)co * 32 + ci) * 128 + h];
    if ((rc = dev_upload(c, &c->d_flat_w, wfl.data(), wfl.size() * 4))) return rc;
    {   // fused flatten (conv2.hip FLAT): per mel row h a 32 -> 4 (padded to 32) 1x1 "conv" in MFMA fragment order
        std::vector<char> all, one;
        for (int h = 0; h < 128; ++h) {
            Folded fr; fr.cout = 32; fr.cin = 32; fr.k = 1; fr.w.assign(32 * 32, 0.f); fr.b.assign(32, 0.f);
            for (int co = 0; co < 4; ++co) for (int ci = 0; ci < 32; ++ci) fr.w[(size_t)co * 32 + ci] = wf[((size_t)co * 32 + ci) * 128 + h];
            pack_conv(nullptr, &fr, c->bf16, 1, one);
            all.insert(all.end(), one.begin(), one.end());
        }
        if ((rc = dev_upload(c, (char**)&c->d_flat_frag, all.data(), all.size()))) return rc;
    }
    if (c->bf16) {   // conv4.hip FLAT: [mel row][step s][lane][slot j] -> channel 16 s + 8 (j >> 2) + 4 (lane >> 5) + (j & 3), row = lane & 31
        std::vector<uint16_t> t4((size_t)128 * 2 * 64 * 8, 0);
        for (int h = 0; h < 128; ++h) for (int s2 = 0; s2 < 2; ++s2) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
            const int co = l & 31, ch = 16 * s2 + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
            if (co < 4) t4[(((size_t)h * 2 + s2) * 64 + l) * 8 + j] = f2bf(wf[((size_t)co * 32 + ch) * 128 + h]);
        }
        if ((rc = dev_upload(c, (char**)&c->d_flat_frag4, (const char*)t4.data(), t4.size() * 2))) return rc;
    }
    if ((rc = dev_upload(c, &c->d_flat_b, bf, 16))) return rc;
    // spec_output_conv.1 (pytorch_neural_nets.py:128): Conv2d(32, 2, 1) with bias
    const float* ws = bl.f32("spec_output_conv.1.weight", 64, err);
    const float* bs = bl.f32("spec_output_conv.1.bias", 2, err);
    if (!ws || !bs) return fail(c, SS_ERR_FORMAT, err);
    if ((rc = dev_upload(c, &c->d_spec_w, ws, 256))) return rc;
    if ((rc = dev_upload(c, &c->d_spec_b, bs, 8))) return rc;
    // mask_output_conv (pytorch_neural_nets.py:137-140): ResBlock1D(4,4) + Conv1d(4,1,1)
    Folded f1, f2, fr;
    const std::string p = "mask_output_conv.0";
    if (!fold_conv_bn(bl, p + ".conv1.0", p + ".conv1.1", 4, 4, 3, f1, err)) return fail(c, SS_ERR_FORMAT, err);
    if (!fold_conv_bn(bl, p + ".conv2.0", p + ".conv2.1", 4, 4, 3, f2, err)) return fail(c, SS_ERR_FORMAT, err);
    if (!fold_conv_bn(bl, p + ".residual.0", p + ".residual.1", 4, 4, 1, fr, err)) return fail(c, SS_ERR_FORMAT, err);
    const float* wo = bl.f32("mask_output_conv.1.weight", 4, err);
    const float* bo = bl.f32("mask_output_conv.1.bias", 1, err);
    if (!wo || !bo) return fail(c, SS_ERR_FORMAT, err);
    for (int co = 0; co < 4; ++co) {
        for (int ci = 0; ci < 4; ++ci) {
            for (int k = 0; k < 3; ++k) {
                c->head.w1[co][ci][k] = f1.w[((size_t)co * 4 + ci) * 3 + k];
                c->head.w2[co][ci][k] = f2.w[((size_t)co * 4 + ci) * 3 + k];
            }
            c->head.wr[co][ci] = fr.w[(size_t)co * 4 + ci];
        }
        c->head.b1[co] = f1.b[co];
        c->head.b2r[co] = f2.b[co] + fr.b[co];
        c->head.wo[co] = wo[co];
    }
    c->head.bo = bo[0];
    return SS_OK;
}

// activation workspace: NHWC tensors for `n` windows
static constexpr size_t kActHeader = 256;
static int ensure_workspace(ss_ctx* c, int n) {
    if (n <= c->ws_chunk) return SS_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto& kv : c->act) hipFree((char*)kv.second - kActHeader);
    c->act.clear();
    if (c->d_feat) hipFree(c->d_feat);
    if (c->d_flat) hipFree(c->d_flat);
    if (c->d_flat_part) hipFree(c->d_flat_part);
    const size_t es = c->bf16 ? 2 : 4;
    struct T { const char* n; int H, W, C; };
    const T ts[] = {{"h1", 128, 256, 32}, {"c1", 128, 256, 32}, {"p1", 64, 128, 32}, {"h2", 64, 128, 64}, {"c2", 64, 128, 64},
                    {"p2", 32, 64, 64},   {"h3", 32, 64, 96},   {"c3", 32, 64, 96},  {"p3", 16, 32, 96},  {"h4", 16, 32, 128},
                    {"c4", 16, 32, 128},  {"p4", 8, 16, 128},   {"hb", 8, 16, 128},  {"bott", 8, 16, 128}, {"he", 8, 16, 128},
                    {"enc", 8, 16, 128},  {"h6", 16, 32, 96},   {"c6", 16, 32, 96},  {"h7", 32, 64, 64},  {"c7", 32, 64, 64},
                    {"h8", 64, 128, 32},  {"c8", 64, 128, 32},  {"h9", 128, 256, 32}, {"c9", 128, 256, 32},
                    {"hs", 128, 256, 32}, {"s9", 128, 256, 32},
                    // r = residual projection written by A launches of the second structure
                    {"r2", 64, 128, 64},  {"r3", 32, 64, 96},   {"r4", 16, 32, 128}, {"rb", 8, 16, 128},  {"re", 8, 16, 128},
                    {"r6", 16, 32, 96},   {"r7", 32, 64, 64},   {"r8", 64, 128, 32}, {"r9", 128, 256, 32}, {"rs", 128, 256, 32}};
    for (const T& t : ts) {        // [kActHeader zero bytes][tensor]: conv4.hip reads the header for out-of-image patch pieces
        void* p = nullptr;
        HIPCHK(c, hipMalloc(&p, kActHeader + (size_t)n * t.H * t.W * t.C * es));
        HIPCHK(c, hipMemsetAsync(p, 0, kActHeader, c->stream));
        c->act[t.n] = (char*)p + kActHeader;
    }
    HIPCHK(c, hipMalloc((void**)&c->d_feat, (size_t)n * 128 * 256 * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_flat, (size_t)n * 4 * 256 * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_flat_part, (size_t)n * 64 * 4 * 256 * 4));
    c->ws_chunk = n;
    return SS_OK;
}

static const char* conv_kernel_name(bool bf16, int NT) {     // first structure, as rocprofv3 prints the instantiation
    static const char* names[2][3] = {{"conv3x3_mfma_kernel<false, 1>", "conv3x3_mfma_kernel<false, 2>", "conv3x3_mfma_kernel<false, 3>"},
                                      {"conv3x3_mfma_kernel<true, 1>", "conv3x3_mfma_kernel<true, 2>", "conv3x3_mfma_kernel<true, 3>"}};
    return names[bf16 ? 1 : 0][NT - 1];
}

struct ConvExtra { const float* first_w = nullptr; const float* first_b = nullptr; const void* flat_w = nullptr; const void* flat_w4 = nullptr; float* flat_part = nullptr; int store_out = 1; };

// One launch of the first structure (conv.hip).
static int run_conv(ss_ctx* c, const ConvPlan& p, int n, const void* s0, const void* s1, const void* r0, const void* r1,
                    const float* rank1_src, void* out, void* pool) {
    ConvArgs a{};
    a.src0 = s0; a.src1 = s1; a.res0 = r0; a.res1 = r1; a.wpk = p.d_w; a.bias = p.d_bias;
    a.rank1_src = rank1_src; a.rank1_w = p.d_rank1; a.out = out; a.pool_out = pool;
    a.N = n; a.H = p.H; a.W = p.W; a.C0 = p.C0; a.C1 = p.C1; a.R0 = p.R0; a.R1 = p.R1; a.Cout = p.Cout; a.relu = p.relu ? 1 : 0;
    a.tiles_y = (p.H + 15) / 16; a.tiles_x = p.W / 16;
    { static const int dbg = getenv("SOFTSPOKEN_DBG") ? atoi(getenv("SOFTSPOKEN_DBG")) : 0; a.dbg = dbg; }
    const double macs = (double)n * p.H * p.W * p.Cout * (9.0 * (p.C0 + p.C1) + (p.R0 + p.R1) + (rank1_src ? 1 : 0));
    const double es = c->bf16 ? 2 : 4;
    const double bytes = (double)n * p.H * p.W * (es * (p.C0 + p.C1 / 4.0 + p.R0 + p.R1 / 4.0 + p.Cout + (pool ? p.Cout / 4.0 : 0)));
    // stat name = "<kernel>/<layer>": bench.py groups by the part before '/'
    ScopedLaunch sl(c, std::string(conv_kernel_name(c->bf16, p.NT)) + "/" + p.name, 2.0 * macs, bytes);
    HIPCHK(c, launch_conv3x3(a, c->bf16, p.NT, c->stream));
    return SS_OK;
}

// One launch of the second structure.  A launches (r_out) compute h and the residual projection r from the block input
// (x0 [+ upsampled x1]); B launches (r_in) compute the block output from h and add r.
static int run_conv2(ss_ctx* c, const ConvPlan& p, int n, const void* x0, const void* x1, void* out, void* pool, void* r_out,
                     const void* r_in, const float* feat, const ConvExtra& ex = ConvExtra()) {
    const bool isA = r_out != nullptr;
    ConvArgs a{};
    a.first_w = ex.first_w; a.first_b = ex.first_b; a.flat_w = ex.flat_w; a.flat_w4 = ex.flat_w4; a.flat_part = ex.flat_part; a.store_out = ex.store_out;
    a.src0 = x0; a.src1 = x1; a.wpk = p.d_w2; a.bias = p.d_bias2;
    a.res_out = r_out; a.res_bias = p.d_res_bias; a.res_in = r_in;
    a.rank1_src = feat; a.rank1_w = p.d_rank1; a.out = out; a.pool_out = pool;
    a.N = n; a.H = p.H; a.W = p.W; a.Cout = p.Cout; a.relu = 1;
    if (isA) { a.C0 = p.C0; a.C1 = p.C1; } else { a.C0 = p.Cout; a.C1 = 0; }       // B's 3x3 input is h
    { static const int dbg = getenv("SOFTSPOKEN_DBG") ? atoi(getenv("SOFTSPOKEN_DBG")) : 0; a.dbg = dbg; }
    const double cin = a.C0 + a.C1;
    const double macs = (double)n * p.H * p.W * p.Cout * (9.0 * cin + (isA ? cin : 0.0) + (feat ? 1 : 0)) +
                        (ex.first_w ? (double)n * p.H * p.W * 32 * 9 : 0.0) + (ex.flat_part ? (double)n * p.H * p.W * 32 * 4 : 0.0);
    const double es = c->bf16 ? 2 : 4;
    // (the fused conv1_1 launch reads the fp32 features, not an h1 tensor: h1 only exists in LDS)
    const double bytes = (double)n * p.H * p.W * es * ((ex.first_w ? 4.0 / es : a.C0) + a.C1 / 4.0 + (ex.flat_part && !ex.store_out ? 0 : p.Cout) + (isA || r_in ? p.Cout : 0) + (pool ? p.Cout / 4.0 : 0));
    // stat name = "<instantiation as rocprofv3 prints it>/<layer>"
    static const int v4_env = getenv("SOFTSPOKEN_CONV4") ? atoi(getenv("SOFTSPOKEN_CONV4")) : 1;
    static const int prio_env = getenv("SOFTSPOKEN_PRIO") ? atoi(getenv("SOFTSPOKEN_PRIO")) : 1;
    if (prio_env) a.dbg |= 32;                            // conv4.hip: raised wave priority inside the MFMA loop
    if (c->bf16 && v4_env && conv_v4_supports(a, p.NT, c->num_cus)) {      // third structure (conv4.hip): bf16 ResBlock launches
        ScopedLaunch sl(c, std::string(conv_v4_variant(a, p.NT, c->num_cus)) + "/" + p.name, 2.0 * macs, bytes);
        HIPCHK(c, launch_conv3x3_v4(a, p.NT, c->num_cus, c->stream));
        if (ex.flat_part) c->flat_groups = conv_v4_flat_groups();
        return SS_OK;
    }
    if (ex.flat_part) c->flat_groups = conv_v2_flat_groups(c->bf16);
    ScopedLaunch sl(c, std::string(conv_v2_variant(a, c->bf16, p.NT, c->num_cus)) + "/" + p.name, 2.0 * macs, bytes);
    HIPCHK(c, launch_conv3x3_v2(a, c->bf16, p.NT, c->num_cus, c->stream));
    return SS_OK;
}

// A ResBlock in the "projection in B" form of the third structure (conv4.hip RP): A writes h alone, B reads h and the centre
// pixels of the block input.  Returns 1 when conv4.hip has no instantiation for this block (the caller then uses A + r / B).
static int run_block_proj(ss_ctx* c, const ConvPlan& pa, const ConvPlan& pb, int n, const void* x0, const void* x1, void* h, void* out,
                          void* pool, const ConvExtra& ex = ConvExtra()) {
    if (!c->bf16 || !pa.d_w3 || !pb.d_proj) return 1;
    ConvArgs a{}, b{};
    a.src0 = x0; a.src1 = x1; a.wpk = pa.d_w3; a.bias = pa.d_bias2; a.out = h; a.plain = 1;
    a.N = n; a.H = pa.H; a.W = pa.W; a.Cout = pa.Cout; a.C0 = pa.C0; a.C1 = pa.C1; a.relu = 1;
    b.src0 = h; b.wpk = pb.d_w2; b.bias = pb.d_bias3; b.out = out; b.pool_out = pool;
    b.N = n; b.H = pb.H; b.W = pb.W; b.Cout = pb.Cout; b.C0 = pb.Cout; b.C1 = 0; b.relu = 1;
    b.proj_w = pb.d_proj; b.xp0 = x0; b.xp1 = x1; b.C0x = pa.C0; b.C1x = pa.C1;
    b.flat_w4 = ex.flat_w4; b.flat_part = ex.flat_part; b.store_out = ex.store_out;
    static const int prio_env = getenv("SOFTSPOKEN_PRIO") ? atoi(getenv("SOFTSPOKEN_PRIO")) : 1;
    if (prio_env) { a.dbg |= 32; b.dbg |= 32; }
    if (!conv_v4_supports(a, pa.NT, c->num_cus) || !conv_v4_supports(b, pb.NT, c->num_cus)) return 1;
    const double px = (double)n * pa.H * pa.W, cin = pa.C0 + pa.C1, cinb = pa.C0 + pa.C1 / 4.0;
    {
        ScopedLaunch sl(c, std::string(conv_v4_variant(a, pa.NT, c->num_cus)) + "/" + pa.name, 2.0 * px * pa.Cout * 9.0 * cin, px * 2.0 * (cinb + pa.Cout));
        HIPCHK(c, launch_conv3x3_v4(a, pa.NT, c->num_cus, c->stream));
    }
    {
        const double flops = 2.0 * px * pb.Cout * (9.0 * pb.Cout + cin) + (ex.flat_part ? 2.0 * px * 32 * 4 : 0.0);
        const double bytes = px * 2.0 * (pb.Cout + cinb + (ex.flat_part && !ex.store_out ? 0 : pb.Cout) + (pool ? pb.Cout / 4.0 : 0));
        ScopedLaunch sl(c, std::string(conv_v4_variant(b, pb.NT, c->num_cus)) + "/" + pb.name, flops, bytes);
        HIPCHK(c, launch_conv3x3_v4(b, pb.NT, c->num_cus, c->stream));
        if (ex.flat_part) c->flat_groups = conv_v4_flat_groups();
    }
    return SS_OK;
}

// A whole ResBlock with 32 output channels in one launch (conv3.hip, bf16): pa / pb are the block's A and B plans.
static int run_fused32(ss_ctx* c, const ConvPlan& pa, const ConvPlan& pb, int n, const void* x0, const void* x1, void* out,
                       const ConvExtra& ex = ConvExtra()) {
    ConvArgs a{};
    a.flat_w = ex.flat_w; a.flat_part = ex.flat_part; a.store_out = ex.store_out;
    a.src0 = x0; a.src1 = x1; a.C0 = pa.C0; a.C1 = pa.C1; a.Cout = 32; a.N = n; a.H = pa.H; a.W = pa.W; a.relu = 1;
    a.wpk = pa.d_w2; a.wpk_b = pb.d_w2; a.bias_a = pa.d_bias2; a.bias = pb.d_bias;      // b1 ; b2 + br
    a.out = out;
    const double cin = pa.C0 + pa.C1;
    const double macs = (double)n * pa.H * pa.W * 32.0 * (9.0 * cin + cin + 9.0 * 32) + (ex.flat_part ? (double)n * pa.H * pa.W * 32 * 4 : 0.0);
    const double bytes = (double)n * pa.H * pa.W * 2.0 * (pa.C0 + pa.C1 / 4.0 + (ex.flat_part && !ex.store_out ? 0 : 32));
    ScopedLaunch sl(c, std::string("resblock32_fused_kernel/") + pa.name.substr(0, pa.name.size() - 2), 2.0 * macs, bytes);
    HIPCHK(c, launch_resblock32_fused(a, c->num_cus, c->stream));
    return SS_OK;
}

// SpecUNet_2D.forward (pytorch_neural_nets.py:142-197) for n <= ws_chunk windows whose arena offsets are d_winoff[0..n)
static int forward_chunk(ss_ctx* c, const int64_t* d_winoff, int n, float* d_logits, float* d_spec, float* d_feat_out) {
    FrontendTables tb{c->d_pretw, c->d_w2048, c->d_mel_start, c->d_mel_count, c->d_mel_off, c->d_mel_w, c->mel_nw, c->d_mel_wp, 0};
    float* feat = d_feat_out ? d_feat_out : c->d_feat;
    {
        ScopedLaunch sl(c, "frontend", 0.0, (double)n * (66150.0 * 4 + 128.0 * 256 * 4));
        HIPCHK(c, launch_frontend(c->d_arena, d_winoff, n, tb, feat, c->num_cus, c->stream));
    }
    if (!d_logits) return SS_OK;
    auto A = [&](const char* k) { return c->act[k]; };
    const double es = c->bf16 ? 2 : 4;
    if (c->conv_version == 2) {
        const std::vector<ConvPlan>& cv = c->convs;
        int rc, i = 0;
#define RC2(x) if ((rc = (x))) return rc
        {   // conv1_1: first conv produced in the loader, 1 -> 32 residual from the staged features
            ConvExtra ex; ex.first_w = c->d_first_w; ex.first_b = c->d_first_b;
            RC2(run_conv2(c, cv[i++], n, nullptr, nullptr, A("c1"), A("p1"), nullptr, nullptr, feat, ex));
        }
        struct Blk { const char *x0, *x1, *h, *r, *y, *pool; };
        const Blk blks[] = {{"p1", nullptr, "h2", "r2", "c2", "p2"},   {"p2", nullptr, "h3", "r3", "c3", "p3"},
                            {"p3", nullptr, "h4", "r4", "c4", "p4"},   {"p4", nullptr, "hb", "rb", "bott", nullptr},
                            {"bott", nullptr, "he", "re", "enc", nullptr}, {"c4", "enc", "h6", "r6", "c6", nullptr},
                            {"c3", "c6", "h7", "r7", "c7", nullptr},   {"c2", "c7", "h8", "r8", "c8", nullptr}};
        static const int fuse_env = getenv("SOFTSPOKEN_FUSE") ? atoi(getenv("SOFTSPOKEN_FUSE")) : 0;
        static const int v4_on = getenv("SOFTSPOKEN_CONV4") ? atoi(getenv("SOFTSPOKEN_CONV4")) : 1;
        static const int proj_env = v4_on && (getenv("SOFTSPOKEN_RPROJ") ? atoi(getenv("SOFTSPOKEN_RPROJ")) : 1);
        const bool fuse32 = c->bf16 && fuse_env && conv_v2_flat_groups(true) == 64;   // conv3.hip: bf16, 8-wave row groups
        for (const Blk& b : blks) {
            if (fuse32 && cv[i].Cout == 32 && cv[i].H % 16 == 0) {
                RC2(run_fused32(c, cv[i], cv[i + 1], n, A(b.x0), b.x1 ? A(b.x1) : nullptr, A(b.y)));
                i += 2;
                continue;
            }
            // (running A and B over Infinity-Cache-sized sub-chunks of windows was measured twice: no gain)
            // Blocks whose input is narrower than their output (encoder) move fewer bytes when B recomputes the 1x1 projection from
            // the block input than when A writes r and B reads it back; conv4.hip has that form for the blocks where it pays.
            if (proj_env) {
                rc = run_block_proj(c, cv[i], cv[i + 1], n, A(b.x0), b.x1 ? A(b.x1) : nullptr, A(b.h), A(b.y), b.pool ? A(b.pool) : nullptr);
                if (rc == SS_OK) { i += 2; continue; }
                if (rc != 1) return rc;
            }
            RC2(run_conv2(c, cv[i], n, A(b.x0), b.x1 ? A(b.x1) : nullptr, A(b.h), nullptr, A(b.r), nullptr, nullptr));
            RC2(run_conv2(c, cv[i + 1], n, A(b.h), nullptr, A(b.y), b.pool ? A(b.pool) : nullptr, nullptr, A(b.r), nullptr));
            i += 2;
        }
        {   // conv9_1 on cat[conv1, up(conv8)]; conv_flatten rides in B's epilogue (c9 itself only when the spec head runs)
            ConvExtra ex; ex.flat_w = c->d_flat_frag; ex.flat_w4 = c->d_flat_frag4; ex.flat_part = c->d_flat_part; ex.store_out = d_spec ? 1 : 0;
            rc = 1;
            if (fuse32) {
                RC2(run_fused32(c, cv[i], cv[i + 1], n, A("c1"), A("c8"), A("c9"), ex));
                c->flat_groups = conv_v2_flat_groups(true);
            } else if (proj_env && (rc = run_block_proj(c, cv[i], cv[i + 1], n, A("c1"), A("c8"), A("h9"), A("c9"), nullptr, ex)) != 1) {
                if (rc) return rc;
            } else {
                RC2(run_conv2(c, cv[i], n, A("c1"), A("c8"), A("h9"), nullptr, A("r9"), nullptr, nullptr));
                RC2(run_conv2(c, cv[i + 1], n, A("h9"), nullptr, A("c9"), nullptr, nullptr, A("r9"), nullptr, ex));
            }
            i += 2;
        }
        if (d_spec) {   // dead head of the reference, on request
            if (fuse32) {
                RC2(run_fused32(c, cv[i], cv[i + 1], n, A("c9"), nullptr, A("s9")));
            } else {
                RC2(run_conv2(c, cv[i], n, A("c9"), nullptr, A("hs"), nullptr, A("rs"), nullptr, nullptr));
                RC2(run_conv2(c, cv[i + 1], n, A("hs"), nullptr, A("s9"), nullptr, nullptr, A("rs"), nullptr));
            }
            ScopedLaunch sl(c, "spec_tail", 2.0 * n * 32768 * 64, (double)n * 32768 * (32 * es + 8));
            HIPCHK(c, launch_spec_tail(A("s9"), c->d_spec_w, c->d_spec_b, d_spec, n, c->bf16, c->stream));
        }
#undef RC2
        const int groups = c->flat_groups;
        ScopedLaunch sl(c, "mask_head_parts", 0.0, (double)n * (groups * 4 * 256 * 4 + 1024));
        HIPCHK(c, launch_mask_head_parts(c->d_flat_part, groups, c->d_flat_b, c->head, d_logits, n, c->stream));
        return SS_OK;
    }
    // ---- first structure (conv.hip): conv_first, then A / B launches per ResBlock with the 1x1 residual as extra K in B ----
    {
        ScopedLaunch sl(c, "conv_first", 2.0 * n * 128 * 256 * 32 * 9, (double)n * 32768 * (4 + 32 * es));
        HIPCHK(c, launch_conv_first(feat, c->d_first_w, c->d_first_b, A("h1"), n, 128, 256, c->bf16, c->stream));
    }
    int rc, i = 0;
    const std::vector<ConvPlan>& cv = c->convs;
#define RC(x) if ((rc = (x))) return rc
    RC(run_conv(c, cv[i++], n, A("h1"), nullptr, nullptr, nullptr, feat, A("c1"), A("p1")));            // conv1_1.B
    struct Blk1 { const char *x0, *x1, *h, *y, *pool; };
    const Blk1 blks1[] = {{"p1", nullptr, "h2", "c2", "p2"},     {"p2", nullptr, "h3", "c3", "p3"},   {"p3", nullptr, "h4", "c4", "p4"},
                          {"p4", nullptr, "hb", "bott", nullptr}, {"bott", nullptr, "he", "enc", nullptr}, {"c4", "enc", "h6", "c6", nullptr},
                          {"c3", "c6", "h7", "c7", nullptr},      {"c2", "c7", "h8", "c8", nullptr},   {"c1", "c8", "h9", "c9", nullptr}};
    for (const Blk1& b : blks1) {
        const void* x1 = b.x1 ? A(b.x1) : nullptr;
        RC(run_conv(c, cv[i], n, A(b.x0), x1, nullptr, nullptr, nullptr, A(b.h), nullptr));
        RC(run_conv(c, cv[i + 1], n, A(b.h), nullptr, A(b.x0), x1, nullptr, A(b.y), b.pool ? A(b.pool) : nullptr));
        i += 2;
    }
    if (d_spec) {                                                                                       // dead head of the reference, on request
        RC(run_conv(c, cv[i], n, A("c9"), nullptr, nullptr, nullptr, nullptr, A("hs"), nullptr));
        RC(run_conv(c, cv[i + 1], n, A("hs"), nullptr, A("c9"), nullptr, nullptr, A("s9"), nullptr));
        ScopedLaunch sl(c, "spec_tail", 2.0 * n * 32768 * 64, (double)n * 32768 * (32 * es + 8));
        HIPCHK(c, launch_spec_tail(A("s9"), c->d_spec_w, c->d_spec_b, d_spec, n, c->bf16, c->stream));
    }
#undef RC
    {
        ScopedLaunch sl(c, "flatten", 2.0 * n * 256 * 4096 * 4, (double)n * 32768 * 32 * es);
        HIPCHK(c, launch_flatten(A("c9"), c->d_flat_w, c->d_flat_b, c->d_flat, n, c->bf16, c->stream));
    }
    {
        ScopedLaunch sl(c, "mask_head", 0.0, (double)n * 5 * 1024);
        HIPCHK(c, launch_mask_head(c->d_flat, c->head, d_logits, n, c->stream));
    }
    return SS_OK;
}

// ------------------------------------------------------------------------------------------------------
// host-only helpers
// ------------------------------------------------------------------------------------------------------
extern "C" int ss_abi_version(void) { return SS_ABI_VERSION; }

extern "C" const char* ss_last_error(const ss_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

// voice_activity.py:23-30 (get_audio_data) needs duration + native rate; load_audio needs the samples.
extern "C" int ss_wav_parse(const void* file_bytes, size_t nbytes, ss_wav_info* out) {
    if (!file_bytes || !out) return fail(nullptr, SS_ERR_ARG, "ss_wav_parse: null argument");
    const unsigned char* b = (const unsigned char*)file_bytes;
    if (nbytes < 12 || memcmp(b, "RIFF", 4) != 0 || memcmp(b + 8, "WAVE", 4) != 0) return fail(nullptr, SS_ERR_FORMAT, "not a RIFF/WAVE file");
    size_t pos = 12;
    bool have_fmt = false;
    uint16_t tag = 0, ch = 0, bits = 0; uint32_t sr = 0;
    while (pos + 8 <= nbytes) {
        uint32_t sz; memcpy(&sz, b + pos + 4, 4);
        const size_t body = pos + 8;
        if (memcmp(b + pos, "fmt ", 4) == 0) {
            if (sz < 16 || body + 16 > nbytes) return fail(nullptr, SS_ERR_FORMAT, "WAV: short fmt chunk");
            memcpy(&tag, b + body, 2); memcpy(&ch, b + body + 2, 2); memcpy(&sr, b + body + 4, 4); memcpy(&bits, b + body + 14, 2);
            if (tag == 0xFFFE && sz >= 26 && body + 26 <= nbytes) memcpy(&tag, b + body + 24, 2);   // WAVE_FORMAT_EXTENSIBLE sub-format
            have_fmt = true;
        } else if (memcmp(b + pos, "data", 4) == 0) {
            if (!have_fmt) return fail(nullptr, SS_ERR_FORMAT, "WAV: data chunk before fmt chunk");
            int fmt = 0;
            if (tag == 1 && bits == 8) fmt = SS_PCM_U8;
            else if (tag == 1 && bits == 16) fmt = SS_PCM_S16;
            else if (tag == 1 && bits == 24) fmt = SS_PCM_S24;
            else if (tag == 1 && bits == 32) fmt = SS_PCM_S32;
            else if (tag == 3 && bits == 32) fmt = SS_PCM_F32;
            else if (tag == 3 && bits == 64) fmt = SS_PCM_F64;
            else return fail(nullptr, SS_ERR_FORMAT, "WAV: unsupported encoding (tag " + std::to_string(tag) + ", " + std::to_string(bits) + " bits)");
            if (ch == 0 || sr == 0) return fail(nullptr, SS_ERR_FORMAT, "WAV: zero channels or sample rate");
            if (sr > 0x7fffffffu) return fail(nullptr, SS_ERR_FORMAT, "WAV: sample rate out of range");     // (found by the header fuzz test)
            const size_t avail = std::min<size_t>(sz, nbytes - body);
            out->format = fmt; out->channels = ch; out->sample_rate = (int32_t)sr; out->bits = bits;
            out->data_offset = (int64_t)body; out->data_bytes = (int64_t)avail;
            out->frames = (int64_t)(avail / ((size_t)ch * bits / 8));
            return SS_OK;
        }
        pos = body + sz + (sz & 1);
    }
    return fail(nullptr, SS_ERR_FORMAT, "WAV: missing fmt or data chunk");
}

extern "C" int64_t ss_resampled_length(int64_t frames, int sample_rate) {
    if (sample_rate <= 0 || frames < 0) return -1;
    if (sample_rate == SS_SAMPLE_RATE) return frames;
    return (frames * SS_SAMPLE_RATE + sample_rate - 1) / sample_rate;
}

// NNDetector.py:66-80
extern "C" int64_t ss_plan_windows(double duration_s, int64_t* starts, int64_t cap) {
    const double L = std::nearbyint(duration_s * 22050.0) + 6.0 * 22050.0;
    int64_t W = (int64_t)std::ceil((L - 66150.0) / 13230.0);
    if (W < 0) W = 0;
    if (starts) for (int64_t i = 0; i < W && i < cap; ++i) starts[i] = i * SS_STEP_SAMPLES;
    return W;
}

static double bin_time(int64_t idx) {    // float(f"{idx / (256 / 3):.4f}")  (NNDetector.py:185, worker.py:100)
    // idx * 3 / 256 = idx * 1171875 / 1e8 exactly; the double the reference formats is within 1e-12 of it, so unless the
    // exact value is a tie at the 4th decimal the rounding is decided by integers, and q / 1e4 in double is what strtod of
    // "q.dddd" returns (both correctly rounded).  A run boundary cost two printf + strtod pairs: 1.2 ms per 256-file job.
    if (idx >= 0 && idx < ((int64_t)1 << 40)) {
        const int64_t N = idx * 1171875;
        int64_t q = N / 10000;
        const int64_t rem = N % 10000;
        if (rem != 5000) return (double)(q + (rem > 5000 ? 1 : 0)) / 10000.0;
        // a tie in exact arithmetic (every 16th index): what is formatted is the DOUBLE d = idx / (256 / 3), which lies a hair to one
        // side of the tie (q + 1/2) / 1e4 -- decide the side exactly: d = m 2^e, compare m * 20000 * 2^e with 2 q + 1 in 128-bit integers
        const double d = (double)idx / (256.0 / 3.0);
        int e; const double fr = std::frexp(d, &e);                  // d = fr * 2^e, 0.5 <= fr < 1
        const __int128 m = (__int128)std::ldexp(fr, 53); e -= 53;    // d = m * 2^e, m < 2^53 (exact)
        if (e <= 0 && e > -100) {
            const __int128 lhs = m * 20000, rhs = (__int128)(2 * q + 1) << (-e);
            if (lhs != rhs) return (double)(q + (lhs > rhs ? 1 : 0)) / 10000.0;
            return (double)(q + (q & 1)) / 10000.0;                  // the double IS the tie: round half to even, as the formatter does
        }
    }
    char buf[64];
    snprintf(buf, sizeof buf, "%.4f", (double)idx / (256.0 / 3.0));
    return strtod(buf, nullptr);
}

// NNDetector.py:112-141 then worker.py:100
extern "C" int ss_find_regions(const double* avg, const int64_t* bin_idx, int64_t n, double threshold, double break_s,
                               ss_region* out, int64_t cap, int64_t* n_out) {
    if ((n > 0 && (!avg || !bin_idx)) || !n_out) return fail(nullptr, SS_ERR_ARG, "ss_find_regions: null argument");
    // a run's start/end are the time strings of its first/last bin: format only at run boundaries
    std::vector<std::pair<double, double>> runs;
    bool open = false; int64_t first = 0, last = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (avg[i] > threshold) {
            if (!open) { first = bin_idx[i]; open = true; }
            last = bin_idx[i];
        } else if (open) { runs.emplace_back(bin_time(first), bin_time(last)); open = false; }
    }
    if (open) runs.emplace_back(bin_time(first), bin_time(last));
    std::vector<std::pair<double, double>> merged;
    if (!runs.empty()) {
        auto cur = runs[0];
        for (size_t i = 1; i < runs.size(); ++i) {
            if (runs[i].first - cur.second <= break_s) cur.second = runs[i].second;
            else { merged.push_back(cur); cur = runs[i]; }
        }
        merged.push_back(cur);
    }
    *n_out = (int64_t)merged.size();
    if ((int64_t)merged.size() > cap) return fail(nullptr, SS_ERR_CAPACITY, "ss_find_regions: output capacity too small");
    for (size_t i = 0; i < merged.size(); ++i) { out[i].start = merged[i].first - 3.0; out[i].end = merged[i].second - 3.0; }
    return SS_OK;
}

// Python repr(float): shortest digits that round-trip, positional for 1e-4 <= |x| < 1e16.
static std::string py_repr(double v) {
    if (v == 0.0) return std::signbit(v) ? "-0.0" : "0.0";
    if (std::isnan(v)) return "nan";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[64];
    int prec = 1;
    for (; prec <= 17; ++prec) { snprintf(buf, sizeof buf, "%.*e", prec - 1, v); if (strtod(buf, nullptr) == v) break; }
    std::string s(buf);
    const size_t epos = s.find('e');
    std::string mant = s.substr(0, epos);
    const int ex = atoi(s.c_str() + epos + 1);
    bool neg = false;
    if (mant[0] == '-') { neg = true; mant = mant.substr(1); }
    std::string digits;
    for (char ch : mant) if (ch != '.') digits.push_back(ch);
    std::string r;
    if (ex >= -4 && ex < 16) {
        if (ex >= 0) {
            if ((int)digits.size() <= ex + 1) r = digits + std::string(ex + 1 - digits.size(), '0') + ".0";
            else r = digits.substr(0, ex + 1) + "." + digits.substr(ex + 1);
        } else r = "0." + std::string(-ex - 1, '0') + digits;
    } else {
        r = digits.substr(0, 1);
        if (digits.size() > 1) r += "." + digits.substr(1);
        char eb[16]; snprintf(eb, sizeof eb, "e%c%02d", ex < 0 ? '-' : '+', std::abs(ex));
        r += eb;
    }
    return neg ? "-" + r : r;
}

static std::string csv_quote(const char* s) {   // csv.QUOTE_MINIMAL, as DataFrame.to_csv
    std::string v(s ? s : "");
    if (v.find_first_of(",\"\r\n") == std::string::npos) return v;
    std::string q = "\"";
    for (char ch : v) { if (ch == '"') q += "\"\""; else q.push_back(ch); }
    return q + "\"";
}

// worker.py:113-123 row dict + silencer_ui.py:816-817 to_csv(index=False)
extern "C" int64_t ss_format_csv_rows(const char* file_path, const char* file_name, const ss_region* regions, int64_t n,
                                      int64_t first_id, char* out, int64_t cap) {
    std::string s;
    const std::string fp = csv_quote(file_path), fn = csv_quote(file_name);
    for (int64_t i = 0; i < n; ++i)
        s += std::to_string(first_id + i) + "," + fp + "," + fn + "," + py_repr(regions[i].start) + "," + py_repr(regions[i].end) + ",0,,\n";
    if (out && cap > 0) {
        const size_t m = std::min<size_t>(s.size(), (size_t)cap - 1);
        memcpy(out, s.data(), m); out[m] = 0;
    }
    return (int64_t)s.size();
}

// ------------------------------------------------------------------------------------------------------
// context lifecycle
// ------------------------------------------------------------------------------------------------------
extern "C" int ss_create(int device_id, const void* weights_blob, size_t nbytes, uint32_t flags, ss_ctx** out) {
    if (!out) return fail(nullptr, SS_ERR_ARG, "ss_create: null argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, SS_ERR_HIP, std::string("ss_create: no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, SS_ERR_ARG, "ss_create: device_id out of range");
    ss_ctx* c = new ss_ctx();
    c->device = device_id; c->flags = flags; c->bf16 = (flags & SS_FLAG_BF16) != 0; c->profile = (flags & SS_FLAG_PROFILE) != 0;
    auto bail = [&](int rc) { std::string m = c->err; ss_destroy(c); g_err = m; return rc; };
    if (hipSetDevice(device_id) != hipSuccess) return bail(fail(c, SS_ERR_HIP, "hipSetDevice failed"));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return bail(fail(c, SS_ERR_HIP, std::string("ss_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName));
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(c, SS_ERR_HIP, "hipStreamCreate failed"));
    hipEventCreate(&c->ev_run0); hipEventCreate(&c->ev_run1);
    if (weights_blob) {
        Blob bl; std::string err;
        if (!parse_blob(weights_blob, nbytes, bl, err)) return bail(fail(c, SS_ERR_FORMAT, err));
        int rc;
        if ((rc = build_tables(c, bl))) return bail(rc);
        if ((rc = build_model(c, bl))) return bail(rc);
        c->has_model = true;
    }   // else: audio-only context (decode / mixdown / resample), every model entry point reports SS_ERR_STATE
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* ev = getenv("SOFTSPOKEN_CONV")) { int v = atoi(ev); if (v == 1 || v == 2) c->conv_version = v; }
    if (const char* ev = getenv("SOFTSPOKEN_CHUNK")) { int v = atoi(ev); if (v > 0) c->chunk = v; }
    *out = c;
    return SS_OK;
}

extern "C" void ss_destroy(ss_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    resolve_events(c);
    for (void* p : c->owned) hipFree(p);
    for (auto& kv : c->act) hipFree((char*)kv.second - kActHeader);
    for (auto& kv : c->taps) hipFree(kv.second.first);
    void* singles[] = {c->d_feat, c->d_flat, c->d_arena, c->d_pcm, c->d_mono, c->d_winoff, c->d_logits, c->d_spec, c->d_avg, c->d_count, c->d_starts, c->d_avgfiles, c->d_batch, c->d_flat_part, c->d_sil_out, c->d_sil_ranges, c->d_sx, c->d_sm};
    for (void* p : singles) if (p) hipFree(p);
    for (hipEvent_t ev : c->evpool) hipEventDestroy(ev);
    if (c->ev_run0) hipEventDestroy(c->ev_run0);
    if (c->ev_run1) hipEventDestroy(c->ev_run1);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->h_above) hipHostFree(c->h_above);
    if (c->h_cov) hipHostFree(c->h_cov);
    if (c->r_above) hipHostFree(c->r_above);
    if (c->r_cov) hipHostFree(c->r_cov);
    if (c->d_above) hipFree(c->d_above);
    if (c->d_cov) hipFree(c->d_cov);
    delete c;
}

extern "C" int ss_set_chunk_windows(ss_ctx* c, int chunk) {
    if (!c || chunk < 1 || chunk > 4096) return fail(c, SS_ERR_ARG, "ss_set_chunk_windows: chunk must be in [1, 4096]");
    c->chunk = chunk;
    return SS_OK;
}

// ------------------------------------------------------------------------------------------------------
// arena
// ------------------------------------------------------------------------------------------------------
extern "C" int ss_reset(ss_ctx* c) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if (c->run_pending) return fail(c, SS_ERR_STATE, "a run is in flight: ss_run_end first");
    hipSetDevice(c->device);
    c->files.clear(); c->arena_used = 0; c->logits_valid = false; c->total_windows = 0;
    return SS_OK;
}

static int arena_slot(ss_ctx* c, int64_t n, FileRec& fr, int64_t stored = -1, bool zero = true) {
    if (c->run_pending) return fail(c, SS_ERR_STATE, "a run is in flight: ss_run_end first");
    fr.n = n < 0 ? 0 : n; fr.n_padded = stored >= 0 ? stored : n + 2 * (int64_t)SS_WINDOW_SAMPLES;
    const size_t need = (size_t)fr.n_padded + 64;       // tail slack, keeps every slot 16-byte aligned
    const size_t off = (c->arena_used + 3) & ~(size_t)3;
    int rc = ensure(c, &c->d_arena, &c->arena_cap, off + need, true);
    if (rc) return rc;
    fr.off = (int64_t)off;
    c->arena_used = off + need;
    if (zero) HIPCHK(c, hipMemsetAsync(c->d_arena + off, 0, need * 4, c->stream));
    return SS_OK;
}

// Kaiser-windowed sinc polyphase taps (the build's own design; oracle/oracle_np.py resample_plan states the same)
static double bessel_i0(double x) {
    double s = 1.0, t = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) { t *= q / ((double)k * k); s += t; if (t < s * 1e-17) break; }
    return s;
}

static int get_taps(ss_ctx* c, int sr_in, int& L, int& M, int& half, float** d_taps) {
    auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
    const int g = gcd(sr_in, SS_SAMPLE_RATE);
    L = SS_SAMPLE_RATE / g; M = sr_in / g;
    const double scale = std::min(1.0, (double)SS_SAMPLE_RATE / sr_in);
    const double fc = 0.95 * scale, beta = 12.0;
    half = (int)std::ceil(32.0 / scale);
    auto key = std::make_pair(sr_in, 0);
    auto it = c->taps.find(key);
    if (it != c->taps.end()) { *d_taps = it->second.first; return SS_OK; }
    const double PI = 3.14159265358979323846;
    std::vector<float> t((size_t)L * 2 * half);
    const double i0b = bessel_i0(beta);
    for (int p = 0; p < L; ++p) {
        const double frac = (double)(((int64_t)p * M) % L) / L;
        for (int jj = 0; jj < 2 * half; ++jj) {
            const double d = (double)(jj - half + 1) - frac;
            const double xx = fc * d;
            const double sinc = xx == 0.0 ? 1.0 : std::sin(PI * xx) / (PI * xx);
            double w = 0.0;
            if (std::fabs(d) <= half) { const double u = 1.0 - (d / half) * (d / half); w = bessel_i0(beta * std::sqrt(u < 0 ? 0 : u)) / i0b; }
            t[(size_t)p * 2 * half + jj] = (float)(fc * sinc * w);
        }
    }
    float* dp = nullptr;
    HIPCHK(c, hipMalloc((void**)&dp, t.size() * 4));
    HIPCHK(c, hipMemcpy(dp, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    c->taps[key] = std::make_pair(dp, half);
    *d_taps = dp;
    return SS_OK;
}

extern "C" int ss_add_pcm_batch_device(ss_ctx* c, const void* pcm_dev, int format, int sr, int ch, const int64_t* frames,
                                       int n_files, int* first_file_id);

// one file == a batch of one (same kernels, same arithmetic)
static int add_pcm_common(ss_ctx* c, const void* d_pcm, int format, int sr, int ch, int64_t frames, int* file_id) {
    return ss_add_pcm_batch_device(c, d_pcm, format, sr, ch, &frames, 1, file_id);
}

static int check_pcm_args(ss_ctx* c, const void* pcm, int format, int sr, int ch, int64_t frames) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if ((!pcm && frames > 0) || format < SS_PCM_U8 || format > SS_PCM_F64 || sr <= 0 || sr > 768000 || ch < 1 || ch > 64 || frames < 0 ||
        frames > ((int64_t)1 << 36))          // (99 h at 192 kHz; keeps frames * channels * bytes and frames * 22050 inside 64 bits)
        return fail(c, SS_ERR_ARG, "ss_add_pcm: bad argument");
    return SS_OK;
}

// voice_activity.py:32-69 (decode -> mono -> resample) + worker.py:58-62 (pad)
extern "C" int ss_add_pcm(ss_ctx* c, const void* pcm, int format, int sr, int ch, int64_t frames, int* file_id) {
    int rc = check_pcm_args(c, pcm, format, sr, ch, frames);
    if (rc) return rc;
    hipSetDevice(c->device);
    const size_t bps = format == SS_PCM_U8 ? 1 : format == SS_PCM_S16 ? 2 : format == SS_PCM_S24 ? 3 : format == SS_PCM_F64 ? 8 : 4;
    const size_t bytes = (size_t)frames * ch * bps;
    size_t cap_b = c->pcm_cap;
    if ((rc = ensure(c, (char**)&c->d_pcm, &cap_b, bytes + 16))) return rc;
    c->pcm_cap = cap_b;
    if (bytes) HIPCHK(c, hipMemcpyAsync(c->d_pcm, pcm, bytes, hipMemcpyHostToDevice, c->stream));
    rc = add_pcm_common(c, c->d_pcm, format, sr, ch, frames, file_id);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));     // d_pcm / d_mono are reused by the next call
    return SS_OK;
}

extern "C" int ss_add_pcm_device(ss_ctx* c, const void* pcm_dev, int format, int sr, int ch, int64_t frames, int* file_id) {
    int rc = check_pcm_args(c, pcm_dev, format, sr, ch, frames);
    if (rc) return rc;
    hipSetDevice(c->device);
    return add_pcm_common(c, pcm_dev, format, sr, ch, frames, file_id);
}

// ------------------------------------------------------------------------------------------------------
// silencer: silencer_ui.py:974-998
// ------------------------------------------------------------------------------------------------------
// Frame ranges the reference's slice assignment touches: int(round(t * sr)) with Python's round (half to
// even), clamped to [0, frames]; sorted and merged so the kernel can binary-search them.
static std::vector<int64_t> silence_ranges(const ss_region* regions, int64_t n, int sr, int64_t frames) {
    std::vector<std::pair<int64_t, int64_t>> r;
    for (int64_t i = 0; i < n; ++i) {
        const double a = std::nearbyint(regions[i].start * (double)sr), b = std::nearbyint(regions[i].end * (double)sr);
        if (std::isnan(a) || std::isnan(b)) continue;
        const int64_t lo = (int64_t)std::min<double>(std::max<double>(a, 0.0), (double)frames);
        const int64_t hi = (int64_t)std::min<double>(std::max<double>(b, 0.0), (double)frames);
        if (hi > lo) r.emplace_back(lo, hi);
    }
    std::sort(r.begin(), r.end());
    std::vector<int64_t> out;
    for (const auto& p : r) {
        if (!out.empty() && p.first <= out.back()) out.back() = std::max(out.back(), p.second);
        else { out.push_back(p.first); out.push_back(p.second); }
    }
    return out;
}

extern "C" int ss_silence_pcm(ss_ctx* c, const void* pcm, int format, int sr, int ch, int64_t frames, const ss_region* regions,
                              int64_t n_regions, int16_t* out) {
    int rc = check_pcm_args(c, pcm, format, sr, ch, frames);
    if (rc) return rc;
    if ((!regions && n_regions > 0) || n_regions < 0 || (!out && frames > 0)) return fail(c, SS_ERR_ARG, "ss_silence_pcm: bad argument");
    if (frames == 0) return SS_OK;
    hipSetDevice(c->device);
    const size_t bps = format == SS_PCM_U8 ? 1 : format == SS_PCM_S16 ? 2 : format == SS_PCM_S24 ? 3 : format == SS_PCM_F64 ? 8 : 4;
    const size_t bytes = (size_t)frames * ch * bps, total = (size_t)frames * ch;
    const std::vector<int64_t> ranges = silence_ranges(regions, n_regions, sr, frames);
    size_t cap_b = c->pcm_cap;
    if ((rc = ensure(c, (char**)&c->d_pcm, &cap_b, bytes + 16))) return rc;
    c->pcm_cap = cap_b;
    if ((rc = ensure(c, &c->d_sil_out, &c->sil_out_cap, total + 8))) return rc;
    if ((rc = ensure(c, &c->d_sil_ranges, &c->sil_ranges_cap, ranges.size() + 2))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_pcm, pcm, bytes, hipMemcpyHostToDevice, c->stream));
    if (!ranges.empty())
        HIPCHK(c, hipMemcpyAsync(c->d_sil_ranges, ranges.data(), ranges.size() * 8, hipMemcpyHostToDevice, c->stream));
    {
        ScopedLaunch sl(c, "silence_encode_kernel", 0.0, (double)bytes + 2.0 * (double)total);
        HIPCHK(c, launch_silence_encode(c->d_pcm, format, ch, frames, c->d_sil_ranges, (int)(ranges.size() / 2), c->d_sil_out, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(out, c->d_sil_out, total * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));     // `ranges` and the caller's buffers are free again
    return SS_OK;
}

// ------------------------------------------------------------------------------------------------------
// review-screen spectrogram: voice_activity.py:148-154
// ------------------------------------------------------------------------------------------------------
extern "C" int64_t ss_stft512_frames(int64_t n) { return n < 0 ? -1 : 1 + n / 256; }

extern "C" int ss_stft512_magnitude(ss_ctx* c, const float* samples, int64_t n, float* out, int64_t cap_frames) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if (n < 0 || (!samples && n > 0) || !out) return fail(c, SS_ERR_ARG, "ss_stft512_magnitude: bad argument");
    const int64_t nf = 1 + n / 256;
    if (cap_frames < nf) return fail(c, SS_ERR_CAPACITY, "ss_stft512_magnitude: capacity < " + std::to_string(nf) + " frames");
    hipSetDevice(c->device);
    int rc;
    if ((rc = ensure(c, &c->d_sx, &c->sx_cap, (size_t)std::max<int64_t>(n, 1)))) return rc;
    if ((rc = ensure(c, &c->d_sm, &c->sm_cap, (size_t)nf * 257))) return rc;
    if (n) HIPCHK(c, hipMemcpyAsync(c->d_sx, samples, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    {
        ScopedLaunch sl(c, "stft512_mag_kernel", 0.0, (double)n * 4 + (double)nf * 257 * 4);
        HIPCHK(c, launch_stft512_mag(c->d_sx, n, nf, c->d_sm, c->num_cus, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(out, c->d_sm, (size_t)nf * 257 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}

// Canonical 44-byte RIFF/WAVE header of a 16-bit PCM file (what libsndfile writes for subtype PCM_16).
extern "C" int ss_wav_header_pcm16(int sr, int ch, int64_t frames, void* out44) {
    const int64_t data = frames * ch * 2;
    if (!out44 || sr <= 0 || ch < 1 || ch > 64 || frames < 0 || data + 36 > 0xFFFFFFFFLL)
        return fail(nullptr, SS_ERR_ARG, "ss_wav_header_pcm16: bad argument (a RIFF file holds < 4 GiB)");
    unsigned char* h = (unsigned char*)out44;
    auto u32 = [&](int at, uint32_t v) { for (int i = 0; i < 4; ++i) h[at + i] = (unsigned char)(v >> (8 * i)); };
    auto u16 = [&](int at, uint32_t v) { h[at] = (unsigned char)v; h[at + 1] = (unsigned char)(v >> 8); };
    memcpy(h, "RIFF", 4); u32(4, (uint32_t)(36 + data)); memcpy(h + 8, "WAVEfmt ", 8); u32(16, 16);
    u16(20, 1); u16(22, (uint32_t)ch); u32(24, (uint32_t)sr); u32(28, (uint32_t)(sr * ch * 2)); u16(32, (uint32_t)(ch * 2)); u16(34, 16);
    memcpy(h + 36, "data", 4); u32(40, (uint32_t)data);
    return SS_OK;
}

// Many files of one format in one device buffer, back to back: two launches for the whole batch.
extern "C" int ss_add_pcm_batch_device(ss_ctx* c, const void* pcm_dev, int format, int sr, int ch, const int64_t* frames,
                                       int n_files, int* first_file_id) {
    if (!frames || n_files < 1) return fail(c, SS_ERR_ARG, "ss_add_pcm_batch_device: bad argument");
    int64_t total_frames = 0, max_frames = 0, max_out = 0;
    for (int i = 0; i < n_files; ++i) {
        int rc = check_pcm_args(c, pcm_dev, format, sr, ch, frames[i]);
        if (rc) return rc;
        total_frames += frames[i]; max_frames = std::max(max_frames, frames[i]);
    }
    hipSetDevice(c->device);
    const size_t bps = format == SS_PCM_U8 ? 1 : format == SS_PCM_S16 ? 2 : format == SS_PCM_S24 ? 3 : format == SS_PCM_F64 ? 8 : 4;
    int rc;
    // reserve every arena slot first (the arena may move while it grows)
    const size_t first = c->files.size();
    const size_t arena_before = (c->arena_used + 3) & ~(size_t)3;
    std::vector<BatchFile> bf(n_files);
    int64_t pcm_off = 0, mono_off = 0;
    for (int i = 0; i < n_files; ++i) {
        FileRec fr;
        fr.duration = (double)frames[i] / (double)sr;
        const int64_t n22 = ss_resampled_length(frames[i], sr);
        if ((rc = arena_slot(c, n22, fr, -1, false))) return rc;
        c->files.push_back(fr);
        bf[i].pcm_off = pcm_off; bf[i].frames = frames[i]; bf[i].mono_off = mono_off; bf[i].n_out = n22;
        bf[i].out_off = fr.off + SS_WINDOW_SAMPLES;
        pcm_off += frames[i] * ch * (int64_t)bps; mono_off += (frames[i] + 3) & ~(int64_t)3;
        max_out = std::max(max_out, n22);
    }
    // one fill for the padding of the whole batch instead of one per file
    HIPCHK(c, hipMemsetAsync(c->d_arena + arena_before, 0, (c->arena_used - arena_before) * 4, c->stream));
    size_t cap = c->batch_cap;
    if ((rc = ensure(c, &c->d_batch, &cap, (size_t)n_files))) return rc;
    c->batch_cap = cap;
    HIPCHK(c, hipMemcpyAsync(c->d_batch, bf.data(), bf.size() * sizeof(BatchFile), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));      // bf is a host temporary
    const double pcm_bytes = (double)total_frames * ch * bps;
    if (sr == SS_SAMPLE_RATE) {
        // decode straight into the arena: mono_off := out_off
        for (int i = 0; i < n_files; ++i) bf[i].mono_off = bf[i].out_off;
        HIPCHK(c, hipMemcpyAsync(c->d_batch, bf.data(), bf.size() * sizeof(BatchFile), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ScopedLaunch sl(c, "decode_mono_batch", 0.0, pcm_bytes + 4.0 * total_frames);
        HIPCHK(c, launch_decode_mono_batch(pcm_dev, format, ch, c->d_batch, n_files, max_frames, c->d_arena, c->stream));
    } else {
        if ((rc = ensure(c, &c->d_mono, &c->mono_cap, (size_t)mono_off + 16))) return rc;
        {
            ScopedLaunch sl(c, "decode_mono_batch", 0.0, pcm_bytes + 4.0 * total_frames);
            HIPCHK(c, launch_decode_mono_batch(pcm_dev, format, ch, c->d_batch, n_files, max_frames, c->d_mono, c->stream));
        }
        int L, M, half; float* d_taps;
        if ((rc = get_taps(c, sr, L, M, half, &d_taps))) return rc;
        double n22sum = 0; for (auto& b : bf) n22sum += (double)b.n_out;
        ScopedLaunch sl(c, "resample_batch", 2.0 * 2 * half * n22sum, 4.0 * total_frames + 4.0 * n22sum);
        HIPCHK(c, launch_resample_batch(c->d_mono, c->d_batch, n_files, max_out, L, M, half, d_taps, c->d_arena, c->num_cus, c->stream));
    }
    if (first_file_id) *first_file_id = (int)first;
    c->logits_valid = false;
    return SS_OK;
}

static int add_f32(ss_ctx* c, const float* s, int64_t n, bool padded, int* file_id) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if ((!s && n > 0) || n < 0) return fail(c, SS_ERR_ARG, "ss_add_f32: bad argument");
    hipSetDevice(c->device);
    FileRec fr;
    const int64_t core = padded ? n - 2 * (int64_t)SS_WINDOW_SAMPLES : n;
    fr.duration = (double)(core < 0 ? 0 : core) / 22050.0;
    int rc;
    if ((rc = arena_slot(c, core, fr, padded ? n : -1))) return rc;
    if (n) HIPCHK(c, hipMemcpyAsync(c->d_arena + fr.off + (padded ? 0 : SS_WINDOW_SAMPLES), s, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->files.push_back(fr);
    if (file_id) *file_id = (int)c->files.size() - 1;
    c->logits_valid = false;
    return SS_OK;
}

extern "C" int ss_add_f32_22k(ss_ctx* c, const float* s, int64_t n, int* file_id) { return add_f32(c, s, n, false, file_id); }
extern "C" int ss_add_padded_f32_22k(ss_ctx* c, const float* s, int64_t n, int* file_id) { return add_f32(c, s, n, true, file_id); }

extern "C" int64_t ss_signal_length(ss_ctx* c, int file_id, int padded) {
    if (!c || file_id < 0 || file_id >= (int)c->files.size()) return -1;
    return padded ? c->files[file_id].n_padded : c->files[file_id].n;
}

extern "C" int ss_read_signal(ss_ctx* c, int file_id, int padded, int64_t offset, int64_t n, float* out) {
    if (!c || file_id < 0 || file_id >= (int)c->files.size() || !out) return fail(c, SS_ERR_ARG, "ss_read_signal: bad argument");
    const FileRec& f = c->files[file_id];
    const int64_t len = padded ? f.n_padded : f.n;
    if (offset < 0 || n < 0 || offset + n > len) return fail(c, SS_ERR_ARG, "ss_read_signal: range outside the signal");
    hipSetDevice(c->device);
    if (n) HIPCHK(c, hipMemcpyAsync(out, c->d_arena + f.off + (padded ? 0 : SS_WINDOW_SAMPLES) + offset, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}

extern "C" int ss_device_alloc(ss_ctx* c, size_t nbytes, void** p) {
    if (!c || !p) return fail(c, SS_ERR_ARG, "ss_device_alloc: null argument");
    hipSetDevice(c->device);
    HIPCHK(c, hipMalloc(p, nbytes ? nbytes : 16));
    return SS_OK;
}
extern "C" int ss_device_free(ss_ctx* c, void* p) {
    if (!c) return fail(c, SS_ERR_ARG, "null context");
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(p));
    return SS_OK;
}
extern "C" int ss_device_upload(ss_ctx* c, void* dst, const void* src, size_t nbytes) {
    if (!c || !dst || !src) return fail(c, SS_ERR_ARG, "ss_device_upload: null argument");
    hipSetDevice(c->device);
    HIPCHK(c, hipMemcpy(dst, src, nbytes, hipMemcpyHostToDevice));
    return SS_OK;
}

// ------------------------------------------------------------------------------------------------------
// compute
// ------------------------------------------------------------------------------------------------------
static int check_windows(ss_ctx* c, int file_id, const int64_t* starts, int n) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if (!c->has_model) return fail(c, SS_ERR_STATE, "context was created without weights (audio-only)");
    if (c->run_pending) return fail(c, SS_ERR_STATE, "a run is in flight: ss_run_end first");
    if (file_id < 0 || file_id >= (int)c->files.size() || !starts || n < 1) return fail(c, SS_ERR_ARG, "bad file_id / starts / n");
    const FileRec& f = c->files[file_id];
    for (int i = 0; i < n; ++i)
        if (starts[i] < 0 || starts[i] + SS_WINDOW_SAMPLES > f.n_padded)
            return fail(c, SS_ERR_ARG, "window start " + std::to_string(starts[i]) + " does not fit the padded signal (" + std::to_string(f.n_padded) + " samples)");
    return SS_OK;
}

static int upload_winoff(ss_ctx* c, const std::vector<int64_t>& off) {
    int rc = ensure(c, &c->d_winoff, &c->winoff_cap, off.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_winoff, off.data(), off.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));     // `off` is a host temporary
    return SS_OK;
}

extern "C" int ss_features(ss_ctx* c, int file_id, const int64_t* starts, int n, float* feat_out) {
    int rc = check_windows(c, file_id, starts, n);
    if (rc) return rc;
    hipSetDevice(c->device);        // feat_out == NULL: run the front-end and discard (timing runs)
    std::vector<int64_t> off(n);
    for (int i = 0; i < n; ++i) off[i] = c->files[file_id].off + starts[i];
    if ((rc = upload_winoff(c, off))) return rc;
    const int ch = std::min(n, c->chunk);
    if ((rc = ensure_workspace(c, ch))) return rc;
    for (int i0 = 0; i0 < n; i0 += ch) {
        const int m = std::min(ch, n - i0);
        if ((rc = forward_chunk(c, c->d_winoff + i0, m, nullptr, nullptr, nullptr))) return rc;
        if (feat_out) {
            HIPCHK(c, hipMemcpyAsync(feat_out + (size_t)i0 * 32768, c->d_feat, (size_t)m * 32768 * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    return SS_OK;
}

// NNDetector.py:84-101
extern "C" int ss_infer_windows(ss_ctx* c, int file_id, const int64_t* starts, int n, float* mask_out, float* spec_out) {
    int rc = check_windows(c, file_id, starts, n);
    if (rc) return rc;
    if (!mask_out) return fail(c, SS_ERR_ARG, "ss_infer_windows: null output");
    hipSetDevice(c->device);
    std::vector<int64_t> off(n);
    for (int i = 0; i < n; ++i) off[i] = c->files[file_id].off + starts[i];
    if ((rc = upload_winoff(c, off))) return rc;
    const int ch = std::min(n, c->chunk);
    if ((rc = ensure_workspace(c, ch))) return rc;
    if ((rc = ensure(c, &c->d_logits, &c->logits_cap, (size_t)n * 256))) return rc;
    if (spec_out && (rc = ensure(c, &c->d_spec, &c->spec_cap, (size_t)ch * 2 * 32768))) return rc;
    c->logits_valid = false;
    for (int i0 = 0; i0 < n; i0 += ch) {
        const int m = std::min(ch, n - i0);
        if ((rc = forward_chunk(c, c->d_winoff + i0, m, c->d_logits + (size_t)i0 * 256, spec_out ? c->d_spec : nullptr, nullptr))) return rc;
        if (spec_out) {
            HIPCHK(c, hipMemcpyAsync(spec_out + (size_t)i0 * 2 * 32768, c->d_spec, (size_t)m * 2 * 32768 * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    HIPCHK(c, hipMemcpyAsync(mask_out, c->d_logits, (size_t)n * 256 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}

// worker.py:49-100 over every file of the arena, in two halves: everything up to the last device -> host copy is enqueued by
// run_begin; run_end waits for it and finds the regions on the host.  ss_run is the two back to back; ss_run_begin / ss_run_end let
// a caller with two contexts overlap one job's host half with the next job's device half.
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int run_begin(ss_ctx* c, double threshold, double break_s, ss_progress_fn progress, void* user, const volatile int* stop_flag) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if (!c->has_model) return fail(c, SS_ERR_STATE, "context was created without weights (audio-only)");
    if (c->run_pending) return fail(c, SS_ERR_STATE, "a run is in flight: ss_run_end first");
    if (c->files.empty()) return fail(c, SS_ERR_STATE, "ss_run: no files added since ss_reset");
    hipSetDevice(c->device);
    int rc;
    c->t_in = now_ms();
    // ---- plan (NNDetector.py:55-82) ----
    int64_t total = 0, total_bins = 0; int max_bins = 0;
    std::vector<int64_t> off;
    std::vector<int32_t> starts;
    std::vector<AvgFile>& af = c->pend_af;
    af.assign(c->files.size(), AvgFile{});
    for (size_t fi = 0; fi < c->files.size(); ++fi) {
        FileRec& f = c->files[fi];
        f.W = ss_plan_windows(f.duration, nullptr, 0);
        // the plan comes from the header duration, the data from the resampler: clamp to what fits
        while (f.W > 0 && (f.W - 1) * (int64_t)SS_STEP_SAMPLES + SS_WINDOW_SAMPLES > f.n_padded) --f.W;
        f.win_base = total;
        const double secs = (double)f.n_padded / 22050.0;
        const int n_bins = (int)std::nearbyint(secs * 256.0 / 3.0);          // NNDetector.py:168
        af[fi].logit_off = total; af[fi].bin_off = total_bins; af[fi].W = (int32_t)f.W; af[fi].n_bins = n_bins; af[fi].start_off = total;
        for (int64_t i = 0; i < f.W; ++i) {
            off.push_back(f.off + i * SS_STEP_SAMPLES);
            starts.push_back((int32_t)std::nearbyint((double)i * 0.6 / (3.0 / 256.0)));   // NNDetector.py:175
        }
        total += f.W; total_bins += n_bins; max_bins = std::max(max_bins, n_bins);
    }
    c->total_windows = total;
    c->logits_valid = false;
    if (total > 0) {
        if ((rc = ensure(c, &c->d_logits, &c->logits_cap, (size_t)total * 256))) return rc;
        if ((rc = ensure(c, &c->d_starts, &c->starts_cap, (size_t)total))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->d_starts, starts.data(), starts.size() * 4, hipMemcpyHostToDevice, c->stream));
        if ((rc = upload_winoff(c, off))) return rc;      // (synchronises: off and starts are host temporaries)
    }
    if ((rc = ensure(c, &c->d_avgfiles, &c->avgfiles_cap, af.size()))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_avgfiles, af.data(), af.size() * sizeof(AvgFile), hipMemcpyHostToDevice, c->stream));   // af lives in the context
    {
        size_t cap = c->avg_cap, cap2 = c->avg_cap;
        if ((rc = ensure(c, &c->d_avg, &cap, (size_t)std::max<int64_t>(total_bins, 1)))) return rc;
        if ((rc = ensure(c, &c->d_count, &cap2, (size_t)std::max<int64_t>(total_bins, 1)))) return rc;
        c->avg_cap = std::min(cap, cap2);
    }
    const size_t words = (size_t)((total_bins + 255) / 256) * 4 + 1;     // bin_masks_kernel writes whole blocks of 4 words
    if ((rc = ensure(c, &c->d_above, &c->mask_cap, words))) return rc;
    if ((rc = ensure(c, &c->d_cov, &c->cov_cap, words))) return rc;
    if (words > c->hmask_cap) {                           // pinned result buffers
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->h_above) hipHostFree(c->h_above);
        if (c->h_cov) hipHostFree(c->h_cov);
        c->h_above = nullptr; c->h_cov = nullptr; c->hmask_cap = 0;
        const size_t cap = words + words / 2;
        HIPCHK(c, hipHostMalloc((void**)&c->h_above, cap * 8, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&c->h_cov, cap * 8, hipHostMallocDefault));
        c->hmask_cap = cap;
    }
    c->total_bins = total_bins; c->avg_on_host = false; ++c->begin_gen;
    c->t_plan = now_ms();
    c->t_sync = c->t_plan;
    // passes of equal size (2560 windows: 3 x 854, not 1024 + 1024 + 512: a short last pass has the launch overheads and tail
    // effects of a full one; within the noise of a same-box A/B on C2) -- unless the caller watches the progress: then a pass is exactly `chunk` windows,
    // as the reference's batches of settings.prediction_batch_size are (worker.py:71-84)
    const int64_t n_pass = std::max<int64_t>(1, (total + c->chunk - 1) / c->chunk);
    const int ch = progress ? (int)std::min<int64_t>(std::max<int64_t>(total, 1), c->chunk) : (int)std::max<int64_t>(1, (total + n_pass - 1) / n_pass);
    if ((rc = ensure_workspace(c, ch))) return rc;        // (waits for the stream itself when it has to reallocate)
    // ---- windows in chunks, across file boundaries (worker.py:71-84 batches per file of 32) ----
    HIPCHK(c, hipEventRecord(c->ev_run0, c->stream));
    for (int64_t i0 = 0; i0 < total; i0 += ch) {
        if (stop_flag && *stop_flag) { hipStreamSynchronize(c->stream); return fail(c, SS_ERR_STOPPED, "stopped on request"); }
        const int m = (int)std::min<int64_t>(ch, total - i0);
        if ((rc = forward_chunk(c, c->d_winoff + i0, m, c->d_logits + (size_t)i0 * 256, nullptr, nullptr))) return rc;
        if (progress) { HIPCHK(c, hipStreamSynchronize(c->stream)); progress(user, i0 + m, total); }
    }
    // ---- overlap averaging on the device (NNDetector.py:153-190) ----
    {
        ScopedLaunch sl(c, "average", 0.0, (double)total * 1024 * 5 + (double)total_bins * 12);
        HIPCHK(c, launch_average(c->d_logits, c->d_avgfiles, (int)af.size(), c->d_starts, c->d_avg, c->d_count, max_bins, c->stream));
    }
    if (total_bins) {
        ScopedLaunch sl(c, "bin_masks", 0.0, (double)total_bins * 12 + (double)words * 16);
        HIPCHK(c, launch_bin_masks(c->d_avg, c->d_count, total_bins, threshold, c->d_above, c->d_cov, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev_run1, c->stream));
    if (total_bins) {
        HIPCHK(c, hipMemcpyAsync(c->h_above, c->d_above, words * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_cov, c->d_cov, words * 8, hipMemcpyDeviceToHost, c->stream));
    }
    c->t_loop = now_ms();
    c->pend_thr = threshold; c->pend_brk = break_s;
    c->run_pending = true;
    return SS_OK;
}

static int run_end(ss_ctx* c) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if (!c->run_pending) return fail(c, SS_ERR_STATE, "ss_run_end: no run in flight");
    hipSetDevice(c->device);
    static const bool timing = getenv("SOFTSPOKEN_TIMING") != nullptr;      // development aid: host-side phases of a run on stderr
    c->run_pending = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    { float ms = 0; if (hipEventElapsedTime(&ms, c->ev_run0, c->ev_run1) == hipSuccess) c->last_run_ms = ms; }
    resolve_events(c);
    const double t_d2h = now_ms();
    // the run's results leave the working set: file bookkeeping is copied, the mask buffers change places with the previous result's
    const std::vector<AvgFile>& af = c->pend_af;
    c->res_files.resize(c->files.size());
    for (size_t fi = 0; fi < c->files.size(); ++fi) {
        FileRec& f = c->files[fi];
        f.bin_off = af[fi].bin_off; f.n_bins = af[fi].n_bins;
        ss_ctx::ResFile& r = c->res_files[fi];
        r.W = f.W; r.win_base = f.win_base; r.bin_off = f.bin_off; r.n_bins = f.n_bins; r.regions.clear();
    }
    std::swap(c->h_above, c->r_above); std::swap(c->h_cov, c->r_cov); std::swap(c->hmask_cap, c->rmask_cap);
    c->res_thr = c->pend_thr; c->res_brk = c->pend_brk;
    c->res_valid = true; c->res_regions = false; c->res_gen = c->begin_gen;
    c->logits_valid = true;
    if (timing)
        fprintf(stderr, "[ss_run] plan+uploads %.3f ms, enqueue %.3f, drain+D2H %.3f (device %.3f), end %.3f\n",
                c->t_plan - c->t_in, c->t_loop - c->t_sync, t_d2h - c->t_loop, c->last_run_ms, now_ms() - t_d2h);
    return SS_OK;
}

// Run lengths + gap merge on the host (NNDetector.py:103-143, worker.py:100) from the two bit masks of the ended run, when a getter
// first asks: a run opens at a bin above the threshold and closes at the next COVERED bin that is not; uncovered bins are absent from
// the reference's series and neither extend nor close a run.  The same decisions ss_find_regions takes on the compacted series.
static void ensure_regions(ss_ctx* c) {
    if (c->res_regions) return;
    static const bool timing = getenv("SOFTSPOKEN_TIMING") != nullptr;
    const double t0 = now_ms();
    const double break_s = c->res_brk;
    const unsigned long long* AB = c->r_above;
    const unsigned long long* CV = c->r_cov;
    // first set bit of (word(k) for k >= pos) in [pos, hi), or hi
    auto next_bit = [&](auto&& word, int64_t pos, int64_t hi) -> int64_t {
        while (pos < hi) {
            unsigned long long w = word(pos >> 6) >> (pos & 63);
            if (w) { const int64_t p = pos + __builtin_ctzll(w); return p < hi ? p : hi; }
            pos = (pos | 63) + 1;
        }
        return hi;
    };
    // last set bit of `above` in [lo, hi), or -1
    auto prev_above = [&](int64_t lo, int64_t hi) -> int64_t {
        int64_t pos = hi - 1;
        while (pos >= lo) {
            unsigned long long w = AB[pos >> 6] << (63 - (pos & 63));
            if (w) { const int64_t p = pos - __builtin_clzll(w); return p >= lo ? p : -1; }
            pos = (pos & ~(int64_t)63) - 1;
        }
        return -1;
    };
    auto above_w = [&](int64_t k) { return AB[k]; };
    auto closer_w = [&](int64_t k) { return CV[k] & ~AB[k]; };
    for (ss_ctx::ResFile& f : c->res_files) {
        f.regions.clear();
        const int64_t lo = f.bin_off, hi = f.bin_off + f.n_bins;
        bool have = false;
        ss_region cur{0, 0};
        int64_t pos = lo;
        while (pos < hi) {
            const int64_t first = next_bit(above_w, pos, hi);
            if (first == hi) break;
            const int64_t q = next_bit(closer_w, first + 1, hi);              // the covered bin that ends the run, or the file's end
            const int64_t last = prev_above(first, q);                         // (>= first: `first` itself is above)
            const double s0 = bin_time(first - lo), e0 = bin_time(last - lo);
            if (have && s0 - cur.end <= break_s) cur.end = e0;
            else { if (have) f.regions.push_back(ss_region{cur.start - 3.0, cur.end - 3.0}); cur.start = s0; cur.end = e0; have = true; }
            pos = q + 1;
        }
        if (have) f.regions.push_back(ss_region{cur.start - 3.0, cur.end - 3.0});
    }
    c->res_regions = true;
    if (timing) fprintf(stderr, "[ss_run] regions %.3f ms\n", now_ms() - t0);
}

extern "C" int ss_run(ss_ctx* c, double threshold, double break_s, ss_progress_fn progress, void* user, const volatile int* stop_flag) {
    const int rc = run_begin(c, threshold, break_s, progress, user, stop_flag);
    return rc ? rc : run_end(c);
}

extern "C" int ss_run_begin(ss_ctx* c, double threshold, double break_s) { return run_begin(c, threshold, break_s, nullptr, nullptr, nullptr); }

extern "C" int ss_run_end(ss_ctx* c) { return run_end(c); }

// The getters below read the last ENDED run (ss_ctx::res_*).  Regions stay readable while the next job is added and in flight;
// averages and per-window logits live in device buffers that the next ss_run_begin reuses, so they are refused after it.
extern "C" int64_t ss_num_windows(ss_ctx* c, int file_id) {
    if (!c || file_id < 0 || !c->res_valid || file_id >= (int)c->res_files.size()) return -1;
    return c->res_files[file_id].W;
}

static int device_results_ok(ss_ctx* c, const char* who) {
    if (!c->res_valid || !c->logits_valid || c->res_gen != c->begin_gen || c->run_pending)
        return fail(c, SS_ERR_STATE, std::string(who) + ": no completed ss_run (or a newer job has taken its device buffers)");
    return SS_OK;
}

extern "C" int ss_get_window_logits(ss_ctx* c, int file_id, float* out, int64_t cap_windows) {
    if (!c || file_id < 0 || !out) return fail(c, SS_ERR_ARG, "ss_get_window_logits: bad argument");
    int rc = device_results_ok(c, "ss_get_window_logits");
    if (rc) return rc;
    if (file_id >= (int)c->res_files.size()) return fail(c, SS_ERR_ARG, "ss_get_window_logits: bad argument");
    const ss_ctx::ResFile& f = c->res_files[file_id];
    if (cap_windows < f.W) return fail(c, SS_ERR_CAPACITY, "ss_get_window_logits: capacity < " + std::to_string(f.W));
    hipSetDevice(c->device);
    if (f.W) HIPCHK(c, hipMemcpyAsync(out, c->d_logits + (size_t)f.win_base * 256, (size_t)f.W * 1024, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}

extern "C" int ss_get_avg(ss_ctx* c, int file_id, double* avg, int64_t* bin_idx, int64_t cap, int64_t* n_out) {
    if (!c || file_id < 0 || !n_out) return fail(c, SS_ERR_ARG, "ss_get_avg: bad argument");
    int rc = device_results_ok(c, "ss_get_avg");
    if (rc) return rc;
    if (file_id >= (int)c->res_files.size()) return fail(c, SS_ERR_ARG, "ss_get_avg: bad argument");
    const ss_ctx::ResFile& f = c->res_files[file_id];
    if (!c->avg_on_host) {                                // the run itself only brought the bin masks back
        hipSetDevice(c->device);
        c->h_avg.resize((size_t)c->total_bins); c->h_cnt.resize((size_t)c->total_bins);
        if (c->total_bins) {
            HIPCHK(c, hipMemcpyAsync(c->h_avg.data(), c->d_avg, (size_t)c->total_bins * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->h_cnt.data(), c->d_count, (size_t)c->total_bins * 4, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->avg_on_host = true;
    }
    const double* av = c->h_avg.data() + f.bin_off;
    const int32_t* cn = c->h_cnt.data() + f.bin_off;
    int64_t covered = 0;
    for (int j = 0; j < f.n_bins; ++j) covered += cn[j] >= 1;
    *n_out = covered;
    if (!avg && !bin_idx) return SS_OK;
    if (cap < covered) return fail(c, SS_ERR_CAPACITY, "ss_get_avg: capacity too small");
    int64_t at = 0;
    for (int j = 0; j < f.n_bins; ++j)
        if (cn[j] >= 1) { if (avg) avg[at] = av[j]; if (bin_idx) bin_idx[at] = j; ++at; }
    return SS_OK;
}

// All files [first_file, first_file + n_files) in one call: counts[i] regions of file first_file + i, back to back in out.
extern "C" int ss_get_regions_batch(ss_ctx* c, int first_file, int n_files, int64_t* counts, ss_region* out, int64_t cap, int64_t* n_out) {
    if (!c || !n_out || first_file < 0 || n_files < 0) return fail(c, SS_ERR_ARG, "ss_get_regions_batch: bad argument");
    if (!c->res_valid) return fail(c, SS_ERR_STATE, "ss_get_regions_batch: no completed ss_run");
    if ((size_t)first_file + (size_t)n_files > c->res_files.size()) return fail(c, SS_ERR_ARG, "ss_get_regions_batch: bad argument");
    ensure_regions(c);
    int64_t total = 0;
    for (int i = 0; i < n_files; ++i) total += (int64_t)c->res_files[first_file + i].regions.size();
    *n_out = total;
    if (!out && !counts) return SS_OK;
    if (out && cap < total) return fail(c, SS_ERR_CAPACITY, "ss_get_regions_batch: capacity < " + std::to_string(total));
    int64_t at = 0;
    for (int i = 0; i < n_files; ++i) {
        const ss_ctx::ResFile& f = c->res_files[first_file + i];
        if (counts) counts[i] = (int64_t)f.regions.size();
        if (out && !f.regions.empty()) memcpy(out + at, f.regions.data(), f.regions.size() * sizeof(ss_region));
        at += (int64_t)f.regions.size();
    }
    return SS_OK;
}

extern "C" int ss_get_regions(ss_ctx* c, int file_id, ss_region* out, int64_t cap, int64_t* n_out) {
    if (!c || file_id < 0 || !n_out) return fail(c, SS_ERR_ARG, "ss_get_regions: bad argument");
    if (!c->res_valid) return fail(c, SS_ERR_STATE, "ss_get_regions: no completed ss_run");
    if (file_id >= (int)c->res_files.size()) return fail(c, SS_ERR_ARG, "ss_get_regions: bad argument");
    ensure_regions(c);
    const ss_ctx::ResFile& f = c->res_files[file_id];
    *n_out = (int64_t)f.regions.size();
    if (!out) return SS_OK;
    if (cap < (int64_t)f.regions.size()) return fail(c, SS_ERR_CAPACITY, "ss_get_regions: capacity too small");
    if (!f.regions.empty()) memcpy(out, f.regions.data(), f.regions.size() * sizeof(ss_region));
    return SS_OK;
}

// ------------------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------------------
extern "C" int ss_sync(ss_ctx* c) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_events(c);
    return SS_OK;
}

extern "C" int ss_reset_kernel_stats(ss_ctx* c) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    ss_sync(c);
    c->stats.clear();
    return SS_OK;
}

extern "C" int ss_get_kernel_stats(ss_ctx* c, ss_kernel_stat* out, int cap, int* n_out) {
    if (!c || !n_out) return fail(c, SS_ERR_ARG, "ss_get_kernel_stats: null argument");
    ss_sync(c);
    *n_out = (int)c->stats.size();
    if (!out) return SS_OK;
    for (int i = 0; i < (int)c->stats.size() && i < cap; ++i) {
        memset(&out[i], 0, sizeof(ss_kernel_stat));
        strncpy(out[i].name, c->stats[i].name.c_str(), sizeof(out[i].name) - 1);
        out[i].launches = c->stats[i].launches; out[i].total_ms = c->stats[i].ms; out[i].flops = c->stats[i].flops; out[i].bytes = c->stats[i].bytes;
    }
    return SS_OK;
}

extern "C" double ss_last_run_device_ms(ss_ctx* c) { return c ? c->last_run_ms : -1.0; }
