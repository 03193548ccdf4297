// Internal state of a context, shared by the host-side units of the library:
//   weights.hip   weights blob -> folded BatchNorm -> MFMA fragment packing, front-end tables      (ss_create's work)
//   engine.hip    activation workspace, per-chunk launch sequence of the U-Net, job halves (plan + enqueue / wait)
//   host.hip      host-only pieces of the path: WAV header walk, window plan, regions, CSV text
//   abi.hip       the C ABI (include/softspoken.h): argument checks, arena, getters, measurement
// Not part of the C ABI.
#pragma once
#include "../../include/softspoken.h"
#include "kernels.h"

#include <map>
#include <string>
#include <vector>

namespace ss {

// bytes of one sample of enum ss_pcm_format (WAV: little endian; AIFF: big endian, 7..12)
inline size_t pcm_bytes_per_sample(int format) {
    switch (format) {
        case SS_PCM_U8: case SS_PCM_S8: return 1;
        case SS_PCM_S16: case SS_PCM_S16BE: return 2;
        case SS_PCM_S24: case SS_PCM_S24BE: return 3;
        case SS_PCM_F64: case SS_PCM_F64BE: return 8;
        default: return 4;
    }
}

// ---- errors --------------------------------------------------------------------------------------------
int fail(ss_ctx* c, int code, const std::string& msg);       // records the message (context + calling thread), returns code
const char* thread_error();

#define HIPCHK(c, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return ss::fail((c), SS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

// Development switches (tools/, tests of alternate kernel forms): compiled into libsoftspoken_hip_dev.so only (-DSS_DEVBUILD).
// The product library has every default fixed at build time and reads two environment variables in all: SOFTSPOKEN_CHUNK and
// SOFTSPOKEN_PRECISION.
#ifdef SS_DEVBUILD
int dev_env(const char* name, int dflt);
#else
inline int dev_env(const char*, int dflt) { return dflt; }
#endif

// ---- weights blob ("SSWBLOB1") ---------------------------------------------------------------------------
struct BlobEntry { char name[96]; uint32_t dtype, ndim; int64_t shape[4]; uint64_t offset, nbytes; };
static_assert(sizeof(BlobEntry) == 152, "blob entry layout");
struct Blob {
    std::map<std::string, BlobEntry> e;
    const char* base = nullptr;
    size_t size = 0;
    bool has(const std::string& k) const { return e.count(k) != 0; }
    const float* f32(const std::string& k, size_t count, std::string& err) const;
};
bool parse_blob(const void* p, size_t n, Blob& b, std::string& err);

// ---- context ---------------------------------------------------------------------------------------------
enum Precision { kFp32 = 0, kBf16 = 1, kF16x2 = 2 };

struct ConvPlan {          // one launch of a ResBlock half
    std::string name;
    // conv2.hip (fp32; also the bf16 fallback): A = conv1 + residual projection (10 taps per chunk), B = conv2 only
    void* d_w2 = nullptr; float* d_bias2 = nullptr; float* d_res_bias = nullptr; float* d_rank1 = nullptr;
    // conv4.hip "projection in B": A = conv1 alone (9 taps per chunk); B = conv2 + the block's 1x1 projection of its own
    // input, weights in MFMA A-operand order per 16-channel step, bias b2 + br
    void* d_w3 = nullptr; void* d_proj = nullptr; float* d_bias3 = nullptr;
    void* d_w_ups = nullptr;                              // conv4_ups.hip: A's banks with the upsampled input half as four taps per parity class
    bool s1_range_proven = false;                         // conv1s.hip: |h1| and |c1| bounded below the f16 limit by the weights alone (weights.hip)
    void* d_w_s1 = nullptr;                               // conv1s.hip: conv1_1's second conv, banks in the K order of the first conv's accumulator registers
    void* d_w_s16 = nullptr;                              // ... and for its 16-pixel form (v_mfma_f32_16x16x32_f16: natural K order, channel rows 8 g + 4 u + r)
    void* d_w_upsr = nullptr;                             // conv4_ups.hip, ring form: the same for the A launch that also writes r (entries in walk order)
    void* d_w_ups32 = nullptr;                            // conv2_ups.hip (fp32): the A launch with the upsampled half at low resolution
    int ups32_nt = 1;                                     // ... packed for this many 32-channel tiles per block
    int Cout = 0, NT = 1, C0 = 0, C1 = 0, R0 = 0, R1 = 0, H = 0, W = 0;
};

struct FileRec {
    int64_t off = 0;        // arena offset of the padded signal
    int64_t n = 0;          // samples at 22 050 Hz (unpadded)
    int64_t n_padded = 0;
    double duration = 0;    // header duration in seconds (frames / sample_rate)
    int64_t W = 0, win_base = 0;                          // plan of the last ss_run_begin
    int64_t bin_off = 0; int n_bins = 0;
};

struct KStat { std::string name; int64_t launches = 0; double ms = 0, flops = 0, bytes = 0, issued = 0; };
struct PendingEvt { int sid; hipEvent_t a, b; };

}  // namespace ss

struct ss_ctx {
    int device = 0;
    uint32_t flags = 0;
    ss::Precision prec = ss::kFp32;
    bool bf16 = false;                                    // storage element is 2 bytes wide in the conv stack (bf16 mode)
    bool profile = false, has_model = false;
    hipStream_t stream = nullptr;
    // ingest: host -> device copies of the NEXT job's files run here, beside the compute stream's kernels; ev_copy is recorded behind
    // the last copy enqueued, and the next ss_add_pcm*_device makes the compute stream wait for it (copy_pending)
    hipStream_t copy_stream = nullptr; hipEvent_t ev_copy = nullptr; bool copy_pending = false;
    std::string err;
    int chunk = 1024;                                      // most windows per pass of the network
    int num_cus = 256;

    // tables + weights on device
    float4* d_pretw = nullptr; float2* d_w2048 = nullptr;
    int *d_mel_start = nullptr, *d_mel_count = nullptr, *d_mel_off = nullptr; float* d_mel_w = nullptr; float* d_mel_wp = nullptr; int mel_nw = 0;
    float2 *d_win2 = nullptr, *d_twt = nullptr, *d_wkt = nullptr; float* d_mel_wq = nullptr; int* d_mel_p0 = nullptr;   // second front-end kernel
    float *d_first_w = nullptr, *d_first_b = nullptr;
    float* d_flat_b = nullptr; void* d_flat_frag = nullptr; void* d_flat_frag4 = nullptr;
    int flat_groups = 0;                                  // row groups the last FLAT launch wrote per window
    float *d_spec_w = nullptr, *d_spec_b = nullptr;
    ss::Head1dWeights head{};
    std::vector<ss::ConvPlan> convs;  // in launch order; pairs (A, B) per ResBlock, conv1_1 has only B
    std::vector<void*> owned;        // device allocations to free
    std::vector<void*> user_dev, user_host;   // ss_device_alloc / ss_host_alloc memory the caller has not freed: released with the context

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
    bool run_pending = false; double pend_thr = 0, pend_brk = 0; std::vector<ss::AvgFile> pend_af;
    double t_in = 0, t_plan = 0, t_sync = 0, t_loop = 0;
    // progress of the run in flight (ss_run_begin_tracked): an event behind every pass and the windows done at it
    std::vector<hipEvent_t> pass_ev; std::vector<int64_t> pass_done_at; size_t pass_reported = 0; int64_t progress_reported = 0, track_total = 0;
    // activation workspace for `ws_chunk` windows
    int ws_chunk = 0;
    std::map<std::string, void*> act;                     // tensor name -> first byte (high plane in f16x2 mode) inside d_act_arena
    void* d_act_arena = nullptr;
    int64_t lo_delta = 0;                                 // f16x2: byte distance from a tensor's high plane to its low plane
    float* d_feat = nullptr; float* d_flat_part = nullptr;
    int64_t ws_bytes = 0;
    // second lane of a run with several passes: its own workspace and stream, so that a pass's launches fill the CUs that the other lane's
    // launch tails leave idle (engine.hip run_begin); allocated when such a run first asks, dropped with the workspace
    struct Lane {
        int chunk = 0; int64_t bytes = 0;
        std::map<std::string, void*> act; void* arena = nullptr; int64_t lo_delta = 0; float* feat = nullptr; float* flat = nullptr;
        hipStream_t stream = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;
    } lane1;
    int* d_range_flag = nullptr; int* h_range_flag = nullptr;    // f16x2: set by the conv kernels when a value does not fit an f16
    int fail_alloc_after = -1;                            // dev build's test hook (ss_debug_fail_workspace_alloc): the n-th workspace allocation from now fails
    bool split_range_ok = true;                           // f16x2: cleared while packing when a folded weight has no f16 representation

    // arena
    float* d_arena = nullptr; size_t arena_cap = 0, arena_used = 0;
    uint64_t reset_gen = 0;                               // bumped by every ss_reset (callers caching file ids compare it)
    std::vector<ss::FileRec> files;
    void* d_pcm = nullptr; size_t pcm_cap = 0;
    float* d_sx = nullptr; size_t sx_cap = 0;                    // review-screen spectrogram: samples in, magnitudes out
    float* d_sm = nullptr; size_t sm_cap = 0;
    short* d_sil_out = nullptr; size_t sil_out_cap = 0;          // silencer output / frame ranges
    int64_t* d_sil_ranges = nullptr; size_t sil_ranges_cap = 0;
    float* d_mono = nullptr; size_t mono_cap = 0;
    ss::BatchFile* d_batch = nullptr; size_t batch_cap = 0;
    std::map<std::pair<int, int>, std::pair<float*, int>> taps;   // (sr_in) -> device taps, half

    // run state
    int64_t* d_winoff = nullptr; size_t winoff_cap = 0;
    float* d_logits = nullptr; size_t logits_cap = 0;
    float* d_spec = nullptr; size_t spec_cap = 0;
    double* d_avg = nullptr; int32_t* d_count = nullptr; size_t avg_cap = 0;
    int32_t* d_starts = nullptr; size_t starts_cap = 0;
    ss::AvgFile* d_avgfiles = nullptr; size_t avgfiles_cap = 0;
    bool logits_valid = false; int64_t total_windows = 0;
    hipEvent_t ev_run0 = nullptr, ev_run1 = nullptr; double last_run_ms = 0;

    // profiling
    std::vector<ss::KStat> stats; std::vector<ss::PendingEvt> pending; std::vector<hipEvent_t> evpool;
};

namespace ss {

struct ScopedLaunch {      // times one launch with HIP events on the context's stream when profiling
    ss_ctx* c; int sid; hipEvent_t a = nullptr, b = nullptr;
    // issued_macs: multiply-adds of the form that runs, when it differs from the layer's algorithmic count (< 0: flops / 2)
    ScopedLaunch(ss_ctx* c_, const std::string& name, double flops, double bytes, double issued_macs = -1.0);
    ~ScopedLaunch();
};
void resolve_events(ss_ctx* c);

template <typename T>
int dev_upload(ss_ctx* c, T** dst, const void* src, size_t bytes) {
    void* p = nullptr;
    HIPCHK(c, hipMalloc(&p, bytes ? bytes : 16));
    c->owned.push_back(p);
    if (bytes) HIPCHK(c, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = (T*)p;
    return SS_OK;
}

// grow-only device buffer (1.5x); keep = carry the old contents over
template <typename T>
int ensure(ss_ctx* c, T** p, size_t* cap, size_t need_elems, bool keep = false) {
    if (need_elems <= *cap && *p) return SS_OK;
    size_t ncap = need_elems > *cap + *cap / 2 ? need_elems : *cap + *cap / 2;
    void* np = nullptr;
    size_t nb = ncap * sizeof(T);
    HIPCHK(c, hipMalloc(&np, nb > 256 ? nb : 256));
    if (keep && *p && *cap) {
        HIPCHK(c, hipMemcpyAsync(np, *p, *cap * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (*p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(*p)); }
    *p = (T*)np; *cap = ncap;
    return SS_OK;
}

// weights.hip
int build_tables(ss_ctx* c, const Blob& bl);
int build_model(ss_ctx* c, const Blob& bl);
// engine.hip
int ensure_workspace(ss_ctx* c, int n);
void free_workspace(ss_ctx* c);
int forward_chunk(ss_ctx* c, const int64_t* d_winoff, int n, float* d_logits, float* d_spec, float* d_feat_out);
int run_begin(ss_ctx* c, double threshold, double break_s, bool track, const volatile int* stop_flag,
              const float* ext_logits = nullptr, int64_t ext_windows = 0);
int run_poll(ss_ctx* c, ss_progress_fn progress, void* user, int block, const volatile int* stop_flag);
int run_end(ss_ctx* c);
void ensure_regions(ss_ctx* c);
int upload_winoff(ss_ctx* c, const std::vector<int64_t>& off);
// host.hip
double bin_time(int64_t idx);                             // float(f"{idx / (256 / 3):.4f}")
std::vector<int64_t> silence_ranges(const ss_region* regions, int64_t n, int sr, int64_t frames);
double now_ms();

}  // namespace ss
