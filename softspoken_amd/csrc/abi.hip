// The C ABI of libsoftspoken_hip.so (include/softspoken.h): argument checks, context lifecycle, the signal arena, compute entry
// points, result getters, measurement.  The work itself is in weights.hip (ss_create), engine.hip (launch sequence, job halves),
// host.hip (host-only helpers) and the kernel files.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace ss;

extern "C" int ss_abi_version(void) { return SS_ABI_VERSION; }

extern "C" const char* ss_last_error(const ss_ctx* ctx) { return ctx ? ctx->err.c_str() : thread_error(); }

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
    c->device = device_id; c->flags = flags; c->profile = (flags & SS_FLAG_PROFILE) != 0;
    if ((flags & SS_FLAG_BF16) && (flags & SS_FLAG_F16X2)) { delete c; return fail(nullptr, SS_ERR_ARG, "ss_create: SS_FLAG_BF16 and SS_FLAG_F16X2 exclude each other"); }
    c->prec = (flags & SS_FLAG_BF16) ? kBf16 : (flags & SS_FLAG_F16X2) ? kF16x2 : kFp32;
    c->bf16 = c->prec == kBf16;
    auto bail = [&](int rc) { std::string m = c->err; ss_destroy(c); fail(nullptr, rc, m); return rc; };
    if (hipSetDevice(device_id) != hipSuccess) return bail(fail(c, SS_ERR_HIP, "hipSetDevice failed"));
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return bail(fail(c, SS_ERR_HIP, "hipGetDeviceProperties failed"));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return bail(fail(c, SS_ERR_HIP, std::string("ss_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName));
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(c, SS_ERR_HIP, "hipStreamCreate failed"));
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(c, SS_ERR_HIP, "hipStreamCreate (copy stream) failed"));
    if (hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) != hipSuccess) return bail(fail(c, SS_ERR_HIP, "hipEventCreate failed"));
    hipEventCreate(&c->ev_run0); hipEventCreate(&c->ev_run1);
    if (weights_blob) {
        Blob bl; std::string err;
        if (!parse_blob(weights_blob, nbytes, bl, err)) return bail(fail(c, SS_ERR_FORMAT, err));
        int rc;
        if ((rc = build_tables(c, bl))) return bail(rc);
        if ((rc = build_model(c, bl))) return bail(rc);
        c->has_model = true;
        if (c->prec == kF16x2) {
            if (hipMalloc((void**)&c->d_range_flag, 4) != hipSuccess || hipMemset(c->d_range_flag, 0, 4) != hipSuccess ||
                hipHostMalloc((void**)&c->h_range_flag, 4, hipHostMallocDefault) != hipSuccess)
                return bail(fail(c, SS_ERR_HIP, "ss_create: range flag allocation failed"));
            *c->h_range_flag = 0;
        }
    }   // else: audio-only context (decode / mixdown / resample), every model entry point reports SS_ERR_STATE
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* ev = getenv("SOFTSPOKEN_CHUNK")) { int v = atoi(ev); if (v > 0) c->chunk = v; }
    *out = c;
    return SS_OK;
}

extern "C" void ss_destroy(ss_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->lane1.stream) hipStreamSynchronize(c->lane1.stream);
    resolve_events(c);
    for (void* p : c->owned) hipFree(p);
    for (void* p : c->user_dev) hipFree(p);
    for (void* p : c->user_host) hipHostFree(p);
    free_workspace(c);
    for (auto& kv : c->taps) hipFree(kv.second.first);
    void* singles[] = {c->d_arena, c->d_pcm, c->d_mono, c->d_winoff, c->d_logits, c->d_spec, c->d_avg, c->d_count, c->d_starts, c->d_avgfiles, c->d_batch, c->d_sil_out, c->d_sil_ranges, c->d_sx, c->d_sm};
    for (void* p : singles) if (p) hipFree(p);
    for (hipEvent_t ev : c->evpool) hipEventDestroy(ev);
    for (hipEvent_t ev : c->pass_ev) hipEventDestroy(ev);
    if (c->ev_run0) hipEventDestroy(c->ev_run0);
    if (c->ev_run1) hipEventDestroy(c->ev_run1);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->lane1.stream) hipStreamDestroy(c->lane1.stream);
    if (c->lane1.ev_in) hipEventDestroy(c->lane1.ev_in);
    if (c->lane1.ev_out) hipEventDestroy(c->lane1.ev_out);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->ev_copy) hipEventDestroy(c->ev_copy);
    if (c->h_above) hipHostFree(c->h_above);
    if (c->h_cov) hipHostFree(c->h_cov);
    if (c->r_above) hipHostFree(c->r_above);
    if (c->r_cov) hipHostFree(c->r_cov);
    if (c->d_range_flag) hipFree(c->d_range_flag);
    if (c->h_range_flag) hipHostFree(c->h_range_flag);
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
    c->files.clear(); c->arena_used = 0; c->logits_valid = false; c->total_windows = 0; ++c->reset_gen;
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
    if ((!pcm && frames > 0) || format < SS_PCM_U8 || format > SS_PCM_F64BE || sr <= 0 || sr > 768000 || ch < 1 || ch > 64 || frames < 0 ||
        frames > ((int64_t)1 << 36))          // (99 h at 192 kHz; keeps frames * channels * bytes and frames * 22050 inside 64 bits)
        return fail(c, SS_ERR_ARG, "ss_add_pcm: bad argument");
    return SS_OK;
}

// voice_activity.py:32-69 (decode -> mono -> resample) + worker.py:58-62 (pad)
extern "C" int ss_add_pcm(ss_ctx* c, const void* pcm, int format, int sr, int ch, int64_t frames, int* file_id) {
    int rc = check_pcm_args(c, pcm, format, sr, ch, frames);
    if (rc) return rc;
    hipSetDevice(c->device);
    const size_t bps = pcm_bytes_per_sample(format);
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
extern "C" int ss_silence_pcm(ss_ctx* c, const void* pcm, int format, int sr, int ch, int64_t frames, const ss_region* regions,
                              int64_t n_regions, int16_t* out) {
    int rc = check_pcm_args(c, pcm, format, sr, ch, frames);
    if (rc) return rc;
    if ((!regions && n_regions > 0) || n_regions < 0 || (!out && frames > 0)) return fail(c, SS_ERR_ARG, "ss_silence_pcm: bad argument");
    if (frames == 0) return SS_OK;
    hipSetDevice(c->device);
    const size_t bps = pcm_bytes_per_sample(format);
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
    if (c->copy_pending) {                                // ingest: the samples may still be crossing PCIe on the copy stream
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copy, 0));
        c->copy_pending = false;
    }
    const size_t bps = pcm_bytes_per_sample(format);
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
        int L, M, half; float* d_taps;
        if ((rc = get_taps(c, sr, L, M, half, &d_taps))) return rc;
        double n22sum = 0; for (auto& b : bf) n22sum += (double)b.n_out;
        if (resample_fused_applies(L, M, half) && dev_env("SOFTSPOKEN_RES3", 1)) {
            // decode + mixdown inside the resampler's LDS staging: one launch, no mono tensor
            ScopedLaunch sl(c, "resample_fused", 2.0 * 2 * half * n22sum, pcm_bytes + 4.0 * n22sum);
            HIPCHK(c, launch_resample_fused(pcm_dev, format, ch, c->d_batch, n_files, max_out, L, M, half, d_taps, c->d_arena, c->num_cus, c->stream));
        } else {
            if ((rc = ensure(c, &c->d_mono, &c->mono_cap, (size_t)mono_off + 16))) return rc;
            {
                ScopedLaunch sl(c, "decode_mono_batch", 0.0, pcm_bytes + 4.0 * total_frames);
                HIPCHK(c, launch_decode_mono_batch(pcm_dev, format, ch, c->d_batch, n_files, max_frames, c->d_mono, c->stream));
            }
            ScopedLaunch sl(c, "resample_batch", 2.0 * 2 * half * n22sum, 4.0 * total_frames + 4.0 * n22sum);
            HIPCHK(c, launch_resample_batch(c->d_mono, c->d_batch, n_files, max_out, L, M, half, d_taps, c->d_arena, c->num_cus, c->stream));
        }
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
    c->user_dev.push_back(*p);
    return SS_OK;
}
extern "C" int ss_device_free(ss_ctx* c, void* p) {
    if (!c) return fail(c, SS_ERR_ARG, "null context");
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto it = std::find(c->user_dev.begin(), c->user_dev.end(), p);
    if (it == c->user_dev.end()) return fail(c, SS_ERR_ARG, "ss_device_free: not a pointer ss_device_alloc of this context returned");
    c->user_dev.erase(it);
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
// ingest (worker.py:57 -> voice_activity.py:37 reads one file at a time in front of its batches): the next job's files cross PCIe
// on the copy stream while the job in flight computes
// ------------------------------------------------------------------------------------------------------
extern "C" int ss_host_alloc(ss_ctx* c, size_t nbytes, void** p) {
    if (!c || !p) return fail(c, SS_ERR_ARG, "ss_host_alloc: null argument");
    hipSetDevice(c->device);
    hipError_t e = hipHostMalloc(p, nbytes ? nbytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) { *p = nullptr; return fail(c, SS_ERR_NOMEM, std::string("ss_host_alloc: ") + hipGetErrorString(e)); }
    c->user_host.push_back(*p);
    return SS_OK;
}
extern "C" int ss_host_free(ss_ctx* c, void* p) {
    if (!c) return fail(c, SS_ERR_ARG, "null context");
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    auto it = std::find(c->user_host.begin(), c->user_host.end(), p);
    if (it == c->user_host.end()) return fail(c, SS_ERR_ARG, "ss_host_free: not a pointer ss_host_alloc of this context returned");
    c->user_host.erase(it);
    HIPCHK(c, hipHostFree(p));
    return SS_OK;
}
extern "C" int ss_device_upload_async(ss_ctx* c, void* dst, const void* src, size_t nbytes) {
    if (!c || !dst || !src) return fail(c, SS_ERR_ARG, "ss_device_upload_async: null argument");
    hipSetDevice(c->device);
    if (nbytes) HIPCHK(c, hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(c, hipEventRecord(c->ev_copy, c->copy_stream));
    c->copy_pending = true;
    return SS_OK;
}
extern "C" int ss_upload_wav_batch_async(ss_ctx* c, const void* const* files, const size_t* nbytes, int n_files, void* dev_dst, size_t cap,
                                         ss_wav_info* infos) {
    if (!c || !files || !nbytes || !infos || !dev_dst || n_files < 1) return fail(c, SS_ERR_ARG, "ss_upload_wav_batch_async: bad argument");
    // every header first: nothing is enqueued for a job with a bad file
    size_t total = 0;
    for (int i = 0; i < n_files; ++i) {
        const int rc = ss_wav_parse(files[i], nbytes[i], &infos[i]);
        if (rc) return fail(c, rc, "file " + std::to_string(i) + " of the batch: " + thread_error());
        total += (size_t)infos[i].frames * infos[i].channels * (infos[i].bits / 8);
    }
    if (total > cap) return fail(c, SS_ERR_CAPACITY, "ss_upload_wav_batch_async: " + std::to_string(total) + " bytes of samples, capacity " + std::to_string(cap));
    hipSetDevice(c->device);
    size_t at = 0;
    for (int i = 0; i < n_files; ++i) {
        const size_t nb = (size_t)infos[i].frames * infos[i].channels * (infos[i].bits / 8);
        if (nb) HIPCHK(c, hipMemcpyAsync((char*)dev_dst + at, (const char*)files[i] + infos[i].data_offset, nb, hipMemcpyHostToDevice, c->copy_stream));
        at += nb;
    }
    HIPCHK(c, hipEventRecord(c->ev_copy, c->copy_stream));
    c->copy_pending = true;
    return SS_OK;
}
extern "C" int ss_upload_wait(ss_ctx* c) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
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
    if (c->d_range_flag) HIPCHK(c, hipMemsetAsync(c->d_range_flag, 0, 4, c->stream));
    for (int i0 = 0; i0 < n; i0 += ch) {
        const int m = std::min(ch, n - i0);
        if ((rc = forward_chunk(c, c->d_winoff + i0, m, c->d_logits + (size_t)i0 * 256, spec_out ? c->d_spec : nullptr, nullptr))) return rc;
        if (spec_out) {
            HIPCHK(c, hipMemcpyAsync(spec_out + (size_t)i0 * 2 * 32768, c->d_spec, (size_t)m * 2 * 32768 * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    HIPCHK(c, hipMemcpyAsync(mask_out, c->d_logits, (size_t)n * 256 * 4, hipMemcpyDeviceToHost, c->stream));
    if (c->d_range_flag) HIPCHK(c, hipMemcpyAsync(c->h_range_flag, c->d_range_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->h_range_flag && *c->h_range_flag)
        return fail(c, SS_ERR_RANGE, "f16x2: an activation left the f16 range (|x| > 65504) or was not finite; run this checkpoint with the fp32 mode");
    return SS_OK;
}
extern "C" int ss_run(ss_ctx* c, double threshold, double break_s, ss_progress_fn progress, void* user, const volatile int* stop_flag) {
    int rc = run_begin(c, threshold, break_s, progress != nullptr, stop_flag);
    if (rc) return rc;
    if (progress && (rc = run_poll(c, progress, user, 1, stop_flag))) {
        const std::string msg = rc == SS_ERR_STOPPED ? "stopped on request" : c->err;
        run_end(c);                                       // (waits for what is enqueued; the results are discarded)
        return fail(c, rc, msg);
    }
    rc = run_end(c);
    if (!rc && progress && stop_flag && *stop_flag) return fail(c, SS_ERR_STOPPED, "stopped on request");
    return rc;
}

extern "C" int ss_run_begin(ss_ctx* c, double threshold, double break_s) { return run_begin(c, threshold, break_s, false, nullptr); }
extern "C" int ss_run_begin_tracked(ss_ctx* c, double threshold, double break_s) { return run_begin(c, threshold, break_s, true, nullptr); }
extern "C" int ss_run_poll(ss_ctx* c, ss_progress_fn progress, void* user, int block) { return run_poll(c, progress, user, block, nullptr); }

extern "C" int ss_run_from_logits(ss_ctx* c, const float* logits, int64_t n_windows, double threshold, double break_s) {
    if (!c || n_windows < 0 || (!logits && n_windows > 0)) return fail(c, SS_ERR_ARG, "ss_run_from_logits: bad argument");
    static const float none = 0.f;
    const int rc = run_begin(c, threshold, break_s, false, nullptr, logits ? logits : &none, n_windows);
    return rc ? rc : run_end(c);
}

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
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
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
        out[i].issued_flops = c->stats[i].issued;
    }
    return SS_OK;
}

extern "C" double ss_last_run_device_ms(ss_ctx* c) { return c ? c->last_run_ms : -1.0; }

extern "C" uint64_t ss_reset_generation(ss_ctx* c) { return c ? c->reset_gen : 0; }

#ifdef SS_DEVBUILD
extern "C" int ss_debug_fail_workspace_alloc(ss_ctx* c, int nth) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    c->fail_alloc_after = nth < 0 ? -1 : nth;
    return SS_OK;
}
#endif

extern "C" int64_t ss_workspace_bytes(ss_ctx* c) { return c ? c->ws_bytes + c->lane1.bytes : -1; }
