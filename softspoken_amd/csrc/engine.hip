// Plan + launch unit: the activation workspace, the launch sequence of SpecUNet_2D for one chunk of windows, and the two
// halves of a job (plan + enqueue, wait).  Reference files are cited per function (paths relative to the reference root).
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ss {

// ------------------------------------------------------------------------------------------------------
// errors, profiling events
// ------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

int fail(ss_ctx* c, int code, const std::string& msg) {
    g_err = msg;
    if (c) c->err = msg;
    return code;
}
const char* thread_error() { return g_err.c_str(); }

#ifdef SS_DEVBUILD
int dev_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#endif

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int stat_id(ss_ctx* c, const std::string& name) {
    for (size_t i = 0; i < c->stats.size(); ++i) if (c->stats[i].name == name) return (int)i;
    KStat k; k.name = name; c->stats.push_back(k);
    return (int)c->stats.size() - 1;
}

ScopedLaunch::ScopedLaunch(ss_ctx* c_, const std::string& name, double flops, double bytes, double issued_macs) : c(c_) {
    sid = stat_id(c, name);
    c->stats[sid].launches++; c->stats[sid].flops += flops; c->stats[sid].bytes += bytes;
    c->stats[sid].issued += 2.0 * (c->prec == kF16x2 ? 3.0 : 1.0) * (issued_macs >= 0 ? issued_macs : flops / 2.0);
    if (c->profile) {
        auto get = [&]() { hipEvent_t e; if (!c->evpool.empty()) { e = c->evpool.back(); c->evpool.pop_back(); } else hipEventCreate(&e); return e; };
        a = get(); b = get();
        hipEventRecord(a, c->stream);
    }
}
ScopedLaunch::~ScopedLaunch() {
    if (c->profile) { hipEventRecord(b, c->stream); c->pending.push_back({sid, a, b}); }
}

void resolve_events(ss_ctx* c) {
    for (auto& p : c->pending) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->stats[p.sid].ms += ms;
        c->evpool.push_back(p.a); c->evpool.push_back(p.b);
    }
    c->pending.clear();
}

// ------------------------------------------------------------------------------------------------------
// activation workspace: NHWC tensors for `n` windows.  Every tensor is preceded by a 256-byte zero header (conv4.hip reads it
// for out-of-image patch pieces).  f16x2 mode keeps two planes per tensor (high and low halves of every value, both f16) in two
// arenas of identical layout, so that one byte distance (ss_ctx::lo_delta) leads from any high plane to its low plane.
// ------------------------------------------------------------------------------------------------------
static constexpr size_t kActHeader = 256;

static void free_lane1(ss_ctx* c) {
    ss_ctx::Lane& L = c->lane1;
    L.chunk = 0; L.bytes = 0; L.act.clear(); L.lo_delta = 0;
    void* arena = L.arena; float* feat = L.feat; float* fp = L.flat;
    L.arena = nullptr; L.feat = nullptr; L.flat = nullptr;
    if (arena) hipFree(arena);
    if (feat) hipFree(feat);
    if (fp) hipFree(fp);
}

void free_workspace(ss_ctx* c) {
    // pointers are cleared BEFORE anything else can fail: a context whose growth failed holds no workspace at all (ws_chunk = 0)
    // and the next call allocates afresh -- never a stale ws_chunk over freed tensors
    c->ws_chunk = 0; c->ws_bytes = 0;
    void* arena = c->d_act_arena; float* feat = c->d_feat; float* fp = c->d_flat_part;
    c->d_act_arena = nullptr; c->d_feat = nullptr; c->d_flat_part = nullptr; c->lo_delta = 0;
    c->act.clear();
    if (arena) hipFree(arena);
    if (feat) hipFree(feat);
    if (fp) hipFree(fp);
    free_lane1(c);
}

static hipError_t ws_malloc(ss_ctx* c, void** p, size_t bytes) {
    if (c->fail_alloc_after >= 0 && c->fail_alloc_after-- == 0) { *p = nullptr; return hipErrorOutOfMemory; }   // test hook
    return hipMalloc(p, bytes);
}

// one workspace for n windows: arena (+ the tensor table), feature and flatten-partial buffers; the zero headers are written on `stream`.
// On failure everything it allocated is freed again and *what names the step.
struct WsAlloc { void* arena = nullptr; float* feat = nullptr; float* flat = nullptr; std::map<std::string, void*> act; int64_t lo_delta = 0, bytes = 0; };
static hipError_t alloc_ws(ss_ctx* c, int n, hipStream_t stream, WsAlloc& w, std::string& what) {
    const size_t es = c->prec == kFp32 ? 4 : 2;
    struct T { const char* n; int H, W, C; };
    const T ts[] = {{"h1", 128, 256, 32}, {"c1", 128, 256, 32}, {"p1", 64, 128, 32}, {"h2", 64, 128, 64}, {"c2", 64, 128, 64},
                    {"p2", 32, 64, 64},   {"h3", 32, 64, 96},   {"c3", 32, 64, 96},  {"p3", 16, 32, 96},  {"h4", 16, 32, 128},
                    {"c4", 16, 32, 128},  {"p4", 8, 16, 128},   {"hb", 8, 16, 128},  {"bott", 8, 16, 128}, {"he", 8, 16, 128},
                    {"enc", 8, 16, 128},  {"h6", 16, 32, 96},   {"c6", 16, 32, 96},  {"h7", 32, 64, 64},  {"c7", 32, 64, 64},
                    {"h8", 64, 128, 32},  {"c8", 64, 128, 32},  {"h9", 128, 256, 32}, {"c9", 128, 256, 32},
                    {"hs", 128, 256, 32}, {"s9", 128, 256, 32},
                    // r = residual projection written by A launches that keep it (conv6, conv8, the spec head; every block in fp32 / f16x2)
                    {"r2", 64, 128, 64},  {"r3", 32, 64, 96},   {"r4", 16, 32, 128}, {"rb", 8, 16, 128},  {"re", 8, 16, 128},
                    {"r6", 16, 32, 96},   {"r7", 32, 64, 64},   {"r8", 64, 128, 32}, {"r9", 128, 256, 32}, {"rs", 128, 256, 32}};
    // one arena, one size (the spec head's tensors included): the workspace is sized once per context and chunk
    size_t total = 0;
    std::vector<size_t> offs;
    for (const T& t : ts) {
        offs.push_back(total + kActHeader);
        total += kActHeader + (((size_t)n * t.H * t.W * t.C * es + 255) & ~(size_t)255);
    }
    const int planes = c->prec == kF16x2 ? 2 : 1;
    auto drop = [&]() { if (w.arena) hipFree(w.arena); if (w.feat) hipFree(w.feat); if (w.flat) hipFree(w.flat); w = WsAlloc(); };
    hipError_t e = ws_malloc(c, &w.arena, total * planes);
    if (e != hipSuccess) { what = "activation workspace (" + std::to_string(total * planes >> 20) + " MiB)"; drop(); return e; }
    if ((e = ws_malloc(c, (void**)&w.feat, (size_t)n * 128 * 256 * 4)) != hipSuccess ||
        (e = ws_malloc(c, (void**)&w.flat, (size_t)n * 64 * 4 * 256 * 4)) != hipSuccess) { what = "activation workspace"; drop(); return e; }
    for (int pl = 0; pl < planes; ++pl)
        for (size_t i = 0; i < offs.size(); ++i) {
            e = hipMemsetAsync((char*)w.arena + pl * total + offs[i] - kActHeader, 0, kActHeader, stream);
            if (e != hipSuccess) { what = "hipMemsetAsync"; drop(); return e; }
        }
    for (size_t i = 0; i < offs.size(); ++i) w.act[ts[i].n] = (char*)w.arena + offs[i];
    w.lo_delta = planes == 2 ? (int64_t)total : 0;
    w.bytes = (int64_t)(total * planes + (size_t)n * 128 * 256 * 4 + (size_t)n * 64 * 4 * 256 * 4);
    return hipSuccess;
}

int ensure_workspace(ss_ctx* c, int n) {
    if (n <= c->ws_chunk) return SS_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->lane1.stream) HIPCHK(c, hipStreamSynchronize(c->lane1.stream));
    free_workspace(c);
    WsAlloc w; std::string what;
    const hipError_t e = alloc_ws(c, n, c->stream, w, what);
    if (e != hipSuccess) return fail(c, what == "hipMemsetAsync" ? SS_ERR_HIP : SS_ERR_NOMEM, what + ": " + hipGetErrorString(e));
    c->d_act_arena = w.arena; c->d_feat = w.feat; c->d_flat_part = w.flat; c->act = std::move(w.act); c->lo_delta = w.lo_delta;
    c->ws_bytes = w.bytes;
    c->ws_chunk = n;
    return SS_OK;
}

// the second lane for passes of n windows; false (and no lane) when the memory is not there -- the run then uses one lane
static bool ensure_lane1(ss_ctx* c, int n) {
    ss_ctx::Lane& L = c->lane1;
    if (L.chunk >= n) return true;
    if (!L.stream) {
        if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess) { L.stream = nullptr; return false; }
        if (hipEventCreateWithFlags(&L.ev_in, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&L.ev_out, hipEventDisableTiming) != hipSuccess) return false;
    }
    hipStreamSynchronize(L.stream);
    free_lane1(c);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (int64_t)free_b < c->ws_bytes + ((int64_t)4 << 30)) return false;   // (leave room for the caller's buffers)
    WsAlloc w; std::string what;
    if (alloc_ws(c, n, L.stream, w, what) != hipSuccess) { (void)hipGetLastError(); return false; }
    L.arena = w.arena; L.feat = w.feat; L.flat = w.flat; L.act = std::move(w.act); L.lo_delta = w.lo_delta; L.bytes = w.bytes; L.chunk = n;
    return true;
}

// ------------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------------
struct ConvExtra { const float* first_w = nullptr; const float* first_b = nullptr; const void* flat_w = nullptr; const void* flat_w4 = nullptr; float* flat_part = nullptr; int store_out = 1; };

static int base_dbg() {
    // product build: 32 (raised wave priority inside conv4.hip's MFMA loop: the measured default); dev build: + SOFTSPOKEN_DBG bits
    return (dev_env("SOFTSPOKEN_PRIO", 1) ? 32 : 0) | dev_env("SOFTSPOKEN_DBG", 0);
}

#ifdef SS_DEVBUILD
// SOFTSPOKEN_STAMP_LAYER=<layer name>: segment times of that launch's stages (shader clock, summed per wave) on stderr
struct StageStamps {
    void* d_st = nullptr;
    static constexpr size_t st_bytes = (size_t)4096 * 8 * 16 * 4;
    int begin(ss_ctx* c, const std::string& name, ConvArgs& a) {
        const char* want = getenv("SOFTSPOKEN_STAMP_LAYER");
        if (want && name == want) { HIPCHK(c, hipMalloc(&d_st, st_bytes)); HIPCHK(c, hipMemsetAsync(d_st, 0, st_bytes, c->stream)); a.stamps = d_st; }
        return SS_OK;
    }
    int end(ss_ctx* c, const std::string& name, int n) {
        if (!d_st) return SS_OK;
        std::vector<uint32_t> h(st_bytes / 4);
        HIPCHK(c, hipMemcpyAsync(h.data(), d_st, st_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        double sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stages = 0; int waves = 0;
        for (size_t w = 0; w < h.size() / 16; ++w) if (h[w * 16 + 5]) {
            ++waves; stages += h[w * 16 + 5];
            for (int i = 0; i < 5; ++i) sum[i] += h[w * 16 + i];
            for (int i = 5; i < 12; ++i) sum[i] += h[w * 16 + i + 1];
        }
        if (waves)
            fprintf(stderr, "[stamps] %s n=%d: %d waves, %.1f stages/wave; cycles per stage: mfma %.0f | barrier1 %.0f | commit+issue %.0f (wait for loads %.0f, LDS writes %.0f, next stage %.0f, patch loads %.0f, one stamp %.0f) | barrier2 %.0f | epilogue %.0f (f16x2: residual add incl. its wait %.0f, the last stage's work %.0f)\n",
                    name.c_str(), n, waves, stages / waves, sum[0] / stages, sum[1] / stages, sum[2] / stages, sum[5] / stages, sum[6] / stages, sum[7] / stages, sum[8] / stages, sum[9] / stages, sum[3] / stages, sum[4] / stages, sum[10] / stages, sum[11] / stages);
        hipFree(d_st); d_st = nullptr;
        return SS_OK;
    }
};
#endif

// One launch of a ResBlock half.  A launches (r_out) compute h and the residual projection r from the block input
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
    a.lo_delta = c->lo_delta; a.range_flag = c->d_range_flag;
    if (isA) { a.C0 = p.C0; a.C1 = p.C1; } else { a.C0 = p.Cout; a.C1 = 0; }       // B's 3x3 input is h
    a.dbg = base_dbg();
    const double cin = a.C0 + a.C1;
    const double macs = (double)n * p.H * p.W * p.Cout * (9.0 * cin + (isA ? cin : 0.0) + (feat ? 1 : 0)) +
                        (ex.first_w ? (double)n * p.H * p.W * 32 * 9 : 0.0) + (ex.flat_part ? (double)n * p.H * p.W * 32 * 4 : 0.0);
    const double es = c->prec == kBf16 ? 2 : 4;           // bytes per stored activation value (f16x2: two f16 planes)
    // (the fused conv1_1 launch reads the fp32 features, not an h1 tensor: h1 only exists in LDS)
    const double bytes = (double)n * p.H * p.W * es * ((ex.first_w ? 4.0 / es : a.C0) + a.C1 / 4.0 + (ex.flat_part && !ex.store_out ? 0 : p.Cout) + (isA || r_in ? p.Cout : 0) + (pool ? p.Cout / 4.0 : 0));
    // stat name = "<instantiation as rocprofv3 prints it>/<layer>"
    const int prec4 = c->prec == kF16x2 ? 2 : 1;
#ifdef SS_DEVBUILD
    StageStamps stamps;
    if (int rcs = stamps.begin(c, p.name, a)) return rcs;
    auto print_stamps = [&]() -> int { return stamps.end(c, p.name, n); };
#endif
    if (isA && c->prec == kF16x2 && p.d_w_upsr) {  // decoder A launches: the upsampled input half at low resolution (conv4_ups.hip, ring form)
        ConvArgs au = a;
        au.wpk = p.d_w_upsr;
        if (conv_upsr_supports(au, c->num_cus)) {
            // (FLOPs booked: the layer's algorithmic ones, as for every launch; this form issues 9 C0 + 4 C1 multiply-adds per output value)
            {
                const double issued = (double)n * p.H * p.W * p.Cout * (9.0 * p.C0 + 4.0 * p.C1 + cin);     // (+ cin: the 1x1 projection)
                ScopedLaunch sl(c, std::string(conv_upsr_variant()) + "/" + p.name, 2.0 * macs, bytes, issued);
                HIPCHK(c, launch_conv3x3_upsr(au, c->num_cus, c->stream));
            }
#ifdef SS_DEVBUILD
            return print_stamps();
#else
            return SS_OK;
#endif
        }
    }
    if (isA && c->prec == kFp32 && p.d_w_ups32 && x1) {   // fp32 decoder A launches: the same idea on the fp32 matrix instruction (conv2_ups.hip)
        ConvArgs au = a;
        au.wpk = p.d_w_ups32;
        const int mtw = dev_env("SOFTSPOKEN_UPS32_MTW", 2) == 2 && p.ups32_nt < 3 ? 2 : 1;
        if (conv_ups32_supports(au, p.ups32_nt, mtw, c->num_cus)) {
            const double issued = (double)n * p.H * p.W * p.Cout * (9.0 * p.C0 + 4.0 * p.C1 + cin);
            ScopedLaunch sl(c, std::string(conv_ups32_variant(p.ups32_nt, mtw)) + "/" + p.name, 2.0 * macs, bytes, issued);
            HIPCHK(c, launch_conv3x3_ups32(au, p.ups32_nt, mtw, c->num_cus, c->stream));
            return SS_OK;
        }
    }
    if (c->prec == kF16x2 && ex.first_w && p.d_w_s16 && dev_env("SOFTSPOKEN_C1S", 1)) {   // conv1_1: the row-streaming form (conv1s.hip)
        ConvArgs as = a;
                const int form = dev_env("SOFTSPOKEN_C1S_FORM", 16) == 32 && p.d_w_s1 ? 32 : 16;     // (32: the development build's other form)
        as.wpk = form == 16 ? p.d_w_s16 : p.d_w_s1;
        as.relu = 1 | (dev_env("SOFTSPOKEN_C1S_ABLATE", 0) << 4);    // (dev build, timing only: 1 no stores, 2 no second-conv products, 4 no pooled rows)
        as.plain = p.s1_range_proven && dev_env("SOFTSPOKEN_C1S_TRACK", 0) == 0;
        if (conv1_stream_supports(as)) {
#ifdef SS_DEVBUILD
            as.stamps = a.stamps;
#endif
            {
                ScopedLaunch sl(c, std::string(conv1_stream_variant(as, form)) + "/" + p.name, 2.0 * macs, bytes);
                HIPCHK(c, launch_conv1_stream(as, form, dev_env("SOFTSPOKEN_C1S_ROWS", 32), c->num_cus, c->stream));
            }
#ifdef SS_DEVBUILD
            if (dev_env("SOFTSPOKEN_C1S", 1) != 2) return print_stamps();    // (2: conv4.hip's form runs as well, behind it -- an A/B aid)
#else
            return SS_OK;
#endif
        }
    }
    if (c->prec != kFp32 && dev_env("SOFTSPOKEN_CONV4", 1) && conv_v4_supports(a, p.NT, c->num_cus, prec4)) {   // conv4.hip: bf16 / f16x2 launches
        {
            ScopedLaunch sl(c, std::string(conv_v4_variant(a, p.NT, c->num_cus, prec4)) + "/" + p.name, 2.0 * macs, bytes);
            HIPCHK(c, launch_conv3x3_v4(a, p.NT, c->num_cus, prec4, c->stream));
        }
#ifdef SS_DEVBUILD
        { const int rcs = print_stamps(); if (rcs) return rcs; }
#endif
        if (ex.flat_part) c->flat_groups = conv_v4_flat_groups();
        return SS_OK;
    }
    if (c->prec == kF16x2) return fail(c, SS_ERR_STATE, "f16x2: no kernel form for " + p.name);
    if (ex.flat_part) c->flat_groups = conv_v2_flat_groups(c->bf16);
    ScopedLaunch sl(c, std::string(conv_v2_variant(a, c->bf16, p.NT, c->num_cus)) + "/" + p.name, 2.0 * macs, bytes);
    HIPCHK(c, launch_conv3x3_v2(a, c->bf16, p.NT, c->num_cus, c->stream));
    return SS_OK;
}

// A ResBlock in the "projection in B" form (conv4.hip RP; bf16, and f16x2 for the blocks conv4.hip has the form for): A writes h
// alone, B reads h and the centre pixels of the block input.  Returns 1 when conv4.hip has no instantiation for this block (the
// caller then uses A + r / B).
static int run_block_proj(ss_ctx* c, const ConvPlan& pa, const ConvPlan& pb, int n, const void* x0, const void* x1, void* h, void* out,
                          void* pool, const ConvExtra& ex = ConvExtra()) {
    if ((c->prec != kBf16 && c->prec != kF16x2) || !pa.d_w3 || !pb.d_proj) return 1;
    const int prec4 = c->prec == kF16x2 ? 2 : 1;
    const double es = c->prec == kBf16 ? 2 : 4;
    ConvArgs a{}, b{};
    a.lo_delta = b.lo_delta = c->lo_delta; a.range_flag = b.range_flag = c->d_range_flag;
    a.src0 = x0; a.src1 = x1; a.wpk = pa.d_w3; a.bias = pa.d_bias2; a.out = h; a.plain = 1;
    a.N = n; a.H = pa.H; a.W = pa.W; a.Cout = pa.Cout; a.C0 = pa.C0; a.C1 = pa.C1; a.relu = 1;
    b.src0 = h; b.wpk = pb.d_w2; b.bias = pb.d_bias3; b.out = out; b.pool_out = pool;
    b.N = n; b.H = pb.H; b.W = pb.W; b.Cout = pb.Cout; b.C0 = pb.Cout; b.C1 = 0; b.relu = 1;
    b.proj_w = pb.d_proj; b.xp0 = x0; b.xp1 = x1; b.C0x = pa.C0; b.C1x = pa.C1;
    b.flat_w4 = ex.flat_w4; b.flat_part = ex.flat_part; b.store_out = ex.store_out;
    a.dbg = b.dbg = base_dbg();
    // f16x2: the A launch with the upsampled input half at low resolution (conv4_ups.hip: four pre-summed taps per output parity class
    // instead of nine on those channels) where that form exists -- conv9_1.A
    ConvArgs au = a;
    au.wpk = pa.d_w_ups;
    const bool ups = c->prec == kF16x2 && pa.d_w_ups && x1 && conv_ups_supports(au, c->num_cus);
    if ((!ups && !conv_v4_supports(a, pa.NT, c->num_cus, prec4)) || !conv_v4_supports(b, pb.NT, c->num_cus, prec4)) return 1;
    const double px = (double)n * pa.H * pa.W, cin = pa.C0 + pa.C1, cinb = pa.C0 + pa.C1 / 4.0;
    if (ups) {
        // FLOPs booked are the layer's algorithmic ones (2 x multiply-adds of the 3x3 as the reference computes it); this
        // form issues 9 C0 + 4 C1 multiply-adds per output value instead of 9 (C0 + C1)
        ScopedLaunch sl(c, std::string(conv_ups_variant()) + "/" + pa.name, 2.0 * px * pa.Cout * 9.0 * cin, px * es * (cinb + pa.Cout),
                        px * pa.Cout * (9.0 * pa.C0 + 4.0 * pa.C1));
        HIPCHK(c, launch_conv3x3_ups(au, c->num_cus, c->stream));
    } else {
#ifdef SS_DEVBUILD
        StageStamps st;
        if (int rcs = st.begin(c, pa.name, a)) return rcs;
#endif
        {
            ScopedLaunch sl(c, std::string(conv_v4_variant(a, pa.NT, c->num_cus, prec4)) + "/" + pa.name, 2.0 * px * pa.Cout * 9.0 * cin, px * es * (cinb + pa.Cout));
            HIPCHK(c, launch_conv3x3_v4(a, pa.NT, c->num_cus, prec4, c->stream));
        }
#ifdef SS_DEVBUILD
        if (int rcs = st.end(c, pa.name, n)) return rcs;
#endif
    }
    {
        const double flops = 2.0 * px * pb.Cout * (9.0 * pb.Cout + cin) + (ex.flat_part ? 2.0 * px * 32 * 4 : 0.0);
        const double bytes = px * es * (pb.Cout + cinb + (ex.flat_part && !ex.store_out ? 0 : pb.Cout) + (pool ? pb.Cout / 4.0 : 0));
#ifdef SS_DEVBUILD
        StageStamps st;
        if (int rcs = st.begin(c, pb.name, b)) return rcs;
#endif
        {
            ScopedLaunch sl(c, std::string(conv_v4_variant(b, pb.NT, c->num_cus, prec4)) + "/" + pb.name, flops, bytes);
            HIPCHK(c, launch_conv3x3_v4(b, pb.NT, c->num_cus, prec4, c->stream));
        }
#ifdef SS_DEVBUILD
        if (int rcs = st.end(c, pb.name, n)) return rcs;
#endif
        if (ex.flat_part) c->flat_groups = conv_v4_flat_groups();
    }
    return SS_OK;
}

// SpecUNet_2D.forward (pytorch_neural_nets.py:142-197) for n <= ws_chunk windows whose arena offsets are d_winoff[0..n)
int forward_chunk(ss_ctx* c, const int64_t* d_winoff, int n, float* d_logits, float* d_spec, float* d_feat_out) {
    FrontendTables tb{c->d_pretw, c->d_w2048, c->d_mel_start, c->d_mel_count, c->d_mel_off, c->d_mel_w, c->mel_nw, c->d_mel_wp, dev_env("SOFTSPOKEN_FEDBG", 0),
                      c->d_win2, c->d_twt, c->d_wkt, c->d_mel_wq, c->d_mel_p0};
    float* feat = d_feat_out ? d_feat_out : c->d_feat;
    {
        ScopedLaunch sl(c, "frontend", 0.0, (double)n * (66150.0 * 4 + 128.0 * 256 * 4));
        HIPCHK(c, launch_frontend(c->d_arena, d_winoff, n, tb, feat, c->num_cus, c->stream));
    }
    if (!d_logits) return SS_OK;
    auto A = [&](const char* k) -> void* {
        auto it = c->act.find(k);
        return it == c->act.end() ? nullptr : it->second;      // (never null after ensure_workspace; a null would be refused by the launch checks)
    };
    const double es = c->prec == kBf16 ? 2 : 4;
    const std::vector<ConvPlan>& cv = c->convs;
    int rc, i = 0;
#define RC2(x) if ((rc = (x))) return rc
    {   // conv1_1: first conv produced in the loader (bf16: MFMA on bf16 features; f16x2: three MFMAs on f16 halves), 1 -> 32 residual
        // from the features (bf16: one more MFMA on hi/lo halves; f16x2: a rank-1 fp32 term in the epilogue; fp32: conv2.hip's own form)
        ConvExtra ex; ex.first_w = c->d_first_w; ex.first_b = c->d_first_b;
        RC2(run_conv2(c, cv[i++], n, nullptr, nullptr, A("c1"), A("p1"), nullptr, nullptr, feat, ex));
    }
    struct Blk { const char *x0, *x1, *h, *r, *y, *pool; };
    const Blk blks[] = {{"p1", nullptr, "h2", "r2", "c2", "p2"},   {"p2", nullptr, "h3", "r3", "c3", "p3"},
                        {"p3", nullptr, "h4", "r4", "c4", "p4"},   {"p4", nullptr, "hb", "rb", "bott", nullptr},
                        {"bott", nullptr, "he", "re", "enc", nullptr}, {"c4", "enc", "h6", "r6", "c6", nullptr},
                        {"c3", "c6", "h7", "r7", "c7", nullptr},   {"c2", "c7", "h8", "r8", "c8", nullptr}};
    const bool proj = c->prec != kFp32 && dev_env("SOFTSPOKEN_CONV4", 1) && dev_env("SOFTSPOKEN_RPROJ", 1);
    for (const Blk& b : blks) {
        // (running A and B over Infinity-Cache-sized sub-chunks of windows was measured twice: no gain)
        // Blocks whose input is narrower than their output (encoder) move fewer bytes when B recomputes the 1x1 projection from
        // the block input than when A writes r and B reads it back; conv4.hip has that form for the blocks where it pays.
        if (proj) {
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
        if (proj && (rc = run_block_proj(c, cv[i], cv[i + 1], n, A("c1"), A("c8"), A("h9"), A("c9"), nullptr, ex)) != 1) {
            if (rc) return rc;
        } else {
            RC2(run_conv2(c, cv[i], n, A("c1"), A("c8"), A("h9"), nullptr, A("r9"), nullptr, nullptr));
            RC2(run_conv2(c, cv[i + 1], n, A("h9"), nullptr, A("c9"), nullptr, nullptr, A("r9"), nullptr, ex));
        }
        i += 2;
    }
    if (d_spec) {   // dead head of the reference (worker.py:78-79 drops it), on request
        RC2(run_conv2(c, cv[i], n, A("c9"), nullptr, A("hs"), nullptr, A("rs"), nullptr, nullptr));
        RC2(run_conv2(c, cv[i + 1], n, A("hs"), nullptr, A("s9"), nullptr, nullptr, A("rs"), nullptr));
        ScopedLaunch sl(c, "spec_tail", 2.0 * n * 32768 * 64, (double)n * 32768 * (32 * es + 8));
        HIPCHK(c, launch_spec_tail(A("s9"), c->lo_delta, c->d_spec_w, c->d_spec_b, d_spec, n, (int)c->prec, c->stream));
    }
#undef RC2
    const int groups = c->flat_groups;
    ScopedLaunch sl(c, "mask_head_parts", 0.0, (double)n * (groups * 4 * 256 * 4 + 1024));
    HIPCHK(c, launch_mask_head_parts(c->d_flat_part, groups, c->d_flat_b, c->head, d_logits, n, c->stream));
    return SS_OK;
}

int upload_winoff(ss_ctx* c, const std::vector<int64_t>& off) {
    int rc = ensure(c, &c->d_winoff, &c->winoff_cap, off.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_winoff, off.data(), off.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));     // `off` is a host temporary
    return SS_OK;
}

// ------------------------------------------------------------------------------------------------------
// a job in two halves (worker.py:49-100 over every file of the arena): everything up to the last device -> host copy is
// enqueued by run_begin; run_end waits for it; the regions are found on the host when first asked for.
// ------------------------------------------------------------------------------------------------------
// device side of the post-processing (NNDetector.py:153-190 averaging, the comparison of :118): logits of `af` files -> two bit masks
static int enqueue_post(ss_ctx* c, const std::vector<AvgFile>& af, int64_t total, int64_t total_bins, int max_bins, double threshold) {
    int rc;
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
    (void)total; (void)max_bins; (void)threshold;
    return SS_OK;
}

static int launch_post(ss_ctx* c, size_t n_files, int64_t total, int64_t total_bins, int max_bins, double threshold) {
    const size_t words = (size_t)((total_bins + 255) / 256) * 4 + 1;
    {
        ScopedLaunch sl(c, "average", 0.0, (double)total * 1024 * 5 + (double)total_bins * 12);
        HIPCHK(c, launch_average(c->d_logits, c->d_avgfiles, (int)n_files, c->d_starts, c->d_avg, c->d_count, max_bins, c->stream));
    }
    if (total_bins) {
        ScopedLaunch sl(c, "bin_masks", 0.0, (double)total_bins * 12 + (double)words * 16);
        HIPCHK(c, launch_bin_masks(c->d_avg, c->d_count, total_bins, threshold, c->d_above, c->d_cov, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev_run1, c->stream));
    if (c->d_range_flag) HIPCHK(c, hipMemcpyAsync(c->h_range_flag, c->d_range_flag, 4, hipMemcpyDeviceToHost, c->stream));
    if (total_bins) {
        HIPCHK(c, hipMemcpyAsync(c->h_above, c->d_above, words * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_cov, c->d_cov, words * 8, hipMemcpyDeviceToHost, c->stream));
    }
    return SS_OK;
}

// ext_logits != nullptr: the windows' logits come from the caller (ss_run_from_logits: a recording whose window ranges were
// inferred on several GPUs) instead of from the network; everything after them is the same code.
// Progress of the run in flight (ss_run_poll).  A caller that watches the progress gets the reference's SEQUENCE of values
// (worker.py:71-84: one emit per batch of settings.prediction_batch_size = 32 windows, done = 32, 64, ..., total) without its pass
// size: the values that fall into a pass are reported, in order, once that pass's event has completed.  block: wait for every
// pass; otherwise report what has completed and return.  (Round 2 ran min(chunk, 32)-window passes for this: a tenth of the rate.)
int run_poll(ss_ctx* c, ss_progress_fn progress, void* user, int block, const volatile int* stop_flag) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if (!c->run_pending) return fail(c, SS_ERR_STATE, "ss_run_poll: no run in flight");
    hipSetDevice(c->device);
    constexpr int64_t kBatch = 32;
    while (c->pass_reported < c->pass_done_at.size()) {
        if (stop_flag && *stop_flag) return SS_ERR_STOPPED;          // (the caller ends the run and discards it)
        hipEvent_t e = c->pass_ev[c->pass_reported];
        if (block) HIPCHK(c, hipEventSynchronize(e));
        else {
            const hipError_t q = hipEventQuery(e);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) return fail(c, SS_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
        }
        const int64_t upto = c->pass_done_at[c->pass_reported], total = c->track_total;
        while (c->progress_reported < upto) {
            const int64_t next = std::min<int64_t>(c->progress_reported + kBatch, total);
            if (next > upto) break;                           // a batch that straddles two passes is reported with the later one
            c->progress_reported = next;
            if (progress) progress(user, next, total);
        }
        ++c->pass_reported;
    }
    return SS_OK;
}

int run_begin(ss_ctx* c, double threshold, double break_s, bool track, const volatile int* stop_flag,
              const float* ext_logits, int64_t ext_windows) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if (!c->has_model && !ext_logits) return fail(c, SS_ERR_STATE, "context was created without weights (audio-only)");
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
    if (ext_logits && ext_windows != total)
        return fail(c, SS_ERR_ARG, "ss_run_from_logits: " + std::to_string(ext_windows) + " windows given, the plan has " + std::to_string(total));
    c->total_windows = total;
    c->logits_valid = false;
    if (total > 0) {
        if ((rc = ensure(c, &c->d_logits, &c->logits_cap, (size_t)total * 256))) return rc;
        if ((rc = ensure(c, &c->d_starts, &c->starts_cap, (size_t)total))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->d_starts, starts.data(), starts.size() * 4, hipMemcpyHostToDevice, c->stream));
        if (ext_logits) HIPCHK(c, hipMemcpyAsync(c->d_logits, ext_logits, (size_t)total * 1024, hipMemcpyHostToDevice, c->stream));
        if ((rc = upload_winoff(c, off))) return rc;      // (synchronises: off, starts and the caller's logits are free again)
    }
    if ((rc = enqueue_post(c, af, total, total_bins, max_bins, threshold))) return rc;
    c->total_bins = total_bins; c->avg_on_host = false; ++c->begin_gen;
    c->t_plan = now_ms();
    c->t_sync = c->t_plan;
    HIPCHK(c, hipEventRecord(c->ev_run0, c->stream));
    if (c->d_range_flag) HIPCHK(c, hipMemsetAsync(c->d_range_flag, 0, 4, c->stream));
    c->pass_done_at.clear(); c->pass_reported = 0; c->progress_reported = 0; c->track_total = total;
    if (!ext_logits) {
        // passes of equal size (2560 windows: 3 x 854, not 1024 + 1024 + 512: a short last pass has the launch overheads and tail
        // effects of a full one)
        const int64_t n_pass = std::max<int64_t>(1, (total + c->chunk - 1) / c->chunk);
        const int ch = (int)std::max<int64_t>(1, (total + n_pass - 1) / n_pass);
        if ((rc = ensure_workspace(c, ch))) return rc;        // (waits for the stream itself when it has to reallocate)
        // Two lanes: odd passes run on a second stream over a second workspace.  Every launch fills the chip with one workgroup per CU, and
        // its last workgroups leave CUs idle until the slowest is done; with another pass's launches queued beside it those CUs take the
        // other lane's workgroups (two processes sharing the card showed it: 30.6 k audio-s/s against 29.6 k).  Same kernels, same
        // results; not with per-launch profiling (the launch times would overlap) and not when the memory for the lane is not there.
        ss_ctx::Lane& L = c->lane1;
        // Off in the product build (SOFTSPOKEN_LANES=2 in the dev build): under rocprofv3 the overlapped launches' durations no longer are
        // the per-launch times the bench line's roofline is computed from (its profiled passes run one lane), and + 0.5-1 % is not worth two
        // sets of numbers that disagree.
        const bool two = n_pass >= 2 && !c->profile && dev_env("SOFTSPOKEN_LANES", 1) >= 2 && ensure_lane1(c, ch);
        struct LaneSwap {                                 // the context's workspace fields <-> the lane's
            ss_ctx* c; ss_ctx::Lane& L; bool on = false;
            void flip() { std::swap(c->act, L.act); std::swap(c->d_feat, L.feat); std::swap(c->d_flat_part, L.flat); std::swap(c->lo_delta, L.lo_delta); std::swap(c->stream, L.stream); on = !on; }
            ~LaneSwap() { if (on) flip(); }
        } lane{c, L};
        if (two) { HIPCHK(c, hipEventRecord(L.ev_in, c->stream)); HIPCHK(c, hipStreamWaitEvent(L.stream, L.ev_in, 0)); }
        // ---- windows in chunks, across file boundaries (worker.py:71-84 batches per file of 32) ----
        // track: an event behind every pass, from which run_poll reports the progress (below).  The passes keep their full size and
        // are all enqueued here; nothing waits for the device.
        int64_t pass_no = 0;
        for (int64_t i0 = 0; i0 < total; i0 += ch, ++pass_no) {
            if (stop_flag && *stop_flag) {
                hipStreamSynchronize(c->stream);
                if (two) hipStreamSynchronize(L.stream);   // (one of the two is the lane's, whichever way the fields are flipped)
                return fail(c, SS_ERR_STOPPED, "stopped on request");
            }
            const int m = (int)std::min<int64_t>(ch, total - i0);
            if (two && ((pass_no & 1) != 0) != lane.on) lane.flip();
            if ((rc = forward_chunk(c, c->d_winoff + i0, m, c->d_logits + (size_t)i0 * 256, nullptr, nullptr))) return rc;
            if (track) {
                const size_t k = c->pass_done_at.size();
                if (k >= c->pass_ev.size()) {
                    hipEvent_t e = nullptr;
                    HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    c->pass_ev.push_back(e);
                }
                HIPCHK(c, hipEventRecord(c->pass_ev[k], c->stream));
                c->pass_done_at.push_back(i0 + m);
            }
        }
        if (lane.on) lane.flip();
        if (two) { HIPCHK(c, hipEventRecord(L.ev_out, L.stream)); HIPCHK(c, hipStreamWaitEvent(c->stream, L.ev_out, 0)); }
    }
    // ---- overlap averaging on the device (NNDetector.py:153-190), then two bits per bin ----
    if ((rc = launch_post(c, af.size(), total, total_bins, max_bins, threshold))) return rc;
    c->t_loop = now_ms();
    c->pend_thr = threshold; c->pend_brk = break_s;
    c->run_pending = true;
    return SS_OK;
}

int run_end(ss_ctx* c) {
    if (!c) return fail(nullptr, SS_ERR_ARG, "null context");
    if (!c->run_pending) return fail(c, SS_ERR_STATE, "ss_run_end: no run in flight");
    hipSetDevice(c->device);
    const bool timing = dev_env("SOFTSPOKEN_TIMING", 0) != 0;      // development aid: host-side phases of a run on stderr
    c->run_pending = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    { float ms = 0; if (hipEventElapsedTime(&ms, c->ev_run0, c->ev_run1) == hipSuccess) c->last_run_ms = ms; }
    resolve_events(c);
    if (c->h_range_flag && *c->h_range_flag)
        return fail(c, SS_ERR_RANGE, "f16x2: an activation left the f16 range (|x| > 65504) or was not finite; run this checkpoint with the fp32 mode");
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
void ensure_regions(ss_ctx* c) {
    if (c->res_regions) return;
    const bool timing = dev_env("SOFTSPOKEN_TIMING", 0) != 0;
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

}  // namespace ss
