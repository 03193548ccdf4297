// Weights of a context: "SSWBLOB1" container -> BatchNorm folded in float64 -> MFMA fragment order on the device; front-end tables.
// Everything here runs once, inside ss_create (reference: NNDetector.py:32-34,42-53 model build + load_checkpoint; the layer list is
// SpecUNet_2D.__init__, pytorch_neural_nets.py:83-140).
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ss {

// ------------------------------------------------------------------------------------------------------
// weights blob ("SSWBLOB1"): header {magic[8], u32 n, u32 0}, n entries {name[96], u32 dtype (0 f32, 1 i64),
// u32 ndim, i64 shape[4], u64 offset, u64 nbytes}, then tensor data (offsets from blob start).
// ------------------------------------------------------------------------------------------------------
const float* Blob::f32(const std::string& k, size_t count, std::string& err) const {
    auto it = e.find(k);
    if (it == e.end()) { err = "weights blob: missing tensor '" + k + "'"; return nullptr; }
    if (it->second.dtype != 0 || it->second.nbytes != count * 4) {
        err = "weights blob: tensor '" + k + "' has wrong dtype/size"; return nullptr;
    }
    return (const float*)(base + it->second.offset);
}

bool parse_blob(const void* p, size_t n, Blob& b, std::string& err) {
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
// build: tables, folded + packed weights, launch plan
// ------------------------------------------------------------------------------------------------------
// f16x2 mode (conv4.hip SPLIT): every weight is two f16 halves, hi = f16(w), lo = f16(w - hi) (hi + lo carries ~22 significant
// bits; the low half may be an f16 subnormal, which the matrix instruction keeps -- tools/probes/mfma_f16_denorm.hip).  Per
// (out-channel group, 32-channel K chunk): a bank of `taps` fragment sets of the high halves, then the same of the low halves;
// inside a bank the layout of pack_conv_v2's bf16 form.
static uint16_t f2h(float x) { const _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

// range_ok is cleared when a weight does not fit an f16 (build_model reports it: ss_ctx::split_range_ok)
static void pack_conv_split(const Folded& w3, const Folded* wr, int NT, std::vector<char>& out, bool& range_ok) {
    const int nch = w3.cin / 32, taps = wr ? 10 : 9, ngroups = w3.cout / (32 * NT);
    const size_t tap_bytes = (size_t)2 * NT * 1024, bank = (size_t)taps * tap_bytes;
    out.assign((size_t)ngroups * nch * 2 * bank, 0);
    for (int g = 0; g < ngroups; ++g)
        for (int ci = 0; ci < nch; ++ci)
            for (int t = 0; t < taps; ++t)
                for (int s = 0; s < 2; ++s)
                    for (int nt = 0; nt < NT; ++nt)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int j = l & 31, h = l >> 5;
                                const int k = ci * 32 + s * 16 + h * 8 + e;
                                const int co = g * 32 * NT + nt * 32 + j;
                                const float v = t < 9 ? w3.w[((size_t)co * w3.cin + k) * 9 + t] : wr->w[(size_t)co * wr->cin + k];
                                const uint16_t hi = f2h(v), lo = f2h(v - h2f(hi));
                                if ((hi & 0x7c00u) == 0x7c00u) range_ok = false;      // infinity / NaN: |w| > 65504
                                const size_t off = ((size_t)g * nch + ci) * 2 * bank + (size_t)t * tap_bytes + ((size_t)(s * NT + nt) * 64 + l) * 16 + (size_t)e * 2;
                                memcpy(&out[off], &hi, 2); memcpy(&out[off + bank], &lo, 2);
                            }
}

// conv1s.hip (round 4): conv1_1's second conv (32 -> 32) for the row-streaming kernel, whose B operand of K step q is the first conv's
// accumulator tile as it stands: element j of lane half h is input channel 16 q + 8 (j >> 2) + 4 h + (j & 3) (MI355X guide, "an accumulator
// tile as the next MFMA's operand").  [plane: high halves, low halves][tap 9][K step 2][lane 64][8 values], A-operand rows = lane & 31.
static void pack_conv_stream(const Folded& w3, std::vector<char>& out, bool& range_ok) {
    const size_t bank = (size_t)9 * 2 * 1024;
    out.assign(2 * bank, 0);
    for (int t = 0; t < 9; ++t)
        for (int q = 0; q < 2; ++q)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int co = l & 31, h = l >> 5, ci = 16 * q + 8 * (j >> 2) + 4 * h + (j & 3);
                    const float v = w3.w[((size_t)co * w3.cin + ci) * 9 + t];
                    const uint16_t hi = f2h(v), lo = f2h(v - h2f(hi));
                    if ((hi & 0x7c00u) == 0x7c00u) range_ok = false;
                    const size_t off = ((size_t)(t * 2 + q) * 64 + l) * 16 + (size_t)j * 2;
                    memcpy(&out[off], &hi, 2); memcpy(&out[off + bank], &lo, 2);
                }
}

// conv1s.hip, 16-pixel form: the second conv for v_mfma_f32_16x16x32_f16.  [plane][tap 9][channel tile u 2][lane 64][8 values]: lane l is
// output row i = l & 15 of tile u = channel 8 (i >> 2) + 4 u + (i & 3), K group g = l >> 4: value j is input channel 8 g + j.
static void pack_conv_stream16(const Folded& w3, std::vector<char>& out, bool& range_ok) {
    const size_t bank = (size_t)9 * 2 * 1024;
    out.assign(2 * bank, 0);
    for (int t = 0; t < 9; ++t)
        for (int u = 0; u < 2; ++u)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int i = l & 15, g = l >> 4, co = 8 * (i >> 2) + 4 * u + (i & 3), ci = 8 * g + j;
                    const float v = w3.w[((size_t)co * w3.cin + ci) * 9 + t];
                    const uint16_t hi = f2h(v), lo = f2h(v - h2f(hi));
                    if ((hi & 0x7c00u) == 0x7c00u) range_ok = false;
                    const size_t off = ((size_t)(t * 2 + u) * 64 + l) * 16 + (size_t)j * 2;
                    memcpy(&out[off], &hi, 2); memcpy(&out[off + bank], &lo, 2);
                }
}

// conv4_ups.hip (round 3): the plain A launch of a decoder block, its upsampled input half at low resolution.  Skip chunks (input
// channels [0, c0)) as pack_conv_split lays them out -- per 32-channel chunk a bank of high halves, then one of low halves, each [tap 9]
// [sub-step 2][lane 64][8 values] --; then the upsampled chunks (input channels [c0, c0 + c1)): per chunk a bank of high halves and one
// of low halves, each [parity class 4][tap 4][sub-step][lane][8], class = a + 2 b for output pixels (2 Y + a, 2 X + b), tap = 2 ty + tx,
// value = sum of w[dy][dx] over the rows dy of set (a, ty) and the columns dx of set (b, tx):
//     a = 0: ty 0 <- {dy = -1}, ty 1 <- {0, +1}        a = 1: ty 0 <- {-1, 0}, ty 1 <- {+1}        (pytorch_neural_nets.py:171-181:
// nearest upsampling by 2 puts the same low-resolution pixel under both members of such a set).  Summed in float64, rounded to fp32, split.
static void pack_conv_split_ups(const Folded& w3, int c0, int c1, std::vector<char>& out, bool& range_ok) {
    const size_t bank_r = (size_t)9 * 2048, bank_u = (size_t)16 * 2048;
    const int nreg = c0 / 32, nups = c1 / 32;
    out.assign((size_t)nreg * 2 * bank_r + (size_t)nups * 2 * bank_u, 0);
    auto put = [&](size_t off_hi, size_t off_lo, float v) {
        const uint16_t hi = f2h(v), lo = f2h(v - h2f(hi));
        if ((hi & 0x7c00u) == 0x7c00u) range_ok = false;
        memcpy(&out[off_hi], &hi, 2); memcpy(&out[off_lo], &lo, 2);
    };
    for (int ci = 0; ci < nreg; ++ci)
        for (int t = 0; t < 9; ++t)
            for (int s = 0; s < 2; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int co = l & 31, k = ci * 32 + s * 16 + (l >> 5) * 8 + e;
                        const size_t off = (size_t)ci * 2 * bank_r + (size_t)t * 2048 + ((size_t)s * 64 + l) * 16 + (size_t)e * 2;
                        put(off, off + bank_r, w3.w[((size_t)co * w3.cin + k) * 9 + t]);
                    }
    static const int lo_of[2][2] = {{0, 1}, {0, 2}}, hi_of[2][2] = {{0, 2}, {1, 2}};   // [parity][tap] -> rows / columns [lo, hi] of the 3 x 3 (index = offset + 1)
    for (int cu = 0; cu < nups; ++cu)
        for (int cls = 0; cls < 4; ++cls)
            for (int t = 0; t < 4; ++t)
                for (int s = 0; s < 2; ++s)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int a = cls & 1, b = cls >> 1, ty = t >> 1, tx = t & 1;
                            const int co = l & 31, k = c0 + cu * 32 + s * 16 + (l >> 5) * 8 + e;
                            double sum = 0.0;
                            for (int dy = lo_of[a][ty]; dy <= hi_of[a][ty]; ++dy)
                                for (int dx = lo_of[b][tx]; dx <= hi_of[b][tx]; ++dx) sum += (double)w3.w[((size_t)co * w3.cin + k) * 9 + dy * 3 + dx];
                            const size_t off = (size_t)nreg * 2 * bank_r + (size_t)cu * 2 * bank_u + ((size_t)cls * 4 + t) * 2048 + ((size_t)s * 64 + l) * 16 + (size_t)e * 2;
                            put(off, off + bank_u, (float)sum);
                        }
}

// conv4_ups.hip, ring form (the A launch that also writes r = conv1x1(x) + br): per 32-channel output group the ring entries in the
// order the kernel walks them -- per skip chunk WH = [tap 9 + the projection's][sub-step 2][lane 64][8] high halves, then [32] bias and
// [32] projection bias of the group (fp32), WL = the same taps' low halves; per upsampled chunk UH = [class 4][tap 4] pre-summed as
// above + the projection's tap, high halves, UL = their low halves.
static void pack_conv_split_upsr(const Folded& w3, const Folded& wr, int c0, int c1, std::vector<char>& out, bool& range_ok) {
    const size_t ent_r = (size_t)10 * 2048, ent_rh = ent_r + 256, ent_u = (size_t)17 * 2048;
    const int nreg = c0 / 32, nups = c1 / 32, ngroups = w3.cout / 32;
    const size_t group = (size_t)nreg * (ent_rh + ent_r) + (size_t)nups * 2 * ent_u;
    out.assign((size_t)ngroups * group, 0);
    auto put = [&](size_t off_hi, size_t off_lo, float v) {
        const uint16_t hi = f2h(v), lo = f2h(v - h2f(hi));
        if ((hi & 0x7c00u) == 0x7c00u) range_ok = false;
        memcpy(&out[off_hi], &hi, 2); memcpy(&out[off_lo], &lo, 2);
    };
    static const int lo_of[2][2] = {{0, 1}, {0, 2}}, hi_of[2][2] = {{0, 2}, {1, 2}};
    for (int g = 0; g < ngroups; ++g) {
        const size_t gb = (size_t)g * group;
        for (int ci = 0; ci < nreg; ++ci) {
            const size_t eh = gb + (size_t)ci * (ent_rh + ent_r), el = eh + ent_rh;
            for (int t = 0; t < 10; ++t)
                for (int s = 0; s < 2; ++s)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int co = g * 32 + (l & 31), k = ci * 32 + s * 16 + (l >> 5) * 8 + e;
                            const float v = t < 9 ? w3.w[((size_t)co * w3.cin + k) * 9 + t] : wr.w[(size_t)co * wr.cin + k];
                            const size_t off = (size_t)t * 2048 + ((size_t)s * 64 + l) * 16 + (size_t)e * 2;
                            put(eh + off, el + off, v);
                        }
            for (int j = 0; j < 32; ++j) {
                memcpy(&out[eh + ent_r + (size_t)j * 4], &w3.b[g * 32 + j], 4);
                memcpy(&out[eh + ent_r + 128 + (size_t)j * 4], &wr.b[g * 32 + j], 4);
            }
        }
        for (int cu = 0; cu < nups; ++cu) {
            const size_t eh = gb + (size_t)nreg * (ent_rh + ent_r) + (size_t)cu * 2 * ent_u, el = eh + ent_u;
            for (int t = 0; t < 17; ++t)                  // 16 = class x tap, then the projection
                for (int s = 0; s < 2; ++s)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int co = g * 32 + (l & 31), k = c0 + cu * 32 + s * 16 + (l >> 5) * 8 + e;
                            float v;
                            if (t < 16) {
                                const int cls = t >> 2, tp = t & 3, a = cls & 1, b = cls >> 1, ty = tp >> 1, tx = tp & 1;
                                double sum = 0.0;
                                for (int dy = lo_of[a][ty]; dy <= hi_of[a][ty]; ++dy)
                                    for (int dx = lo_of[b][tx]; dx <= hi_of[b][tx]; ++dx) sum += (double)w3.w[((size_t)co * w3.cin + k) * 9 + dy * 3 + dx];
                                v = (float)sum;
                            } else v = wr.w[(size_t)co * wr.cin + k];
                            const size_t off = (size_t)t * 2048 + ((size_t)s * 64 + l) * 16 + (size_t)e * 2;
                            put(eh + off, el + off, v);
                        }
        }
    }
}

// conv2_ups.hip (fp32): the A launch of a decoder block, its upsampled input half at low resolution.  Per 16-channel chunk, in
// pack_conv_v2's lane / sub-step layout: skip chunks [tap 9 + the 1x1 projection's]; upsampled chunks [parity class 4][tap 4] pre-summed
// as in pack_conv_split_ups (float64 sums rounded to fp32) + the projection's tap, class = 2 (Y & 1) + (X & 1), tap = 2 ty + tx.
static void pack_conv_v2_ups(const Folded& w3, const Folded& wr, int c0, int c1, int NT, std::vector<char>& out) {
    const int ngroups = w3.cout / (32 * NT), nreg = c0 / 16, nups = c1 / 16, group_taps = nreg * 10 + nups * 17;
    const size_t tap_bytes = (size_t)2 * NT * 1024;
    out.assign((size_t)ngroups * group_taps * tap_bytes, 0);
    static const int lo_of[2][2] = {{0, 1}, {0, 2}}, hi_of[2][2] = {{0, 2}, {1, 2}};
    for (int g = 0; g < ngroups; ++g)
    for (int ci = 0; ci < nreg + nups; ++ci) {
        const bool ups = ci >= nreg;
        const size_t cbase = ((size_t)g * group_taps + (size_t)(ups ? nreg * 10 + (ci - nreg) * 17 : ci * 10)) * tap_bytes;
        for (int t = 0; t < (ups ? 17 : 10); ++t)
            for (int s = 0; s < 2; ++s)
                for (int nt = 0; nt < NT; ++nt)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 4; ++e) {
                            const int co = (g * NT + nt) * 32 + (l & 31), k = ci * 16 + (l >> 5) * 8 + s * 4 + e;
                            float v;
                            if (t == (ups ? 16 : 9)) v = wr.w[(size_t)co * wr.cin + k];
                            else if (!ups) v = w3.w[((size_t)co * w3.cin + k) * 9 + t];
                            else {
                                const int cls = t >> 2, tp = t & 3, a = cls >> 1, b = cls & 1, ty = tp >> 1, tx = tp & 1;
                                double sum = 0.0;
                                for (int dy = lo_of[a][ty]; dy <= hi_of[a][ty]; ++dy)
                                    for (int dx = lo_of[b][tx]; dx <= hi_of[b][tx]; ++dx) sum += (double)w3.w[((size_t)co * w3.cin + k) * 9 + dy * 3 + dx];
                                v = (float)sum;
                            }
                            memcpy(&out[cbase + (size_t)t * tap_bytes + ((size_t)(s * NT + nt) * 64 + l) * 16 + (size_t)e * 4], &v, 4);
                        }
    }
}

int build_tables(ss_ctx* c, const Blob& bl) {
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
    // ---- tables of the second front-end kernel (frontend.hip, frontend_kernel: 16 lanes per frame, four passes r = 0..3) ----
    {
        std::vector<float2> win2(256), twt(4 * 256), wkt(4 * 256);
        for (int n = 0; n < 256; ++n) win2[n] = make_float2(win[2 * n], win[2 * n + 1]);
        for (int r = 0; r < 4; ++r)
            for (int i = 0; i < 16; ++i)          // i = n0 (inter-pass twiddle) / m1 (untangle twiddle); j = the lane's m0
                for (int j = 0; j < 16; ++j) {
                    const double a1 = -2.0 * PI * (double)((i * (4 * j + r)) % 1024) / 1024.0;       // W1024^(n0 (4 m0 + r))
                    twt[(r * 16 + i) * 16 + j] = make_float2((float)std::cos(a1), (float)std::sin(a1));
                    const double a2 = -2.0 * PI * (double)(4 * (j + 16 * i) + r) / 2048.0;           // W2048^k, k = 4 (m0 + 16 m1) + r
                    wkt[(r * 16 + i) * 16 + j] = make_float2((float)std::cos(a2), (float)std::sin(a2));
                }
        // mel weights per lane for paired (8-byte) reads of the power spectrum, which the kernel stores with 4 pad words behind
        // every 64 bins (phys(k) = k + 4 (k / 64): lanes whose filters start 16 or 32 bins apart would otherwise share banks).
        // A lane's run starts at the group of four bins that holds its filter's first bin; weights are zero on the alignment slot, on pad
        // words and behind the filter's end, so the kernel's loops have fixed trip counts (kMelPairsLo / kMelPairsHi pairs).
        std::vector<float> wq((size_t)64 * kMelRow, 0.f);
        std::vector<int> p0(128, 0);
        // inside every group of four bins the buffer holds them in the order (0, 2, 1, 3): the kernel's two pass groups produce the
        // even and the odd quarter-spectra as register pairs and store them as such
        auto phys = [](int k) { static const int perm[4] = {0, 2, 1, 3}; return (k & ~3) + perm[k & 3] + 4 * (k / 64); };
        for (int l = 0; l < 64; ++l)
            for (int which = 0; which < 2; ++which) {
                const int j = which ? 127 - l : l, pairs = which ? kMelPairsHi : kMelPairsLo;
                float* dst = wq.data() + (size_t)l * kMelRow + (which ? 2 * kMelPairsLo : 0);
                const int s_even = mstart[j] & ~3;           // a run starts at its filter's group of four
                const int ph0 = phys(s_even);
                p0[2 * l + which] = ph0;
                int placed = 0;
                for (int bidx = 0; bidx < mcount[j]; ++bidx) {
                    const int slot = phys(mstart[j] + bidx) - ph0;
                    if (slot < 0 || slot >= 2 * pairs)
                        return fail(c, SS_ERR_FORMAT, "mel filterbank: a filter does not fit the front-end kernel's fixed trip counts");
                    dst[slot] = mw[moff[j] + bidx]; ++placed;
                }
                if (ph0 + 2 * pairs > kPwWords) return fail(c, SS_ERR_FORMAT, "mel filterbank: a filter's run leaves the power-spectrum buffer");
                (void)placed;
            }
        if ((rc = dev_upload(c, &c->d_win2, win2.data(), win2.size() * sizeof(float2)))) return rc;
        if ((rc = dev_upload(c, &c->d_twt, twt.data(), twt.size() * sizeof(float2)))) return rc;
        if ((rc = dev_upload(c, &c->d_wkt, wkt.data(), wkt.size() * sizeof(float2)))) return rc;
        if ((rc = dev_upload(c, &c->d_mel_wq, wq.data(), wq.size() * 4))) return rc;
        if ((rc = dev_upload(c, &c->d_mel_p0, p0.data(), p0.size() * 4))) return rc;
    }
    c->mel_nw = (int)mw.size();
    if ((rc = dev_upload(c, &c->d_mel_w, mw.data(), mw.size() * 4))) return rc;
    return SS_OK;
}

// ------------------------------------------------------------------------------------------------------
// f16x2: power-of-two channel normalisation.
// A value split as hi = f16(x), lo = f16(x - hi) carries 22 bits only while lo is a normal f16, i.e. for |x| >~ 0.06; below that the
// pair has an ABSOLUTE resolution of 6e-8 (14 bits at 1e-3, nothing at 1e-7), and beyond 65504 it does not exist.  A checkpoint
// whose BatchNorm gains put a tensor far from 1 and undo it downstream is the same function as far as the reference's fp32 is
// concerned (pytorch_neural_nets.py:7-41: conv, BN, ReLU; ReLU, max-pool, nearest upsampling and concat commute with a positive
// per-channel factor) but not for the split.  So every activation tensor T is STORED as 2^s_T[c] x its value, with the exponents
// chosen here, once, from the folded weights alone: a conv from tensor X into tensor Y gets w'[c][k] = w[c][k] 2^(s_Y[c] - s_X[k])
// and b'[c] = b[c] 2^s_Y[c], the two branches of a block share s_Y, conv_flatten (and the spec head's 1x1) take 2^-s[k] of their
// input and deliver unscaled values.  Powers of two commute with every fp32 rounding, so in exact-fp32 terms nothing changes;
// what changes is where the values sit in the f16 range.  s_Y[c] = -round(log2(est)), est^2 = |row of w', inputs normalised|^2 / 2
// + b^2: the second moment of the channel for unit-second-moment inputs behind a ReLU.  No kernel knows about any of it.
// ------------------------------------------------------------------------------------------------------
static int norm_exponent(double sumsq_w, double bias) {
    const double est = std::sqrt(0.5 * sumsq_w + bias * bias);
    if (!(est > 0.0) || !std::isfinite(est)) return 0;
    const long e = -std::lround(std::log2(est));
    return (int)std::max<long>(-60, std::min<long>(60, e));
}
// scale a folded conv: rows by 2^so[c], columns by 2^-si[k]
static void scale_folded(Folded& f, const std::vector<int>& so, const std::vector<int>& si) {
    for (int c = 0; c < f.cout; ++c) {
        for (int k = 0; k < f.cin; ++k)
            for (int t = 0; t < f.k; ++t) {
                float& v = f.w[((size_t)c * f.cin + k) * f.k + t];
                v = std::ldexp(v, so[c] - si[k]);
            }
        f.b[c] = std::ldexp(f.b[c], so[c]);
    }
}
// sum of squares of row c with the columns brought to normalised inputs
static double row_sumsq(const Folded& f, int c, const std::vector<int>& si) {
    double q = 0;
    for (int k = 0; k < f.cin; ++k)
        for (int t = 0; t < f.k; ++t) { const double v = std::ldexp((double)f.w[((size_t)c * f.cin + k) * f.k + t], -si[k]); q += v * v; }
    return q;
}

// 32-channel tiles per block; the 8x16 bottom level uses NT = 1 so that 4 x more blocks exist
static int pick_nt(int cout, int H) { return H <= 8 ? 1 : (cout == 96 ? 3 : (cout >= 64 ? 2 : 1)); }

// One ResBlock (pytorch_neural_nets.py:7-41) -> launch A (conv1+BN+ReLU) and launch B (conv2+BN + residual+BN, add, ReLU).
// s_in: exponents of the block input's channels (cat[x0, x1]); s_out receives the block output's (all zero unless f16x2)
static int build_resblock(ss_ctx* c, const Blob& bl, const std::string& name, int cin0, int cin1, int cout, int H, int W,
                          const std::vector<int>& s_in, std::vector<int>& s_out) {
    std::string err;
    const int cin = cin0 + cin1;
    Folded f1, f2, fr;
    if (!fold_conv_bn(bl, name + ".conv1.0", name + ".conv1.1", cout, cin, 9, f1, err)) return fail(c, SS_ERR_FORMAT, err);
    if (!fold_conv_bn(bl, name + ".conv2.0", name + ".conv2.1", cout, cout, 9, f2, err)) return fail(c, SS_ERR_FORMAT, err);
    if (!fold_conv_bn(bl, name + ".residual.0", name + ".residual.1", cout, cin, 1, fr, err)) return fail(c, SS_ERR_FORMAT, err);
    s_out.assign(cout, 0);
    if (c->prec == kF16x2 && dev_env("SOFTSPOKEN_NORM", 1)) {      // power-of-two channel normalisation (above)
        std::vector<int> s_h(cout, 0);
        for (int co = 0; co < cout; ++co) s_h[co] = norm_exponent(row_sumsq(f1, co, s_in), f1.b[co]);
        for (int co = 0; co < cout; ++co) s_out[co] = norm_exponent(row_sumsq(f2, co, s_h) + row_sumsq(fr, co, s_in), (double)f2.b[co] + (double)fr.b[co]);
        scale_folded(f1, s_h, s_in);
        scale_folded(f2, s_out, s_h);
        scale_folded(fr, s_out, s_in);
    }
    int NT = pick_nt(cout, H);
    // f16x2: the 64- and 128-channel layers run one 32-channel tile per block: with two, the A launches (two accumulator sets) spilled
    // under the 128-register cap of two blocks per CU (conv2_1.A 2336 -> 1851 us, conv7.A 2162 -> 1563 us per 1005 windows)
    if (c->prec == kF16x2 && NT == 2) NT = 1;
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
        if ((rc = dev_upload(c, &B.d_bias2, b2r.data(), cout * 4))) return rc;   // b2 + br (the rank-1 residual has no separate tensor)
        if ((rc = dev_upload(c, &B.d_rank1, fr.w.data(), cout * 4))) return rc;
        if (c->prec == kF16x2) pack_conv_split(f2, nullptr, NT, pk, c->split_range_ok); else pack_conv_v2(f2, nullptr, c->bf16, NT, pk);
        if ((rc = dev_upload(c, (char**)&B.d_w2, pk.data(), pk.size()))) return rc;
        if (c->prec == kF16x2) {                         // the row-streaming form (conv1s.hip)
            pack_conv_stream16(f2, pk, c->split_range_ok);
            if (pk.size() != conv1_stream_weight_bytes()) return fail(c, SS_ERR_STATE, "pack_conv_stream16: size");
            if ((rc = dev_upload(c, (char**)&B.d_w_s16, pk.data(), pk.size()))) return rc;
#ifdef SS_DEVBUILD
            pack_conv_stream(f2, pk, c->split_range_ok);          // the 32-column form (development build: SOFTSPOKEN_C1S_FORM=32)
            if ((rc = dev_upload(c, (char**)&B.d_w_s1, pk.data(), pk.size()))) return rc;
#endif
            // Can a stored value of this block leave the f16 range?  A finite feature is sqrt(log10(mel + 1)) <= sqrt(log10(FLT_MAX)):
            // with it |h1[c]| <= sum |w1[c]| fmax + |b1[c]| and |c1[co]| <= sum |w2[co][c]| |h1[c]| + |b2 + br| + |wr[co]| fmax, in the
            // normalised units the kernel stores.  Below the limit with a margin, the kernel needs no run-time test (conv1s.hip TRACK).
            const double fmax = 6.2076;
            std::vector<double> bh(cout);
            double worst = 0;
            for (int co = 0; co < cout; ++co) {
                double q = std::fabs((double)f1.b[co]);
                for (int t = 0; t < 9; ++t) q += std::fabs((double)f1.w[(size_t)co * 9 + t]) * fmax;
                bh[co] = q; worst = std::max(worst, q);
            }
            for (int co = 0; co < cout; ++co) {
                double q = std::fabs((double)b2r[co]) + std::fabs((double)fr.w[co]) * fmax;
                for (int ci = 0; ci < cout; ++ci)
                    for (int t = 0; t < 9; ++t) q += std::fabs((double)f2.w[((size_t)co * cout + ci) * 9 + t]) * bh[ci];
                worst = std::max(worst, q);
            }
            B.s1_range_proven = std::isfinite(worst) && worst < 60000.0;
        }
        c->convs.push_back(B);
        return SS_OK;
    }
    // f16x2, A launch of the 96-channel blocks: three 32-channel groups through the shared two-slot bank ring (conv4.hip RING)
    // instead of one 96-channel tile per block with 120 KB of banks streamed per chunk and tile
    const int NTA = (c->prec == kF16x2 && NT == 3) ? 1 : NT;
    ConvPlan A; A.name = name + ".A"; A.Cout = cout; A.NT = NTA; A.C0 = cin0; A.C1 = cin1; A.H = H; A.W = W;
    if ((rc = dev_upload(c, &A.d_bias2, f1.b.data(), cout * 4))) return rc;
    if (c->prec == kF16x2) pack_conv_split(f1, &fr, NTA, pk, c->split_range_ok); else pack_conv_v2(f1, &fr, c->bf16, NT, pk);
    if ((rc = dev_upload(c, (char**)&A.d_w2, pk.data(), pk.size()))) return rc;
    if ((rc = dev_upload(c, &A.d_res_bias, fr.b.data(), cout * 4))) return rc;
    if (c->prec == kF16x2 && NTA == 1 && cin0 >= 64 && cin0 % 32 == 0 && cin1 % 32 == 0 && (cin1 >= 32 || dev_env("SOFTSPOKEN_UPSR", 1) >= 2)) {
        // the same launch with the upsampled input half at low resolution (conv4_ups.hip, ring form: conv6 / conv7 / conv8)
        pack_conv_split_upsr(f1, fr, cin0, cin1, pk, c->split_range_ok);
        if (pk.size() != conv_upsr_weight_bytes(cin0, cin1, cout)) return fail(c, SS_ERR_STATE, "pack_conv_split_upsr: size");
        if ((rc = dev_upload(c, (char**)&A.d_w_upsr, pk.data(), pk.size()))) return rc;
    }
    if (c->prec == kFp32 && cin1 >= 16 && cin0 % 16 == 0 && cin1 % 16 == 0 && cout % 32 == 0 && H % 16 == 0 && dev_env("SOFTSPOKEN_UPS32", 1)) {
        // fp32: the same launch with the upsampled input half at low resolution (conv2_ups.hip)
        A.ups32_nt = dev_env("SOFTSPOKEN_UPS32_NT", 1);
        if (A.ups32_nt < 1 || A.ups32_nt > 3 || (cout / 32) % A.ups32_nt) A.ups32_nt = 1;
        pack_conv_v2_ups(f1, fr, cin0, cin1, A.ups32_nt, pk);
        if (pk.size() != conv_ups32_weight_bytes(cin0, cin1, cout)) return fail(c, SS_ERR_STATE, "pack_conv_v2_ups: size");
        if ((rc = dev_upload(c, (char**)&A.d_w_ups32, pk.data(), pk.size()))) return rc;
    }
    c->convs.push_back(A);
    if (c->bf16) {
        pack_conv_v2(f1, nullptr, true, NT, pk);
        if ((rc = dev_upload(c, (char**)&A.d_w3, pk.data(), pk.size()))) return rc;
        c->convs.back().d_w3 = A.d_w3;
    }
    if (c->prec == kF16x2) {                              // "projection in B" form: the 3x3 banks alone (no 1x1 tap)
        pack_conv_split(f1, nullptr, NTA, pk, c->split_range_ok);
        if ((rc = dev_upload(c, (char**)&A.d_w3, pk.data(), pk.size()))) return rc;
        c->convs.back().d_w3 = A.d_w3;
        if (cin1 >= 32 && cin0 >= 32 && cout == 32) {     // ... and the same with the upsampled half at low resolution (conv4_ups.hip)
            pack_conv_split_ups(f1, cin0, cin1, pk, c->split_range_ok);
            if (pk.size() != conv_ups_weight_bytes(cin0, cin1)) return fail(c, SS_ERR_STATE, "pack_conv_split_ups: size");
            if ((rc = dev_upload(c, (char**)&A.d_w_ups, pk.data(), pk.size()))) return rc;
            c->convs.back().d_w_ups = A.d_w_ups;
        }
    }
    // f16x2, B launch of the 96-channel blocks: three 32-channel groups over the four tiles' shared ring as well (SOFTSPOKEN_NTB1=0 in the
    // dev build: one 96-channel tile per 8-wave block)
    const int NTB = (c->prec == kF16x2 && NT == 3 && dev_env("SOFTSPOKEN_NTB1", 1)) ? 1 : NT;
    ConvPlan B; B.name = name + ".B"; B.Cout = cout; B.NT = NTB; B.C0 = cout; B.R0 = cin0; B.R1 = cin1; B.H = H; B.W = W;
    if (c->bf16 && cin % 16 == 0) {
        // [step][32-channel tile][lane][slot j]: row (output channel) = 32 tile + (lane & 31), input channel = 16 step + 8 (lane >> 5) + j
        const int steps = cin / 16, tiles = cout / 32;
        std::vector<uint16_t> pj((size_t)steps * tiles * 64 * 8);
        for (int st = 0; st < steps; ++st) for (int t = 0; t < tiles; ++t) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j)
            pj[(((size_t)st * tiles + t) * 64 + l) * 8 + j] = f2bf(fr.w[(size_t)(32 * t + (l & 31)) * cin + 16 * st + 8 * (l >> 5) + j]);
        if ((rc = dev_upload(c, (char**)&B.d_proj, (const char*)pj.data(), pj.size() * 2))) return rc;
        if ((rc = dev_upload(c, &B.d_bias3, b2r.data(), cout * 4))) return rc;
    }
    if (c->prec == kF16x2 && cin % 16 == 0) {
        // the same fragments as two banks of f16 halves: [bank][step][32-channel tile][lane][slot j]
        const int steps = cin / 16, tiles = cout / 32;
        const size_t bank = (size_t)steps * tiles * 64 * 8;
        std::vector<uint16_t> pj(2 * bank);
        for (int st = 0; st < steps; ++st) for (int t = 0; t < tiles; ++t) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
            const float v = fr.w[(size_t)(32 * t + (l & 31)) * cin + 16 * st + 8 * (l >> 5) + j];
            const uint16_t hi = f2h(v);
            if ((hi & 0x7c00u) == 0x7c00u) c->split_range_ok = false;
            const size_t at = (((size_t)st * tiles + t) * 64 + l) * 8 + j;
            pj[at] = hi; pj[bank + at] = f2h(v - h2f(hi));
        }
        if ((rc = dev_upload(c, (char**)&B.d_proj, (const char*)pj.data(), pj.size() * 2))) return rc;
        if ((rc = dev_upload(c, &B.d_bias3, b2r.data(), cout * 4))) return rc;
    }
    if (c->prec == kF16x2) pack_conv_split(f2, nullptr, NTB, pk, c->split_range_ok); else pack_conv_v2(f2, nullptr, c->bf16, NT, pk);
    if ((rc = dev_upload(c, (char**)&B.d_w2, pk.data(), pk.size()))) return rc;
    if ((rc = dev_upload(c, &B.d_bias2, f2.b.data(), cout * 4))) return rc;
    c->convs.push_back(B);
    return SS_OK;
}

int build_model(ss_ctx* c, const Blob& bl) {
    int rc;
    c->split_range_ok = true;
    // launch order == pytorch_neural_nets.py:156-181
    // x0 / x1: the blocks whose outputs are concatenated into this block's input (pytorch_neural_nets.py:171-180: [skip, upsampled]);
    // their channel exponents (f16x2 normalisation) travel with them
    struct RB { const char* n; int c0, c1, co, H, W; const char *x0, *x1; };
    const RB rbs[] = {{"conv1_1", 1, 0, 32, 128, 256, nullptr, nullptr},  {"conv2_1", 32, 0, 64, 64, 128, "conv1_1", nullptr},
                      {"conv3_1", 64, 0, 96, 32, 64, "conv2_1", nullptr}, {"conv4_1", 96, 0, 128, 16, 32, "conv3_1", nullptr},
                      {"conv_bottleneck", 128, 0, 128, 8, 16, "conv4_1", nullptr}, {"encoder_out", 128, 0, 128, 8, 16, "conv_bottleneck", nullptr},
                      {"conv6", 128, 128, 96, 16, 32, "conv4_1", "encoder_out"},  {"conv7", 96, 96, 64, 32, 64, "conv3_1", "conv6"},
                      {"conv8", 64, 64, 32, 64, 128, "conv2_1", "conv7"},         {"conv9_1", 32, 32, 32, 128, 256, "conv1_1", "conv8"},
                      {"spec_output_conv.0", 32, 0, 32, 128, 256, "conv9_1", nullptr}};
    std::map<std::string, std::vector<int>> sc;           // block name -> exponents of its output channels
    for (const RB& r : rbs) {
        std::vector<int> s_in;
        if (r.x0) s_in = sc[r.x0]; else s_in.assign(r.c0, 0);          // (conv1_1: the features, as they are)
        if (r.x1) s_in.insert(s_in.end(), sc[r.x1].begin(), sc[r.x1].end());
        if ((int)s_in.size() != r.c0 + r.c1) return fail(c, SS_ERR_STATE, "build_model: channel bookkeeping");
        if ((rc = build_resblock(c, bl, r.n, r.c0, r.c1, r.co, r.H, r.W, s_in, sc[r.n]))) return rc;
    }
    std::string err;
    // conv_flatten (pytorch_neural_nets.py:133): weight (4, 32, 128, 1) -> [h][ci][c]
    const float* wf0 = bl.f32("conv_flatten.weight", 4 * 32 * 128, err);
    const float* bf = bl.f32("conv_flatten.bias", 4, err);
    if (!wf0 || !bf) return fail(c, SS_ERR_FORMAT, err);
    // conv9_1's channels arrive as 2^s[ci] x their values.  The filter takes the differences between the channels back; their common
    // part (the median exponent: a checkpoint whose scores are huge has it far from 0) stays in the partial sums and is taken out in
    // fp32 by the head kernel (Head1dWeights::fscale, an exact power of two), so that the filter's own values stay near their size
    std::vector<float> wfs(wf0, wf0 + 4 * 32 * 128);
    std::vector<int> s9 = sc["conv9_1"];
    std::vector<int> srt = s9; std::sort(srt.begin(), srt.end());
    const int s_common = srt[srt.size() / 2];
    for (int co = 0; co < 4; ++co) for (int ci = 0; ci < 32; ++ci) for (int h = 0; h < 128; ++h)
        wfs[((size_t)co * 32 + ci) * 128 + h] = std::ldexp(wfs[((size_t)co * 32 + ci) * 128 + h], s_common - s9[ci]);
    c->head.fscale = std::ldexp(1.0f, -s_common);
    const float* wf = wfs.data();
    {   // fused flatten (conv2.hip FLAT): per mel row PAIR a 32 -> 8 (padded to 32) 1x1 "conv" in MFMA fragment order -- columns 0..3 are
        // the four flatten channels with the weights of row 2 p, columns 4..7 the same with those of row 2 p + 1: one product per pair of
        // rows, a pixel takes the columns of its own row.  [pair 64][fragment: K chunk, sub-step][lane 64][16 bytes], pack_conv's K order
        const int KC = c->bf16 ? 32 : 16, per = c->bf16 ? 8 : 4, nfs = (32 / KC) * 2;
        std::vector<char> all((size_t)64 * nfs * 1024, 0);
        for (int pr = 0; pr < 64; ++pr) for (int f = 0; f < nfs; ++f) for (int l = 0; l < 64; ++l) for (int e = 0; e < per; ++e) {
            const int cc = f >> 1, sub = f & 1, j = l & 31, h = l >> 5;
            if (j >= 8) continue;
            const int k = c->bf16 ? cc * 32 + sub * 16 + h * 8 + e : cc * 16 + h * 8 + sub * 4 + e;
            const float v = wf[((size_t)(j & 3) * 32 + k) * 128 + 2 * pr + (j >> 2)];
            const size_t off = (((size_t)pr * nfs + f) * 64 + l) * 16 + (size_t)e * (c->bf16 ? 2 : 4);
            if (c->bf16) { const uint16_t hv = f2bf(v); memcpy(&all[off], &hv, 2); } else memcpy(&all[off], &v, 4);
        }
        if ((rc = dev_upload(c, (char**)&c->d_flat_frag, all.data(), all.size()))) return rc;
    }
    // conv4.hip FLAT: [mel row pair][step s][lane][slot j] -> channel 16 s + 8 (j >> 2) + 4 (lane >> 5) + (j & 3); fragment row
    // lane & 31: rows 0..3 = the 4 flatten channels with the weights of mel row 2 pair, rows 4..7 with those of row 2 pair + 1
    if (c->bf16) {
        std::vector<uint16_t> t4((size_t)64 * 2 * 64 * 8, 0);
        for (int pr = 0; pr < 64; ++pr) for (int s2 = 0; s2 < 2; ++s2) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
            const int row = l & 31, ch = 16 * s2 + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
            if (row < 8) t4[(((size_t)pr * 2 + s2) * 64 + l) * 8 + j] = f2bf(wf[((size_t)(row & 3) * 32 + ch) * 128 + 2 * pr + (row >> 2)]);
        }
        if ((rc = dev_upload(c, (char**)&c->d_flat_frag4, (const char*)t4.data(), t4.size() * 2))) return rc;
    }
    if (c->prec == kF16x2) {   // the same, two banks of f16 halves: [bank][mel row pair][step][lane][slot]
        const size_t bank = (size_t)64 * 2 * 64 * 8;
        std::vector<uint16_t> t4(2 * bank, 0);
        for (int pr = 0; pr < 64; ++pr) for (int s2 = 0; s2 < 2; ++s2) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
            const int row = l & 31, ch = 16 * s2 + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
            if (row >= 8) continue;
            const float v = wf[((size_t)(row & 3) * 32 + ch) * 128 + 2 * pr + (row >> 2)];
            const uint16_t hi = f2h(v);
            if ((hi & 0x7c00u) == 0x7c00u) c->split_range_ok = false;
            const size_t at = (((size_t)pr * 2 + s2) * 64 + l) * 8 + j;
            t4[at] = hi; t4[bank + at] = f2h(v - h2f(hi));
        }
        if ((rc = dev_upload(c, (char**)&c->d_flat_frag4, (const char*)t4.data(), t4.size() * 2))) return rc;
    }
    if ((rc = dev_upload(c, &c->d_flat_b, bf, 16))) return rc;
    // spec_output_conv.1 (pytorch_neural_nets.py:128): Conv2d(32, 2, 1) with bias
    const float* ws = bl.f32("spec_output_conv.1.weight", 64, err);
    const float* bs = bl.f32("spec_output_conv.1.bias", 2, err);
    if (!ws || !bs) return fail(c, SS_ERR_FORMAT, err);
    float wss[64];
    for (int co = 0; co < 2; ++co) for (int ci = 0; ci < 32; ++ci) wss[co * 32 + ci] = std::ldexp(ws[co * 32 + ci], -sc["spec_output_conv.0"][ci]);
    if ((rc = dev_upload(c, &c->d_spec_w, wss, 256))) return rc;
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
    if (c->prec == kF16x2 && !c->split_range_ok)
        return fail(c, SS_ERR_RANGE, "f16x2: a folded conv weight is outside the f16 range (|w| > 65504 after the channel normalisation) or not finite; create the context in the fp32 mode");
    return SS_OK;
}

}  // namespace ss
