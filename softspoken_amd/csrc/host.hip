// Host-only pieces of the path (no device work): RIFF/WAVE header walk, window plan, threshold / run-length / gap-merge, the
// reference's time strings and CSV text, silencer frame ranges, WAV header writer.  Reference lines are cited per function.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace ss;

// voice_activity.py:23-30 (get_audio_data) needs duration + native rate; load_audio needs the samples.
// AIFF / AIFF-C ("FORM" size "AIFF" | "AIFC"; big-endian chunk sizes): COMM = channels u16, frames u32, bits u16, sample rate as an 80-bit
// extended float [, compression type, name]; SSND = offset u32, block size u32, samples.  8-bit samples are signed.
static int aiff_parse(const unsigned char* b, size_t nbytes, ss_wav_info* out) {
    auto be16 = [&](size_t p) { return (uint32_t)b[p] << 8 | b[p + 1]; };
    auto be32 = [&](size_t p) { return (uint32_t)b[p] << 24 | (uint32_t)b[p + 1] << 16 | (uint32_t)b[p + 2] << 8 | b[p + 3]; };
    const bool aifc = memcmp(b + 8, "AIFC", 4) == 0;
    size_t pos = 12;
    bool have_comm = false;
    uint32_t ch = 0, bits = 0, n_frames = 0; double sr = 0; char comp[5] = "NONE";
    while (pos + 8 <= nbytes) {
        const uint32_t sz = be32(pos + 4);
        const size_t body = pos + 8;
        if (memcmp(b + pos, "COMM", 4) == 0) {
            if (sz < 18 || body + 18 > nbytes) return fail(nullptr, SS_ERR_FORMAT, "AIFF: short COMM chunk");
            ch = be16(body); n_frames = be32(body + 2); bits = be16(body + 6);
            const int e = (int)(be16(body + 8) & 0x7fff) - 16383;                  // 80-bit extended: sign + 15-bit exponent, 64-bit mantissa with its leading 1
            const uint64_t mant = (uint64_t)be32(body + 10) << 32 | be32(body + 14);
            sr = (be16(body + 8) & 0x7fff) == 0x7fff ? 0.0 : std::ldexp((double)mant, e - 63);
            if (aifc) {
                if (sz < 22 || body + 22 > nbytes) return fail(nullptr, SS_ERR_FORMAT, "AIFF-C: COMM chunk without a compression type");
                memcpy(comp, b + body + 18, 4);
            }
            have_comm = true;
        } else if (memcmp(b + pos, "SSND", 4) == 0) {
            if (!have_comm) return fail(nullptr, SS_ERR_FORMAT, "AIFF: SSND chunk before COMM chunk");
            if (sz < 8 || body + 8 > nbytes) return fail(nullptr, SS_ERR_FORMAT, "AIFF: short SSND chunk");
            const uint32_t off = be32(body);
            int fmt = 0;
            const bool le = memcmp(comp, "sowt", 4) == 0;
            if (memcmp(comp, "NONE", 4) == 0 || le) {
                if (bits == 8) fmt = SS_PCM_S8;
                else if (bits == 16) fmt = le ? SS_PCM_S16 : SS_PCM_S16BE;
                else if (bits == 24) fmt = le ? SS_PCM_S24 : SS_PCM_S24BE;
                else if (bits == 32) fmt = le ? SS_PCM_S32 : SS_PCM_S32BE;
            } else if (memcmp(comp, "fl32", 4) == 0 || memcmp(comp, "FL32", 4) == 0) { fmt = SS_PCM_F32BE; bits = 32; }
            else if (memcmp(comp, "fl64", 4) == 0 || memcmp(comp, "FL64", 4) == 0) { fmt = SS_PCM_F64BE; bits = 64; }
            if (!fmt) return fail(nullptr, SS_ERR_FORMAT, std::string("AIFF: unsupported encoding (") + comp + ", " + std::to_string(bits) + " bits)");
            if (ch == 0 || !(sr >= 1.0) || sr > 2147483647.0) return fail(nullptr, SS_ERR_FORMAT, "AIFF: zero channels or sample rate out of range");
            if ((uint64_t)body + 8 + off > nbytes) return fail(nullptr, SS_ERR_FORMAT, "AIFF: SSND offset beyond the file");
            const size_t start = body + 8 + off;
            const size_t avail = std::min<size_t>(sz - 8 > off ? sz - 8 - off : 0, nbytes - start);
            const size_t fb = (size_t)ch * bits / 8;
            out->format = fmt; out->channels = (int32_t)ch; out->sample_rate = (int32_t)std::nearbyint(sr); out->bits = (int32_t)bits;
            out->data_offset = (int64_t)start; out->data_bytes = (int64_t)avail;
            out->frames = (int64_t)std::min<size_t>(avail / fb, n_frames);           // (COMM's count is the truth; a truncated file has fewer)
            return SS_OK;
        }
        pos = body + sz + (sz & 1);
    }
    return fail(nullptr, SS_ERR_FORMAT, "AIFF: missing COMM or SSND chunk");
}

extern "C" int ss_wav_parse(const void* file_bytes, size_t nbytes, ss_wav_info* out) {
    if (!file_bytes || !out) return fail(nullptr, SS_ERR_ARG, "ss_wav_parse: null argument");
    const unsigned char* b = (const unsigned char*)file_bytes;
    if (nbytes >= 12 && memcmp(b, "FORM", 4) == 0 && (memcmp(b + 8, "AIFF", 4) == 0 || memcmp(b + 8, "AIFC", 4) == 0)) return aiff_parse(b, nbytes, out);
    if (nbytes < 12 || memcmp(b, "RIFF", 4) != 0 || memcmp(b + 8, "WAVE", 4) != 0) return fail(nullptr, SS_ERR_FORMAT, "not a RIFF/WAVE (or AIFF) file");
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

double ss::bin_time(int64_t idx) {    // float(f"{idx / (256 / 3):.4f}")  (NNDetector.py:185, worker.py:100)
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
// silencer: silencer_ui.py:974-998
// ------------------------------------------------------------------------------------------------------
// Frame ranges the reference's slice assignment touches: int(round(t * sr)) with Python's round (half to
// even), clamped to [0, frames]; sorted and merged so the kernel can binary-search them.
std::vector<int64_t> ss::silence_ranges(const ss_region* regions, int64_t n, int sr, int64_t frames) {
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
