// conv3x3 over cat[x0, up2(x1)] in the f16x2 arithmetic with the UPSAMPLED HALF AT LOW RESOLUTION ("sub-pixel" form), round 3.
//
// Reference: the first conv of a decoder ResBlock, pytorch_neural_nets.py:171-181 -- Upsample(scale 2, nearest), cat([skip, upsampled]),
// Conv2d 3x3 pad 1.  For the channels that come through the upsampling, u[r][c] = x1[r >> 1][c >> 1], so for an output pixel
// (2 Y + a, 2 X + b) the nine taps meet only four distinct low-resolution pixels:
//     rows:  a = 0: dy = -1 -> Y - 1,  dy = 0, +1 -> Y          a = 1: dy = -1, 0 -> Y,  dy = +1 -> Y + 1        (columns alike with b)
// i.e. per parity class (a, b) a 2 x 2 conv over x1 with pre-summed taps  W_ab[ty][tx] = sum over the dy of row set (a, ty) and the dx of
// column set (b, tx) of w[dy][dx]  (csrc/weights.hip sums them in float64 from the folded weights).  4 taps instead of 9 on those channels:
// 2.25 x fewer matrix products, exact up to the summation order of the weights; and the stage's patch is the 6 x 10 low-resolution pixels
// under an 8 x 16 tile instead of 10 x 18 replicated ones (3 x fewer loads).  The part is at its power cap in these launches
// (DESIGN.md section 6): what moves the step is fewer products and fewer bytes, which is what this form is.
//
// Structure = conv4.hip's four-tile workgroup (four 4-wave tiles of 8 x 16 pixels a beat apart over shared resident banks; a skip chunk
// of 32 channels is two stages: low halves against wh, then high halves against wh and wl; an upsampled chunk is ONE stage -- both halves
// of its small patch are staged together: 24 products), with three differences:
//   * an MFMA's 32 pixels must share their weights, so an M-tile is one PARITY CLASS of the tile: wave w = class (a, b) = (w & 1, w >> 1),
//     lane m = (X, Y) = (m & 7, m >> 3) owns pixel (2 Y + a, 2 X + b);
//   * pixels two apart would put every lane of a 16-byte LDS read on an even slot (2-way conflicts at best), so the patch of the skip
//     channels lies de-interleaved by column parity -- row pitch 1472 = [9 even columns x 80 B][9 odd columns x 80 B] + 32: slot index
//     (8 Y + 5 X) mod 16 over a service group, conflict-free for every tap; the low-resolution patch is dense (row pitch 896, the same
//     residues);
//   * the upsampled chunks' banks hold 4 classes x 4 taps (a wave reads its class's): 32 KB per bank instead of 18.
// Form built: the plain A launch (h alone: the block's projection is computed by its B launch), one 32-channel output group, banks
// resident -- conv9_1.A (32 skip + 32 upsampled channels: 36 + 64 KB of banks beside 4 x 14.4 KB of patches).  The other decoder blocks'
// A launches stream their banks through conv4.hip's two-slot ring, which the class banks outgrow.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace ss {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kHdr = 256;            // zero bytes in front of every activation tensor (engine.hip ensure_workspace)
constexpr int kPix = 80;             // pixel pitch in LDS: 64 bytes of channels + 16
constexpr int kRowR = 1472;          // skip-channel patch row: 9 even columns, 9 odd columns, 32 bytes of padding
constexpr int kOdd = 720;            // offset of a row's odd columns
constexpr int kRowU = 896;           // low-resolution patch row: 10 pixels + padding
constexpr int kLowPlane = 6 * kRowU;                      // low-resolution patch: high halves, then low halves
constexpr int kPatchBytes = 10 * kRowR;                   // (the two low-resolution planes, 2 x 5376, live in the same buffer)
constexpr int kBankR = 9 * 2048, kBankU = 16 * 2048;      // one bank (high or low halves) of a skip chunk / of an upsampled chunk
constexpr int NW = 4, NH = 4, NTHR = 64 * NW;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void vm_lds_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
__device__ __forceinline__ f32x2 unpack_f16(uint32_t v) { return __builtin_convertvector(__builtin_bit_cast(f16x2, v), f32x2); }
// the largest high half seen so far, per 16-bit lane (the values are >= 0 behind the ReLU: as unsigned integers they order like the
// values, infinity and NaN on top): one instruction per pair; the test for "all exponent bits set" happens once, on the maximum
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// lo = f16(x - hi) of a pair whose high halves are packed in `hi`: the mixed-precision FMA reads the f16 half and the fp32 value, subtracts in
// fp32 (exactly: hi is x rounded) and rounds to f16 into one half of the destination -- two instructions for the pair instead of two
// conversions back, two subtractions and a pack; bit for bit the same (tools/probes/fma_mix_split.hip)
__device__ __forceinline__ uint32_t split_lo(uint32_t hi, float x0, float x1) {
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
    return l;
}
__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void half_swap(uint32_t& x, uint32_t& y) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0]; y = r[1];
}
struct Packed { uint32_t p[4][2]; };
// as conv4.hip: after the swaps a lane holds channels [8 hh, 8 hh + 8) in lo and [16 + 8 hh, 16 + 8 hh + 8) in hi
__device__ __forceinline__ void to_runs(Packed& k, u32x4& lo, u32x4& hi) {
    half_swap(k.p[0][0], k.p[1][0]); half_swap(k.p[0][1], k.p[1][1]);
    half_swap(k.p[2][0], k.p[3][0]); half_swap(k.p[2][1], k.p[3][1]);
    lo = u32x4{k.p[0][0], k.p[0][1], k.p[1][0], k.p[1][1]};
    hi = u32x4{k.p[2][0], k.p[2][1], k.p[3][0], k.p[3][1]};
}

#ifdef SS_DEVBUILD
int dev_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int dev_env(const char*, int dflt) { return dflt; }
#endif

}  // namespace

__global__ __launch_bounds__(NTHR * NH) __attribute__((amdgpu_waves_per_eu(4)))
void conv3x3_ups_kernel(ConvArgs a, int total_tiles, int lds_b_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTHR);          // this thread's tile of the workgroup
    const int tid = (int)threadIdx.x % NTHR, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, m = lane & 31;
    const int ca = wave & 1, cb = wave >> 1;              // the wave's parity class: output rows 2 Y + ca, columns 2 X + cb
    const int X = m & 7, Y = m >> 3;
    char* sA = smem + half * kPatchBytes;
    char* sB = smem + NH * kPatchBytes;
    const float* sBias = (const float*)(sB + lds_b_bytes);

    const int H = a.H, W = a.W, Cout = a.Cout;            // Cout == 32
    const int nreg = a.C0 / 32, nups = a.C1 / 32;
    const int nch = 2 * nreg + nups;                      // stages per tile: two per skip chunk, one per upsampled chunk

    const int xcd = blockIdx.x & 7, local = (int)(blockIdx.x >> 3) * NH + half, gper = (int)(gridDim.x >> 3) * NH;
    const int per = (total_tiles + 7) >> 3;
    auto tile_of = [&](int loc, int it) -> int {
        const int idx = loc + it * gper;
        const int t = xcd * per + idx;
        return (idx < per && t < total_tiles) ? t : -1;
    };
    struct Tile { int n, y0, x0; };
    auto decode = [&](int t) -> Tile {
        Tile d;
        d.x0 = (t % a.tiles_x) * 16; t /= a.tiles_x;
        d.y0 = (t % a.tiles_y) * 8;
        d.n = t / a.tiles_y;
        return d;
    };

    // ---- this thread's patch pieces: geometry fixed for the kernel's lifetime ----
    constexpr int AIT = 3;                                // 10 x 18 pixels x 4 pieces = 720 pieces over 256 threads
    uint32_t pix_full[AIT], lds_off[AIT], flags = 0;
    const uint32_t part16 = (tid & 3) * 16;
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
        const int p = tid + NTHR * it, pix = p >> 2;
        const int pyy = pix / 18, pxx = pix - pyy * 18;
        lds_off[it] = pyy * kRowR + (pxx & 1) * kOdd + (pxx >> 1) * kPix + part16;
        pix_full[it] = pyy * W + pxx;                                       // from the patch origin (y0 - 1, x0 - 1)
        const uint32_t f = (pyy == 0 ? 1u : 0u) | (pyy == 9 ? 2u : 0u) | (pxx == 0 ? 4u : 0u) | (pxx == 17 ? 8u : 0u) | (p >= 720 ? 16u : 0u);
        flags |= f << (8 * it);
    }
    uint32_t pix_low, lds_low, flags_low;                 // 6 x 10 low-resolution pixels x 4 pieces = 240 pieces: one per thread
    {
        const int pix = tid >> 2;
        const int ly = pix / 10, lx = pix - ly * 10;
        lds_low = ly * kRowU + lx * kPix + part16;
        pix_low = ly * (W >> 1) + lx;                                       // from (y0 / 2 - 1, x0 / 2 - 1)
        flags_low = (ly == 0 ? 1u : 0u) | (ly == 5 ? 2u : 0u) | (lx == 0 ? 4u : 0u) | (lx == 9 ? 8u : 0u) | (tid >= 240 ? 16u : 0u);
    }
    u32x4 ra[AIT];
    // stage ci of a tile: ci < 2 nreg: skip chunk ci >> 1, part ci & 1 (0: low halves, 1: high halves); else upsampled chunk ci - 2 nreg
    auto issue_patch = [&](const Tile& d, int ci) {
        const uint32_t tm = (d.y0 == 0 ? 1u : 0u) | (d.y0 + 8 == H ? 2u : 0u) | (d.x0 == 0 ? 4u : 0u) | (d.x0 + 16 == W ? 8u : 0u) | 16u;
        if (ci < 2 * nreg) {
            const int chunk = ci >> 1;
            const int64_t plane = (ci & 1) ? 0 : a.lo_delta;
            const char* base = (const char*)a.src0 - kHdr + plane;
            const uint32_t cs2 = 2u * a.C0;
            const uint32_t tp = kHdr + ((((uint32_t)d.n * H + d.y0 - 1) * W + d.x0 - 1) * a.C0 + chunk * 32) * 2u + part16;   // mod 2^32
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                uint32_t off = __umul24(pix_full[it], cs2) + tp;
                if (flags & (tm << (8 * it))) off = 0;    // outside the picture (or past the patch): the zero header
                ra[it] = *(const u32x4*)(base + off);
            }
        } else {
            const char* base = (const char*)a.src1 - kHdr;
            const uint32_t cs2 = 2u * a.C1;
            const uint32_t tp = kHdr + ((((uint32_t)d.n * (H >> 1) + (d.y0 >> 1) - 1) * (W >> 1) + (d.x0 >> 1) - 1) * a.C1 + (ci - 2 * nreg) * 32) * 2u + part16;
            uint32_t off = __umul24(pix_low, cs2) + tp;
            if (flags_low & tm) off = 0;
            ra[0] = *(const u32x4*)(base + off);                          // high halves
            ra[1] = *(const u32x4*)(base + (off ? a.lo_delta + off : 0)); // low halves (the zero header has one plane's worth of zeros only)
        }
    };
    auto commit = [&](int ci) {                           // the patch of stage ci from ra into this tile's buffer
        if (ci < 2 * nreg) {
#pragma unroll
            for (int it = 0; it < AIT; ++it)
                if (!(flags & (16u << (8 * it)))) *(u32x4*)(sA + lds_off[it]) = ra[it];
        } else if (!(flags_low & 16u)) {
            *(u32x4*)(sA + lds_low) = ra[0];
            *(u32x4*)(sA + kLowPlane + lds_low) = ra[1];
        }
    };

    int it_tile = 0;
    struct Stage { int ci; Tile d; };
    auto next_stage = [&](const Stage& s0, Stage& n) -> bool {
        n = s0; n.ci = s0.ci + 1;
        if (n.ci == nch) {
            n.ci = 0;
            const int t = tile_of(local, ++it_tile);
            if (t < 0) return false;
            n.d = decode(t);
        }
        return true;
    };
    int my_stages = 0, max_stages = 0;
    {
        const int l0 = (int)(blockIdx.x >> 3) * NH;       // (the tile with the lowest index has the most positions)
        int n0 = 0, nmine = 0;
        while (tile_of(l0, n0) >= 0) ++n0;
        while (tile_of(l0 + half, nmine) >= 0) ++nmine;
        my_stages = nmine * nch;
        max_stages = n0 * nch;
        if (max_stages == 0) return;                      // whole workgroup idle
    }
    // (a tile without positions runs the prologue on position 0 -- valid addresses, results unused -- and then only keeps the beat)
    Stage cs{0, decode(my_stages == 0 ? 0 : tile_of(local, 0))}, n1 = cs, n2 = cs;

    for (int p = (int)threadIdx.x; p < lds_b_bytes / 16; p += NTHR * NH) *(u32x4*)(sB + p * 16) = *(const u32x4*)((const char*)a.wpk + (size_t)p * 16);
    for (int i = (int)threadIdx.x; i < Cout; i += NTHR * NH) ((float*)sBias)[i] = a.bias[i];
    issue_patch(cs.d, 0);
    commit(0);
    bool ok1 = next_stage(cs, n1), ok2 = false;
    if (ok1) issue_patch(n1.d, n1.ci);
    __syncthreads();

    f32x16 acc;
    // fragment bases of this lane: skip patch (rows 2 Y + ..., even / odd column halves), low-resolution patch
    const int base_r = 2 * Y * kRowR + X * kPix + hh * 16;
    const int base_u = Y * kRowU + X * kPix + hh * 16;
    const int boff0 = lane * 16;
    // where this lane's runs of its pixel (2 Y + ca, 2 X + cb) go, relative to the tile's origin
    const uint32_t st_off = (uint32_t)(((2 * Y + ca) * W + 2 * X + cb) * Cout + hh * 8) * 2u;
    // per tap of the skip patch: (ca + dy) rows down, column cb + dx: its parity half and its place there
    int tap_r[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        tap_r[t] = (ca + dy) * kRowR + ((cb + dx) & 1) * kOdd + ((cb + dx) >> 1) * kPix;
    }
    int tap_u[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) tap_u[t] = (ca + (t >> 1)) * kRowU + (cb + (t & 1)) * kPix;

    // Timing perturbation for the tests (development build only; ConvArgs::dbg bit 10, pattern in bits 11-12, as in conv4.hip): chosen waves
    // sleep ~1 us at the stage's synchronisation points.  Results must not change; a missing barrier shows up as a changed bit.
    int jit_n = 0;
    auto jitter = [&](int site) {
#ifdef SS_DEVBUILD
        if (a.dbg & 1024) {
            const int pat = (a.dbg >> 11) & 3, w16 = half * NW + wave;
            const bool z = pat == 0 ? ((w16 + site + jit_n) & 3) == 0 : pat == 1 ? w16 == 0 : pat == 2 ? w16 != 0 : (w16 & 1) != 0;
            if (z) __builtin_amdgcn_s_sleep(32);
        }
#else
        (void)site;
#endif
    };
    auto stage = [&]() -> bool {
        ++jit_n; jitter(0);
        const Tile cur = cs.d;
        const int ci = cs.ci;
        const bool last = ci == nch - 1;
        if (ci == 0) {                                    // accumulators start from the bias (the MFMA's C operand)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *(const f32x4*)(sBias + 8 * g + 4 * hh);
                acc[4 * g] = b4[0]; acc[4 * g + 1] = b4[1]; acc[4 * g + 2] = b4[2]; acc[4 * g + 3] = b4[3];
            }
        }
        if (a.dbg & 32) __builtin_amdgcn_s_setprio(2);
        // skip channels: nine taps x two 16-channel sub-steps; part 0: low halves against wh; part 1: high halves against wh and wl
        auto skip_stage = [&](auto part_c) {
            constexpr int PART = decltype(part_c)::value;
            const char* bb0 = sB + boff0 + (ci >> 1) * 2 * kBankR;
            const char* bb1 = bb0 + kBankR;
            constexpr int PM = 3;
            u32x4 pf[PM], wh_[PM], wl_[PM];
            auto load3 = [&](int st, int slot) {
                const int tap = st >> 1, sub = st & 1;
                pf[slot] = *(const u32x4*)(sA + base_r + tap_r[tap] + sub * 32);
                wh_[slot] = *(const u32x4*)(bb0 + tap * 2048 + sub * 1024);
                if constexpr (PART == 1) wl_[slot] = *(const u32x4*)(bb1 + tap * 2048 + sub * 1024);
            };
#pragma unroll
            for (int st = 0; st < PM - 1; ++st) load3(st, st);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + PM - 1 < 18) load3(st + PM - 1, (st + PM - 1) % PM);
                const u32x4 pixv = pf[st % PM];
                acc = mfma(wh_[st % PM], pixv, acc);
                if constexpr (PART == 1) acc = mfma(wl_[st % PM], pixv, acc);
            }
        };
        using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
        if (ci < 2 * nreg) {
            if (ci & 1) skip_stage(P1{}); else skip_stage(P0{});
        } else {
            // upsampled channels: the class's four pre-summed taps over the low-resolution patch, both halves at once: wh xl + wl xh + wh xh
            const char* bb0 = sB + boff0 + nreg * 2 * kBankR + (ci - 2 * nreg) * 2 * kBankU + wave * 4 * 2048;
            const char* bb1 = bb0 + kBankU;
            constexpr int PM = 2;
            u32x4 ph[PM], pl[PM], wh_[PM], wl_[PM];
            auto load4 = [&](int st, int slot) {
                const int tap = st >> 1, sub = st & 1;
                ph[slot] = *(const u32x4*)(sA + base_u + tap_u[tap] + sub * 32);
                pl[slot] = *(const u32x4*)(sA + kLowPlane + base_u + tap_u[tap] + sub * 32);
                wh_[slot] = *(const u32x4*)(bb0 + tap * 2048 + sub * 1024);
                wl_[slot] = *(const u32x4*)(bb1 + tap * 2048 + sub * 1024);
            };
            load4(0, 0);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (st + 1 < 8) load4(st + 1, (st + 1) % PM);
                acc = mfma(wh_[st % PM], pl[st % PM], acc);
                acc = mfma(wl_[st % PM], ph[st % PM], acc);
                acc = mfma(wh_[st % PM], ph[st % PM], acc);
            }
        }
        if (a.dbg & 32) __builtin_amdgcn_s_setprio(0);
        jitter(1);
        lds_barrier();                                    // every wave of the workgroup is done with this beat's LDS reads
        jitter(2);
        // ---- the tile's off-phase: commit the next stage's patch, request the one after, epilogue -- while other tiles multiply ----
        if (ok1) {
            commit(n1.ci);
            ok2 = next_stage(n1, n2);
            if (ok2) issue_patch(n2.d, n2.ci);
        }
        if (last) {
            __builtin_amdgcn_sched_barrier(0);
            uint32_t ovf = 0;
            Packed kh, kl;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // ReLU as an integer max (negative floats are negative integers), then hi = f16(v), lo = f16(v - hi)
                    const float a0 = acc[4 * g + 2 * h], a1 = acc[4 * g + 2 * h + 1];
                    const int b0 = __builtin_bit_cast(int, a0), b1 = __builtin_bit_cast(int, a1);
                    const float x0 = __builtin_bit_cast(float, b0 > 0 ? b0 : 0), x1 = __builtin_bit_cast(float, b1 > 0 ? b1 : 0);
                    kh.p[g][h] = pack_f16(x0, x1);
                    ovf = pk_max_u16(ovf, kh.p[g][h]);
                    kl.p[g][h] = split_lo(kh.p[g][h], x0, x1);
                }
            char* op = (char*)a.out + ((((uint32_t)cur.n * H + cur.y0) * W + cur.x0) * Cout) * 2u + st_off;
            u32x4 lo, hi;
            to_runs(kh, lo, hi);
            *(u32x4*)(op) = lo;
            *(u32x4*)(op + 32) = hi;
            to_runs(kl, lo, hi);
            *(u32x4*)(op + a.lo_delta) = lo;
            *(u32x4*)(op + a.lo_delta + 32) = hi;
            if (((ovf & 0x7fff7fffu) + 0x04000400u) & 0x80008000u) atomicOr(a.range_flag, 1);       // (rare: the engine turns it into SS_ERR_RANGE)
        }
        jitter(3);
        lds_barrier();
        jitter(4);
        cs = n1; n1 = n2;
        const bool more = ok1;
        ok1 = ok1 && ok2;
        return more;
    };
    // beats: tile q starts q barriers late and ends NH - 1 - q barriers late; a tile that runs out of stages keeps the beat
    for (int i = 0; i < half; ++i) lds_barrier();
    for (int s = 0; s < max_stages; ++s) {
        if (s < my_stages) stage();
        else { lds_barrier(); lds_barrier(); }
    }
    for (int i = half; i < NH - 1; ++i) lds_barrier();
}

// =========================================================================================================
// Ring form: the RES A launches of conv6 / conv7 / conv8 (h and the block's 1x1 projection r out, 32 output channels per position,
// several channel groups, banks streamed).  conv4.hip's two-slot ring holds whole 32-channel chunks (both halves: 40 KB) and a chunk
// of class banks would be 68 KB; here a ring ENTRY is one half (high or low) of a chunk's banks and a stage reads ONE entry:
//     skip chunk:       R0 = wh xl (18 products + 2 of the projection) | R1a = wh xh (18 + 2) | R1b = wl xh (18 + 2)
//                       entries WH (R0, R1a: 9 taps + the projection tap = 20 KB, + the group's two bias rows) and WL (R1b: 20 KB)
//     upsampled chunk:  U1 = wh xl + wh xh (16 + 4) | U2 = wl xh (8 + 2)
//                       entries UH (U1) and UL (U2): 4 classes x 4 taps + the projection tap = 34 KB each
// Stage s of the walk multiplies in beat 2 s + q on tile q, so at any beat the entries of three consecutive stages are live: THREE
// slots of 34 KB (102 KB) beside the four patches (57.5 KB) -- the LDS to 512 bytes.  Entry e takes the slot of entry e - 3, whose
// last stage lies at least three stages before e's first (two whole entries in between): its last read is in beat 2 s_e - 3 at the
// latest, and e is written in beats 2 s_e - 2 and 2 s_e - 1, a quarter by each tile in the off-phase it has there (tile q: the
// off-phase of its stage s_e - d_q, d = 1, 2, 2, 3), from registers requested one off-phase earlier.
// R1b and U2 use the patch of the stage before them, so a chunk costs as many patch loads as in the resident form.
// r leaves as launch B's fp32 accumulator fragments (conv4.hip r_mtile): B's M-tile is two rows x 16 pixels with lane
// m' = (x & 1) | (y << 1) | ((x >> 1) << 2); the lane that owns pixel (2 Y + ca, 2 X + cb) here writes its 16 values where B's lane of
// that pixel reads them (M-tile Y of the tile, lane 32 hh + 4 X + 2 ca + cb): two 32-byte pieces per lane.
// =========================================================================================================
namespace {
constexpr int kEntR = 10 * 2048;                          // one half of a skip chunk's banks: 9 taps + the projection's
constexpr int kEntRH = kEntR + 256;                       // the high half carries [32] bias and [32] projection bias of the group (fp32)
constexpr int kEntU = 17 * 2048;                          // one half of an upsampled chunk's class banks + the projection's
constexpr int kSlot = kEntU;
constexpr int kGL = (kEntU / 32 + NTHR - 1) / NTHR;       // LDS-DMA instructions per thread for half of the largest entry: 5 (the fifth for one wave)
}  // namespace

__global__ __launch_bounds__(NTHR * NH) __attribute__((amdgpu_waves_per_eu(4)))
void conv3x3_upsr_kernel(ConvArgs a, int total_tiles, int) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // timing-only ablations (development build, SOFTSPOKEN_DBG bits 16..20: results are wrong): no matrix products / no operand reads from
    // the LDS / no patch loads from memory / no ring DMA / no epilogue stores -- what a beat is made of
#ifdef SS_DEVBUILD
#define SS_ABL(bit) (a.dbg & (1 << (bit)))
#else
#define SS_ABL(bit) false
#endif
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTHR);
    const int tid = (int)threadIdx.x % NTHR, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, m = lane & 31;
    const int ca = wave & 1, cb = wave >> 1;
    const int X = m & 7, Y = m >> 3;
    char* sA = smem + half * kPatchBytes;
    char* sR = smem + NH * kPatchBytes;                   // three slots

    const int H = a.H, W = a.W, Cout = a.Cout;
    const int ngroups = Cout / 32;
    const int nreg = a.C0 / 32, nups = a.C1 / 32;
    const int nst = 3 * nreg + 2 * nups;                  // stages per (position, channel group)
    const int group_bytes = nreg * (kEntRH + kEntR) + nups * 2 * kEntU;

    // A stage of the walk, kept as counters (no division on the way): sec 0 = skip chunk `chunk`, part 0 (R0: low plane, WH) / 1 (R1a:
    // high plane, WH) / 2 (R1b: high plane again, WL); sec 1 = upsampled chunk, part 0 (U1: both planes, UH) / 1 (U2: high plane again, UL)
    struct Sp { int sec, chunk, part; };
    auto sp_next = [&](Sp& p) -> bool {                   // true when it wrapped to the next (position, group)
        if (++p.part == (p.sec ? 2 : 3)) {
            p.part = 0;
            if (++p.chunk == (p.sec ? nups : nreg)) {
                p.chunk = 0; p.sec ^= 1;
                if (p.sec && nups == 0) p.sec = 0;        // (a block without an upsampled input: skip chunks only)
                return p.sec == 0;
            }
        }
        return false;
    };
    auto starts_entry = [&](const Sp& p) -> bool { return p.sec || p.part != 1; };
    auto needs_patch = [&](const Sp& p) -> bool { return p.sec ? p.part == 0 : p.part != 2; };
    auto entry_off = [&](const Sp& p) -> int {
        return p.sec ? nreg * (kEntRH + kEntR) + (2 * p.chunk + p.part) * kEntU : p.chunk * (kEntRH + kEntR) + (p.part == 2 ? kEntRH : 0);
    };
    auto entry_size = [&](const Sp& p) -> int { return p.sec ? kEntU : (p.part == 2 ? kEntR : kEntRH); };

    // positions: tile j of the (quarter-)workgroup with index loc on this XCD is position xcd per_pos + loc + j gper; every tile of the
    // workgroup walks a position's channel groups in the same order
    const int xcd = blockIdx.x & 7, local = (int)(blockIdx.x >> 3) * NH + half, gper = (int)(gridDim.x >> 3) * NH;
    const int total_pos = total_tiles / ngroups, per_pos = (total_pos + 7) >> 3;
    const int lim = min(per_pos, total_pos - xcd * per_pos);
    auto count_pos = [&](int loc) -> int { return lim > loc ? (lim - loc + gper - 1) / gper : 0; };
    struct Tile { int n, y0, x0, g; };
    auto decode_pos = [&](int pos, int g) -> Tile {
        Tile d;
        d.g = g;
        d.x0 = (pos % a.tiles_x) * 16; pos /= a.tiles_x;
        d.y0 = (pos % a.tiles_y) * 8;
        d.n = pos / a.tiles_y;
        return d;
    };
    const int my_pos = count_pos(local);
    const int my_stages = my_pos * ngroups * nst;
    const int max_stages = count_pos((int)(blockIdx.x >> 3) * NH) * ngroups * nst;     // (the tile with the lowest index has the most positions)
    if (max_stages == 0) return;                          // whole workgroup idle

    // ---- this thread's patch pieces (as in the resident form) ----
    constexpr int AIT = 3;
    uint32_t pix_full[AIT], lds_off[AIT], flags = 0;
    const uint32_t part16 = (tid & 3) * 16;
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
        const int p = tid + NTHR * it, pix = p >> 2;
        const int pyy = pix / 18, pxx = pix - pyy * 18;
        lds_off[it] = pyy * kRowR + (pxx & 1) * kOdd + (pxx >> 1) * kPix + part16;
        pix_full[it] = pyy * W + pxx;
        const uint32_t f = (pyy == 0 ? 1u : 0u) | (pyy == 9 ? 2u : 0u) | (pxx == 0 ? 4u : 0u) | (pxx == 17 ? 8u : 0u) | (p >= 720 ? 16u : 0u);
        flags |= f << (8 * it);
    }
    uint32_t pix_low, lds_low, flags_low;
    {
        const int pix = tid >> 2;
        const int ly = pix / 10, lx = pix - ly * 10;
        lds_low = ly * kRowU + lx * kPix + part16;
        pix_low = ly * (W >> 1) + lx;
        flags_low = (ly == 0 ? 1u : 0u) | (ly == 5 ? 2u : 0u) | (lx == 0 ? 4u : 0u) | (lx == 9 ? 8u : 0u) | (tid >= 240 ? 16u : 0u);
    }
    u32x4 ra[AIT];
    auto issue_patch = [&](const Tile& d, const Sp& p) {
        if (SS_ABL(18)) return;
        const uint32_t tm = (d.y0 == 0 ? 1u : 0u) | (d.y0 + 8 == H ? 2u : 0u) | (d.x0 == 0 ? 4u : 0u) | (d.x0 + 16 == W ? 8u : 0u) | 16u;
        if (p.sec == 0) {
            const int64_t plane = p.part ? 0 : a.lo_delta;                     // part 0: low halves; part 1: high halves
            const char* base = (const char*)a.src0 - kHdr + plane;
            const uint32_t cs2 = 2u * a.C0;
            const uint32_t tp = kHdr + ((((uint32_t)d.n * H + d.y0 - 1) * W + d.x0 - 1) * a.C0 + p.chunk * 32) * 2u + part16;   // mod 2^32
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                uint32_t off = __umul24(pix_full[it], cs2) + tp;
                if (flags & (tm << (8 * it))) off = 0;    // outside the picture (or past the patch): the zero header
                ra[it] = *(const u32x4*)(base + off);
            }
        } else {
            const char* base = (const char*)a.src1 - kHdr;
            const uint32_t cs2 = 2u * a.C1;
            const uint32_t tp = kHdr + ((((uint32_t)d.n * (H >> 1) + (d.y0 >> 1) - 1) * (W >> 1) + (d.x0 >> 1) - 1) * a.C1 + p.chunk * 32) * 2u + part16;
            uint32_t off = __umul24(pix_low, cs2) + tp;
            if (flags_low & tm) off = 0;
            ra[0] = *(const u32x4*)(base + off);                          // high halves
            ra[1] = *(const u32x4*)(base + (off ? a.lo_delta + off : 0)); // low halves (the zero header has one plane's worth of zeros only)
        }
    };
    auto commit = [&](int sec) {
        if (sec == 0) {
#pragma unroll
            for (int it = 0; it < AIT; ++it)
                if (!(flags & (16u << (8 * it)))) *(u32x4*)(sA + lds_off[it]) = ra[it];
        } else if (!(flags_low & 16u)) {
            *(u32x4*)(sA + lds_low) = ra[0];
            *(u32x4*)(sA + kLowPlane + lds_low) = ra[1];
        }
    };

    // ---- the walk: the current (position, group), the one behind it (decoded once per position, in a light off-phase), and the stage
    // whose patch is in flight in `ra` (it lies in the current position or in the next) ----
    const int pos0 = xcd * per_pos + local;
    Tile cur = decode_pos(my_pos ? pos0 : 0, 0), nxt = cur;
    int cur_g = 0, cur_j = 0, nxt_g = 0, nxt_j = 0;
    bool nxt_ok = false;
    auto compute_next = [&]() {
        nxt_g = cur_g + 1; nxt_j = cur_j;
        if (nxt_g == ngroups) { nxt_g = 0; ++nxt_j; }
        nxt_ok = nxt_j < my_pos;
        if (nxt_ok) nxt = decode_pos(pos0 + nxt_j * gper, nxt_g);
    };
    struct Cur { Sp sp; Tile d; bool ok; };
    auto advance = [&](Cur& c) {
        if (sp_next(c.sp)) { c.d = nxt; c.ok = nxt_ok; }
    };
    Cur ip{Sp{0, 0, 0}, cur, true};

    // ---- the bank ring: prologue = the walk's first three entries (group 0's), by every thread ----
    {
        Sp p{0, 0, 0};
        for (int e = 0; e < 3; ++e) {
            const char* src = (const char*)a.wpk + entry_off(p);
            const int np = entry_size(p) / 16;
            for (int q = (int)threadIdx.x; q < np; q += NTHR * NH) *(u32x4*)(sR + e * kSlot + q * 16) = *(const u32x4*)(src + (size_t)q * 16);
            do sp_next(p); while (!starts_entry(p));
        }
    }
    // loader duty (tiles 1 and 3, half an entry each): at the end of the off-phase that lies in beat 2 s_e - 2 -- stage s_e - 2 of tile
    // 1, s_e - 3 of tile 3 -- the entry that starts at stage s_e goes straight from memory into its slot (LDS-DMA: no registers, no
    // write pass); the issuing waves retire it with the vmcnt(0) in front of the barrier that ends beat 2 s_e - 1, the readers pass
    // that barrier before their first read.  The cursor runs d stages ahead of the tile's own stage count.
    Sp rq{0, 0, 0};
    int rq_g = 0, rq_t = 0, rq_e = 0, rq_slot = 0;        // target stage: group, absolute number; entries started before it: count, slot of the next
    auto rq_advance = [&]() {
        if (starts_entry(rq)) { ++rq_e; rq_slot = rq_slot == 2 ? 0 : rq_slot + 1; }
        ++rq_t;
        if (sp_next(rq)) { if (++rq_g == ngroups) rq_g = 0; }
    };
    if (half & 1) for (int i = 0; i < (half == 1 ? 2 : 3); ++i) rq_advance();
    auto ring_issue = [&]() {
        if (!(half & 1)) return;
        if (!SS_ABL(19) && rq_t < max_stages && rq_e >= 3 && starts_entry(rq)) {
            const int total = entry_size(rq) / 16;                             // 16-byte pieces
            const int h0 = ((total / 2 + 63) / 64) * 64;                       // tile 1's share: whole wave-instructions
            const int base = half == 1 ? 0 : h0, cnt = half == 1 ? h0 : total - h0;
            const char* src = (const char*)a.wpk + (size_t)rq_g * group_bytes + entry_off(rq) + base * 16;
            char* dst = sR + rq_slot * kSlot + base * 16 + wave * 1024;        // the wave's 1 KB of an instruction: base + 16 x lane by the hardware
#pragma unroll
            for (int it = 0; it < kGL; ++it) {
                const int idx = tid + NTHR * it;
                if (idx < cnt)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)idx * 16),
                                                     (__attribute__((address_space(3))) void*)(uintptr_t)(dst + it * NTHR * 16), 16, 0, 0);
            }
        }
        rq_advance();
    };

    issue_patch(ip.d, ip.sp);
    commit(0);
    if (my_stages == 0) ip.ok = false;                    // (a tile without positions ran the prologue on position 0 and only keeps the beat)
    else do advance(ip); while (ip.ok && !needs_patch(ip.sp));
    if (ip.ok) issue_patch(ip.d, ip.sp);
    __syncthreads();

    f32x16 acc, racc;
    auto mm = [&](const u32x4& w, const u32x4& x, const f32x16& c) -> f32x16 { return SS_ABL(16) ? c : mfma(w, x, c); };
    const int base_r = 2 * Y * kRowR + X * kPix + hh * 16;
    const int base_u = Y * kRowU + X * kPix + hh * 16;
    const int boff0 = lane * 16;
    const uint32_t st_off = (uint32_t)(((2 * Y + ca) * W + 2 * X + cb) * Cout + hh * 8) * 2u;
    const uint32_t r_lane = (uint32_t)(32 * hh + 4 * X + 2 * ca + cb) * 32u;
    int tap_r[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        tap_r[t] = (ca + dy) * kRowR + ((cb + dx) & 1) * kOdd + ((cb + dx) >> 1) * kPix;
    }
    int tap_u[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) tap_u[t] = (ca + (t >> 1)) * kRowU + (cb + (t & 1)) * kPix;
    constexpr int kCentreU = kRowU + kPix;                // the low-resolution pixel under this lane's output pixel: patch (Y + 1, X + 1)

    // Timing perturbation for the tests (development build only; ConvArgs::dbg bit 10, pattern in bits 11-12, as in conv4.hip)
    int jit_n = 0;
#ifdef SS_DEVBUILD
    // stamps (ConvArgs::stamps, SOFTSPOKEN_STAMP_LAYER): shader-clock time between the stage's synchronisation points, summed per wave:
    // [0] multiply, [1] wait at barrier 1, [2] off-phase, [3] wait at barrier 2, [4] loop turn; of the off-phase: [5] commit, [6] epilogue,
    // [7] next patch's request, [8] ring duty
    uint32_t st_prev = 0, st_sub = 0, st_sum[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#endif
    auto substamp = [&](int seg) {
#ifdef SS_DEVBUILD
        if (a.stamps) { const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime(); if (seg >= 5) st_sum[seg] += t - st_sub; st_sub = t; }
#else
        (void)seg;
#endif
    };
    auto jitter = [&](int site) {
#ifdef SS_DEVBUILD
        if (a.stamps) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime();
            const int seg = site == 0 ? 4 : site - 1;
            if (site != 0 || jit_n > 1) st_sum[seg] += t - st_prev;
            st_prev = t;
        }
        if (a.dbg & 1024) {
            const int pat = (a.dbg >> 11) & 3, w16 = half * NW + wave;
            const bool z = pat == 0 ? ((w16 + site + jit_n) & 3) == 0 : pat == 1 ? w16 == 0 : pat == 2 ? w16 != 0 : (w16 & 1) != 0;
            if (z) __builtin_amdgcn_s_sleep(32);
        }
#else
        (void)site;
#endif
    };
    // h of a finished position: ReLU, split, store (its r went out in the position's last off-phase)
    f32x16 acc_done;
    uint32_t done_off = 0;                                // element offset of the finished position's tile origin and channel group
    bool have_done = false;
    auto store_h = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        uint32_t ovf = 0;
        Packed kh, kl;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // ReLU as an integer max (negative floats are negative integers), then hi = f16(v), lo = f16(v - hi)
                const float a0 = acc_done[4 * g + 2 * h], a1 = acc_done[4 * g + 2 * h + 1];
                const int b0 = __builtin_bit_cast(int, a0), b1 = __builtin_bit_cast(int, a1);
                const float x0 = __builtin_bit_cast(float, b0 > 0 ? b0 : 0), x1 = __builtin_bit_cast(float, b1 > 0 ? b1 : 0);
                kh.p[g][h] = pack_f16(x0, x1);
                ovf = pk_max_u16(ovf, kh.p[g][h]);
                kl.p[g][h] = split_lo(kh.p[g][h], x0, x1);
            }
        char* op = (char*)a.out + done_off * 2u + st_off;
        u32x4 lo, hi;
        to_runs(kh, lo, hi);
        *(u32x4*)(op) = lo;
        *(u32x4*)(op + 32) = hi;
        to_runs(kl, lo, hi);
        *(u32x4*)(op + a.lo_delta) = lo;
        *(u32x4*)(op + a.lo_delta + 32) = hi;
        if (((ovf & 0x7fff7fffu) + 0x04000400u) & 0x80008000u) atomicOr(a.range_flag, 1);           // (rare: the engine turns it into SS_ERR_RANGE)
        __builtin_amdgcn_sched_barrier(0);
    };
    int e_slot = 0;                                       // ring slot of the current stage's entry
    int stage_no = 0;
    // One stage; its kind is a compile-time constant (the position's stages are written out below), the chunk a loop counter:
    // KIND 0 = R0, 1 = R1a, 2 = R1b of skip chunk c; 3 = U1, 4 = U2 of upsampled chunk c
    auto stage = [&](auto kind_c, int c) {
        constexpr int KIND = decltype(kind_c)::value;
        ++jit_n; jitter(0);
        const bool last = nups ? (KIND == 4 && c == nups - 1) : (KIND == 2 && c == nreg - 1);
        const char* ent = sR + e_slot * kSlot;
        if (KIND == 0 && c == 0) {                        // accumulators start from the biases (the MFMA's C operand): tail of the group's first entry
            const float* sBias = (const float*)(ent + kEntR);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *(const f32x4*)(sBias + 8 * g + 4 * hh);
                const f32x4 r4 = *(const f32x4*)(sBias + 32 + 8 * g + 4 * hh);
                acc[4 * g] = b4[0]; acc[4 * g + 1] = b4[1]; acc[4 * g + 2] = b4[2]; acc[4 * g + 3] = b4[3];
                racc[4 * g] = r4[0]; racc[4 * g + 1] = r4[1]; racc[4 * g + 2] = r4[2]; racc[4 * g + 3] = r4[3];
            }
        }
        if (a.dbg & 32) __builtin_amdgcn_s_setprio(2);
        if constexpr (KIND <= 2) {
            // a skip stage: nine taps x two 16-channel sub-steps of ONE plane against ONE half of the weights; the centre tap also feeds r
            const char* bb = ent + boff0;
            constexpr int PM = 3;
            u32x4 pf[PM], wf[PM];
            auto load2 = [&](int st, int slot) {
                if (SS_ABL(17)) return;
                const int tap = st >> 1, sub = st & 1;
                pf[slot] = *(const u32x4*)(sA + base_r + tap_r[tap] + sub * 32);
                wf[slot] = *(const u32x4*)(bb + tap * 2048 + sub * 1024);
            };
            u32x4 wr0, wr1;
#pragma unroll
            for (int st = 0; st < PM - 1; ++st) load2(st, st);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + PM - 1 < 18) load2(st + PM - 1, (st + PM - 1) % PM);
                if (st == 6) wr0 = *(const u32x4*)(bb + 9 * 2048);             // the projection's fragments, two steps ahead of their products
                if (st == 7) wr1 = *(const u32x4*)(bb + 9 * 2048 + 1024);
                acc = mm(wf[st % PM], pf[st % PM], acc);
                if (st == 8) racc = mm(wr0, pf[st % PM], racc);
                if (st == 9) racc = mm(wr1, pf[st % PM], racc);
            }
        } else if constexpr (KIND == 3) {
            // U1: the class's four pre-summed taps over both planes of the low-resolution patch against the high halves: wh xl + wh xh
            const char* bb = ent + boff0 + wave * 4 * 2048;
            constexpr int PM = 2;
            u32x4 ph[PM], pl[PM], wf[PM];
            auto load3 = [&](int st, int slot) {
                if (SS_ABL(17)) return;
                const int tap = st >> 1, sub = st & 1;
                ph[slot] = *(const u32x4*)(sA + base_u + tap_u[tap] + sub * 32);
                pl[slot] = *(const u32x4*)(sA + kLowPlane + base_u + tap_u[tap] + sub * 32);
                wf[slot] = *(const u32x4*)(bb + tap * 2048 + sub * 1024);
            };
            load3(0, 0);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (st + 1 < 8) load3(st + 1, (st + 1) % PM);
                acc = mm(wf[st % PM], pl[st % PM], acc);
                acc = mm(wf[st % PM], ph[st % PM], acc);
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const u32x4 w = *(const u32x4*)(ent + boff0 + 16 * 2048 + sub * 1024);
                const u32x4 xh = *(const u32x4*)(sA + base_u + kCentreU + sub * 32), xl = *(const u32x4*)(sA + kLowPlane + base_u + kCentreU + sub * 32);
                racc = mm(w, xl, racc);
                racc = mm(w, xh, racc);
            }
        } else {
            // U2: the high plane against the low halves: wl xh
            const char* bb = ent + boff0 + wave * 4 * 2048;
            constexpr int PM = 3;
            u32x4 ph[PM], wf[PM];
            auto load2 = [&](int st, int slot) {
                if (SS_ABL(17)) return;
                const int tap = st >> 1, sub = st & 1;
                ph[slot] = *(const u32x4*)(sA + base_u + tap_u[tap] + sub * 32);
                wf[slot] = *(const u32x4*)(bb + tap * 2048 + sub * 1024);
            };
#pragma unroll
            for (int st = 0; st < PM - 1; ++st) load2(st, st);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (st + PM - 1 < 8) load2(st + PM - 1, (st + PM - 1) % PM);
                acc = mm(wf[st % PM], ph[st % PM], acc);
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const u32x4 w = *(const u32x4*)(ent + boff0 + 16 * 2048 + sub * 1024);
                const u32x4 xh = *(const u32x4*)(sA + base_u + kCentreU + sub * 32);
                racc = mm(w, xh, racc);
            }
        }
        if (a.dbg & 32) __builtin_amdgcn_s_setprio(0);
        jitter(1);
        vm_lds_barrier();                                 // every wave of the workgroup is done with this beat's LDS reads; this wave's LDS-DMA has landed
        jitter(2);
        // ---- off-phase: the next stage's patch (when it has one of its own) into the LDS, the epilogue with the patch registers free,
        // the next patch requested, this tile's share of a ring entry set off ----
        const bool more = stage_no + 1 < my_stages;
        // the stage behind this one: R0 -> R1a (own patch) | R1a -> R1b (none) | R1b -> the next chunk's R0, or U1 | U1 -> U2 (none) | U2 -> U1, or R0
        constexpr bool next_has_patch = KIND == 0 || KIND == 2 || KIND == 4;
        const int next_sec = KIND == 0 ? 0 : KIND == 2 ? ((c == nreg - 1 && nups) ? 1 : 0) : (c == nups - 1 ? 0 : 1);
        const bool np = next_has_patch && more;
        substamp(0);
        if (np) commit(next_sec);                         // (`ra` holds exactly this stage's patch: ip is the first stage with a patch behind the last commit)
        substamp(5);
        if (last && !SS_ABL(20)) {
            // r now: un-activated (its bias came in through C), as launch B's accumulator fragments.  h two stages later, from a copy of
            // the accumulator, in the off-phase of the next position's R1a, which has no patch to commit or request: the off-phase behind
            // a position's last stage has both, and with the whole epilogue in it that beat was twice as long as the others
            char* rq_p = (char*)a.res_out +
                         ((((((uint32_t)cur.n * (H >> 1) + (cur.y0 >> 1) + Y) * (W >> 4) + (cur.x0 >> 4)) * (uint32_t)ngroups) + cur.g) * 2048u + r_lane);
            *(f32x4*)(rq_p) = f32x4{racc[0], racc[1], racc[2], racc[3]};
            *(f32x4*)(rq_p + 16) = f32x4{racc[4], racc[5], racc[6], racc[7]};
            *(f32x4*)(rq_p + a.lo_delta) = f32x4{racc[8], racc[9], racc[10], racc[11]};
            *(f32x4*)(rq_p + a.lo_delta + 16) = f32x4{racc[12], racc[13], racc[14], racc[15]};
        }
        if (last) { acc_done = acc; done_off = (((uint32_t)cur.n * H + cur.y0) * W + cur.x0) * Cout + cur.g * 32; have_done = true; }
        if (KIND == 1 && c == 0 && have_done && !SS_ABL(20)) { store_h(); have_done = false; }
        if (KIND == 1 && c == 0) compute_next();          // (before the patch requests reach into the next position: the second chunk's R0 at the earliest)
        if (last && more) { cur = nxt; cur_g = nxt_g; cur_j = nxt_j; }
        substamp(6);
        if (np) {
            do advance(ip); while (ip.ok && !needs_patch(ip.sp));
            if (ip.ok) issue_patch(ip.d, ip.sp);
        }
        substamp(7);
        ring_issue();
        substamp(8);
        jitter(3);
        lds_barrier();
        jitter(4);
        if (KIND != 0) e_slot = e_slot == 2 ? 0 : e_slot + 1;         // (every stage but R1a starts a new entry)
        ++stage_no;
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>; using K4 = std::integral_constant<int, 4>;
    // beats: tile q starts q barriers late and ends NH - 1 - q barriers late; a tile that runs out of positions keeps the beat and its loader duty
    const int my_pg = my_pos * ngroups, max_pg = max_stages / nst;
    for (int i = 0; i < half; ++i) lds_barrier();
    for (int pg = 0; pg < max_pg; ++pg) {
        if (pg < my_pg) {
            for (int c = 0; c < nreg; ++c) { stage(K0{}, c); stage(K1{}, c); stage(K2{}, c); }
            for (int c = 0; c < nups; ++c) { stage(K3{}, c); stage(K4{}, c); }
        } else {
            for (int st = 0; st < nst; ++st) { vm_lds_barrier(); ring_issue(); lds_barrier(); }
        }
    }
    for (int i = half; i < NH - 1; ++i) lds_barrier();
    if (have_done && !SS_ABL(20)) store_h();              // the tile's last position
#ifdef SS_DEVBUILD
    if (a.stamps && lane == 0) {
        uint32_t* sp = (uint32_t*)a.stamps + (((size_t)blockIdx.x * NH + half) * NW + wave) * 16;
        for (int i = 0; i < 5; ++i) sp[i] = st_sum[i];
        sp[5] = (uint32_t)jit_n; sp[6] = st_sum[5]; sp[7] = st_sum[6]; sp[8] = st_sum[7]; sp[9] = st_sum[8];
    }
#endif
#undef SS_ABL
}

namespace {

struct UpsChoice { bool ok; int total, lds_b, grid; size_t lds; };

UpsChoice choose_ups(ConvArgs& a, int num_cus) {
    UpsChoice c{};
    static const int on = dev_env("SOFTSPOKEN_UPS", 1);
    if (!on) return c;
    if (!a.plain || !a.src0 || !a.src1 || !a.out || !a.wpk || !a.bias || !a.range_flag || a.lo_delta <= 0) return c;
    if (a.res_out || a.res_in || a.pool_out || a.rank1_src || a.first_w || a.flat_part || a.proj_w || !a.relu || a.R0 || a.R1) return c;
    if (a.Cout != 32 || a.C0 < 32 || a.C1 < 32 || a.C0 % 32 || a.C1 % 32 || a.H % 8 || a.W % 16 || ((a.H | a.W) & 1)) return c;
    if ((double)a.N * a.H * a.W * std::max(a.Cout, std::max(a.C0, a.C1)) * 2.0 + kHdr >= 4294967296.0) return c;   // 32-bit byte offsets
    a.tiles_y = a.H / 8; a.tiles_x = a.W / 16;
    const long total_l = (long)a.N * a.tiles_y * a.tiles_x;
    if (total_l <= 0 || total_l > 0x7fffffff) return c;
    c.total = (int)total_l;
    c.lds_b = (a.C0 / 32) * 2 * kBankR + (a.C1 / 32) * 2 * kBankU;
    c.lds = (size_t)NH * kPatchBytes + c.lds_b + (size_t)a.Cout * 4;
    if (c.lds > 160 * 1024) return c;
    c.grid = (num_cus + 7) / 8 * 8;
    if (c.grid * NH > c.total) c.grid = ((c.total + NH - 1) / NH + 7) / 8 * 8;
    c.ok = true;
    return c;
}

}  // namespace

namespace {

struct UpsrChoice { bool ok; int total, grid; size_t lds; };

UpsrChoice choose_upsr(ConvArgs& a, int num_cus) {
    UpsrChoice c{};
    static const int on = dev_env("SOFTSPOKEN_UPSR", 1);
    if (!on) return c;
    if (a.plain || !a.src0 || !a.out || !a.res_out || !a.wpk || !a.range_flag || a.lo_delta <= 0) return c;
    if (a.res_in || a.pool_out || a.rank1_src || a.first_w || a.flat_part || a.proj_w || !a.relu || a.R0 || a.R1) return c;
    if (a.C1 ? !a.src1 : on < 2) return c;                // (development build, SOFTSPOKEN_UPSR=2: the encoder's A launches through this ring as well)
    if (a.Cout < 32 || a.Cout % 32 || a.C0 < 64 || a.C0 % 32 || a.C1 % 32 || a.H % 8 || a.W % 16 || ((a.H | a.W) & 1)) return c;
    if ((double)a.N * a.H * a.W * std::max(a.Cout, std::max(a.C0, a.C1)) * 2.0 + kHdr >= 4294967296.0) return c;   // 32-bit byte offsets
    a.tiles_y = a.H / 8; a.tiles_x = a.W / 16;
    const long total_l = (long)a.N * a.tiles_y * a.tiles_x * (a.Cout / 32);
    if (total_l <= 0 || total_l > 0x7fffffff) return c;
    c.total = (int)total_l;
    c.lds = (size_t)NH * kPatchBytes + 3 * (size_t)kSlot;
    if (c.lds > 160 * 1024) return c;
    const long pos = total_l / (a.Cout / 32);
    c.grid = (num_cus + 7) / 8 * 8;
    if ((long)c.grid * NH > pos) c.grid = (int)(((pos + NH - 1) / NH + 7) / 8 * 8);
    c.ok = true;
    return c;
}

}  // namespace

bool conv_upsr_supports(const ConvArgs& a_in, int num_cus) {
    ConvArgs a = a_in;
    return choose_upsr(a, num_cus).ok;
}

const char* conv_upsr_variant() { return "conv3x3_upsr_kernel"; }

// bytes of the packed weights (weights.hip pack_conv_split_upsr): per 32-channel output group the entries in walk order
size_t conv_upsr_weight_bytes(int C0, int C1, int Cout) {
    return (size_t)(Cout / 32) * ((size_t)(C0 / 32) * (kEntRH + kEntR) + (size_t)(C1 / 32) * 2 * kEntU);
}

hipError_t launch_conv3x3_upsr(const ConvArgs& a_in, int num_cus, hipStream_t s) {
    ConvArgs a = a_in;
    const UpsrChoice c = choose_upsr(a, num_cus);
    if (!c.ok) return hipErrorInvalidValue;
    static std::atomic<uint64_t> attr_done{0};              // (per device: kernels.h allow_full_lds)
    if (hipError_t e = allow_full_lds((const void*)conv3x3_upsr_kernel, attr_done)) return e;
    hipLaunchKernelGGL(conv3x3_upsr_kernel, dim3(c.grid), dim3(NTHR * NH), c.lds, s, a, c.total, 0);
    return hipGetLastError();
}

bool conv_ups_supports(const ConvArgs& a_in, int num_cus) {
    ConvArgs a = a_in;
    return choose_ups(a, num_cus).ok;
}

const char* conv_ups_variant() { return "conv3x3_ups_kernel"; }

// bytes of the packed weights this launch reads (weights.hip pack_conv_split_ups)
size_t conv_ups_weight_bytes(int C0, int C1) { return (size_t)(C0 / 32) * 2 * kBankR + (size_t)(C1 / 32) * 2 * kBankU; }

hipError_t launch_conv3x3_ups(const ConvArgs& a_in, int num_cus, hipStream_t s) {
    ConvArgs a = a_in;
    const UpsChoice c = choose_ups(a, num_cus);
    if (!c.ok) return hipErrorInvalidValue;
    static std::atomic<uint64_t> attr_done{0};              // (per device: kernels.h allow_full_lds)
    if (hipError_t e = allow_full_lds((const void*)conv3x3_ups_kernel, attr_done)) return e;
    hipLaunchKernelGGL(conv3x3_ups_kernel, dim3(c.grid), dim3(NTHR * NH), c.lds, s, a, c.total, c.lds_b);
    return hipGetLastError();
}

}  // namespace ss
