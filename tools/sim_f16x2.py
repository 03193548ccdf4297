"""CPU simulation of the conv stack's f16x2 arithmetic (development aid; nothing here is on the product path).

    python tools/sim_f16x2.py [--windows N] [--hostile] [--no-norm] [--winograd LAYERS|all] [--wino-f16-transform]

What is simulated, op for op what csrc/conv4.hip + csrc/weights.hip do in the f16x2 mode:
  * BatchNorm folded in float64, cast to fp32 (weights.hip fold_conv_bn);
  * the power-of-two channel normalisation of weights.hip (norm_exponent / scale_folded), unless --no-norm;
  * every activation tensor stored as two f16 halves (hi = f16(x), lo = f16(x - hi)), every weight likewise, a term w x as the
    three products wh xh + wh xl + wl xh accumulated in fp32 (torch's fp32 conv on f16-representable operands: every product is
    exact in fp32, the summation order differs from the kernel's, which is what any fp32 path differs by);
  * the residual branch, ReLU, pool, upsample, concat on fp32 values; conv_flatten on the halves; the 1-D head in fp32.
--winograd: the named blocks' 3x3 convs as Winograd F(2x2, 3x3): filter transform G g G^T in float64 then split, input transform
  B^T d B in fp32 on the reconstructed values then split (or, --wino-f16-transform, on the two planes separately in fp32 --
  what a kernel that transforms hi and lo on their own would do), 16 element-wise products per tile (three f16 products each),
  output transform A^T m A in fp32.  VERDICT r02 item 2 asks for this number before any kernel is written.
Truth = the oracle's network (oracle/oracle_np.py) in float64; the fp32 oracle's own distance to it is printed beside.
"""
import argparse
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from softspoken_amd import synth            # noqa: E402
from oracle import oracle_np as O           # noqa: E402

BLOCKS = [("conv1_1", None, None), ("conv2_1", "conv1_1", None), ("conv3_1", "conv2_1", None), ("conv4_1", "conv3_1", None),
          ("conv_bottleneck", "conv4_1", None), ("encoder_out", "conv_bottleneck", None), ("conv6", "conv4_1", "encoder_out"),
          ("conv7", "conv3_1", "conv6"), ("conv8", "conv2_1", "conv7"), ("conv9_1", "conv1_1", "conv8")]


def split(x):
    hi = x.half().float()
    lo = (x - hi).half().float()
    return hi, lo


def q22(x):
    hi, lo = split(x)
    return hi + lo


def split_lo_bits(x, bits):
    """--lo-bits: the low half as a `bits`-bit signed integer in units of ulp(hi) / 2^bits (a 3-byte value for bits = 8): what-if for a
    cheaper stored format; the f16 floor of 2^-24 stays."""
    hi = x.half().float()
    lo = x - hi
    e = torch.floor(torch.log2(hi.abs().clamp_min(2.0 ** -14)))
    q = torch.clamp(torch.exp2(e - 10 - bits), min=2.0 ** -24)
    return hi, (torch.round(lo / q) * q).half().float()


def fold(sd, conv, bn):
    w = sd[conv + ".weight"].double()
    sc = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
    wf = (w * sc.view(-1, *([1] * (w.dim() - 1)))).float()
    bf = (sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * sc).float()
    return wf, bf


def norm_exponent(sumsq, bias):
    est = math.sqrt(0.5 * sumsq + bias * bias)
    if not (est > 0.0) or not math.isfinite(est):
        return 0
    # C's lround: half away from zero
    v = math.log2(est)
    r = math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)
    return int(max(-60, min(60, -r)))


def row_sumsq(w, si):
    ws = w.double() * torch.pow(2.0, -torch.tensor(si, dtype=torch.float64)).view(1, -1, *([1] * (w.dim() - 2)))
    return (ws * ws).flatten(1).sum(1)


def scale(w, b, so, si):
    e = torch.tensor(so, dtype=torch.float32).view(-1, 1) - torch.tensor(si, dtype=torch.float32).view(1, -1)
    w2 = torch.ldexp(w, e.view(e.shape[0], e.shape[1], *([1] * (w.dim() - 2))).expand_as(w).to(torch.int32))
    b2 = torch.ldexp(b, torch.tensor(so, dtype=torch.int32))
    return w2, b2


def build(sd, norm=True):
    net, sc = {}, {}
    for name, x0, x1 in BLOCKS:
        w1, b1 = fold(sd, name + ".conv1.0", name + ".conv1.1")
        w2, b2 = fold(sd, name + ".conv2.0", name + ".conv2.1")
        wr, br = fold(sd, name + ".residual.0", name + ".residual.1")
        cin, cout = w1.shape[1], w1.shape[0]
        s_in = (sc[x0] if x0 else [0] * cin) + (sc[x1] if x1 else [])
        s_h, s_y = [0] * cout, [0] * cout
        if norm:
            q1 = row_sumsq(w1, s_in)
            s_h = [norm_exponent(float(q1[c]), float(b1[c])) for c in range(cout)]
            q2 = row_sumsq(w2, s_h) + row_sumsq(wr, s_in)
            s_y = [norm_exponent(float(q2[c]), float(b2[c]) + float(br[c])) for c in range(cout)]
        w1, b1 = scale(w1, b1, s_h, s_in)
        w2, b2 = scale(w2, b2, s_y, s_h)
        wr, br = scale(wr, br, s_y, s_in)
        sc[name] = s_y
        net[name] = (w1, b1, w2, b2, wr, br)
    wf = sd["conv_flatten.weight"].clone()
    wf = torch.ldexp(wf, (-torch.tensor(sc["conv9_1"], dtype=torch.int32)).view(1, -1, 1, 1).expand_as(wf))
    net["flatten"] = (wf, sd["conv_flatten.bias"])
    net["scales"] = sc
    return net


def conv3(xh, xl, w, b, pad):
    wh, wl = split(w)
    return F.conv2d(xl, wh, None, padding=pad) + F.conv2d(xh, wl, None, padding=pad) + F.conv2d(xh, wh, b, padding=pad)


# ---- Winograd F(2x2, 3x3) ----
G_ = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def wino_conv3(x, w, b, f16_transform=False):
    """x (B,C,H,W) fp32 (22-bit values), w (O,C,3,3) fp32 -> (B,O,H,W): F(2x2,3x3) with f16x2 products."""
    Bn, C, H, W = x.shape
    U = (G_ @ w.double() @ G_.t()).float()                         # (O,C,4,4), exact transform rounded once to fp32
    Uh, Ul = split(U)
    xp = F.pad(x, (1, 1, 1, 1))
    def tiles(t):
        return t.unfold(2, 4, 2).unfold(3, 4, 2)                   # (B,C,H/2,W/2,4,4)
    def transform(d):
        return torch.einsum("ij,bcyxjk,lk->bcyxil", BT, d, BT)
    if f16_transform:
        xh, xl = split(xp)
        Vh0, Vl0 = transform(tiles(xh)), transform(tiles(xl))       # fp32 sums of f16 values, planes on their own
        V = Vh0 + Vl0
    else:
        V = transform(tiles(xp))
    Vh, Vl = split(V)
    M = (torch.einsum("ocij,bcyxij->boyxij", Uh, Vl) + torch.einsum("ocij,bcyxij->boyxij", Ul, Vh) + torch.einsum("ocij,bcyxij->boyxij", Uh, Vh))
    Y = torch.einsum("ij,boyxjk,lk->boyxil", AT, M, AT)             # (B,O,H/2,W/2,2,2)
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, -1, H, W)
    return Y + b.view(1, -1, 1, 1)


def forward(net, sd, feats, wino=(), f16_transform=False, stats=None, h_hi_only=(), lo_bits=0, lo_scope="h"):
    def up(t):
        return F.interpolate(t, scale_factor=2, mode="nearest")
    out = {}
    x = feats.unsqueeze(1)
    for name, x0, x1 in BLOCKS:
        w1, b1, w2, b2, wr, br = net[name]
        if x0 is None:
            xin = x
        else:
            a = out[x0]
            if name in ("conv2_1", "conv3_1", "conv4_1", "conv_bottleneck"):
                a = F.max_pool2d(a, 2, 2)
            xin = a if x1 is None else torch.cat([a, up(out[x1])], dim=1)
        xh, xl = split_lo_bits(xin, lo_bits) if (lo_bits and lo_scope == "all" and x0 is not None) else split(xin)
        use_w = name in wino and xin.shape[1] > 1
        if use_w:
            h = F.relu(wino_conv3(xh + xl, w1, b1, f16_transform))
        else:
            h = F.relu(conv3(xh, xl, w1, b1, 1))
        h = q22(h)
        hh, hl = split_lo_bits(h, lo_bits) if lo_bits else split(h)
        if name in h_hi_only:                                        # --h-hi-only: the block's intermediate keeps its high half alone (11 bits)
            hl = torch.zeros_like(hl)
        if xin.shape[1] == 1:
            r = F.conv2d(xin, wr, br)                               # conv1_1: fp32 rank-1 term
        else:
            r = conv3(xh, xl, wr, br, 0)
        if name in wino:
            y = wino_conv3(hh + hl, w2, b2, f16_transform) + r
        else:
            y = conv3(hh, hl, w2, b2, 1) + r
        y = q22(F.relu(y))
        if stats is not None:
            stats[name] = (float(xin.abs().max()), float(h.abs().max()), float(y.abs().max()), float(y[y > 0].median()) if (y > 0).any() else 0.0)
        out[name] = y
    c9h, c9l = split(out["conv9_1"])
    wf, bfl = net["flatten"]
    flat = F.relu(conv3(c9h, c9l, wf, bfl, 0)).squeeze(2)
    m = O._resblock1d(flat, sd, "mask_output_conv.0")
    return F.conv1d(m, sd["mask_output_conv.1.weight"], sd["mask_output_conv.1.bias"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=8)
    ap.add_argument("--hostile", action="store_true")
    ap.add_argument("--no-norm", action="store_true")
    ap.add_argument("--winograd", default="")
    ap.add_argument("--wino-f16-transform", action="store_true")
    ap.add_argument("--h-hi-only", default="", help="blocks (or all) whose intermediate tensor h is stored as ONE f16 (what-if: half its bytes, two products instead of three in the second conv)")
    ap.add_argument("--lo-bits", type=int, default=0, help="what-if: stored low halves as N-bit integers relative to ulp(hi) (8 = a 3-byte value)")
    ap.add_argument("--lo-scope", default="h", help="h: the blocks' intermediate tensors only; all: every stored activation")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    torch.set_num_threads(a.threads)
    sd_np = synth.make_state_dict(0, hostile=a.hostile) if a.hostile else synth.make_state_dict(0)
    sd = synth.to_torch_state_dict(sd_np)
    pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
    sig, _, _ = O.load_audio_from_bytes(synth.wav_bytes(pcm, 16000))
    padded = O.pad_3s(sig)
    starts = O.plan_windows(60.0)
    pick = np.linspace(0, len(starts) - 1, a.windows).astype(int)
    sl = torch.stack([torch.from_numpy(padded[int(starts[i]): int(starts[i]) + 66150]) for i in pick])
    feats = O.mel_features(sl, sd["mel_spectrogram.spectrogram.window"], sd["mel_spectrogram.mel_scale.fb"])
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _, truth = O.unet_forward(sd64, feats.double(), want_spec=False)
    _, m32 = O.unet_forward(sd, feats, want_spec=False)
    print(f"fp32 oracle vs float64: max {float((m32.double() - truth).abs().max()):.3e}   logits in [{float(truth.min()):.2f}, {float(truth.max()):.2f}]", flush=True)
    wino = tuple(n for n, _, _ in BLOCKS) if a.winograd == "all" else tuple(x for x in a.winograd.split(",") if x)
    net = build(sd, norm=not a.no_norm)
    stats = {}
    hho = tuple(n for n, _, _ in BLOCKS) if a.h_hi_only == "all" else tuple(x for x in a.h_hi_only.split(",") if x)
    m = forward(net, sd, feats, wino, a.wino_f16_transform, stats, hho, a.lo_bits, a.lo_scope)
    d = (m.double() - truth).abs()
    tag = (f"lo-bits {a.lo_bits} ({a.lo_scope}) " if a.lo_bits else "") + (f"h-hi-only[{a.h_hi_only}] " if hho else "") + ("hostile " if a.hostile else "") + ("no-norm " if a.no_norm else "norm ") + (f"winograd[{a.winograd}]" + (" planes transformed separately" if a.wino_f16_transform else "") if wino else "direct")
    print(f"f16x2 {tag}: max |logit - float64| {float(d.max()):.3e}  mean {float(d.mean()):.3e}  vs fp32 oracle {float((m - m32).abs().max()):.3e}", flush=True)
    for k, (xi, hm, ym, ymed) in stats.items():
        s = net["scales"][k]
        print(f"   {k:16s} stored |x|max in {xi:9.3g}  h {hm:9.3g}  out {ym:9.3g} (median of >0: {ymed:8.3g})   exponents {min(s):+d}..{max(s):+d}")


if __name__ == "__main__":
    main()
