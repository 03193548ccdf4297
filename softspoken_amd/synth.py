"""Deterministic synthetic inputs: a checkpoint with the reference key layout and seeded WAV audio.

The real checkpoint (root/models/spec_unet_2d_pytorch/model_checkpoint.pth, reference
settings.py:19-20) is a missing blob, and there is no network for datasets, so tests, the bench
and smoke() all run on inputs made here.  Everything is generated from
numpy.random.Generator(PCG64(seed)) so the same bytes come out in the build container and on
the GPU box (same image, same numpy).

Key layout follows SpecUNet_2D.state_dict() of the reference
(root/code/backend/pytorch_neural_nets.py:83-140): 222 conv/BN entries
plus the two torchaudio buffers `mel_spectrogram.spectrogram.window` and
`mel_spectrogram.mel_scale.fb`.
"""
from __future__ import annotations

import math
import struct
from collections import OrderedDict

import numpy as np

from .layout import SR, WINDOW   # noqa: F401
from .layout import RESBLOCKS_2D, hann_window_512, mel_filterbank, state_dict_layout   # noqa: F401  (re-exported: tests and tools use synth.*)


# "Hostile-scale" variant of the synthetic checkpoint: the SAME function up to fp32 rounding, with BatchNorm
# gains that push whole tensors three decades away from 1 and convolutions downstream that undo it.  (block, where, factor):
# "out" scales the block's output (gamma and beta of conv2.1 and residual.1), "h" the tensor between its two 3x3 convs (gamma and
# beta of conv1.1, undone in conv2.0.weight).  An "out" factor is undone in the input-channel slice of every consumer's conv1.0 /
# residual.0 weights (pytorch_neural_nets.py:156-181: the concat order is [skip, upsampled]).
HOSTILE_GAINS = (("conv3_1", "out", 1e-3), ("conv4_1", "out", 1e3), ("conv_bottleneck", "h", 1e-3), ("encoder_out", "out", 1e-3),
                 ("conv6", "out", 1e3), ("conv7", "h", 1e3), ("conv7", "out", 1e-3))
_CONSUMERS = {"conv3_1": (("conv4_1", 0, 96), ("conv7", 0, 96)), "conv4_1": (("conv_bottleneck", 0, 128), ("conv6", 0, 128)),
              "encoder_out": (("conv6", 128, 256),), "conv6": (("conv7", 96, 192),), "conv7": (("conv8", 64, 128),)}


def _apply_hostile_gains(sd):
    f32 = np.float32
    for block, where, g in HOSTILE_GAINS:
        if where == "h":
            for k in ("weight", "bias"):
                sd[f"{block}.conv1.1.{k}"] = (sd[f"{block}.conv1.1.{k}"] * f32(g)).astype(f32)
            sd[f"{block}.conv2.0.weight"] = (sd[f"{block}.conv2.0.weight"] * f32(1.0 / g)).astype(f32)
        else:
            for bn in ("conv2.1", "residual.1"):
                for k in ("weight", "bias"):
                    sd[f"{block}.{bn}.{k}"] = (sd[f"{block}.{bn}.{k}"] * f32(g)).astype(f32)
            for cons, lo, hi in _CONSUMERS[block]:
                for conv in ("conv1.0", "residual.0"):
                    w = sd[f"{cons}.{conv}.weight"].copy()
                    w[:, lo:hi] = (w[:, lo:hi] * f32(1.0 / g)).astype(f32)
                    sd[f"{cons}.{conv}.weight"] = w
    return sd


def make_state_dict(seed: int = 0, hostile: bool = False):
    """Synthetic checkpoint tensors as an OrderedDict of numpy arrays (reference key layout).

    Conv weights are He-scaled normals; BatchNorm statistics are non-trivial so the folding is
    exercised (gamma~U(0.5,1.5), beta~N(0,0.1), mean~N(0,0.1), var~U(0.5,1.5)).
    hostile=True: the same network with the HOSTILE_GAINS applied (activations of conv3_1 ... conv7 near 1e-3 or 1e3).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    for key, (shape, kind) in state_dict_layout().items():
        if kind == "window":
            sd[key] = hann_window_512()
        elif kind == "fb":
            sd[key] = mel_filterbank()
        elif kind == "bn_count":
            sd[key] = np.array(1000, dtype=np.int64)
        elif kind == "bn_var":
            sd[key] = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif kind == "bn_gamma":
            sd[key] = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif kind in ("bn_mean", "bn_beta", "conv_b"):
            sd[key] = (0.1 * rng.standard_normal(size=shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            # residual + main branch are summed, so scale each branch down a little
            std = 0.75 * math.sqrt(2.0 / fan_in)
            sd[key] = (std * rng.standard_normal(size=shape)).astype(np.float32)
    # Head calibration: the He-scaled 4-channel head is nearly flat,
    # so give it gain and centre the logits on settings.threshold (0.1); then both classes of bin
    # occur on the synthetic audio and errors upstream are visible in the logits.
    sd["conv_flatten.weight"] *= np.float32(SYNTH_HEAD_GAINS[0])
    for k in ("residual.0.weight", "conv1.0.weight", "conv2.0.weight"):
        sd["mask_output_conv.0." + k] *= np.float32(SYNTH_HEAD_GAINS[1])
    sd["mask_output_conv.1.weight"] *= np.float32(SYNTH_HEAD_GAINS[2])
    sd["mask_output_conv.1.bias"] = np.array([SYNTH_HEAD_BIAS], dtype=np.float32)
    if hostile:
        _apply_hostile_gains(sd)
    return sd


# chosen once against the seed-0 weights and the seed-1001 audio; see tests/golden/make_golden.py
SYNTH_HEAD_GAINS = (4.0, 2.0, 2.0)
SYNTH_HEAD_BIAS = -0.68


def to_torch_state_dict(sd):
    import torch
    out = OrderedDict()
    for k, v in sd.items():
        out[k] = torch.from_numpy(np.array(v, copy=True, order='C'))
    return out


def save_checkpoint(path, seed: int = 0, epoch: int = 0):
    """Write a checkpoint file the way the reference expects to read it
    (NNDetector.py:42-53: torch.load(weights_only=True)['model_state_dict'], ['epoch'])."""
    import torch
    torch.save({"model_state_dict": to_torch_state_dict(make_state_dict(seed)), "epoch": epoch}, path)


# ----------------------------------------------------------------------------------------------
# audio
# ----------------------------------------------------------------------------------------------

def synth_audio(seed: int, seconds: float, sr: int = 16000, channels: int = 1,
                bursts: int | None = None, with_silence: bool = True):
    """Seeded test signal in [-1, 1): pink-ish noise floor + voiced harmonic bursts (+ exact
    digital silence + one 1e-3-scaled burst), after SURVEY.md section 8(d) 'Synthetic inputs'.
    Returns float64 array (channels, n)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(round(seconds * sr))
    t = np.arange(n, dtype=np.float64) / sr
    out = np.zeros((channels, n))
    if bursts is None:
        bursts = max(1, int(round(seconds / 10.0)))
    for ch in range(channels):
        white = rng.standard_normal(n)
        # one-pole low-pass mixed with white: cheap pink-ish tilt
        lp = np.empty(n)
        acc = 0.0
        a = 0.97
        # vectorised one-pole via cumulative filter in blocks (exact recursion, float64)
        from scipy.signal import lfilter
        lp = lfilter([1.0 - a], [1.0, -a], white)
        x = 0.02 * (0.6 * white + 2.0 * lp)
        for b in range(bursts):
            dur = rng.uniform(0.4, 2.5)
            if seconds <= dur + 0.2:
                dur = max(0.1, seconds * 0.4)
            start = rng.uniform(0.0, max(1e-3, seconds - dur))
            f0 = rng.uniform(90.0, 250.0)
            i0, i1 = int(start * sr), min(n, int((start + dur) * sr))
            tt = t[i0:i1] - t[i0]
            voiced = np.zeros(i1 - i0)
            for k in range(1, 21):
                if k * f0 < 0.45 * sr:
                    voiced += np.sin(2 * math.pi * k * f0 * tt + rng.uniform(0, 2 * math.pi)) / k
            am = 0.5 * (1.0 - np.cos(2 * math.pi * 4.0 * tt))
            voiced *= am
            peak = np.max(np.abs(voiced)) + 1e-12
            scale = 0.3 / peak
            if with_silence and b == bursts - 1 and bursts > 1:
                scale *= 1e-3            # the quiet burst: exercises log10(x + 1) cancellation
            x[i0:i1] += scale * voiced
        if with_silence and seconds >= 20.0:
            s0 = int(0.55 * n)
            x[s0:s0 + int(5.0 * sr)] = 0.0   # exact digital silence -> exactly-zero features
        out[ch] = x
    return np.clip(out, -0.999, 0.999)


def to_pcm16(x):
    """float [-1,1) -> int16 (round-half-even), shape (n,) or (n, ch) interleaved."""
    y = np.rint(np.asarray(x) * 32767.0).astype(np.int16)
    if y.ndim == 2:
        y = np.ascontiguousarray(y.T)
        if y.shape[1] == 1:
            y = y[:, 0]
    return y


def wav_bytes(pcm, sr: int, fmt: str = "pcm16"):
    """Minimal RIFF/WAVE writer. `pcm`: (n,) or (n, ch) array; fmt in pcm16|pcm24|pcm32|f32|u8."""
    a = np.asarray(pcm)
    ch = 1 if a.ndim == 1 else a.shape[1]
    if fmt == "pcm16":
        data = a.astype("<i2").tobytes(); bits, tag = 16, 1
    elif fmt == "pcm32":
        data = a.astype("<i4").tobytes(); bits, tag = 32, 1
    elif fmt == "u8":
        data = a.astype(np.uint8).tobytes(); bits, tag = 8, 1
    elif fmt == "f32":
        data = a.astype("<f4").tobytes(); bits, tag = 32, 3
    elif fmt == "pcm24":
        v = a.astype("<i4").reshape(-1)
        b = v.view(np.uint8).reshape(-1, 4)[:, :3]
        data = np.ascontiguousarray(b).tobytes(); bits, tag = 24, 1
    else:
        raise ValueError(fmt)
    block = ch * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE"
    hdr += b"fmt " + struct.pack("<IHHIIHH", 16, tag, ch, sr, sr * block, block, bits)
    hdr += b"data" + struct.pack("<I", len(data))
    pad = b"\x00" if len(data) & 1 else b""
    return hdr + data + pad


def aiff_bytes(pcm, sr: int, bits: int = 16, compression: bytes | None = None):
    """Minimal AIFF (compression None) / AIFF-C writer for tests: `pcm` (n,) or (n, ch) integers (or floats for fl32 / fl64); big-endian
    samples, b"sowt" little-endian, 8-bit samples signed."""
    a = np.asarray(pcm)
    if a.ndim == 1:
        a = a[:, None]
    n, ch = a.shape
    if compression in (b"fl32", b"fl64"):
        body = a.astype(">f4" if compression == b"fl32" else ">f8").tobytes()
        bits = 32 if compression == b"fl32" else 64
    elif bits == 24:
        v = a.astype(np.int64) & 0xFFFFFF
        tri = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], axis=-1).astype(np.uint8)
        body = (tri[..., ::-1] if compression == b"sowt" else tri).tobytes()
    else:
        dt = {8: "i1", 16: "i2", 32: "i4"}[bits]
        body = a.astype(("<" if compression == b"sowt" else ">") + dt if bits > 8 else dt).tobytes()
    ex = 16383 + 63
    mant = int(sr)
    while mant < (1 << 63):
        mant <<= 1
        ex -= 1
    comm = struct.pack(">hIhHQ", ch, n, bits, ex, mant)
    if compression is not None:
        comm += compression + b"\x00\x00"                  # compression type + an empty pascal string (padded to even)
    ssnd = struct.pack(">II", 0, 0) + body
    chunks = b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd + (b"\x00" if len(ssnd) & 1 else b"")
    form = (b"AIFC" if compression is not None else b"AIFF") + chunks
    return b"FORM" + struct.pack(">I", len(form)) + form


def write_wav(path, pcm, sr: int, fmt: str = "pcm16"):
    with open(path, "wb") as f:
        f.write(wav_bytes(pcm, sr, fmt))
