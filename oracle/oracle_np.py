"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not product code.

CPU restatement (numpy + torch-CPU ops) of the reference's "Run Voice Detector" hot path.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
path (softspoken_amd/, root/code/...) never does and fails loudly when the HIP library is missing.

Parity status of each stage (see DESIGN.md "Oracle"):
  * A4 CNN body, A7 averaging, A8 regions      -- PINNED: checked against goldens produced by the
    reference's own SpecUNet_2D / NNDetector methods (tests/golden/make_golden.py).
  * A3 mel front-end                           -- "parity unpinned" at the torchaudio boundary:
    torchaudio is not in the image and the reference has no tests/fixtures; this restates
    torchaudio.transforms.MelSpectrogram's documented algorithm on torch.stft.
  * A2 decode / mixdown / resample             -- "parity unpinned": soundfile/librosa/soxr absent,
    no fixtures; order and dtypes follow voice_activity.py:32-69, the resampler is the build's own.
  * A9 CSV text                                -- restated from worker.py:99-128 and
    silencer_ui.py:775-817; checked against pandas here.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
import struct

import numpy as np
import torch
import torch.nn.functional as F

SR = 22050                      # settings.py:16  vad_resample
WINDOW = 3 * SR                 # NNDetector.py:74
STEP = math.floor(SR * 0.6)     # NNDetector.py:75  (settings.py:9 step_size)
N_FFT = 512 * 4                 # pytorch_neural_nets.py:94  n_fft=settings.n_fft*4
WIN_LENGTH = 512                # settings.py:5
HOP = 256                       # settings.py:6
N_MELS = 128                    # pytorch_neural_nets.py:87
F_MAX = 8000.0                  # pytorch_neural_nets.py:98
N_FRAMES = 256                  # pytorch_neural_nets.py:150
THRESHOLD = 0.1                 # settings.py:13
BATCH = 32                      # settings.py:12


# ------------------------------------------------------------------------------------------------
# A1  window planning  (NNDetector.py:55-82)
# ------------------------------------------------------------------------------------------------
def plan_windows(duration_s: float) -> np.ndarray:
    """NNDetector.plan_detection_job body for one file (NNDetector.py:66-80)."""
    audio_data_length = round(duration_s * SR) + (3 * 2 * SR)
    num_windows = int(np.ceil((audio_data_length - WINDOW) / STEP))
    return np.arange(num_windows) * STEP


# ------------------------------------------------------------------------------------------------
# A2  decode / mixdown / resample  (voice_activity.py:32-69)  -- parity unpinned
# ------------------------------------------------------------------------------------------------
def parse_wav(buf: bytes):
    """Minimal RIFF/WAVE chunk walk -> dict(fmt_tag, channels, sr, bits, data_off, data_len, frames).
    Stands where soundfile/libsndfile stands in the reference (voice_activity.py:37)."""
    if len(buf) >= 12 and buf[:4] == b"FORM" and buf[8:12] in (b"AIFF", b"AIFC"):
        return parse_aiff(buf)
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid = buf[pos:pos + 4]
        (sz,) = struct.unpack_from("<I", buf, pos + 4)
        body = pos + 8
        if cid == b"fmt ":
            tag, ch, sr, _br, _ba, bits = struct.unpack_from("<HHIIHH", buf, body)
            if tag == 0xFFFE and sz >= 26:          # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
                (tag,) = struct.unpack_from("<H", buf, body + 24)
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            data = (body, min(sz, len(buf) - body))
            break
        pos = body + sz + (sz & 1)
    if fmt is None or data is None:
        raise ValueError("missing fmt or data chunk")
    tag, ch, sr, bits = fmt
    frames = data[1] // (ch * bits // 8)
    return dict(fmt_tag=tag, channels=ch, sr=sr, bits=bits, data_off=data[0], data_len=data[1], frames=frames)


def parse_aiff(buf: bytes):
    """AIFF / AIFF-C chunk walk (big-endian sizes; COMM: channels, frames, bits, 80-bit extended sample rate [, compression]; SSND: offset,
    block size, samples) -> the same dict as parse_wav with fmt_tag 'aiff:<compression>' (libsndfile reads these containers too)."""
    pos, comm, data = 12, None, None
    aifc = buf[8:12] == b"AIFC"
    while pos + 8 <= len(buf):
        cid = buf[pos:pos + 4]
        (sz,) = struct.unpack_from(">I", buf, pos + 4)
        body = pos + 8
        if cid == b"COMM":
            ch, frames, bits, ex, mant = struct.unpack_from(">hIhHQ", buf, body)
            sr = float(mant) * 2.0 ** ((ex & 0x7fff) - 16383 - 63)
            comp = buf[body + 18:body + 22] if aifc else b"NONE"
            comm = (ch, frames, bits, int(round(sr)), comp)
        elif cid == b"SSND":
            (off,) = struct.unpack_from(">I", buf, body)
            start = body + 8 + off
            data = (start, max(0, min(sz - 8 - off, len(buf) - start)))
            break
        pos = body + sz + (sz & 1)
    if comm is None or data is None:
        raise ValueError("missing COMM or SSND chunk")
    ch, frames, bits, sr, comp = comm
    if comp in (b"fl32", b"FL32"):
        bits = 32
    if comp in (b"fl64", b"FL64"):
        bits = 64
    frames = min(frames, data[1] // (ch * bits // 8))
    return dict(fmt_tag="aiff:" + comp.decode("ascii", "replace"), channels=ch, sr=sr, bits=bits, data_off=data[0], data_len=data[1], frames=frames)


def decode_pcm(buf: bytes, info) -> np.ndarray:
    """PCM -> float32 (frames, ch), libsndfile's float conversion (x / 2^(bits-1); u8 is offset-128)."""
    raw = np.frombuffer(buf, dtype=np.uint8, count=info["frames"] * info["channels"] * info["bits"] // 8,
                        offset=info["data_off"])
    ch, bits, tag = info["channels"], info["bits"], info["fmt_tag"]
    if isinstance(tag, str) and tag.startswith("aiff:"):          # big endian (sowt: little endian), 8-bit samples signed
        comp = tag[5:]
        if comp == "sowt" and bits > 8:
            return decode_pcm(buf, dict(info, fmt_tag=1))
        if comp in ("NONE", "sowt") and bits == 8:
            x = raw.view(np.int8).astype(np.float32) / np.float32(128.0)
        elif comp == "NONE" and bits == 16:
            x = raw.view(">i2").astype(np.float32) / np.float32(32768.0)
        elif comp == "NONE" and bits == 24:
            b = raw.reshape(-1, 3).astype(np.int32)
            v = (b[:, 2] | (b[:, 1] << 8) | (b[:, 0] << 16))
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / np.float32(8388608.0)
        elif comp == "NONE" and bits == 32:
            x = (raw.view(">i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        elif comp in ("fl32", "FL32"):
            x = raw.view(">f4").astype(np.float32)
        elif comp in ("fl64", "FL64"):
            x = raw.view(">f8").astype(np.float32)
        else:
            raise ValueError(f"unsupported AIFF encoding {comp} bits={bits}")
        return x.reshape(-1, ch)
    if tag == 1 and bits == 16:
        x = raw.view("<i2").astype(np.float32) / np.float32(32768.0)
    elif tag == 1 and bits == 8:
        x = (raw.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif tag == 1 and bits == 24:
        b = raw.reshape(-1, 3).astype(np.int32)
        v = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / np.float32(8388608.0)
    elif tag == 1 and bits == 32:
        x = (raw.view("<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 3 and bits == 32:
        x = raw.view("<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = raw.view("<f8").astype(np.float32)
    else:
        raise ValueError(f"unsupported WAV encoding tag={tag} bits={bits}")
    return x.reshape(-1, ch)


def to_mono(x: np.ndarray) -> np.ndarray:
    """librosa.to_mono == np.mean(axis=0) on (ch, n) float32 (voice_activity.py:61-62): float32
    running sum over channels, then one float32 divide."""
    if x.ndim == 1 or x.shape[1] == 1:
        return x.reshape(-1).astype(np.float32)
    acc = x[:, 0].astype(np.float32).copy()
    for c in range(1, x.shape[1]):
        acc = (acc + x[:, c]).astype(np.float32)
    return (acc / np.float32(x.shape[1])).astype(np.float32)


RESAMPLE_ZEROS = 32        # sinc zero crossings kept each side (at the lower of the two rates)
RESAMPLE_BETA = 12.0       # Kaiser beta
RESAMPLE_ROLLOFF = 0.95    # cutoff as a fraction of the lower Nyquist


def resample_plan(sr_in: int, sr_out: int = SR):
    """The build's own Kaiser-windowed-sinc polyphase design (stands where librosa.resample ->
    soxr_hq stands, voice_activity.py:65-67; soxr cannot be reproduced here -> parity unpinned).
    Output sample m sits at input time m*M/L (L/M = sr_out/sr_in reduced).  Returns
    (L, M, half, taps[L, 2*half]) with float32 taps computed in float64."""
    g = math.gcd(sr_in, sr_out)
    L, M = sr_out // g, sr_in // g
    scale = min(1.0, sr_out / sr_in)                # < 1 when down-sampling: widen the kernel
    fc = RESAMPLE_ROLLOFF * scale                   # cutoff in cycles per input-sample * 2
    half = int(math.ceil(RESAMPLE_ZEROS / scale))
    p = np.arange(L, dtype=np.float64)[:, None]
    frac = ((p * M) % L) / L                        # fractional input position of phase p
    j = np.arange(-half + 1, half + 1, dtype=np.float64)[None, :]    # input taps base+j
    d = j - frac                                    # distance (input samples) tap -> output instant
    h = fc * np.sinc(fc * d)
    w = np.i0(RESAMPLE_BETA * np.sqrt(np.clip(1.0 - (d / half) ** 2, 0.0, None))) / np.i0(RESAMPLE_BETA)
    w = np.where(np.abs(d) <= half, w, 0.0)
    return L, M, half, (h * w).astype(np.float32)


def resample(x: np.ndarray, sr_in: int, sr_out: int = SR) -> np.ndarray:
    """Polyphase FIR, float32 taps and samples, float32 accumulation in tap order (the device
    kernel accumulates in the same order).  n_out = ceil(n_in * sr_out / sr_in), librosa's rule."""
    if sr_in == sr_out:
        return x.astype(np.float32)
    L, M, half, taps = resample_plan(sr_in, sr_out)
    n_in = x.shape[0]
    n_out = int(math.ceil(n_in * sr_out / sr_in))
    m = np.arange(n_out, dtype=np.int64)
    base = (m * M) // L
    phase = (m * M) % L
    xp = np.concatenate([np.zeros(half, np.float32), x.astype(np.float32), np.zeros(half + 1, np.float32)])
    acc = np.zeros(n_out, dtype=np.float32)
    for jj in range(2 * half):
        # input index base + (jj - half + 1); +half for the left pad
        acc = (acc + taps[phase, jj] * xp[base + jj + 1]).astype(np.float32)
    return acc


def load_audio_from_bytes(buf: bytes):
    """voice_activity.load_audio (:32-69): decode float32 -> transpose -> mono -> resample."""
    info = parse_wav(buf)
    x = decode_pcm(buf, info)
    mono = to_mono(x)
    return resample(mono, info["sr"], SR), SR, info


def pad_3s(x: np.ndarray) -> np.ndarray:
    """worker.py:58-62: 3 s of zeros on each side."""
    out = np.zeros(len(x) + 2 * WINDOW, dtype=np.float32)
    out[WINDOW:WINDOW + len(x)] = x
    return out


# ------------------------------------------------------------------------------------------------
# A3  mel front-end  (pytorch_neural_nets.py:92-99, 144-153)  -- restated torchaudio algorithm
# ------------------------------------------------------------------------------------------------
def mel_features(x: torch.Tensor, window: torch.Tensor, fb: torch.Tensor) -> torch.Tensor:
    """(B, 66150) f32 -> (B, 128, 256) f32.
    torchaudio Spectrogram(power=2) == torch.stft(center, reflect, onesided).abs().pow(2);
    MelScale == (spec^T @ fb)^T; then sqrt(log10(.+1)) (:80-81,147) and [:, :, :256] (:150)."""
    spec = torch.stft(x, n_fft=N_FFT, hop_length=HOP, win_length=WIN_LENGTH, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.abs().pow(2.0)
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)
    return torch.sqrt(torch.log10(mel + 1))[:, :, :N_FRAMES]


# ------------------------------------------------------------------------------------------------
# A4  SpecUNet_2D body, eval mode  (pytorch_neural_nets.py:7-77, 142-197)
# ------------------------------------------------------------------------------------------------
def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, momentum=0.1, eps=1e-5)


def _resblock2d(x, sd, p):
    """ResBlock.forward (:32-41); Dropout2d is identity in eval."""
    idt = _bn(F.conv2d(x, sd[p + ".residual.0.weight"]), sd, p + ".residual.1")
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.0.weight"], padding=1), sd, p + ".conv1.1"))
    out = _bn(F.conv2d(out, sd[p + ".conv2.0.weight"], padding=1), sd, p + ".conv2.1")
    return F.relu(out + idt)


def _resblock1d(x, sd, p):
    """ResBlock1D.forward (:68-77)."""
    idt = _bn(F.conv1d(x, sd[p + ".residual.0.weight"]), sd, p + ".residual.1")
    out = F.relu(_bn(F.conv1d(x, sd[p + ".conv1.0.weight"], padding=1), sd, p + ".conv1.1"))
    out = _bn(F.conv1d(out, sd[p + ".conv2.0.weight"], padding=1), sd, p + ".conv2.1")
    return F.relu(out + idt)


def unet_forward(sd, feats: torch.Tensor, want_spec: bool = True, taps: dict | None = None):
    """feats (B,128,256) -> (spec (B,2,128,256) | None, mask (B,1,256)).  :153-197.
    `taps`, if given, is filled with every block's output for layer-wise checks."""
    def up(t):
        return F.interpolate(t, scale_factor=2, mode="nearest")

    def rec(name, t):
        if taps is not None:
            taps[name] = t
        return t

    x = feats.unsqueeze(1)
    c1 = rec("conv1_1", _resblock2d(x, sd, "conv1_1"))
    c2 = rec("conv2_1", _resblock2d(F.max_pool2d(c1, 2, 2), sd, "conv2_1"))
    c3 = rec("conv3_1", _resblock2d(F.max_pool2d(c2, 2, 2), sd, "conv3_1"))
    c4 = rec("conv4_1", _resblock2d(F.max_pool2d(c3, 2, 2), sd, "conv4_1"))
    bott = rec("conv_bottleneck", _resblock2d(F.max_pool2d(c4, 2, 2), sd, "conv_bottleneck"))
    enc = rec("encoder_out", _resblock2d(bott, sd, "encoder_out"))
    c6 = rec("conv6", _resblock2d(torch.cat([c4, up(enc)], dim=1), sd, "conv6"))
    c7 = rec("conv7", _resblock2d(torch.cat([c3, up(c6)], dim=1), sd, "conv7"))
    c8 = rec("conv8", _resblock2d(torch.cat([c2, up(c7)], dim=1), sd, "conv8"))
    c9 = rec("conv9_1", _resblock2d(torch.cat([c1, up(c8)], dim=1), sd, "conv9_1"))
    spec = None
    if want_spec:
        s = _resblock2d(c9, sd, "spec_output_conv.0")
        spec = F.relu(F.conv2d(s, sd["spec_output_conv.1.weight"], sd["spec_output_conv.1.bias"]))
    flat = F.relu(F.conv2d(c9, sd["conv_flatten.weight"], sd["conv_flatten.bias"])).squeeze(2)
    rec("flatten", flat)
    m = _resblock1d(flat, sd, "mask_output_conv.0")
    mask = F.conv1d(m, sd["mask_output_conv.1.weight"], sd["mask_output_conv.1.bias"])
    return spec, mask


def model_forward(sd, x: torch.Tensor, want_spec: bool = False):
    """SpecUNet_2D.forward (:142-197) on raw windows (B, 66150)."""
    feats = mel_features(x, sd["mel_spectrogram.spectrogram.window"], sd["mel_spectrogram.mel_scale.fb"])
    return unet_forward(sd, feats, want_spec)


def infer_windows(sd, padded: np.ndarray, starts: np.ndarray, batch: int = BATCH, want_spec: bool = False):
    """NNDetector.process_batch (:84-101) driven as worker.py:71-79 drives it -> (W, 1, 256) f32."""
    out = []
    sig = torch.from_numpy(np.ascontiguousarray(padded, dtype=np.float32))
    with torch.no_grad():
        for s0 in range(0, len(starts), batch):
            idx = starts[s0:s0 + batch]
            sl = torch.stack([sig[int(i): int(i) + WINDOW] for i in idx])
            _, mask = model_forward(sd, sl, want_spec)
            out.append(mask.numpy())
    return np.vstack(out) if out else np.zeros((0, 1, 256), np.float32)


# ------------------------------------------------------------------------------------------------
# A7  overlap averaging  (NNDetector.py:153-190)
# ------------------------------------------------------------------------------------------------
def average_overlapping(window_logits: np.ndarray, audio_length_seconds: float):
    """-> (avg float64[n_kept], idx int64[n_kept]); the reference keeps bins with count >= 1 and
    labels bin idx with the string f"{idx/(256/3):.4f}" (:185)."""
    output_length = int(round(audio_length_seconds * 256 / 3))
    s = np.zeros(output_length)
    c = np.zeros(output_length)
    time_resolution = 3 / 256
    for i, w in enumerate(window_logits):
        start = int(round(i * 0.6 / time_resolution))
        s[start:start + 256] += w.reshape(-1)      # raises, as the reference does, if it does not fit
        c[start:start + 256] += 1
    keep = np.nonzero(c >= 1)[0]
    return s[keep] / c[keep], keep


def time_str(idx: int) -> str:
    return f"{idx / (256 / 3):.4f}"          # NNDetector.py:185


# ------------------------------------------------------------------------------------------------
# A8  threshold + run-length + gap merge  (NNDetector.py:103-143), then worker.py:100
# ------------------------------------------------------------------------------------------------
def find_regions(avg: np.ndarray, idx: np.ndarray, threshold: float = THRESHOLD, break_duration: float = 0.5):
    """-> list of (start_str, end_str) exactly as find_speech_regions returns them."""
    regions, start, end = [], None, None
    for v, i in zip(avg, idx):
        t = time_str(int(i))
        if v > threshold:
            if start is None:
                start = t
            end = t
        elif start is not None:
            regions.append((start, end))
            start = None
    if start is not None:
        regions.append((start, end))
    if not regions:
        return []
    merged, cur = [], regions[0]
    for nxt in regions[1:]:
        if float(nxt[0]) - float(cur[1]) <= break_duration:
            cur = (cur[0], nxt[1])
        else:
            merged.append(cur)
            cur = nxt
    merged.append(cur)
    return merged


def regions_minus_pad(regions):
    """worker.py:100: float(str) - 3 in double."""
    return [(float(s) - 3, float(e) - 3) for (s, e) in regions]


# ------------------------------------------------------------------------------------------------
# A9  CSV text  (worker.py:103-125, silencer_ui.py:779-788, 816-817)
# ------------------------------------------------------------------------------------------------
CSV_HEADER = "ID,file_path,file_name,start_time,end_time,erase,user_comment,review_datetime"


def csv_text(rows) -> str:
    """rows: iterable of (id, file_path, file_name, start, end).  Mirrors DataFrame.to_csv(index=False)
    for the reference's frame: floats by repr(), erase 0, two empty trailing fields; paths are
    quoted only when they contain a comma, quote or newline (csv.QUOTE_MINIMAL)."""
    def q(s):
        s = str(s)
        if any(ch in s for ch in ',"\n\r'):
            return '"' + s.replace('"', '""') + '"'
        return s
    lines = [CSV_HEADER]
    for (i, fp, fn, st, en) in rows:
        lines.append(f"{int(i)},{q(fp)},{q(fn)},{repr(float(st))},{repr(float(en))},0,,")
    return "\n".join(lines) + "\n"


def detect_signal(sd, signal_22k: np.ndarray, duration_s: float | None = None):
    """Whole per-file path of worker.py:57-100 on a 22 050 Hz mono float32 signal.
    -> dict(window_logits, avg, idx, regions_str, regions)."""
    if duration_s is None:
        duration_s = len(signal_22k) / SR
    padded = pad_3s(signal_22k)
    starts = plan_windows(duration_s)
    logits = infer_windows(sd, padded, starts)
    avg, idx = average_overlapping(logits, len(padded) / SR)
    reg = find_regions(avg, idx)
    return dict(starts=starts, window_logits=logits, avg=avg, idx=idx, regions_str=reg,
                regions=regions_minus_pad(reg))


# ------------------------------------------------------------------------------------------------
# Silencer -- silencer_ui.py:974-998.  "parity unpinned": the reference holds no
# fixture for it and neither librosa nor libsndfile is in this image, so this restates
#   librosa.load(sr=None, mono=False)  -> float32 (ch, n)          (decode_pcm above)
#   audio[:, int(round(st*sr)) : int(round(et*sr))] = 0.0          (clamped to [0, n]; Python round)
#   sf.write(path, audio.T, sr)        -> WAV, default subtype PCM_16: libsndfile's f2les_array,
#                                         lrintf(x * 0x7FFF), no clipping (low 16 bits kept)
# from the reference's call sites and libsndfile's published conversion.
# ------------------------------------------------------------------------------------------------
def silence_pcm16(x: np.ndarray, sr: int, regions) -> np.ndarray:
    """x: float32 (frames, ch) as decode_pcm returns it -> int16 (frames, ch)."""
    x = np.array(x, dtype=np.float32, copy=True)
    n = x.shape[0]
    for st, et in regions:
        a, b = float(st) * sr, float(et) * sr
        if a != a or b != b:
            continue
        lo = max(0, min(int(round(a)), n))
        hi = max(0, min(int(round(b)), n))
        x[lo:hi, :] = 0.0
    scaled = (x * np.float32(32767.0)).astype(np.float32)
    return np.rint(scaled).astype(np.int64).astype(np.int16)     # wraps like the C cast does


def wav_pcm16_bytes(pcm: np.ndarray, sr: int) -> bytes:
    """Canonical 44-byte header + little-endian samples (what libsndfile writes for WAV / PCM_16)."""
    import struct
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    ch = pcm.shape[1] if pcm.ndim == 2 else 1
    data = pcm.tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
            struct.pack("<IHHIIHH", 16, 1, ch, sr, sr * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)) + data)


# ------------------------------------------------------------------------------------------------
# Review-screen spectrogram -- voice_activity.py:148-154:
#   np.abs(librosa.stft(data, n_fft=512, win_length=512, hop_length=256))      (settings.py:4-6)
# "parity unpinned": librosa is not in this image and the reference holds no spectrogram fixture.  librosa.stft's defaults
# restated: window = scipy.signal.get_window('hann', 512, fftbins=True) (periodic), center=True with pad_mode='constant'
# (zeros, n_fft // 2 each side), frames at hop 256, float64 window x frame product, rfft; a float32 input gives complex64.
# ------------------------------------------------------------------------------------------------
def stft512_magnitude(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x).reshape(-1)
    out_dtype = np.float32 if x.dtype == np.float32 else np.float64
    n = x.size
    nf = 1 + n // 256
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(512) / 512.0)
    pad = np.zeros(n + 512, dtype=np.float64)
    pad[256:256 + n] = x
    frames = np.stack([pad[256 * t: 256 * t + 512] for t in range(nf)], axis=1)      # [512][nf]
    spec = np.fft.rfft(win[:, None] * frames, axis=0)                                  # [257][nf]
    if out_dtype == np.float32:
        spec = spec.astype(np.complex64)
    return np.abs(spec).astype(out_dtype)


def load_audio_startstop_from_bytes(buf: bytes, start: float, stop: float):
    """voice_activity.py:72-143: frames [int(start*sr), int(stop*sr)) clipped to the file, float32, mono, resampled to 22 050 Hz."""
    if start < 0 or stop <= start:
        return None, None
    info = parse_wav(buf)
    x = decode_pcm(buf, info)
    a, b = int(start * info["sr"]), min(int(stop * info["sr"]), info["frames"])
    x = x[a:b]
    if x.size == 0:
        return None, None
    m = to_mono(x)
    if info["sr"] != SR:
        m = resample(m, info["sr"], SR)
    return m, SR
