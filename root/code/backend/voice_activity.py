"""Audio loading for the voice-detector path on MI355X.

Mirrors the two functions of the reference's root/code/backend/voice_activity.py that the detector
path calls -- get_audio_data (:23-30) and load_audio (:32-69) -- with the same names, arguments,
return values and failure behaviour.  The header walk is ss_wav_parse and decode / mixdown /
resample run on the GPU (ss_add_pcm) instead of libsndfile + librosa + soxr.  The rest of the
reference file (augmentation, plotting, training-clip loaders) is not part of this path.

Formats: RIFF/WAVE with PCM 8/16/24/32-bit or IEEE float 32/64 (what the field recorders the
project targets produce).  Other containers that libsndfile would open are reported as a decode
failure, i.e. load_audio returns (None, None) exactly as the reference does on a failed read.
"""
from __future__ import annotations

import logging
import threading

import numpy as np

from root.code.backend import settings
from softspoken_amd import native as _native

_audio_ctx = None
_audio_lock = threading.Lock()


def _map_file(path):
    return np.memmap(path, dtype=np.uint8, mode="r")


def wav_info(file):
    """Header walk only (the memory map never touches the sample data)."""
    return _native.wav_parse(_map_file(file))


def get_audio_data(file):
    """-> (duration in seconds, native sample rate), from the header alone (reference :23-30)."""
    info = wav_info(file)
    return (info.frames / info.sample_rate, info.sample_rate)


def audio_context(device_index: int = 0):
    """A model-less context for decode/mixdown/resample (created on first use)."""
    global _audio_ctx
    if _audio_ctx is None:
        _audio_ctx = _native.Context(None, device_index)
    return _audio_ctx


def add_file_to_context(ctx, path):
    """Upload a WAV file's samples and decode/mixdown/resample/pad them on the device. -> (file_id, WavInfo)"""
    buf = _map_file(path)
    info = _native.wav_parse(buf)
    pcm = buf[info.data_offset: info.data_offset + info.data_bytes]
    return ctx.add_pcm(pcm, info.format, info.sample_rate, info.channels, info.frames), info


def load_audio(directory, start=None):
    """-> (float32 mono signal at settings.vad_resample, settings.vad_resample), or (None, None) when
    the file cannot be decoded (reference :32-69; order there: decode float32 -> mono -> resample).

    `start` is a sample position at the 22 050 Hz scale; as in the reference (:44-55) the excerpt is cut from the file at its NATIVE
    rate -- frames [int(start * sr / 22050), + int(3 sr)), clipped to the file's end -- and only that excerpt is mixed down and
    resampled."""
    try:
        with _audio_lock:
            ctx = audio_context()
            ctx.reset()
            if start is None:
                fid, info = add_file_to_context(ctx, directory)
            else:
                buf = _map_file(directory)
                info = _native.wav_parse(buf)
                sr = info.sample_rate
                a = max(0, min(int(start * (sr / settings.vad_resample)), info.frames))
                b = max(a, min(a + int(sr * 3), info.frames))
                bpf = info.channels * (info.bits // 8)
                pcm = buf[info.data_offset + a * bpf: info.data_offset + b * bpf]
                fid = ctx.add_pcm(pcm, info.format, sr, info.channels, b - a)
            data = ctx.read_signal(fid, padded=False)
    except Exception as e:       # the reference prints and returns (None, None) (:39-41,57-58)
        logging.error("load_audio failed for %s: %s", directory, e)
        print(f'EXCEPTION EXCEPTION EXCEPTION: \n\t{directory}\n\t{str({e})}')
        return (None, None)
    return (data, settings.vad_resample)


def load_audio_startstop(full_path, start_stop):
    """-> (float32 mono excerpt at settings.vad_resample, settings.vad_resample) for start_stop = (start_s, stop_s) in the
    file's own time base, or (None, None) (reference :72-143: frames [int(start*sr), int(stop*sr)) clipped to the file's
    end, float32, mono, resampled).  Decode, mixdown and resampling of the excerpt run on the device."""
    start, stop = start_stop
    if start < 0 or stop <= start:
        print(f"Invalid start ({start}) and stop ({stop}) times. Ensure that 0 <= start < stop.")
        return None, None
    try:
        buf = _map_file(full_path)
        info = _native.wav_parse(buf)
        sr = info.sample_rate
        a, b = int(start * sr), int(stop * sr)
        if b > info.frames:
            print(f"Requested stop time ({stop}s) exceeds file duration. Adjusting to file's end.")
            b = info.frames
        if b - a <= 0:
            print(f"No data read from {full_path} between {start}s and {stop}s.")
            return None, None
        bpf = info.channels * (info.bits // 8)       # (not data_bytes // frames: a truncated data chunk would give a wrong stride)
        pcm = buf[info.data_offset + a * bpf: info.data_offset + b * bpf]
        with _audio_lock:
            ctx = audio_context()
            ctx.reset()
            fid = ctx.add_pcm(pcm, info.format, sr, info.channels, b - a)
            data = ctx.read_signal(fid, padded=False)
    except Exception as e:
        print(f'EXCEPTION EXCEPTION EXCEPTION:\n\t{full_path}\n\t{str(e)}')
        return None, None
    return data, settings.vad_resample


def wav_to_spec(data, trim_edges=True):
    """|STFT| of a mono signal with the review screen's settings (n_fft = win_length = 512, hop 256; reference :148-154),
    [257, frames] (cut to [256, 256] with trim_edges), computed on the device in float32."""
    data = np.asarray(data)
    with _audio_lock:
        D = audio_context().stft512_magnitude(data.astype(np.float32, copy=False))
    if data.dtype == np.float64:
        D = D.astype(np.float64)
    if trim_edges:
        D = D[..., 0:256, 0:256]
    return D
