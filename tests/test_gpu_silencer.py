"""SURVEY.md 8(f) N3 -- the silencer through the C ABI against oracle/oracle_np.silence_pcm16.  Integer output:
bit-exact.  ("parity unpinned" at the libsndfile boundary: see the oracle's header.)"""
import os

import numpy as np
import pandas as pd
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native(build_all):
    from softspoken_amd import native
    return native


@pytest.fixture(scope="module")
def ctx(native):
    c = native.Context(None, 0)            # audio-only: the silencer needs no weights
    yield c
    c.close()


def _make(fmt, sr, ch, seconds, seed):
    from softspoken_amd import synth
    x = synth.synth_audio(seed, seconds, sr, ch, with_silence=False).T.reshape(-1, ch)
    x = (x * 1.6).astype(np.float32)                       # reach full scale
    if fmt == "pcm16":
        pcm = np.clip(np.rint(x * 32768), -32768, 32767).astype(np.int16)
    elif fmt == "pcm24":
        pcm = np.clip(np.rint(x * 8388608), -8388608, 8388607).astype(np.int32)
    elif fmt == "pcm32":
        pcm = np.clip(np.rint(x.astype(np.float64) * 2147483648), -2147483648, 2147483647).astype(np.int64).astype(np.int32)
    elif fmt == "u8":
        pcm = np.clip(np.rint(x * 128 + 128), 0, 255).astype(np.uint8)
    else:
        pcm = (x * 1.3).astype(np.float32)                 # float files may exceed full scale
    return synth.wav_bytes(pcm.squeeze(), sr, fmt)


REGIONS = [(0.25, 0.5), (0.4, 0.75), (1.9, 99.0), (-2.0, 0.0105), (1.2, 1.1), (1.00005, 1.00015), (0.3, 0.35)]


@pytest.mark.parametrize("fmt,sr,ch", [("pcm16", 48000, 2), ("pcm24", 44100, 1), ("pcm32", 96000, 4), ("u8", 8000, 1),
                                       ("f32", 22050, 2), ("pcm16", 10000, 3)])
def test_silence_matches_oracle(ctx, native, fmt, sr, ch):
    wav = _make(fmt, sr, ch, 2.0, 77)
    info = native.wav_parse(wav)
    oinfo = O.parse_wav(wav)
    x = O.decode_pcm(wav, oinfo)
    want = O.silence_pcm16(x, sr, REGIONS)
    pcm = np.frombuffer(wav, dtype=np.uint8, count=info.data_bytes, offset=info.data_offset)
    got = ctx.silence_pcm(pcm, info.format, sr, ch, info.frames, REGIONS)
    assert got.shape == want.shape and got.dtype == np.int16
    assert np.array_equal(got, want)
    assert not got[int(round(0.25 * sr)):int(round(0.75 * sr))].any() and got[int(round(0.75 * sr)) + 1:int(round(sr))].any()
    # no regions: a plain transcode; idempotent on its own 16-bit output below 16384
    plain = ctx.silence_pcm(pcm, info.format, sr, ch, info.frames, [])
    assert np.array_equal(plain, O.silence_pcm16(x, sr, []))
    assert native.wav_header_pcm16(sr, ch, info.frames) + got.tobytes() == O.wav_pcm16_bytes(want, sr)


def test_silence_ragged_and_empty(ctx):
    for frames in (0, 1, 3, 4, 5, 1023, 1025):
        pcm = (np.arange(frames * 3, dtype=np.int64) * 2311 % 65536 - 32768).astype(np.int16).reshape(frames, 3)
        x = pcm.astype(np.float32) / np.float32(32768)
        regs = [(0.0, 1e-4), (0.05, 0.0503)]
        got = ctx.silence_pcm(pcm, 2, 10000, 3, frames, regs)
        assert np.array_equal(got, O.silence_pcm16(x, 10000, regs))


def test_silence_large_roundtrip_property(ctx):
    """A 10-minute 48 kHz stereo recording: zero inside, untouched (|s| < 16384 is the identity) outside."""
    sr, ch, frames = 48000, 2, 48000 * 600
    rng = np.random.default_rng(5)
    pcm = rng.integers(-16000, 16000, size=(frames, ch), dtype=np.int16)
    regs = [(float(a), float(a) + 7.3) for a in range(5, 590, 31)]
    got = ctx.silence_pcm(pcm, 2, sr, ch, frames, regs)
    mask = np.zeros(frames, dtype=bool)
    for a, b in regs:
        mask[int(round(a * sr)):int(round(b * sr))] = True
    assert not got[mask].any()
    assert np.array_equal(got[~mask], pcm[~mask])


def test_silence_job_writes_reference_layout(tmp_path, build_all):
    from softspoken_amd import silence, synth
    from root.code.backend import voice_activity
    src = tmp_path / "in" / "deep"
    src.mkdir(parents=True)
    out = tmp_path / "out"
    out.mkdir()
    wavs = {}
    for name, fmt, sr, ch in (("a.wav", "pcm16", 16000, 1), ("b b.WAV", "pcm24", 44100, 2), ("keep.wav", "pcm16", 8000, 1)):
        wavs[name] = _make(fmt, sr, ch, 1.5, 3)
        (src / name).write_bytes(wavs[name])
    df = pd.DataFrame({
        "ID": [1, 2, 3, 4, 5], "file_path": [str(src)] * 5,
        "file_name": ["a.wav", "b b.WAV", "a.wav", "keep.wav", "missing.wav"],
        "start_time": [0.2, 0.1, 0.9, 0.3, 0.0], "end_time": [0.4, 1.2, 1.0, 0.6, 1.0], "erase": [1, 1, 1, 0, 1]})
    seen = []
    job = silence.SilenceJob(df, str(out), file_started=lambda p: seen.append(("start", p)),
                             file_complete=lambda p: seen.append(("done", p)),
                             overall_progress=lambda p: seen.append(("pct", p)), finished=lambda: seen.append(("fin",)))
    paths = job.run()
    assert sorted(os.listdir(out)) == ["a_silenced.wav", "b b_silenced.wav"]          # keep.wav has no erase row
    assert [s for s in seen if s[0] == "pct"] == [("pct", 33), ("pct", 66), ("pct", 100)] and seen[-1] == ("fin",)
    assert list(job.errors) == [os.path.join(str(src), "missing.wav")]
    for name, regs in (("a.wav", [(0.2, 0.4), (0.9, 1.0)]), ("b b.WAV", [(0.1, 1.2)])):
        o = O.parse_wav(wavs[name])
        want = O.wav_pcm16_bytes(O.silence_pcm16(O.decode_pcm(wavs[name], o), o["sr"], regs), o["sr"])
        got = (out / (os.path.splitext(name)[0] + "_silenced.wav")).read_bytes()
        assert got == want
        assert voice_activity.get_audio_data(str(out / (os.path.splitext(name)[0] + "_silenced.wav")))[0] == o["frames"] / o["sr"]
    assert len(paths) == 2
    assert silence.silence_files(df[df["erase"] == 0], str(out)) == []
