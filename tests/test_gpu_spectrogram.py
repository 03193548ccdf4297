"""SURVEY.md 8(f) N4 -- the review screen's spectrogram helper and excerpt loader on the device against
oracle/oracle_np.py ("parity unpinned" at the librosa boundary: see the oracle's header).  Floating point: the bar is
1e-5 of the spectrogram's peak (float32 FFT against the oracle's float64 one)."""
import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(build_all):
    from softspoken_amd import native
    c = native.Context(None, 0)
    yield c
    c.close()


@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 4095, 4096, 22050 * 3, 22050 * 61 + 17])
def test_stft512_magnitude_matches_oracle(ctx, n):
    from softspoken_amd import synth
    x = synth.synth_audio(9, max(n, 1) / 22050.0 + 0.01, 22050, 1, with_silence=False)[0].astype(np.float32)[:n]
    want = O.stft512_magnitude(x)
    got = ctx.stft512_magnitude(x)
    assert got.shape == want.shape == (257, 1 + n // 256) and got.dtype == np.float32
    peak = max(float(want.max()), 1e-6)
    assert np.abs(got - want).max() <= 1e-5 * peak + 1e-7


def test_stft512_properties_full_length(ctx):
    """10 minutes: a pure tone sits in its bin in every interior frame; doubling the input doubles the output exactly."""
    sr, n = 22050, 22050 * 600
    t = np.arange(n, dtype=np.float64) / sr
    x = (0.25 * np.cos(2 * np.pi * (sr / 512 * 37) * t)).astype(np.float32)
    S = ctx.stft512_magnitude(x)
    assert S.shape == (257, 1 + n // 256)
    assert np.all(S[:, 2:-2].argmax(axis=0) == 37)
    assert np.allclose(S[37, 2:-2], 0.25 * 128.0, rtol=2e-4)
    assert np.array_equal(ctx.stft512_magnitude(2.0 * x), 2.0 * S)


def test_wav_to_spec_and_load_audio_startstop(tmp_path, build_all):
    from root.code.backend import voice_activity, settings
    from softspoken_amd import synth
    assert (settings.n_fft, settings.win_length, settings.hop_length) == (512, 512, 256)
    sr, ch = 44100, 2
    x = synth.synth_audio(4, 8.0, sr, ch, with_silence=False)
    wav = synth.wav_bytes(synth.to_pcm16(x), sr, "pcm16")
    path = tmp_path / "rec.wav"
    path.write_bytes(wav)

    data, rate = voice_activity.load_audio_startstop(str(path), (2, 5))
    want, _ = O.load_audio_startstop_from_bytes(wav, 2, 5)
    assert rate == settings.vad_resample == 22050 and data.dtype == np.float32
    assert np.array_equal(data, want)                                  # decode / mixdown / resample are bit-exact
    late, _ = voice_activity.load_audio_startstop(str(path), (6, 30))  # stop clipped to the file's end
    assert np.array_equal(late, O.load_audio_startstop_from_bytes(wav, 6, 30)[0]) and len(late) == 2 * 22050
    assert voice_activity.load_audio_startstop(str(path), (5, 5)) == (None, None)
    assert voice_activity.load_audio_startstop(str(path), (-1, 2)) == (None, None)
    assert voice_activity.load_audio_startstop(str(path), (9, 10)) == (None, None)
    assert voice_activity.load_audio_startstop(str(tmp_path / "nope.wav"), (0, 1)) == (None, None)

    full = voice_activity.wav_to_spec(data, trim_edges=False)
    ref = O.stft512_magnitude(data)
    assert full.shape == ref.shape and np.abs(full - ref).max() <= 1e-5 * ref.max()
    assert voice_activity.wav_to_spec(data).shape == (256, 256)
    # the review screen hands over a zero-padded float64 buffer (review_detections.py:859-865) and gets float64 back
    padded = np.zeros(4 * 22050)
    padded[:len(data)] = data
    D = voice_activity.wav_to_spec(padded, trim_edges=False)
    assert D.dtype == np.float64 and D.shape == (257, 1 + len(padded) // 256)
    assert np.abs(D - O.stft512_magnitude(padded)).max() <= 1e-5 * ref.max()
