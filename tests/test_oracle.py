"""The oracles against the goldens the reference's own classes produced (tests/golden/make_golden.py).
CPU only.  Tolerances: the numpy/torch oracle runs the same torch ops as the reference (differences
are thread-count summation order only); the C oracle is an independent fp32 implementation."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O


def test_signal_generator_is_stable(c1, gold):
    g = gold["c1_signal_check"]
    assert len(c1["sig"]) == int(g["n"]) == 1323000
    np.testing.assert_array_equal(c1["sig"][:64], g["head"])
    np.testing.assert_array_equal(c1["sig"][500000:500064], g["mid"])
    assert abs(float(np.sum(c1["sig"].astype(np.float64) ** 2)) - float(g["sumsq"])) < 1e-6


def test_mel_tables(sd_np, gold):
    g = gold["mel_tables"]
    fb = sd_np["mel_spectrogram.mel_scale.fb"]
    r, c = np.nonzero(fb)
    assert len(r) == 1469 and r.max() == 743
    np.testing.assert_array_equal(r, g["rows"]); np.testing.assert_array_equal(c, g["cols"])
    np.testing.assert_array_equal(fb[r, c], g["vals"])
    np.testing.assert_array_equal(sd_np["mel_spectrogram.spectrogram.window"], g["window"])
    # independent statement of the same filterbank (float64 linspace): agrees to 2e-5
    from transformers.audio_utils import mel_filter_bank
    ref = mel_filter_bank(1025, 128, 0.0, 8000.0, 22050, norm=None, mel_scale="htk")
    assert np.abs(ref - fb).max() < 5e-5
    assert np.array_equal(torch.hann_window(512).numpy(), sd_np["mel_spectrogram.spectrogram.window"]) or \
        np.abs(torch.hann_window(512).numpy() - sd_np["mel_spectrogram.spectrogram.window"]).max() < 1e-7


def test_plan_counts():
    # 3 s -> 10, 60 s -> 105, 10 min -> 1005
    assert [len(O.plan_windows(d)) for d in (3.0, 60.0, 600.0)] == [10, 105, 1005]
    assert O.plan_windows(60.0)[-1] == 104 * 13230


def test_np_oracle_features_and_logits(c1, sd_torch, gold):
    torch.set_grad_enabled(False)
    gf, gl = gold["c1_features"], gold["c1_logits"]
    pick = gf["window_index"]
    x = torch.stack([torch.from_numpy(c1["padded"][s:s + 66150]) for s in c1["starts"][pick]])
    feats = O.mel_features(x, sd_torch["mel_spectrogram.spectrogram.window"], sd_torch["mel_spectrogram.mel_scale.fb"])
    assert np.abs(feats.numpy() - gf["feats"]).max() < 1e-6
    res = O.detect_signal(sd_torch, c1["sig"], 60.0)
    assert res["window_logits"].shape == (105, 1, 256)
    assert np.abs(res["window_logits"] - gl["logits"]).max() < 5e-5
    assert len(res["avg"]) == 5581                           # bins never covered are dropped (A7)
    assert np.abs(res["avg"] - gl["avg"]).max() < 5e-5
    assert [list(r) for r in res["regions_str"]] == gl["regions_str"].tolist()
    assert np.array_equal(np.array(res["regions"]), gl["regions"])
    rows = [(i + 1, "/data/site a", "c1_seed1001.wav", s, e) for i, (s, e) in enumerate(res["regions"])]
    assert O.csv_text(rows) == str(gl["csv"])
    assert O.csv_text([]) == str(gl["empty_csv"])


def test_np_oracle_postprocessing_is_exact_on_golden_logits(gold):
    gl = gold["c1_logits"]
    avg, idx = O.average_overlapping(gl["logits"], int(gl["n_padded"]) / 22050)
    np.testing.assert_array_equal(avg, gl["avg"])
    assert [O.time_str(int(i)) for i in idx] == gl["avg_time_str"].tolist()
    assert [list(r) for r in O.find_regions(avg, idx)] == gl["regions_str"].tolist()
    a0, i0 = O.average_overlapping(np.zeros((0, 1, 256), np.float32), 6.0)
    assert len(a0) == 0 and O.find_regions(a0, i0) == []


def test_c_oracle(c1, sd_np, blob, gold):
    from oracle import oracle_c as OC
    OC.build()
    gf, gl, gy = gold["c1_features"], gold["c1_logits"], gold["c1_layers"]
    info = c1["info"]
    raw = np.frombuffer(c1["wav"], np.uint8)[info["data_off"]:]
    sig = OC.decode_resample(raw, 2, 1, info["frames"], 16000)
    np.testing.assert_array_equal(sig, c1["sig"])            # same taps, same float32 op order
    x = np.stack([c1["padded"][s:s + 66150] for s in c1["starts"][gf["window_index"]]])
    f = OC.mel_features(x, sd_np["mel_spectrogram.spectrogram.window"], sd_np["mel_spectrogram.mel_scale.fb"])
    assert np.abs(f - gf["feats"]).max() < 5e-6
    spec, mask, flat = OC.unet_forward(blob, gf["feats"], want_spec=True)
    assert np.abs(mask - gy["mask"]).max() < 1e-4
    assert np.abs(flat - gy["flatten"].reshape(2, 4, 256)).max() < 1e-4
    assert np.abs(spec[:, :, 64, :] - gy["spec_row64"]).max() < 1e-4
    a, i = OC.average(gl["logits"], int(gl["n_padded"]))
    np.testing.assert_array_equal(a, gl["avg"])
    assert OC.regions(a, i) == [tuple(r) for r in gl["regions"].tolist()]
    assert [OC.plan_windows(d) for d in (3.0, 60.0, 600.0)] == [10, 105, 1005]


@pytest.mark.parametrize("sr,ch,fmt", [(48000, 2, "pcm16"), (44100, 1, "pcm24"), (8000, 1, "u8"), (22050, 2, "f32"),
                                       (16000, 1, "pcm32")])
def test_decode_paths_agree_between_oracles(sr, ch, fmt):
    from softspoken_amd import synth
    from oracle import oracle_c as OC
    x = synth.synth_audio(11, 1.5, sr, ch, with_silence=False)
    if fmt == "pcm16":
        pcm = synth.to_pcm16(x)
    elif fmt == "pcm24":
        pcm = np.rint(x.T * 8388607).astype(np.int32).reshape(-1, ch).squeeze()
    elif fmt == "pcm32":
        pcm = np.rint(x.T * 2147483000).astype(np.int64).astype(np.int32).reshape(-1, ch).squeeze()
    elif fmt == "u8":
        pcm = np.clip(np.rint(x.T * 127 + 128), 0, 255).astype(np.uint8).reshape(-1, ch).squeeze()
    else:
        pcm = x.T.astype(np.float32).reshape(-1, ch).squeeze()
    wav = synth.wav_bytes(pcm, sr, fmt)
    sig, _, info = O.load_audio_from_bytes(wav)
    code = {"u8": 1, "pcm16": 2, "pcm24": 3, "pcm32": 4, "f32": 5}[fmt]
    raw = np.frombuffer(wav, np.uint8)[info["data_off"]: info["data_off"] + info["data_len"]]
    sig_c = OC.decode_resample(raw, code, ch, info["frames"], sr)
    assert len(sig) == len(sig_c) == int(np.ceil(info["frames"] * 22050 / sr))
    assert np.abs(sig - sig_c).max() < 2e-7
    # resampler sanity: a 1 kHz tone keeps its amplitude
    t = np.arange(sr) / sr
    tone = (0.5 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32)
    y = O.resample(tone, sr)
    mid = y[2000:-2000]
    assert abs(np.sqrt(2 * np.mean(mid.astype(np.float64) ** 2)) - 0.5) < 2e-3


# ---- silencer: known answers for the restatement --------------------------------
def test_silencer_oracle_known_answers():
    x = np.array([[0.5, -0.5], [1.0, -1.0], [0.25, 0.75], [-0.999969482421875, 0.0], [0.1, 0.2]], dtype=np.float32)
    out = O.silence_pcm16(x, 10, [(0.15, 0.25)])              # round(1.5)=2, round(2.5)=2 -> nothing cut
    assert out.tolist() == [[16384, -16384], [32767, -32767], [8192, 24575], [-32766, 0], [3277, 6553]]
    out = O.silence_pcm16(x, 10, [(0.05, 0.25), (0.4, 9.0), (-3.0, 0.04)])   # round(.5)=0, [0,2) and [4,5)
    assert out.tolist() == [[0, 0], [0, 0], [8192, 24575], [-32766, 0], [0, 0]]
    out = O.silence_pcm16(x, 10, [(0.3, 0.1)])                # end before start: empty slice
    assert out[3].tolist() == [-32766, 0]
    loud = np.array([[1.5], [-1.5]], dtype=np.float32)         # beyond full scale wraps, as an unclipped C cast does
    assert O.silence_pcm16(loud, 8000, []).tolist() == [[49150 - 65536], [-49150 + 65536]]
    wav = O.wav_pcm16_bytes(out, 10)
    info = O.parse_wav(wav)
    assert (info["channels"], info["sr"], info["bits"], info["frames"], len(wav)) == (2, 10, 16, 5, 44 + 20)


# ---- review-screen spectrogram: known answers for the restatement ------------------
def test_stft512_oracle_known_answers():
    sr, n = 22050, 22050
    t = np.arange(n) / sr
    f0 = 43.06640625 * 20                          # bin 20 exactly (sr / 512 per bin)
    x = np.cos(2 * np.pi * f0 * t).astype(np.float32)
    S = O.stft512_magnitude(x)
    assert S.shape == (257, 1 + n // 256) and S.dtype == np.float32
    mid = S[:, 10:70]
    assert np.all(mid.argmax(axis=0) == 20)
    assert np.allclose(mid[20], 128.0, rtol=1e-4)          # A/2 * sum(hann) = 0.5 * 256
    assert np.allclose(mid[19], 64.0, rtol=1e-3) and np.allclose(mid[21], 64.0, rtol=1e-3)   # Hann side lobes = half
    assert mid[40:].max() < 1e-2
    # centred frames over zero padding: frame 0 sees only the second half of its window
    assert 40.0 < S[20, 0] < 90.0
    # float64 in -> float64 out, one frame for an empty signal
    assert O.stft512_magnitude(x.astype(np.float64)).dtype == np.float64
    assert O.stft512_magnitude(np.zeros(0, dtype=np.float32)).shape == (257, 1)
    # linear: |STFT(2x)| = 2 |STFT(x)|
    assert np.allclose(O.stft512_magnitude(2 * x), 2 * S, rtol=1e-6, atol=1e-6)


def test_np_oracle_on_the_hostile_scale_checkpoint(c1):
    """tests/golden/c1_hostile.npz: the reference's SpecUNet_2D on the checkpoint whose BatchNorm gains put conv3_1 ... conv7 near
    1e-3 / 1e3 (synth.HOSTILE_GAINS).  The oracle runs the same torch ops: same logits up to the thread-count summation order."""
    import os
    from softspoken_amd import synth
    torch.set_grad_enabled(False)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_hostile.npz"))
    am = dict(zip(g["absmax_names"].tolist(), g["absmax"].tolist()))
    assert am["conv3_1"] < 0.02 and am["conv4_1"] > 3e3 and am["encoder_out"] < 0.03 and am["conv6"] > 3e3 and am["conv7"] < 0.02
    sd = synth.to_torch_state_dict(synth.make_state_dict(0, hostile=True))
    got = O.infer_windows(sd, c1["padded"], c1["starts"][40:56])
    assert np.abs(got - g["logits"][40:56]).max() < 5e-5
    base = synth.to_torch_state_dict(synth.make_state_dict(0))
    same = O.infer_windows(base, c1["padded"], c1["starts"][40:44])       # the gains are undone downstream: the same function up to rounding
    assert np.abs(same - g["logits"][40:44]).max() < 1e-4
