"""Host logic and the C-ABI surface, no GPU: the library loads, exports every symbol the header
declares, and its host-only entry points (WAV walk, planning, region finding, CSV text) agree with the
oracle / goldens.  No compute entry point is called here."""
import os
import re

import numpy as np
import pytest

from oracle import oracle_np as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native(build_all):
    from softspoken_amd import native
    return native


def test_library_exports_every_declared_symbol(native):
    hdr = open(os.path.join(ROOT, "include", "softspoken.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    dev_only = set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", "".join(re.findall(r"#ifdef SS_DEVBUILD(.*?)#endif", hdr, flags=re.S))))
    assert dev_only == set(native._DEV_SIGS)              # the development build's extras: not in the product library
    hdr = re.sub(r"#ifdef SS_DEVBUILD.*?#endif", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", hdr)) - {"ss_progress_fn"}
    assert declared == set(native.EXPORTS)
    L = native.lib()
    for name in declared:
        assert getattr(L, name) is not None
    for name in dev_only:
        assert not hasattr(L, name)
    assert L.ss_abi_version() == 3 == native.ABI_VERSION


def test_create_fails_loudly_without_gpu(native, blob):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.NativeError) as e:
        native.Context(blob)
    assert "no CPU fallback" in str(e.value)


@pytest.mark.parametrize("fmt,code,bits", [("pcm16", 2, 16), ("pcm24", 3, 24), ("pcm32", 4, 32), ("u8", 1, 8), ("f32", 5, 32)])
def test_wav_parse(native, fmt, code, bits):
    from softspoken_amd import synth
    pcm = (np.arange(2 * 1000) % 200).reshape(1000, 2)
    wav = synth.wav_bytes(pcm, 48000, fmt)
    i = native.wav_parse(wav)
    assert (i.format, i.channels, i.sample_rate, i.bits, i.frames, i.data_offset) == (code, 2, 48000, bits, 1000, 44)
    ref = O.parse_wav(wav)
    assert (ref["frames"], ref["data_off"], ref["sr"]) == (i.frames, i.data_offset, i.sample_rate)
    # a LIST chunk before data, odd-sized (pad byte) -> still found
    extra = wav[:36] + b"LIST" + (5).to_bytes(4, "little") + b"abcde\x00" + wav[36:]
    j = native.wav_parse(extra)
    assert (j.frames, j.data_offset) == (1000, 44 + 14)


@pytest.mark.parametrize("bits,comp,code", [(8, None, 7), (16, None, 8), (24, None, 9), (32, None, 10), (16, b"sowt", 2), (24, b"sowt", 3),
                                            (32, b"fl32", 11), (64, b"fl64", 12)])
def test_aiff_parse_agrees_with_the_stdlib_reader(native, bits, comp, code):
    """AIFF / AIFF-C (round 4): the header walk against Python's own `aifc` module (an independent reader) on files from
    synth.aiff_bytes, and the oracle's decode against the samples `aifc` returns (big endian, 8-bit samples signed)."""
    import aifc, io
    from softspoken_amd import synth
    rng = np.random.default_rng(bits)
    if comp in (b"fl32", b"fl64"):
        x = rng.standard_normal((333, 2))
    else:
        x = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(333, 2))
    b = synth.aiff_bytes(x, 44100, bits, comp)
    i = native.wav_parse(b)
    info = O.parse_wav(b)
    assert (i.format, i.channels, i.sample_rate, i.bits, i.frames) == (code, 2, 44100, bits, 333)
    assert (info["frames"], info["data_off"], info["sr"], info["channels"]) == (i.frames, i.data_offset, i.sample_rate, i.channels)
    assert i.data_offset + i.frames * i.channels * (i.bits // 8) <= len(b)
    dec = O.decode_pcm(b, info)
    if comp is None:                                        # `aifc` reads uncompressed AIFF (and AIFF-C 'NONE') only
        f = aifc.open(io.BytesIO(b))
        assert (f.getnchannels(), f.getframerate(), f.getnframes(), f.getsampwidth()) == (2, 44100, 333, bits // 8)
        raw = np.frombuffer(f.readframes(333), dtype=np.uint8)
        if bits == 24:
            t = raw.reshape(-1, 3).astype(np.int64)
            v = (t[:, 0] << 16) | (t[:, 1] << 8) | t[:, 2]
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
        else:
            v = raw.view({8: "i1", 16: ">i2", 32: ">i4"}[bits]).astype(np.int64)
        assert np.array_equal(v.reshape(-1, 2), x)
        assert np.array_equal(dec, (x.astype(np.float64) / float(1 << (bits - 1))).astype(np.float32))
    elif comp == b"sowt":
        assert np.array_equal(dec, (x.astype(np.float64) / float(1 << (bits - 1))).astype(np.float32))
    else:
        assert np.array_equal(dec, x.astype(">f4" if comp == b"fl32" else ">f8").astype(np.float32))


def test_wav_parse_rejects_garbage(native):
    for bad in (b"", b"RIFFxxxxWAVE", b"not a wav file at all, really not", b"RIFF\x00\x00\x00\x00WAVEdata\x04\x00\x00\x00abcd"):
        with pytest.raises(native.NativeError):
            native.wav_parse(bad + b"\x00" * 4)


def test_wav_parse_survives_mutated_headers(native):
    """Random byte flips, truncations and chunk-size edits of valid files: the header walk either reports a format error or returns a
    description that stays inside the buffer (the device decode trusts offset and byte count)."""
    from softspoken_amd import synth
    rng = np.random.default_rng(11)
    base = [synth.wav_bytes((np.arange(2 * 300) % 97).reshape(300, 2), 44100, f) for f in ("pcm16", "pcm24", "u8", "f32")]
    base.append(base[0][:36] + b"LIST" + (7).to_bytes(4, "little") + b"abcdefg\x00" + base[0][36:])
    pat = (np.arange(2 * 300) % 97).reshape(300, 2)
    base += [synth.aiff_bytes(pat, 44100, 16), synth.aiff_bytes(pat, 22050, 24), synth.aiff_bytes(pat, 48000, 16, b"sowt"),
             synth.aiff_bytes(pat.astype(np.float64), 8000, 32, b"fl32")]
    ok = bad = 0
    for trial in range(3000):
        b = bytearray(base[trial % len(base)])
        for _ in range(int(rng.integers(1, 4))):
            kind = rng.integers(0, 4)
            if kind == 0:                                   # flip a byte in the first 64
                b[int(rng.integers(0, min(64, len(b))))] = int(rng.integers(0, 256))
            elif kind == 1:                                 # truncate
                b = b[:int(rng.integers(0, len(b) + 1))]
            elif kind == 2 and len(b) >= 8:                 # a huge / odd little-endian size somewhere in the header area
                at = int(rng.integers(4, max(5, min(60, len(b) - 4))))
                b[at:at + 4] = int(rng.choice([0, 1, 0x7fffffff, 0xffffffff, 0xfffffffe, int(rng.integers(0, 1 << 32))])).to_bytes(4, "little")
            else:                                           # garbage appended
                b += bytes(rng.integers(0, 256, int(rng.integers(0, 9)), dtype=np.uint8))
        try:
            i = native.wav_parse(bytes(b) if len(b) else b"\x00")
        except native.NativeError:
            bad += 1
            continue
        ok += 1
        assert i.channels >= 1 and i.sample_rate >= 1 and i.frames >= 0 and i.data_offset >= 20
        assert i.data_offset + i.frames * i.channels * (i.bits // 8) <= len(b)
    assert ok > 200 and bad > 200


def test_plan_windows_matches_oracle(native):
    rng = np.random.default_rng(0)
    durs = [0.0, 0.01, 0.59, 0.6, 0.61, 3.0, 59.99, 60.0, 600.0, 3600.0] + list(rng.uniform(0, 900, 200)) + \
           [n / sr for n, sr in [(48001, 16000), (1, 8000), (13230001, 22050), (7919, 44100)]]
    for d in durs:
        ref = O.plan_windows(d)
        got = native.plan_windows(d)
        assert np.array_equal(ref, got), d
    L = native.lib()
    assert L.ss_resampled_length(48000, 16000) == 66150 and L.ss_resampled_length(48001, 16000) == 66152
    assert L.ss_resampled_length(1234, 22050) == 1234


def test_find_regions_matches_oracle_on_random_series(native, gold):
    gl = gold["c1_logits"]
    idx = np.nonzero(np.ones(len(gl["avg"])))[0]
    got = native.find_regions(gl["avg"], idx)
    assert got == [tuple(r) for r in gl["regions"].tolist()]
    rng = np.random.default_rng(1)
    for trial in range(60):
        n = int(rng.integers(0, 4000))
        # smooth-ish random walk so that runs and short gaps both occur
        x = np.cumsum(rng.standard_normal(n)) * 0.05 + 0.1 + rng.standard_normal(n) * 0.02 if n else np.zeros(0)
        keep = np.sort(rng.choice(np.arange(n + 50), size=n, replace=False)) if n else np.zeros(0, np.int64)
        ref = O.regions_minus_pad(O.find_regions(x, keep))
        assert native.find_regions(x, keep) == ref
    # threshold is strict, exact value is not a detection
    assert native.find_regions(np.array([0.1, 0.1]), np.array([0, 1])) == []
    assert native.find_regions(np.zeros(0), np.zeros(0, np.int64)) == []


def test_csv_rows_match_pandas(native, gold):
    import pandas as pd
    gl = gold["c1_logits"]
    regs = [tuple(r) for r in gl["regions"].tolist()]
    hdr = "ID,file_path,file_name,start_time,end_time,erase,user_comment,review_datetime\n"
    assert hdr + native.format_csv_rows("/data/site a", "c1_seed1001.wav", regs, 1) == str(gl["csv"])
    rng = np.random.default_rng(2)
    vals = list(rng.uniform(-3, 4000, 300)) + [0.0, -0.0, 1e-5, 123456789012345680.0, 1e16, 9999999999999998.0, 0.0001,
                                               0.00009999, -2.9883, 5e-324, 1.7976931348623157e308, 0.011699999999999822]
    regs = [(float(a), float(b)) for a, b in zip(vals[::2], vals[1::2])]
    text = native.format_csv_rows('/d,ir/with "quote"', "na\nme.wav", regs, 7)
    from softspoken_amd.detections import COLUMN_TYPES
    df = pd.DataFrame(columns=COLUMN_TYPES.keys()).astype(COLUMN_TYPES)
    for k, (s, e) in enumerate(regs):
        df.loc[len(df)] = {'ID': 7 + k, 'file_path': '/d,ir/with "quote"', 'file_name': "na\nme.wav", 'start_time': s,
                           'end_time': e, 'erase': 0, 'user_comment': '', 'review_datetime': ''}
    assert hdr + text == df.to_csv(index=False)
    assert O.csv_text([(7 + k, '/d,ir/with "quote"', "na\nme.wav", s, e) for k, (s, e) in enumerate(regs)]) == hdr + text


def test_checkpoint_pack_roundtrip(sd_np, blob, tmp_path):
    import struct
    assert blob[:8] == b"SSWBLOB1"
    (n,) = struct.unpack_from("<I", blob, 8)
    assert n == 224
    from softspoken_amd import synth, checkpoint
    p = tmp_path / "model_checkpoint.pth"
    synth.save_checkpoint(str(p), 0, epoch=4)
    sd, epoch = checkpoint.load_checkpoint_file(str(p))
    assert epoch == 4 and checkpoint.pack_state_dict(sd) == blob


def test_model_state_dict_layout_and_strict_load(sd_torch, build_all):
    """Key layout == the reference's (the goldens script asserts synth layout == reference state_dict)."""
    from root.code.backend.pytorch_neural_nets import SpecUNet_2D
    m = SpecUNet_2D()
    sd = m.state_dict()
    assert list(sd.keys()).sort() == list(sd_torch.keys()).sort() and len(sd) == 224
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(sd_torch[k].shape) and v.dtype == sd_torch[k].dtype, k
    m.load_state_dict(sd_torch, strict=True)
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 1713555
    assert sum(v.numel() for k, v in sd.items() if not k.startswith("mel_spectrogram")) == 1718607


class _PM:
    def __init__(self, files):
        self.files = files

    def get_unprocessed_list(self):
        return list(self.files)


def test_detector_host_methods_match_reference_goldens(gold, tmp_path, build_all):
    """NNDetector.average_overlapping_detections / find_speech_regions on the reference's own logits."""
    from softspoken_amd import synth
    from root.code.frontend.NNDetector import NNDetector
    from root.code.backend import settings
    gl = gold["c1_logits"]
    wavp = tmp_path / "a.wav"
    synth.write_wav(str(wavp), synth.to_pcm16(synth.synth_audio(3, 60.0, 16000, 1)), 16000)
    ck = tmp_path / "ck.pth"
    synth.save_checkpoint(str(ck), 0, epoch=11)
    with pytest.raises(FileNotFoundError):
        NNDetector(_PM([str(wavp)]), checkpoint_path=str(tmp_path / "missing.pth"))
    det = NNDetector(_PM([str(wavp)]), checkpoint_path=str(ck))
    assert det.load_checkpoint(det.model, str(ck)) == 12 and det.load_checkpoint(det.model, "/nonexistent") == -1
    plan = det.plan_detection_job()
    assert list(plan) == [str(wavp)] and np.array_equal(plan[str(wavp)], O.plan_windows(60.0)) and plan[str(wavp)].dtype == np.int64
    f = str(gl["file_key"])
    avg = det.average_overlapping_detections({f: gl["logits"]}, int(gl["n_padded"]) / settings.vad_resample)
    assert [t for _, t in avg[f]] == gl["avg_time_str"].tolist()
    assert np.array_equal(np.array([a for a, _ in avg[f]]), gl["avg"])
    reg = det.find_speech_regions({f: avg}, break_duration=0.5)
    assert [list(r) for r in reg[f]] == gl["regions_str"].tolist()
    assert det.average_overlapping_detections({f: np.array([])}, 66.0)[f] == []
    assert det.find_speech_regions({f: {f: []}})[f] == []
    assert det.extract_filename("/a/b/c.d.wav") == "c.d"


def test_settings_names_and_values():
    from root.code.backend import settings as s
    assert (s.n_fft, s.win_length, s.hop_length, s.step_size, s.prediction_batch_size, s.threshold, s.vad_resample,
            s.model_name, s.minimum_detection_len) == (512, 512, 256, 0.6, 32, 0.1, 22050, 'model_checkpoint.pth', 0.1)
    assert s.model_dir.replace("\\", "/").endswith("root/models/spec_unet_2d_pytorch")


def test_shard_files_lpt():
    from softspoken_amd.parallel import shard_files
    d = [600, 10, 600, 300, 300, 5, 600, 1]
    sh = shard_files(d, 3)
    assert sorted(sum(sh, [])) == list(range(8))
    loads = [sum(d[i] for i in s) for s in sh]
    assert max(loads) - min(loads) <= 300 and max(loads) <= 900
    assert shard_files(d, 3) == sh and shard_files([], 2) == [[], []]
    assert [len(s) for s in shard_files([1.0] * 1000, 8)] == [125] * 8


def test_wav_header_pcm16_matches_oracle(native):
    from oracle import oracle_np as O
    for sr, ch, frames in ((8000, 1, 0), (44100, 2, 12345), (96000, 6, 7)):
        want = O.wav_pcm16_bytes(np.zeros((frames, ch), dtype=np.int16), sr)[:44]
        assert native.wav_header_pcm16(sr, ch, frames) == want
        info = native.wav_parse(want + bytes(frames * ch * 2))
        assert (info.sample_rate, info.channels, info.frames, info.format) == (sr, ch, frames, 2)
    with pytest.raises(native.NativeError):
        native.wav_header_pcm16(48000, 2, 1 << 31)            # would not fit a RIFF file


def test_bin_time_fast_path_equals_the_reference_expression(native):
    """Region bounds are float(f"{idx / (256 / 3):.4f}") - 3 (NNDetector.py:185, worker.py:100); the library rounds with integers
    except at exact ties.  Every bin index of a 2.7-hour file, isolated bins so that each one becomes a region of its own."""
    n = 240000
    idx = np.arange(n, dtype=np.int64)
    for parity in (0, 1):
        avg = np.where((idx & 1) == parity, 1.0, 0.0)
        got = native.find_regions(avg, idx, threshold=0.5, break_s=-1.0)
        want_idx = idx[(idx & 1) == parity]
        assert len(got) == len(want_idx)
        want = np.array([float(f"{i / (256 / 3):.4f}") - 3.0 for i in want_idx.tolist()])
        g = np.array(got)
        assert np.array_equal(g[:, 0], want) and np.array_equal(g[:, 1], want)


def test_pcm_buffer_shorter_than_the_frame_count_is_refused():
    """The C ABI copies frames * channels * bytes_per_sample from the caller's pointer; the binding checks the numpy buffer first."""
    from softspoken_amd import native
    buf = np.zeros(1000, dtype=np.int16)
    native.Context._need_bytes(buf, native.PCM_S16, 1, 1000)
    native.Context._need_bytes(buf, native.PCM_S16, 2, 500)
    for fmt, ch, frames in ((native.PCM_S16, 1, 1001), (native.PCM_S16, 2, 501), (native.PCM_S24, 1, 700), (native.PCM_F64, 1, 251),
                            (native.PCM_S16, 1, np.array([600, 401]))):
        with pytest.raises(ValueError):
            native.Context._need_bytes(buf, fmt, ch, frames)
    with pytest.raises(ValueError):
        native.Context._need_bytes(buf, 99, 1, 1)
