"""Parity of the HIP path (through the C ABI) with the oracle and the reference-made goldens, on a
real MI355X.  Bars: fp32 scores within 1e-4 of the reference CPU path (north star), detection
tables / CSV text identical, integer/index work bit-exact; bf16 is a throughput mode and is held to
a looser, stated tolerance.  Full-size cases (BASELINE configs 2/3) are covered by size-independent
properties: chunking invariance, shift equivariance, job batching invariance, silence."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

TOL_FP32 = 1e-4      # BASELINE.json north_star: "within 1e-4 fp32"
TOL_FEAT = 1e-5


@pytest.fixture(scope="module")
def native(build_all):
    from softspoken_amd import native
    return native


@pytest.fixture(scope="module")
def ctx(native, blob):
    c = native.Context(blob, 0, bf16=False)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_bf16(native, blob):
    c = native.Context(blob, 0, bf16=True)
    yield c
    c.close()


def test_frontend_features(ctx, c1, gold, sd_torch):
    gf = gold["c1_features"]
    ctx.reset()
    fid = ctx.add_f32_22k(c1["sig"])
    feats = ctx.features(fid, c1["starts"][gf["window_index"]])
    assert np.abs(feats - gf["feats"]).max() < TOL_FEAT
    allf = ctx.features(fid, c1["starts"])                          # 105 windows, incl. silence and the 1e-3 burst
    x = torch.stack([torch.from_numpy(c1["padded"][s:s + 66150]) for s in c1["starts"]])
    ref = O.mel_features(x, sd_torch["mel_spectrogram.spectrogram.window"], sd_torch["mel_spectrogram.mel_scale.fb"]).numpy()
    d = np.abs(allf - ref)
    # quiet bins sit on the float32 grid of log10(x + 1) (steps of ~2.3e-4 near 0):
    # a one-ulp difference in the mel sum may move a bin by one such step; everything else is ~1e-6
    assert (d > TOL_FEAT).sum() <= 4 and d.max() < 3e-4
    assert np.array_equal(allf[0], np.zeros_like(allf[0]))           # leading 3 s pad: exactly zero features
    stats = gf["feat_stats"]
    assert np.abs(allf.mean(axis=(1, 2)) - stats[:, 0]).max() < 1e-6


def test_fp32_logits_vs_reference_goldens(ctx, c1, gold):
    gl, gy = gold["c1_logits"], gold["c1_layers"]
    ctx.reset()
    fid = ctx.add_f32_22k(c1["sig"])
    spec, mask = ctx.infer_windows(fid, c1["starts"])
    assert mask.shape == (105, 1, 256) and spec is None
    assert np.abs(mask - gl["logits"]).max() < TOL_FP32
    spec, mask2 = ctx.infer_windows(fid, c1["starts"][gy["window_index"]], want_spec=True)
    assert np.abs(mask2 - gy["mask"]).max() < TOL_FP32
    assert spec.shape == (2, 2, 128, 256)
    assert np.abs(spec[:, :, 64, :] - gy["spec_row64"]).max() < TOL_FP32
    for b in range(2):
        assert abs(spec[b].astype(np.float64).mean() - gy["spec_stats"][b, 0]) < 1e-5
        assert abs(np.abs(spec[b]).max() - gy["spec_stats"][b, 1]) < TOL_FP32


def test_whole_job_table_and_csv_identical(native, ctx, c1, gold):
    gl = gold["c1_logits"]
    ctx.reset()
    fid = ctx.add_f32_22k(c1["sig"])
    assert ctx.run(0.1, 0.5)
    assert ctx.num_windows(fid) == 105
    assert np.abs(ctx.window_logits(fid) - gl["logits"]).max() < TOL_FP32
    avg, idx = ctx.avg(fid)
    assert len(avg) == 5581 and np.array_equal(idx, np.arange(5581))
    assert np.abs(avg - gl["avg"]).max() < TOL_FP32
    # device averaging is exact given the device logits (float64 sums of float32)
    a2, i2 = O.average_overlapping(ctx.window_logits(fid), int(gl["n_padded"]) / 22050)
    assert np.array_equal(a2, avg) and np.array_equal(i2, idx)
    regs = ctx.regions(fid)
    assert regs == [tuple(r) for r in gl["regions"].tolist()]
    text = O.CSV_HEADER + "\n" + native.format_csv_rows("/data/site a", "c1_seed1001.wav", regs, 1)
    assert text == str(gl["csv"])


def test_device_decode_resample_matches_oracle(ctx, c1):
    ctx.reset()
    fid, info = ctx.add_wav_bytes(c1["wav"])
    dev = ctx.read_signal(fid)
    assert len(dev) == len(c1["sig"]) and np.array_equal(dev, c1["sig"])       # same taps, same op order
    pad = ctx.read_signal(fid, padded=True)
    assert len(pad) == len(dev) + 2 * 66150 and not pad[:66150].any() and not pad[-66150:].any()


def test_decode_resample_bit_exact_over_sample_rates(ctx):
    """Common rates, their odd relatives (44 056, 22 051: a 22 050-phase polyphase bank), primes and extremes, a few samples and a
    second and a bit: the device signal equals the oracle's bit for bit (same taps, same order of roundings)."""
    from softspoken_amd import synth
    for sr in (4000, 7919, 8000, 11025, 12000, 16000, 22050, 22051, 24000, 32000, 44056, 44100, 48000, 50000, 88200, 96000, 192000, 384000):
        for secs in (0.013, 1.37):
            wav = synth.wav_bytes(synth.to_pcm16(synth.synth_audio(5, secs, sr, 1, with_silence=False)), sr, "pcm16")
            ref, _, _ = O.load_audio_from_bytes(wav)
            ctx.reset()
            fid, _ = ctx.add_wav_bytes(wav)
            dev = ctx.read_signal(fid)
            assert len(dev) == len(ref) and np.array_equal(dev, ref), (sr, secs)


@pytest.mark.parametrize("sr,ch,bits,comp,code", [(48000, 2, 16, None, 8), (44100, 1, 24, None, 9), (8000, 1, 8, None, 7), (96000, 3, 32, None, 10),
                                                  (16000, 1, 16, b"sowt", 2), (22050, 2, 32, b"fl32", 11), (32000, 1, 64, b"fl64", 12)])
def test_device_decode_aiff(ctx, sr, ch, bits, comp, code):
    """AIFF / AIFF-C containers (big-endian samples, signed 8-bit; round 4): decode + mixdown + resample on the device equal the
    oracle's bit for bit; the reference reads them through soundfile like any WAV (voice_activity.py:37)."""
    from softspoken_amd import synth
    x = synth.synth_audio(23, 1.5, sr, ch, with_silence=False).T                 # (n, ch) in [-1, 1)
    if comp in (b"fl32", b"fl64"):
        pcm = x
    else:
        pcm = np.rint(x * ((1 << (bits - 1)) - 1)).astype(np.int64)
    img = synth.aiff_bytes(pcm, sr, bits, comp)
    ref, _, info = O.load_audio_from_bytes(img)
    ctx.reset()
    fid, winfo = ctx.add_wav_bytes(img)
    assert (winfo.format, winfo.channels, winfo.sample_rate, winfo.frames) == (code, ch, sr, len(pcm))
    dev = ctx.read_signal(fid)
    assert len(dev) == len(ref) and np.array_equal(dev, ref)


@pytest.mark.parametrize("sr,ch,fmt,code", [(48000, 2, "pcm16", 2), (44100, 1, "pcm24", 3), (8000, 1, "u8", 1),
                                            (22050, 2, "f32", 5), (96000, 4, "pcm32", 4), (22050, 1, "pcm16", 2)])
def test_device_decode_formats(ctx, sr, ch, fmt, code):
    from softspoken_amd import synth
    x = synth.synth_audio(21, 2.0, sr, ch, with_silence=False)
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
    ref, _, info = O.load_audio_from_bytes(wav)
    ctx.reset()
    fid, winfo = ctx.add_wav_bytes(wav)
    assert (winfo.format, winfo.channels, winfo.sample_rate) == (code, ch, sr)
    dev = ctx.read_signal(fid)
    assert len(dev) == len(ref)
    assert np.abs(dev - ref).max() < 2e-7


@pytest.mark.parametrize("bf16", [False, True])
def test_ragged_batches_and_chunking_are_bit_identical(native, blob, c1, bf16):
    """n = 1, 2, 33 windows in chunks of 1 / 7 / 64 -> the same bits in both precisions (windows are independent and every
    reduction runs in a fixed order)."""
    outs = []
    for chunk in (1, 7, 64):
        c = native.Context(blob, 0, bf16=bf16, chunk=chunk)
        fid = c.add_f32_22k(c1["sig"])
        _, m = c.infer_windows(fid, c1["starts"][:33])
        outs.append(m)
        if chunk == 7:
            _, one = c.infer_windows(fid, c1["starts"][20:21])
            _, two = c.infer_windows(fid, c1["starts"][[20, 5]])
            assert np.array_equal(one[0], m[20]) and np.array_equal(two[0], m[20]) and np.array_equal(two[1], m[5])
        c.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_shift_equivariance_full_length(ctx, sd_torch):
    """BASELINE config-3 sized property (one 10-min recording, 1005 windows): delaying the signal by one
    step (13230 samples) moves every window's logits by one index, bit for bit."""
    from softspoken_amd import synth
    x = synth.synth_audio(3000, 600.0, 22050, 1)[0].astype(np.float32)
    ctx.reset()
    f0 = ctx.add_f32_22k(x)
    f1 = ctx.add_f32_22k(np.concatenate([np.zeros(13230, np.float32), x]))
    assert ctx.run()
    a, b = ctx.window_logits(f0), ctx.window_logits(f1)
    assert a.shape[0] == 1005 and b.shape[0] == 1006
    assert np.array_equal(a, b[1:])
    # a few windows spot-checked against the oracle at full length
    pick = np.array([0, 4, 5, 500, 1000, 1004])
    padded = O.pad_3s(x)
    ref = O.infer_windows(sd_torch, padded, O.plan_windows(600.0)[pick])
    assert np.abs(a[pick] - ref).max() < TOL_FP32


def test_job_batching_invariance_config2_shape(native, blob, ctx):
    """BASELINE config-2 shape (256 x 3 s clips, 2560 windows): one job over all clips == each clip alone."""
    from softspoken_amd import synth
    clips = [synth.to_pcm16(synth.synth_audio(2000 + k, 3.0, 16000, 1, with_silence=False)) for k in range(256)]
    ctx.reset()
    ids = [ctx.add_pcm(p, native.PCM_S16, 16000, 1, len(p)) for p in clips]
    assert ctx.run()
    assert all(ctx.num_windows(i) == 10 for i in ids)
    joint = [ctx.window_logits(i) for i in ids]
    joint_regs = [ctx.regions(i) for i in ids]
    counts, reg = ctx.regions_batch(ids[0], len(ids))            # one call for the whole job == the per-file calls
    assert counts.tolist() == [len(r) for r in joint_regs] and int(counts.sum()) == len(reg) > 0
    assert [tuple(x) for x in reg.tolist()] == [r for regs in joint_regs for r in regs]
    with pytest.raises(native.NativeError):
        ctx.regions_batch(ids[0], len(ids) + 1)
    for k in (0, 17, 255):
        ctx.reset()
        i = ctx.add_pcm(clips[k], native.PCM_S16, 16000, 1, len(clips[k]))
        assert ctx.run()
        assert np.array_equal(ctx.window_logits(i), joint[k]) and ctx.regions(i) == joint_regs[k]
        avg, idx = ctx.avg(i)
        assert len(avg) == 717 and idx[-1] == 716               # round(51.2 * 9) + 256 covered bins of 768


def test_edge_cases(native, ctx):
    # empty file: plan has 5 windows over the 6 s of padding; features exactly zero; no crash
    ctx.reset()
    f_empty = ctx.add_f32_22k(np.zeros(0, np.float32))
    f_short = ctx.add_f32_22k(np.full(100, 0.25, np.float32))
    f_sil = ctx.add_f32_22k(np.zeros(66150, np.float32))
    assert ctx.run()
    assert ctx.num_windows(f_empty) == len(O.plan_windows(0.0)) == 5
    assert ctx.num_windows(f_short) == len(O.plan_windows(100 / 22050))
    assert ctx.num_windows(f_sil) == 10
    lg = ctx.window_logits(f_sil)
    assert np.array_equal(lg[0], lg[9]) and np.array_equal(lg, np.broadcast_to(lg[0], lg.shape))   # silence: all windows equal
    assert np.array_equal(ctx.window_logits(f_empty)[0], lg[0])
    # bad window start is an error, not a fault
    with pytest.raises(native.NativeError):
        ctx.infer_windows(f_short, np.array([10 ** 9]))
    with pytest.raises(native.NativeError):
        ctx.infer_windows(f_short, np.array([-1]))
    with pytest.raises(native.NativeError):
        ctx.features(99, np.array([0]))
    # stop flag observed between chunks -> run reports stopped, context stays usable
    stop = ctypes.c_int(1)
    assert ctx.run(stop_flag=stop) is False
    stop.value = 0
    assert ctx.run(stop_flag=stop) is True
    # audio-only context refuses model calls
    a = native.Context(None, 0)
    fid = a.add_f32_22k(np.zeros(1000, np.float32))
    with pytest.raises(native.NativeError):
        a.run()
    a.close()
    with pytest.raises(native.NativeError):
        native.Context(b"SSWBLOB1" + b"\x00" * 8, 0)                 # no tensors in the blob


def test_corrupt_weights_blob_is_an_error_not_a_fault(native, blob):
    """Entry table edits (sizes, offsets, dtypes, names, counts, truncation): ss_create reports SS_ERR_FORMAT; it never reads outside
    the buffer.  (An edit that leaves a well-formed table simply gives other weights.)"""
    import struct
    rng = np.random.default_rng(5)
    n_entries = struct.unpack_from("<I", blob, 8)[0]
    errors = 0
    for trial in range(24):
        b = bytearray(blob)
        ent = 16 + 152 * int(rng.integers(0, n_entries))
        kind = trial % 6
        if kind == 0: struct.pack_into("<Q", b, ent + 136, (1 << 64) - 8)            # offset that wraps with nbytes
        elif kind == 1: struct.pack_into("<Q", b, ent + 144, 1 << 40)                # nbytes far past the end
        elif kind == 2: struct.pack_into("<I", b, ent + 96, 7)                       # unknown dtype
        elif kind == 3: b[ent:ent + 4] = b"zzzz"                                     # a tensor goes missing under its name
        elif kind == 4: struct.pack_into("<I", b, 8, 0xffffffff)                     # entry count
        else: b = b[:int(rng.integers(16, 16 + 152 * n_entries))]                    # truncated inside the table
        try:
            native.Context(bytes(b), 0).close()
        except native.NativeError as e:
            errors += 1
            assert "weights blob" in str(e) or "tensor" in str(e), str(e)
    assert errors == 24


def test_progress_callback(ctx, c1):
    ctx.reset()
    ctx.set_chunk(32)
    fid = ctx.add_f32_22k(c1["sig"])
    seen = []
    assert ctx.run(progress=lambda d, t: seen.append((d, t)))
    ctx.set_chunk(64)
    assert seen == [(32, 105), (64, 105), (96, 105), (105, 105)]


def test_progress_values_do_not_depend_on_the_pass_size(native, blob, c1):
    """worker.py:82-84 emits after every batch of settings.prediction_batch_size = 32 windows.  The library reports that sequence of
    values whatever the pass size (one pass of 105 windows, passes of 40, of 7): a batch that straddles two passes is reported with
    the later one, the last value is the total; the logits do not depend on any of it."""
    want = [(32, 105), (64, 105), (96, 105), (105, 105)]
    logits = []
    for chunk in (1024, 40, 7):
        c = native.Context(blob, 0, precision="f16x2", chunk=chunk)
        fid = c.add_f32_22k(c1["sig"])
        seen = []
        assert c.run(progress=lambda d, t: seen.append((d, t)))
        assert seen == want, (chunk, seen)
        logits.append(c.window_logits(fid))
        c.close()
    assert np.array_equal(logits[0], logits[1]) and np.array_equal(logits[0], logits[2])


def test_tracked_run_and_poll(native, blob, c1):
    """ss_run_begin_tracked + ss_run_poll + ss_run_end == ss_run with a progress callback: poll without blocking returns at once
    with whatever has completed (possibly nothing), a blocking poll delivers the rest, values never repeat; outside a run it is
    an error; an untracked run has nothing to report."""
    c = native.Context(blob, 0, precision="f16x2", chunk=16)
    fid = c.add_f32_22k(c1["sig"])
    with pytest.raises(native.NativeError) as e:
        c.run_poll(lambda d, t: None)
    assert e.value.code == 4                                        # SS_ERR_STATE
    seen = []
    c.run_begin(0.1, 0.5, track=True)
    c.run_poll(lambda d, t: seen.append(d), block=False)            # may or may not have anything yet
    n_early = len(seen)
    c.run_poll(lambda d, t: seen.append(d), block=True)
    c.run_poll(lambda d, t: seen.append(d), block=True)             # nothing left: no repeats
    c.run_end()
    assert seen == [32, 64, 96, 105] and 0 <= n_early <= 4
    regs = c.regions(fid)
    c.reset(); fid = c.add_f32_22k(c1["sig"])
    seen2 = []
    c.run_begin(0.1, 0.5)                                           # untracked
    c.run_poll(lambda d, t: seen2.append(d), block=True)
    c.run_end()
    assert seen2 == [] and c.regions(fid) == regs
    c.close()


def test_ingest_wav_batch_upload(native, blob):
    """ss_upload_wav_batch_async (header walk + asynchronous host -> device copies on the copy stream) followed by
    ss_add_pcm_batch_device, which waits for the copies on the device: the signals equal, bit for bit, those of ss_add_pcm of the same
    files; page-locked and ordinary host memory; ragged lengths; a file that is not a WAV refuses the whole batch before anything is
    enqueued; a staging buffer that is too small is SS_ERR_CAPACITY; uploads are allowed while a run is in flight."""
    from softspoken_amd import synth
    durs = [2.5, 0.3, 7.0, 1.0]
    pcms = [synth.to_pcm16(synth.synth_audio(700 + k, d, 16000, 1, with_silence=False)) for k, d in enumerate(durs)]
    wavs = [np.frombuffer(synth.wav_bytes(p, 16000), dtype=np.uint8).copy() for p in pcms]
    c = native.Context(blob, 0, precision="f16x2")
    ref = []
    for p in pcms:
        c.reset()
        fid = c.add_pcm(p, native.PCM_S16, 16000, 1, len(p))
        ref.append(c.read_signal(fid))
    total = sum(p.nbytes for p in pcms)
    dev = c.device_alloc(total + 64)
    pinned = []
    for w in wavs:
        h = c.host_alloc(w.nbytes); h[:] = w; pinned.append(h)
    for files in (pinned, wavs):
        c.reset()
        infos = c.upload_wav_batch_async(files, dev, total + 64)
        assert [i.frames for i in infos] == [len(p) for p in pcms] and all(i.format == native.PCM_S16 and i.sample_rate == 16000 for i in infos)
        first = c.add_pcm_batch_device(dev, native.PCM_S16, 16000, 1, [i.frames for i in infos])
        for k in range(len(pcms)):
            assert np.array_equal(c.read_signal(first + k), ref[k])
    with pytest.raises(native.NativeError) as e:
        c.upload_wav_batch_async(pinned, dev, total - 2)
    assert e.value.code == native.SS_ERR_CAPACITY
    bad = np.frombuffer(b"RIFF....WAVEjunkjunkjunk", dtype=np.uint8).copy()
    with pytest.raises(native.NativeError) as e:
        c.upload_wav_batch_async([pinned[0], bad, pinned[1]], dev, total + 64)
    assert e.value.code == 3 and "file 1" in str(e.value)          # SS_ERR_FORMAT, nothing enqueued
    c.reset()
    first = c.add_pcm_batch_device(dev, native.PCM_S16, 16000, 1, [len(p) for p in pcms])
    c.run_begin(0.1, 0.5)
    dev2 = c.device_alloc(total + 64)
    c.upload_wav_batch_async(pinned, dev2, total + 64)             # while the run is in flight
    with pytest.raises(native.NativeError):
        c.add_pcm_batch_device(dev2, native.PCM_S16, 16000, 1, [len(p) for p in pcms])   # ... but the arena is the run's
    c.run_end()
    regs = [c.regions(first + k) for k in range(len(pcms))]
    c.reset()
    first = c.add_pcm_batch_device(dev2, native.PCM_S16, 16000, 1, [len(p) for p in pcms])
    assert c.run()
    assert [c.regions(first + k) for k in range(len(pcms))] == regs
    c.upload_wait()
    for h in pinned:
        c.host_free(h)
    c.device_free(dev); c.device_free(dev2)
    c.close()


def test_bf16_mode(ctx_bf16, c1, gold):
    """bf16 activations/weights with fp32 accumulation: throughput mode (BASELINE config 2), not the parity
    mode.  Stated tolerance: logits within 0.15 absolute (logit std 0.53); every averaged bin whose
    reference value is more than 0.05 away from the threshold lands on the same side (the synthetic head
    is centred on the threshold, so many bins sit within a few 1e-3 of it); >= 97 % of all bins agree;
    same number of regions with boundaries within 0.1 s."""
    gl = gold["c1_logits"]
    ctx_bf16.reset()
    fid = ctx_bf16.add_f32_22k(c1["sig"])
    assert ctx_bf16.run()
    lg = ctx_bf16.window_logits(fid)
    d = np.abs(lg - gl["logits"])
    assert d.max() < 0.15 and d.mean() < 0.02
    avg, _ = ctx_bf16.avg(fid)
    same = (avg > 0.1) == (gl["avg"] > 0.1)
    assert same[np.abs(gl["avg"] - 0.1) > 0.05].all()
    assert same.mean() >= 0.97
    assert np.abs(avg - gl["avg"]).max() < 0.1
    regs = ctx_bf16.regions(fid)
    ref = gl["regions"]
    assert len(regs) == len(ref)
    assert np.abs(np.array(regs) - ref).max() <= 0.1


_ALT_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from softspoken_amd import synth, native, checkpoint
from oracle import oracle_np as O
g = np.load({gold!r})
pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
sig, _, _ = O.load_audio_from_bytes(synth.wav_bytes(pcm, 16000))
ctx = native.Context(checkpoint.pack_state_dict(synth.make_state_dict(0)), 0, precision={mode!r})
fid = ctx.add_f32_22k(sig)
assert ctx.run()
d = np.abs(ctx.window_logits(fid) - g["logits"])
print("MAXDIFF", float(d.max()), "REGIONS", len(ctx.regions(fid)))
"""


@pytest.mark.parametrize("env,mode,tol", [({"SOFTSPOKEN_CONV4": "0"}, "bf16", 0.15), ({"SOFTSPOKEN_CONV4": "0", "SOFTSPOKEN_NW": "4"}, "bf16", 0.15),
                                          ({"SOFTSPOKEN_NW": "4"}, "fp32", 1e-4), ({"SOFTSPOKEN_UPS32": "0"}, "fp32", 1e-4), ({"SOFTSPOKEN_UPS32_NT": "2"}, "fp32", 1e-4),
                                          ({"SOFTSPOKEN_UPS32_NT": "3"}, "fp32", 1e-4),
                                          ({"SOFTSPOKEN_RPROJ": "0"}, "bf16", 0.15), ({"SOFTSPOKEN_RPROJ": "0", "SOFTSPOKEN_PF2": "0"}, "bf16", 0.15),
                                          ({"SOFTSPOKEN_DUO": "0"}, "f16x2", 1e-4), ({"SOFTSPOKEN_DUO": "2"}, "f16x2", 1e-4), ({"SOFTSPOKEN_DUO_H8": "0"}, "f16x2", 1e-4), ({"SOFTSPOKEN_UPS": "0"}, "f16x2", 1e-4), ({"SOFTSPOKEN_UPSR": "0"}, "f16x2", 1e-4),
                                          ({"SOFTSPOKEN_RING": "0"}, "f16x2", 1e-4), ({"SOFTSPOKEN_RING": "2"}, "f16x2", 1e-4), ({"SOFTSPOKEN_NTB1": "0"}, "f16x2", 1e-4),
                                          ({"SOFTSPOKEN_RPROJ": "0"}, "f16x2", 1e-4), ({"SOFTSPOKEN_RPROJ": "1"}, "f16x2", 1e-4)])
def test_alternate_kernel_structures(env, mode, tol, build_all):
    """Kernel forms the product library does not select but still contains code for: conv2.hip in bf16 (the fp32 path's structure;
    its 4-wave geometry, also in fp32), conv4.hip with the r tensors everywhere / without the two-stage prefetch, and the f16x2
    launches with resident banks as independent 8-wave blocks / as two 8-wave tiles per workgroup (the product: four 4-wave tiles), the
    shared two-slot bank ring off / on for every B launch (the product: A launches, B launches from the 32 x 64 level down), the 96-channel
    blocks' B launches as one 96-channel tile per 8-wave block (the product: three groups over the ring), the f16x2 "projection in B" form for no
    block / for conv2_1 alone (the product: conv2_1 and conv9_1), conv9_1.A / conv6.A, conv7.A and conv8.A in conv4.hip's direct form
    instead of conv4_ups.hip's resident / ring form (the upsampled input half at low resolution, four taps per parity class).  The switches
    exist in the development build of the library only (libsoftspoken_hip_dev.so, -DSS_DEVBUILD) and are read once per process, so
    each case runs in its own interpreter."""
    import os, subprocess, sys
    from softspoken_amd import build as hip_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _ALT_SCRIPT.format(root=root, gold=os.path.join(root, "tests", "golden", "c1_logits.npz"), mode=mode)
    e = dict(os.environ); e.update(env); e["SOFTSPOKEN_LIB"] = hip_build.DEV_LIB
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("MAXDIFF")][0].split()
    assert float(line[1]) < tol and int(line[3]) == 6


_LANES_SCRIPT = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, {root!r})
from softspoken_amd import synth, native, checkpoint
from oracle import oracle_np as O
g = np.load({gold!r})
pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
sig, _, _ = O.load_audio_from_bytes(synth.wav_bytes(pcm, 16000))
ctx = native.Context(checkpoint.pack_state_dict(synth.make_state_dict(0)), 0, precision="f16x2", chunk=24)
fid = ctx.add_f32_22k(sig)
seen = []
assert ctx.run(progress=lambda done, total: seen.append(done))       # 105 windows in five passes
lg = ctx.window_logits(fid)
print("MAXDIFF", float(np.abs(lg - g["logits"]).max()), "REGIONS", len(ctx.regions(fid)), "HASH", hashlib.sha256(lg.tobytes()).hexdigest(),
      "PROGRESS", ",".join(str(int(x)) for x in seen), "WS", ctx.workspace_bytes())
"""


def test_two_lanes_give_the_same_bits(build_all):
    """Development build, SOFTSPOKEN_LANES=2: the passes of a run alternate between two streams and two workspaces (engine.hip run_begin;
    off in the product).  Same logits bit for bit as on one lane, the same per-32-window progress values in the same order, twice the
    workspace."""
    import os, subprocess, sys
    from softspoken_amd import build as hip_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _LANES_SCRIPT.format(root=root, gold=os.path.join(root, "tests", "golden", "c1_logits.npz"))
    outs = []
    for lanes in ("1", "2"):
        e = dict(os.environ); e["SOFTSPOKEN_LANES"] = lanes; e["SOFTSPOKEN_LIB"] = hip_build.DEV_LIB
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("MAXDIFF")][0].split())
    one, two = outs
    assert float(two[1]) < 1e-4 and int(two[3]) == 6
    assert one[5] == two[5] and one[7] == two[7]                   # logits hash, progress sequence
    assert int(two[9]) == 2 * int(one[9]) > 0


_REPEAT_SCRIPT = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, {root!r})
from softspoken_amd import synth, native, checkpoint
x = synth.to_pcm16(synth.synth_audio(77, 600.0, 16000, 1))
ctx = native.Context(checkpoint.pack_state_dict(synth.make_state_dict(0)), 0, bf16=True, chunk={chunk})
for rep in range(2):
    ctx.reset()
    fid = ctx.add_pcm(x, native.PCM_S16, 16000, 1, len(x))
    assert ctx.run()
    print("HASH", hashlib.sha256(np.ascontiguousarray(ctx.window_logits(fid)).tobytes()).hexdigest(), ctx.num_windows(fid))
"""


def test_long_recording_same_bits_every_run_and_chunking(build_all):
    """1005 windows in bf16: twice in one process, and in fresh processes with 1024 / 300 / 64 windows per pass (blocks of the conv
    launches then walk 170, 50 or 11 tiles each): the same bits every time."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for chunk in (1024, 300, 64):
        r = subprocess.run([sys.executable, "-c", _REPEAT_SCRIPT.format(root=root, chunk=chunk)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l.split() for l in r.stdout.splitlines() if l.startswith("HASH")]
        assert len(lines) == 2 and lines[0][2] == "1005"
        hashes += [l[1] for l in lines]
    assert len(set(hashes)) == 1, hashes


_JITTER_SCRIPT = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, {root!r})
from softspoken_amd import synth, native, checkpoint
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
x = synth.to_pcm16(synth.synth_audio(78, 600.0, 16000, 1))
sig = synth.synth_audio(7, 40.0, 22050, 1).astype(np.float32).ravel()
starts = (np.arange(33) * 13230).astype(np.int64)
def h(a): return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
out = []
c = native.Context(blob, 0, bf16=True)                    # long recording, one chunk: many tiles per block, claimed tiles
fid = c.add_pcm(x, native.PCM_S16, 16000, 1, len(x)); assert c.run(); out.append(h(c.window_logits(fid))); c.close()
for bf16 in (True, False):                                # 33 windows in chunks of 7: one or two tiles per block, many last stages
    c = native.Context(blob, 0, bf16=bf16, chunk=7)
    fid = c.add_f32_22k(sig); _, m = c.infer_windows(fid, starts); out.append(h(m)); c.close()
c = native.Context(blob, 0, bf16=False)                   # fp32, 120 windows in one chunk
fid = c.add_pcm(x[:16000 * 75], native.PCM_S16, 16000, 1, 16000 * 75); assert c.run(); out.append(h(c.window_logits(fid))); c.close()
c = native.Context(blob, 0, precision="f16x2")            # f16x2: three stages per K chunk, 120 windows in one chunk and 33 in chunks of 7
fid = c.add_pcm(x[:16000 * 75], native.PCM_S16, 16000, 1, 16000 * 75); assert c.run(); out.append(h(c.window_logits(fid))); c.close()
c = native.Context(blob, 0, precision="f16x2", chunk=7)
fid = c.add_f32_22k(sig); _, m = c.infer_windows(fid, starts); out.append(h(m)); c.close()
print("HASHES", " ".join(out))
"""


def test_results_do_not_depend_on_wave_timing(build_all):
    """The conv kernels (conv4.hip in bf16 and f16x2, conv2.hip in fp32) synchronise their waves with LDS-only barriers.  In the development
    build of the library (-DSS_DEVBUILD) ConvArgs::dbg bit 10 makes chosen waves sleep about a microsecond at every synchronisation point of a stage
    (a rotating wave, wave 0 only, all but wave 0, the odd waves).  A missing barrier then shows as changed bits -- the flatten
    launch's last-stage race was found that way.  dbg = 0 is the product library itself."""
    import os, subprocess, sys
    from softspoken_amd import build as hip_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.exists(hip_build.JITTER_LIB)
    seen = {}
    for dbg in (0, 1024, 1024 + 2048, 1024 + 4096, 1024 + 6144):
        e = dict(os.environ); e["SOFTSPOKEN_DBG"] = str(dbg)
        if dbg:
            e["SOFTSPOKEN_LIB"] = hip_build.JITTER_LIB
        r = subprocess.run([sys.executable, "-c", _JITTER_SCRIPT.format(root=root)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        seen[dbg] = [l for l in r.stdout.splitlines() if l.startswith("HASHES")][0]
    assert len(set(seen.values())) == 1, seen


def test_regions_from_bin_masks_equal_the_host_series(native, blob):
    """ss_run brings two bits per bin back (covered, average > threshold) and finds the runs in them; ss_find_regions does it the
    reference's way on the compacted (averages, bin index) series.  Ragged files (bin offsets off word boundaries), several thresholds
    and break durations, including thresholds that make every bin / no bin a detection."""
    from softspoken_amd import synth
    durs = [0.4, 3.0, 7.3, 12.1, 1.7, 20.0, 5.55, 9.0]
    files = [synth.to_pcm16(synth.synth_audio(900 + k, d, 16000, 1)) for k, d in enumerate(durs)]
    c = native.Context(blob, 0, bf16=False)
    for thr, brk in ((0.1, 0.5), (0.0, 0.05), (0.35, 1.0), (-1e9, 0.5), (1e9, 0.5), (0.2, -1.0)):
        c.reset()
        fids = [c.add_pcm(x, native.PCM_S16, 16000, 1, len(x)) for x in files]
        assert c.run(thr, brk)
        counts, reg = c.regions_batch(fids[0], len(fids))
        at = 0
        for f, n in zip(fids, counts.tolist()):
            avg, idx = c.avg(f)
            want = native.find_regions(avg, idx, threshold=thr, break_s=brk)
            assert c.regions(f) == want and [tuple(r) for r in reg[at:at + n].tolist()] == want, (thr, brk, f)
            at += n
        if thr == -1e9:
            assert all(len(c.regions(f)) == 1 for f in fids)
        if thr == 1e9:
            assert counts.sum() == 0
    c.close()


def test_run_in_two_halves_and_two_contexts(native, blob):
    """ss_run_begin + ss_run_end == ss_run; between the halves the context refuses other work; two contexts alternating on one
    device (job k's host half while job k+1's kernels run) give what one context gives job by job."""
    from softspoken_amd import synth
    jobs = [[synth.to_pcm16(synth.synth_audio(500 + 10 * j + k, 6.0 + 3 * k, 16000, 1)) for k in range(4)] for j in range(5)]
    def add(c, job):
        return [c.add_pcm(x, native.PCM_S16, 16000, 1, len(x)) for x in job]
    def results(c, fids):
        return [(c.regions(f), c.avg(f)[0].tobytes(), c.window_logits(f).tobytes()) for f in fids]
    one = native.Context(blob, 0, bf16=True)
    want = []
    for job in jobs:
        one.reset(); fids = add(one, job); assert one.run(); want.append(results(one, fids))
    # halves on one context + the state rules
    one.reset(); fids = add(one, jobs[0])
    with pytest.raises(native.NativeError):
        one.run_end()                                    # nothing in flight
    one.run_begin()
    for call in (one.reset, lambda: add(one, jobs[1]), one.run_begin, lambda: one.infer_windows(fids[0], np.zeros(1, np.int64))):
        with pytest.raises(native.NativeError):
            call()
    one.run_end()
    assert results(one, fids) == want[0]
    # two contexts, jobs in flight on both
    from softspoken_amd import pipeline
    two = [one, native.Context(blob, 0, bf16=True)]
    def submit(c, job):
        c.reset(); fids = add(c, job); c.run_begin()
        return fids
    def end(c, job, fids):
        c.run_end()
    def regions_of(c, job, fids):
        return [c.regions(f) for f in fids]
    want_regions = [[r for r, _, _ in w] for w in want]
    assert list(pipeline.run_jobs(two, jobs, submit, end, regions_of)) == want_regions
    # one context: the ended job's regions are read while the next job is already in flight
    assert list(pipeline.run_jobs(two[:1], jobs, submit, end, regions_of)) == want_regions
    one.reset(); fids = add(one, jobs[2]); one.run_begin(); one.run_end()
    assert results(one, fids) == want[2]                # averages and logits: readable until ...
    one.reset()
    assert [one.regions(f) for f in fids] == want_regions[2] and one.num_windows(fids[0]) > 0
    with pytest.raises(native.NativeError):
        one.avg(fids[0])                                 # ... the arena is reset
    fids2 = add(one, jobs[3]); one.run_begin()
    assert [one.regions(f) for f in fids] == want_regions[2]      # still job 2's, job 3 in flight
    with pytest.raises(native.NativeError):
        one.window_logits(fids[0])
    one.run_end()
    assert results(one, fids2) == want[3]
    for c in two:
        c.close()


def test_bf16_and_fp32_agree_on_a_long_recording(native, blob):
    """Product-level check on a 10-minute recording (1005 windows): the bf16 throughput mode finds the regions the fp32 parity mode
    finds -- same count, every boundary within two bins (3/256 s each) -- except where the averaged score sits on the threshold."""
    from softspoken_amd import synth
    x = synth.to_pcm16(synth.synth_audio(4242, 600.0, 16000, 1))
    res = {}
    for bf16 in (False, True):
        c = native.Context(blob, 0, bf16=bf16)
        fid = c.add_pcm(x, native.PCM_S16, 16000, 1, len(x))
        assert c.run()
        res[bf16] = (c.regions(fid), c.avg(fid))
        c.close()
    (r32, (a32, i32)), (r16, (a16, i16)) = res[False], res[True]
    assert np.array_equal(i32, i16)
    margin = np.abs(a32 - 0.1)
    assert np.mean(np.abs(a16 - a32)) < 0.05
    assert np.all((a16 > 0.1)[margin > 0.1] == (a32 > 0.1)[margin > 0.1])
    assert len(r32) > 20 and abs(len(r16) - len(r32)) <= max(2, len(r32) // 20)
    s32 = np.array([s for s, _ in r32]); s16 = np.array([s for s, _ in r16])
    near = np.array([np.abs(s16 - s).min() for s in s32])
    assert np.mean(near <= 2 * 3.0 / 256 + 1e-9) > 0.9


_WS_FAIL_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from softspoken_amd import synth, native, checkpoint
from oracle import oracle_np as O
pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
sig, _, _ = O.load_audio_from_bytes(synth.wav_bytes(pcm, 16000))
starts = O.plan_windows(60.0)
c = native.Context(checkpoint.pack_state_dict(synth.make_state_dict(0)), 0, precision="fp32", chunk=8)
fid = c.add_f32_22k(sig)
_, want = c.infer_windows(fid, starts[:8])
assert c.workspace_bytes() > 0
for nth in (0, 1, 2):
    c.set_chunk(16 + 8 * nth)                                 # the next call has to grow the workspace
    c.debug_fail_workspace_alloc(nth)
    try:
        c.infer_windows(fid, starts[:16 + 8 * nth])
        raise SystemExit("no error")
    except native.NativeError as e:
        assert e.code == 6 and "workspace" in str(e), e
    assert c.workspace_bytes() == 0
    _, got = c.infer_windows(fid, starts[:16 + 8 * nth])     # allocates afresh
    assert c.workspace_bytes() > 0 and np.array_equal(got[:8], want)
c.debug_fail_workspace_alloc(-1)
c.close()
print("WS_OK")
"""


def test_workspace_growth_failure_leaves_a_usable_context(build_all):
    """ss_debug_fail_workspace_alloc (development build only): every allocation of the activation workspace fails in turn.  The call
    reports SS_ERR_NOMEM, the context holds no workspace afterwards (no stale size over freed tensors), and the next call allocates
    afresh and gives the bits of an undisturbed context."""
    import os, subprocess, sys
    from softspoken_amd import build as hip_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e["SOFTSPOKEN_LIB"] = hip_build.DEV_LIB
    r = subprocess.run([sys.executable, "-c", _WS_FAIL_SCRIPT.format(root=root)], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WS_OK" in r.stdout, r.stderr[-2000:]


def test_f16x2_mode_meets_the_parity_bar(native, blob, c1, gold):
    """f16x2: fp32 operands as two f16 halves on the f16 matrix cores (three products per term, fp32 accumulation).  Same bar as
    the fp32 mode: logits, the spec head and the averages within 1e-4 of the fixtures made by the reference's classes; regions
    and CSV identical; ragged batches and chunkings bit-identical."""
    gl, gy = gold["c1_logits"], gold["c1_layers"]
    c = native.Context(blob, 0, precision="f16x2")
    fid = c.add_f32_22k(c1["sig"])
    spec, mask = c.infer_windows(fid, c1["starts"])
    assert np.isfinite(mask).all() and np.abs(mask - gl["logits"]).max() < TOL_FP32
    spec, mask2 = c.infer_windows(fid, c1["starts"][gy["window_index"]], want_spec=True)
    assert np.abs(mask2 - gy["mask"]).max() < TOL_FP32
    assert np.abs(spec[:, :, 64, :] - gy["spec_row64"]).max() < TOL_FP32
    for b in range(2):
        assert abs(spec[b].astype(np.float64).mean() - gy["spec_stats"][b, 0]) < 1e-5
    assert c.run(0.1, 0.5)
    avg, idx = c.avg(fid)
    assert np.abs(avg - gl["avg"]).max() < TOL_FP32
    regs = c.regions(fid)
    assert regs == [tuple(r) for r in gl["regions"].tolist()]
    assert O.CSV_HEADER + "\n" + native.format_csv_rows("/data/site a", "c1_seed1001.wav", regs, 1) == str(gl["csv"])
    c.close()
    outs = []
    for chunk in (1, 7, 64):
        c = native.Context(blob, 0, precision="f16x2", chunk=chunk)
        fid = c.add_f32_22k(c1["sig"])
        _, m = c.infer_windows(fid, c1["starts"][:33])
        outs.append(m)
        c.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.array_equal(outs[0], mask[:33])


def test_f16x2_reports_values_outside_the_f16_range(native, sd_np, c1):
    """f16x2 cannot represent a value beyond 65504 or a non-finite one: the conv kernels flag it where the value is split and the call
    reports SS_ERR_RANGE (8) instead of returning scores; a weight without an f16 representation is refused when the context is
    created; the fp32 mode runs the same inputs.  Magnitude alone no longer gets there: BatchNorm gains of 400 and 4e6, which round 2
    refused, are taken out by the power-of-two channel normalisation (weights.hip) and the scores agree with the fp32 mode's."""
    from softspoken_amd import checkpoint
    blob0 = checkpoint.pack_state_dict(sd_np)
    bad = c1["sig"].copy()
    bad[300000] = np.float32("nan")                                                       # one NaN sample: every window over it has NaN features
    c = native.Context(blob0, 0, precision="f16x2")
    fid = c.add_f32_22k(bad)
    hit = [i for i, s0 in enumerate(c1["starts"]) if s0 <= 300000 + 66150 < s0 + 66150]
    with pytest.raises(native.NativeError) as e:
        c.infer_windows(fid, c1["starts"][hit[0]:hit[0] + 2])
    assert e.value.code == 8 == native.SS_ERR_RANGE and "f16 range" in str(e.value)
    with pytest.raises(native.NativeError) as e:
        c.run()
    assert e.value.code == 8
    assert c.features(fid, c1["starts"][:2]).shape == (2, 128, 256)                       # the context stays usable
    _, ok = c.infer_windows(fid, c1["starts"][:4])                                        # windows clear of the NaN still run
    assert np.isfinite(ok).all()
    c.close()
    sd = {k: v.copy() for k, v in sd_np.items()}
    sd["conv2_1.conv2.0.weight"][3, 5, 1, 1] = np.float32("inf")                          # a weight that is not finite: refused at creation
    with pytest.raises(native.NativeError) as e:
        native.Context(checkpoint.pack_state_dict(sd), 0, precision="f16x2")
    assert e.value.code == 8 and "weight" in str(e.value)
    for gain2 in (1.0, 1e4):                                                              # x 400, and x 4e6 on conv2_1's second BatchNorm
        sd = {k: v.copy() for k, v in sd_np.items()}
        for name in ("conv1_1.conv2.1.weight", "conv2_1.conv2.1.weight"):
            sd[name] = sd[name] * np.float32(400.0)
        sd["conv2_1.conv2.1.weight"] = sd["conv2_1.conv2.1.weight"] * np.float32(gain2)
        big = checkpoint.pack_state_dict(sd)
        out = {}
        for prec in ("f16x2", "fp32"):
            c = native.Context(big, 0, precision=prec)
            fid = c.add_f32_22k(c1["sig"])
            _, out[prec] = c.infer_windows(fid, c1["starts"][10:14])
            c.close()
        assert np.isfinite(out["fp32"]).all() and np.isfinite(out["f16x2"]).all()
        assert np.abs(out["f16x2"] - out["fp32"]).max() <= 1e-4 * max(1.0, float(np.abs(out["fp32"]).max()))


def test_f16x2_hostile_scale_checkpoint(native, c1):
    """the checkpoint whose BatchNorm gains put the outputs of conv3_1 ... conv7 near 1e-3 / 1e3 (undone
    downstream; synth.HOSTILE_GAINS), against logits made by the reference's SpecUNet_2D on it (tests/golden/c1_hostile.npz):
    f16x2 and fp32 within 1e-4 on all 105 windows, averages within 1e-4, regions identical."""
    import os
    from softspoken_amd import synth, checkpoint
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_hostile.npz"))
    blob = checkpoint.pack_state_dict(synth.make_state_dict(0, hostile=True))
    for prec in ("f16x2", "fp32"):
        c = native.Context(blob, 0, precision=prec)
        fid = c.add_f32_22k(c1["sig"])
        assert c.run(0.1, 0.5)
        lg = c.window_logits(fid)
        assert np.isfinite(lg).all() and np.abs(lg - g["logits"]).max() < TOL_FP32, (prec, float(np.abs(lg - g["logits"]).max()))
        avg, _ = c.avg(fid)
        assert np.abs(avg - g["avg"]).max() < TOL_FP32
        assert c.regions(fid) == [tuple(r) for r in g["regions"].tolist()]
        c.close()


_NONORM_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from softspoken_amd import synth, native, checkpoint
from oracle import oracle_np as O
pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
sig, _, _ = O.load_audio_from_bytes(synth.wav_bytes(pcm, 16000))
starts = O.plan_windows(60.0)
try:                                                      # hostile gains, no normalisation: a folded weight beyond 65504
    native.Context(checkpoint.pack_state_dict(synth.make_state_dict(0, hostile=True)), 0, precision="f16x2")
    raise SystemExit("hostile checkpoint accepted without the normalisation")
except native.NativeError as e:
    assert e.code == 8 and "weight" in str(e), e
sd = synth.make_state_dict(0)
for name in ("conv1_1.conv2.1.weight", "conv2_1.conv2.1.weight"):
    sd[name] = sd[name] * np.float32(400.0)
c = native.Context(checkpoint.pack_state_dict(sd), 0, precision="f16x2")
fid = c.add_f32_22k(sig)
try:                                                      # gains of 400, no normalisation: activations of conv2_1 beyond 65504
    c.infer_windows(fid, starts[10:14])
    raise SystemExit("no range error")
except native.NativeError as e:
    assert e.code == 8 and "f16 range" in str(e), e
c.close()
g = np.load({gold!r})                                     # the plain checkpoint without the normalisation: round 2's arithmetic
c = native.Context(checkpoint.pack_state_dict(synth.make_state_dict(0)), 0, precision="f16x2")
fid = c.add_f32_22k(sig)
_, m = c.infer_windows(fid, starts)
assert np.abs(m - g["logits"]).max() < 1e-4
print("NONORM_OK")
"""


def test_without_the_channel_normalisation_the_hostile_checkpoints_are_refused(build_all):
    """What the power-of-two channel normalisation buys, shown by switching it off (development build, SOFTSPOKEN_NORM=0): the
    hostile-scale checkpoint is refused at creation (a folded weight beyond 65504), BatchNorm gains of 400 overflow at run time --
    both SS_ERR_RANGE, which the drop-in answers by running the checkpoint in fp32 --, and the plain checkpoint scores as in round 2."""
    import os, subprocess, sys
    from softspoken_amd import build as hip_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e["SOFTSPOKEN_LIB"] = hip_build.DEV_LIB; e["SOFTSPOKEN_NORM"] = "0"
    code = _NONORM_SCRIPT.format(root=root, gold=os.path.join(root, "tests", "golden", "c1_logits.npz"))
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NONORM_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])




_C1S_SCRIPT = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, {root!r})
from softspoken_amd import synth, native, checkpoint
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
sig = synth.synth_audio(7, 40.0, 22050, 1).astype(np.float32).ravel()
starts = (np.arange(33) * 13230).astype(np.int64)
out = []
for chunk in (None, 7):                                   # 33 windows in one pass (the grid's waves get one or two units) and in passes of 7
    c = native.Context(blob, 0, precision="f16x2", chunk=chunk)
    fid = c.add_f32_22k(sig); _, m = c.infer_windows(fid, starts); out.append(m); c.close()
np.save({out!r}, np.stack(out))
"""


def test_conv1_streaming_kernel_variants_agree(build_all, tmp_path):
    """conv1_1 in f16x2 runs as a row-streaming kernel (csrc/conv1s.hip; the product: two interleaved 16-pixel tiles per strip row on
    v_mfma_f32_16x16x32_f16).  Its work units (window, band of rows, strip) are independent and its arithmetic does not depend on how they
    are cut: 16- and 32-row units, the instantiation with the per-pair range test and the one without (range proven from the weights)
    give the same BITS, whatever the pass size; the round's first version of the kernel (one 32-column tile on v_mfma_f32_32x32x16_f16,
    dev build, SOFTSPOKEN_C1S_FORM=32: same bits for both unit sizes) and conv4.hip's tile form of the same block (SOFTSPOKEN_C1S=0) give
    the same scores up to the summation order of fp32."""
    import os, subprocess, sys
    from softspoken_amd import build as hip_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env in (("product", {}), ("rows16", {"SOFTSPOKEN_C1S_ROWS": "16"}), ("track", {"SOFTSPOKEN_C1S_TRACK": "1"}), ("tiles", {"SOFTSPOKEN_C1S": "0"}),
                     ("form32", {"SOFTSPOKEN_C1S_FORM": "32"}), ("form32_rows16", {"SOFTSPOKEN_C1S_FORM": "32", "SOFTSPOKEN_C1S_ROWS": "16"})):
        e = dict(os.environ); e.update(env)
        if env:
            e["SOFTSPOKEN_LIB"] = hip_build.DEV_LIB
        out = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([sys.executable, "-c", _C1S_SCRIPT.format(root=root, out=out)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(out)
    assert np.array_equal(res["product"][0], res["product"][1])                  # pass size
    assert np.array_equal(res["product"], res["rows16"]) and np.array_equal(res["product"], res["track"])
    assert np.isfinite(res["tiles"]).all() and np.abs(res["tiles"] - res["product"]).max() < 2e-5
    # the 32-column form of the streaming kernel (v_mfma_f32_32x32x16_f16; the product runs two interleaved 16-pixel tiles on 16x16x32)
    assert np.array_equal(res["form32"], res["form32_rows16"]) and np.abs(res["form32"] - res["product"]).max() < 2e-5
