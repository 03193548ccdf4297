"""BASELINE configs at their full sizes on a real MI355X, through the C ABI:

  C3  one full 10-minute recording against the fixture the REFERENCE's own classes produced for it (tests/golden/make_golden.py
      make_c3: all 1005 windows, every averaged bin, regions, CSV) in both parity modes (fp32 and f16x2); the 100-recording job
      through its size-independent properties (a recording gives the same bits wherever it sits in the job and whatever shares its
      passes with it).
  C2  the 256-clip job on the bf16 context (the kernels the throughput number is quoted on) and on the f16x2 context.
  C5  a 10-minute 48 kHz stereo recording: decode + mixdown + resample on the device bit for bit the C oracle's, features of windows
      along the whole file against the torch oracle.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

TOL = 1e-4           # BASELINE.json north_star: "within 1e-4 fp32"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def native(build_all):
    from softspoken_amd import native
    return native


@pytest.fixture(scope="module")
def c3_pcm():
    from softspoken_amd import synth
    return synth.to_pcm16(synth.synth_audio(3000, 600.0, 16000, 1))


def _h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:20]


@pytest.mark.parametrize("precision", ["fp32", "f16x2"])
def test_c3_recording_against_the_reference_fixture(native, blob, c3_pcm, precision):
    """Every window, every averaged bin, the regions and the CSV of one 10-minute recording (1005 windows, 51 661 bins)."""
    g = np.load(os.path.join(GOLDEN, "c3_recording.npz"))
    c = native.Context(blob, 0, precision=precision)
    fid = c.add_pcm(c3_pcm, native.PCM_S16, 16000, 1, len(c3_pcm))
    assert c.run(0.1, 0.5)
    sig = c.read_signal(fid)
    assert abs(float(np.sum(sig.astype(np.float64))) - float(g["sig_sum"])) < 1e-9 * len(sig)      # the generator has not drifted
    lg = c.window_logits(fid)
    assert lg.shape == g["logits"].shape == (1005, 1, 256)
    assert np.isfinite(lg).all()
    d = np.abs(lg - g["logits"])
    assert d.max() < TOL, (precision, float(d.max()))
    avg, idx = c.avg(fid)
    assert len(avg) == len(g["avg"]) and np.array_equal(idx, np.arange(len(avg)))
    assert np.abs(avg - g["avg"]).max() < TOL
    # device averaging is exact given the device logits (float64 sums of float32): the oracle's loop over them gives the same doubles
    a2, i2 = O.average_overlapping(lg, int(g["n_padded"]) / 22050)
    assert np.array_equal(a2, avg) and np.array_equal(i2, idx)
    regs = c.regions(fid)
    assert regs == native.find_regions(avg, idx, 0.1, 0.5)              # mask walk == the reference's walk over the series
    # the table is the reference's unless a bin's average sits within the score tolerance of the threshold on the other side of it
    flipped = np.nonzero((avg > 0.1) != (g["avg"] > 0.1))[0]
    assert np.all(np.abs(g["avg"][flipped] - 0.1) < TOL)
    ref_regs = [tuple(r) for r in g["regions"].tolist()]
    if len(flipped) == 0:
        assert regs == ref_regs
        text = O.CSV_HEADER + "\n" + native.format_csv_rows(os.path.dirname(str(g["file_key"])), os.path.basename(str(g["file_key"])), regs, 1)
        assert text == str(g["csv"])
    else:                                                                 # (not expected with the committed fixture; kept honest)
        assert abs(len(regs) - len(ref_regs)) <= 2 * len(flipped)
    c.close()


@pytest.mark.parametrize("precision,n_files", [("f16x2", 100), ("fp32", 24)])
def test_c3_job_repeated_recordings_and_batching(native, blob, precision, n_files):
    """The 100-recording job (100 500 windows; 24 recordings in the slower exact-fp32 mode): the job repeats 4 distinct recordings,
    so file k and file k + 4 must give the same bits although they sit in different passes next to different neighbours, and each
    must give the bits of the recording processed alone (windows are independent; every reduction has a fixed order)."""
    from softspoken_amd import synth
    base = [synth.to_pcm16(synth.synth_audio(3000 + k, 600.0, 16000, 1)) for k in range(4)]
    files = [base[k % 4] for k in range(n_files)]
    frames = np.array([len(f) for f in files], dtype=np.int64)
    pcm = np.concatenate(files)
    c = native.Context(blob, 0, precision=precision)
    d = c.device_alloc(pcm.nbytes)
    c.device_upload(d, pcm)
    first = c.add_pcm_batch_device(d, native.PCM_S16, 16000, 1, frames, host_copy=pcm)
    assert c.run()
    counts, reg = c.regions_batch(first, n_files)
    assert all(c.num_windows(first + k) == 1005 for k in range(n_files))
    assert int(counts.sum()) == len(reg) > 0
    offs = np.concatenate([[0], np.cumsum(counts)])
    tables = [reg[offs[k]:offs[k + 1]].tobytes() for k in range(n_files)]
    hashes = [_h(c.window_logits(first + k)) for k in range(n_files)]
    for k in range(4, n_files):
        assert hashes[k] == hashes[k % 4] and tables[k] == tables[k % 4], k
    avg_hash = [_h(c.avg(first + k)[0]) for k in range(4)]
    for k in range(4):                                   # each distinct recording alone, through the host-buffer entry point
        c.reset()
        fid = c.add_pcm(base[k], native.PCM_S16, 16000, 1, len(base[k]))
        assert c.run()
        assert _h(c.window_logits(fid)) == hashes[k]
        assert _h(c.avg(fid)[0]) == avg_hash[k]
        assert np.array(c.regions(fid), dtype=np.float64).reshape(-1, 2).tobytes() == tables[k]
    c.device_free(d)
    c.close()


@pytest.mark.parametrize("precision", ["bf16", "f16x2"])
def test_c2_job_on_the_throughput_kernels(native, blob, precision):
    """BASELINE config 2 (256 x 3 s clips, 2560 windows) on the bf16 context -- the kernels bench.py's C2 line times -- and on the
    f16x2 context: one job over all clips == each clip alone, bit for bit; f16x2 also within 1e-4 of the torch oracle."""
    from softspoken_amd import synth
    clips = [synth.to_pcm16(synth.synth_audio(2000 + k, 3.0, 16000, 1, with_silence=False)) for k in range(256)]
    frames = np.array([len(p) for p in clips], dtype=np.int64)
    pcm = np.concatenate(clips)
    c = native.Context(blob, 0, precision=precision)
    d = c.device_alloc(pcm.nbytes)
    c.device_upload(d, pcm)
    first = c.add_pcm_batch_device(d, native.PCM_S16, 16000, 1, frames, host_copy=pcm)
    assert c.run()
    assert all(c.num_windows(first + k) == 10 for k in range(256))
    joint = [c.window_logits(first + k) for k in range(256)]
    joint_regs = [c.regions(first + k) for k in range(256)]
    assert np.isfinite(np.stack(joint)).all()
    for k in (0, 17, 128, 255):
        c.reset()
        i = c.add_pcm(clips[k], native.PCM_S16, 16000, 1, len(clips[k]))
        assert c.run()
        assert np.array_equal(c.window_logits(i), joint[k]) and c.regions(i) == joint_regs[k]
    if precision == "f16x2":
        from softspoken_amd import synth as S
        sd = S.to_torch_state_dict(S.make_state_dict(0))
        for k in (3, 200):
            x = clips[k].astype(np.float32) / np.float32(32768.0)
            ref = O.detect_signal(sd, O.resample(x, 16000), 3.0)
            assert np.abs(joint[k] - ref["window_logits"]).max() < TOL
    c.device_free(d)
    c.close()


def test_c5_48k_stereo_recording_front_end(native, blob, sd_torch):
    """BASELINE config 5's source format at full length: 10 min of 48 kHz stereo PCM16.  Mixdown-then-resample (voice_activity.py:61-67)
    on the device equals the plain-C oracle bit for bit over all 13.23 M samples; mel features of windows along the file (first,
    last, around the middle) are within 1e-5 of the torch oracle's, apart from single steps of the float32 grid of log10(x + 1)."""
    from softspoken_amd import synth
    from oracle import oracle_c
    x2 = synth.to_pcm16(synth.synth_audio(5000, 60.0, 48000, 2, with_silence=False))        # (frames, 2) int16
    x = np.ascontiguousarray(np.concatenate([x2] * 10))                                       # 10 min
    frames = x.shape[0]
    c = native.Context(blob, 0, precision="fp32")
    fid = c.add_pcm(x, native.PCM_S16, 48000, 2, frames)
    dev = c.read_signal(fid)
    ref = oracle_c.decode_resample(x.view(np.uint8).reshape(-1), native.PCM_S16, 2, frames, 48000)
    assert len(dev) == len(ref) == 13230000
    assert np.array_equal(dev, ref)
    starts = native.plan_windows(600.0)
    assert len(starts) == 1005
    pick = np.array([0, 5, 400, 401, 777, 1000, 1004])
    feats = c.features(fid, starts[pick])
    padded = O.pad_3s(ref)
    xw = torch.stack([torch.from_numpy(padded[s:s + 66150]) for s in starts[pick]])
    want = O.mel_features(xw, sd_torch["mel_spectrogram.spectrogram.window"], sd_torch["mel_spectrogram.mel_scale.fb"]).numpy()
    d = np.abs(feats - want)
    assert (d > 1e-5).sum() <= 8 and d.max() < 3e-4
    # all windows of the file through the front-end alone (the C5 bench leg): same values as the picked ones, no fault at full size
    allf = c.features(fid, starts)
    assert np.array_equal(allf[pick], feats) and np.isfinite(allf).all()
    c.close()


def test_run_from_logits_is_the_tail_of_run(native, blob, c3_pcm):
    """ss_run_from_logits (the owner's half of a recording sharded by window ranges): fed the per-window logits of a normal run it
    gives the same averages, bit for bit, and the same regions -- on the model context and on an audio-only one."""
    c = native.Context(blob, 0, precision="f16x2")
    x = c3_pcm[: 16000 * 95]
    fid = c.add_pcm(x, native.PCM_S16, 16000, 1, len(x))
    assert c.run(0.1, 0.5)
    lg, (avg, idx), regs = c.window_logits(fid), c.avg(fid), c.regions(fid)
    assert len(regs) > 3
    for ctx2 in (c, native.Context(None, 0)):
        ctx2.reset()
        f2 = ctx2.add_pcm(x, native.PCM_S16, 16000, 1, len(x))
        ctx2.run_from_logits(lg, 0.1, 0.5)
        a2, i2 = ctx2.avg(f2)
        assert np.array_equal(a2, avg) and np.array_equal(i2, idx) and ctx2.regions(f2) == regs
        assert np.array_equal(ctx2.window_logits(f2), lg)
        with pytest.raises(native.NativeError):
            ctx2.run_from_logits(lg[:-1], 0.1, 0.5)                 # not the plan's window count
    c.close()


_SHARD_SCRIPT = r"""
import os, sys, json, numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from softspoken_amd import synth, native, checkpoint, parallel
dist.init_process_group("gloo")                      # two ranks share the one card of the test box; the exchange itself is backend-agnostic
rank = dist.get_rank()
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
x = synth.to_pcm16(synth.synth_audio(3001, 240.0, 16000, 1))
ctx = native.Context(blob, 0, precision="f16x2")
regs = parallel.detect_recording_sharded(ctx, x, native.PCM_S16, 16000, 1, len(x))
if rank == 0:
    ctx.reset(); fid = ctx.add_pcm(x, native.PCM_S16, 16000, 1, len(x)); assert ctx.run()
    serial = ctx.regions(fid)
    print("SHARDED", json.dumps(dict(equal=(regs == serial), n=len(serial), windows=int(ctx.num_windows(fid)))))
else:
    assert regs is None
dist.barrier(); dist.destroy_process_group()
"""


def test_one_recording_on_two_ranks_gives_the_serial_table(build_all, tmp_path):
    """a single recording sharded by contiguous window ranges over two ranks (both on this box's one GPU), logits
    gathered to rank 0, which runs the tail of the path: the detection table equals the one-GPU run's, exactly."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "shard.py"
    script.write_text(_SHARD_SCRIPT.format(root=root))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("SHARDED")][0].split(" ", 1)[1])
    assert out["equal"] and out["n"] > 5 and out["windows"] == 405


_FILE_SHARD_SCRIPT = r"""
import os, sys, json, numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from softspoken_amd import synth, parallel
from softspoken_amd.detections import DetectionProject
from root.code.frontend.NNDetector import NNDetector
from root.code.backend.worker import ProcessWorker
from root.code.backend.voice_activity import get_audio_data
dist.init_process_group("gloo")                      # two ranks share the one card of the test box; the exchange itself is backend-agnostic
rank = dist.get_rank()
tmp = {tmp!r}
durs = [75.0, 31.0, 120.0, 9.5, 60.0, 44.0, 3.0]      # ragged: LPT gives rank 0 files 2, 3, 5 and rank 1 files 0, 1, 4, 6
files = [os.path.join(tmp, "site", f"rec_{{k}}.wav") for k in range(len(durs))]
ck = os.path.join(tmp, "model_checkpoint.pth")
if rank == 0:
    os.makedirs(os.path.join(tmp, "site"), exist_ok=True)
    for k, (f, d) in enumerate(zip(files, durs)):
        synth.write_wav(f, synth.to_pcm16(synth.synth_audio(3100 + k, d, 16000, 1)), 16000)
    synth.save_checkpoint(ck, 0, epoch=0)
dist.barrier()
class PM:
    def __init__(self, fl, csv): self.files = fl; self.current_project = {{'detections_file': csv}}
    def get_unprocessed_list(self): return list(self.files)
det = NNDetector(PM(files, os.path.join(tmp, f"unused_{{rank}}.csv")), checkpoint_path=ck)
header_durs = [get_audio_data(f)[0] for f in files]
rows = parallel.run_sharded(files, header_durs, det.detect_files)            # files across ranks, ONE gather of (file, start, end) rows
shards = parallel.shard_files(header_durs, 2)
if rank == 0:
    from softspoken_amd.detections import COLUMN_TYPES
    import pandas as pd
    df = pd.DataFrame(rows, columns=list(COLUMN_TYPES))
    sharded_csv = df.to_csv(index=False)
    csv = os.path.join(tmp, "serial_detections.csv")                         # the serial job: the drop-in's own worker over all files
    pm = PM(files, csv)
    ProcessWorker(det, DetectionProject(pm), det.plan_detection_job()).run()
    serial_csv = open(csv).read()
    print("FILESHARD", json.dumps(dict(equal=(sharded_csv == serial_csv), rows=len(rows), shards=shards,
                                       ids=[r["ID"] for r in rows] == list(range(1, len(rows) + 1)))))
else:
    assert rows is None
dist.barrier(); dist.destroy_process_group()
"""


def test_files_sharded_over_two_ranks_give_the_serial_csv(build_all, tmp_path):
    """SURVEY.md 8(e), first half (BASELINE config 4's shape): files sharded LPT over two ranks (both on this box's one GPU), every rank
    runs NNDetector.detect_files on its shard, ONE gather of detection rows, rank 0 numbers them in file-list order: the numbered rows
    written as CSV equal, byte for byte, what the serial ProcessWorker job writes for the same files (worker.py:107-125)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fileshard.py"
    script.write_text(_FILE_SHARD_SCRIPT.format(root=root, tmp=str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29643", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("FILESHARD")][0].split(" ", 1)[1])
    assert out["equal"] and out["ids"] and out["rows"] > 8 and all(len(s) >= 3 for s in out["shards"])
