"""The reference's own Python surface (root/code/...) driven the way silencer_ui.py / worker.py drive it,
on a real MI355X: same calls, same return types, detections CSV identical to the reference-made golden."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


class _PM:
    """The two things NNDetector / DetectionProject read from the GUI's ProjectManager."""

    def __init__(self, files, detections_file):
        self.files = files
        self.current_project = {'detections_file': detections_file}

    def get_unprocessed_list(self):
        return list(self.files)


@pytest.fixture(scope="module")
def project(tmp_path_factory, c1, build_all):
    from softspoken_amd import synth
    d = tmp_path_factory.mktemp("proj") / "site a"
    d.mkdir()
    wav = d / "c1_seed1001.wav"
    wav.write_bytes(c1["wav"])
    ck = d / "model_checkpoint.pth"
    synth.save_checkpoint(str(ck), 0, epoch=0)
    return dict(dir=str(d), wav=str(wav), ck=str(ck), csv=str(d / "p_detections.csv"))


def test_get_audio_data_and_load_audio(project, c1):
    from root.code.backend.voice_activity import get_audio_data, load_audio
    assert get_audio_data(project["wav"]) == (60.0, 16000)
    data, sr = load_audio(project["wav"])
    assert sr == 22050 and data.dtype == np.float32 and np.array_equal(data, c1["sig"])
    bad = os.path.join(project["dir"], "broken.wav")
    open(bad, "wb").write(b"RIFF....WAVEjunk")
    assert load_audio(bad) == (None, None)                       # reference behaviour on a failed read


def test_model_forward_contract(project, c1, gold, sd_torch):
    from root.code.backend.pytorch_neural_nets import SpecUNet_2D
    gy = gold["c1_layers"]
    m = SpecUNet_2D()
    m.load_state_dict(sd_torch)
    m.eval()
    x = torch.stack([torch.from_numpy(c1["padded"][s:s + 66150]) for s in c1["starts"][gy["window_index"]]])
    spec, mask = m(x)
    assert tuple(spec.shape) == (2, 2, 128, 256) and tuple(mask.shape) == (2, 1, 256) and mask.dtype == torch.float32
    assert np.abs(mask.numpy() - gy["mask"]).max() < 1e-4
    assert np.abs(spec.numpy()[:, :, 64, :] - gy["spec_row64"]).max() < 1e-4
    with pytest.raises(ValueError):
        m(torch.zeros(2, 100))


def test_reference_call_sequence_process_batch(project, c1, gold):
    """worker.py:57-100 spelled out against the drop-in detector: load, pad, batches of 32, average, regions."""
    from root.code.frontend.NNDetector import NNDetector
    from root.code.backend.voice_activity import load_audio
    from root.code.backend import settings
    gl = gold["c1_logits"]
    det = NNDetector(_PM([project["wav"]], project["csv"]), checkpoint_path=project["ck"])
    plan = det.plan_detection_job()
    audio, _ = load_audio(project["wav"])
    pad = settings.vad_resample * 3
    padded = np.zeros(len(audio) + 2 * pad, dtype=audio.dtype)
    padded[pad:pad + len(audio)] = audio
    idxs = plan[project["wav"]]
    preds = []
    for s in range(0, len(idxs), settings.prediction_batch_size):
        speech, mask = det.process_batch(padded, idxs[s:s + settings.prediction_batch_size])
        assert speech.shape[1:] == (2, 128, 256) and mask.shape[1:] == (1, 256) and mask.dtype == np.float32
        preds.append(mask)
    logits = np.vstack(preds)
    assert np.abs(logits - gl["logits"]).max() < 1e-4
    avg = det.average_overlapping_detections({project["wav"]: logits}, len(padded) / settings.vad_resample)
    reg = det.find_speech_regions({project["wav"]: avg}, break_duration=0.5)
    assert [list(r) for r in reg[project["wav"]]] == gl["regions_str"].tolist()


def test_worker_writes_the_reference_csv(project, gold):
    from root.code.frontend.NNDetector import NNDetector
    from root.code.backend.worker import ProcessWorker
    from softspoken_amd.detections import DetectionProject
    gl = gold["c1_logits"]
    if os.path.exists(project["csv"]):
        os.remove(project["csv"])
    pm = _PM([project["wav"]], project["csv"])
    det = NNDetector(pm, checkpoint_path=project["ck"])
    dp = DetectionProject(pm)
    w = ProcessWorker(det, dp, det.plan_detection_job())
    ev = []
    w.signals.fileStarted.connect(lambda f: ev.append(("start", f)))
    w.signals.fileProgressChanged.connect(lambda p: ev.append(("prog", p)))
    w.signals.fileDone.connect(lambda f: ev.append(("done", f)))
    w.signals.overallProgressChanged.connect(lambda p: ev.append(("overall", p)))
    w.signals.finished.connect(lambda: ev.append(("finished",)))
    w.run()
    assert ev[0] == ("start", project["wav"]) and ev[-1] == ("finished",) and ev[-2] == ("overall", 100.0)
    progs = [e[1] for e in ev if e[0] == "prog"]
    assert progs and progs[-1] == 100.0 and progs == sorted(progs)
    text = open(project["csv"]).read()
    assert text == str(gl["csv"]).replace("/data/site a", project["dir"])
    # second run appends with continuing IDs (worker.py:107-111), as the reference does
    dp2 = DetectionProject(pm)
    w2 = ProcessWorker(det, dp2, det.plan_detection_job())
    w2.run()
    lines = open(project["csv"]).read().splitlines()
    assert len(lines) == 1 + 12 and lines[7].startswith("7,")
    # stop before start: nothing processed, finished still emitted
    w3 = ProcessWorker(det, DetectionProject(pm), det.plan_detection_job())
    fin = []
    w3.signals.finished.connect(lambda: fin.append(1))
    w3.stop()
    w3.run()
    assert fin == [1] and len(open(project["csv"]).read().splitlines()) == 13


def test_load_audio_start_reads_the_native_excerpt(project, c1):
    """reference voice_activity.py:44-55: `start` is a position at the 22 050 Hz scale; the excerpt is frames
    [int(start * sr / 22050), + int(3 sr)) of the file at its own rate, and only that excerpt is resampled."""
    from root.code.backend.voice_activity import load_audio
    from softspoken_amd import synth
    sr = 16000
    for start in (0, 22050 * 7 + 333, 22050 * 58):              # the last one runs into the file's end (60 s): a shorter excerpt
        data, rate = load_audio(project["wav"], start=start)
        a = int(start * (sr / 22050))
        b = min(a + 3 * sr, len(c1["pcm"]))
        want, _, _ = O.load_audio_from_bytes(synth.wav_bytes(c1["pcm"][a:b], sr))
        assert rate == 22050 and data.dtype == np.float32
        assert len(data) == len(want) and np.array_equal(data, want), start


def test_process_batch_never_serves_a_stale_signal(project, c1, gold):
    """The signal kept in HBM between process_batch calls is keyed on its whole content and on the arena's reset count: a reused
    buffer with other samples (same address, same length, same probe points), an in-place edit, and a direct model(x) call in
    between must all lead to a fresh upload."""
    from root.code.frontend.NNDetector import NNDetector
    det = NNDetector(_PM([project["wav"]], project["csv"]), checkpoint_path=project["ck"])
    idx = c1["starts"][20:24]
    buf = c1["padded"].copy()
    _, m0 = det.process_batch(buf, idx)
    assert np.abs(m0 - gold["c1_logits"]["logits"][20:24]).max() < 1e-4
    uploads = det._resident[3]
    _, m1 = det.process_batch(buf, idx)                          # unchanged: no new upload, same bits
    assert det._resident[3] == uploads and np.array_equal(m0, m1)
    n = len(buf)
    step = max(1, n // 64)
    p = (int(idx[1]) // step + 1) * step                         # a probe point of round 1's key (64 strided samples) inside window idx[1]
    edit = slice(p + 1, p + 3001)                                # ... and 3000 samples right behind it: between two probe points
    assert int(idx[1]) <= edit.start and edit.stop <= int(idx[1]) + 66150
    assert not any(edit.start <= k * step < edit.stop for k in range(65))      # (what that key would have missed)
    buf[edit] *= -0.5                                            # in-place edit of the caller's array
    _, m2 = det.process_batch(buf, idx)
    assert not np.array_equal(m2, m0)
    ref = O.infer_windows(det.model.state_dict(), buf, idx)
    assert np.abs(m2 - ref).max() < 1e-4
    x = torch.from_numpy(np.stack([buf[i:i + 66150] for i in idx[:2]]))
    det.model(x)                                                 # resets the arena behind the detector's back
    _, m3 = det.process_batch(buf, idx)
    assert np.array_equal(m3, m2)


def test_worker_progress_granularity_and_skipped_file(project, tmp_path):
    """worker.py:71-84: the file progress moves after every batch of settings.prediction_batch_size = 32 windows; a file that
    cannot be decoded is reported, skipped and still counted in the overall progress (which must reach 100 %)."""
    from root.code.frontend.NNDetector import NNDetector
    from root.code.backend.worker import ProcessWorker
    from softspoken_amd.detections import DetectionProject
    bad = os.path.join(project["dir"], "broken2.wav")
    open(bad, "wb").write(b"RIFF....WAVEjunk")
    csv = str(tmp_path / "q_detections.csv")
    pm = _PM([project["wav"]], csv)
    det = NNDetector(pm, checkpoint_path=project["ck"])
    plan = det.plan_detection_job()                              # (planning itself reads the header: the broken file joins the work list below)
    assert len(plan[project["wav"]]) == 105
    w = ProcessWorker(det, DetectionProject(pm), {bad: np.arange(3), project["wav"]: plan[project["wav"]]})
    ev = []
    w.signals.fileProgressChanged.connect(lambda p: ev.append(("prog", p)))
    w.signals.overallProgressChanged.connect(lambda p: ev.append(("overall", p)))
    w.signals.message.connect(lambda m: ev.append(("msg", m)))
    w.signals.fileDone.connect(lambda f: ev.append(("done", f)))
    w.run()
    assert [e[1] for e in ev if e[0] == "overall"] == [50.0, 100.0]
    assert any(e[0] == "msg" and "broken2.wav" in e[1] for e in ev)
    assert [e[1] for e in ev if e[0] == "done"] == [project["wav"]]
    progs = [e[1] for e in ev if e[0] == "prog"]
    assert progs == [pytest.approx(100.0 * min(32 * (k + 1), 105) / 105) for k in range(4)]


def _run_worker(files, csv, ck, precision=None):
    """One ProcessWorker job over `files` -> (csv text, detector, message signals)."""
    from root.code.frontend.NNDetector import NNDetector
    from root.code.backend.worker import ProcessWorker
    from softspoken_amd.detections import DetectionProject
    if os.path.exists(csv):
        os.remove(csv)
    pm = _PM(files, csv)
    det = NNDetector(pm, checkpoint_path=ck)
    if precision:
        det.model.precision = precision
    w = ProcessWorker(det, DetectionProject(pm), det.plan_detection_job())
    msgs = []
    w.signals.message.connect(msgs.append)
    w.run()
    return open(csv).read(), det, msgs


def test_worker_runs_one_unrepresentable_file_in_fp32_and_keeps_f16x2(project, c1, tmp_path, caplog):
    """VERDICT r02 item 4a / ADVICE r03: SS_ERR_RANGE must not turn into "message + skip", and ONE input the f16x2 mode cannot
    represent must not move the whole detector to fp32.  A float32 WAV with one NaN sample makes the f16x2 mode report SS_ERR_RANGE
    at run time: that file alone is run again on the model's fp32 side context (one log line), the other files stay in f16x2, and
    the job's CSV equals the one of a detector that ran in fp32 from the start (the reference's fp32 gives NaN scores for the
    windows over the sample and detections everywhere else: pytorch_neural_nets.py:142-197 cannot fail on magnitude).  A SECOND
    such file says something about the checkpoint or the data: the detector switches to fp32 for good."""
    import logging
    from softspoken_amd import synth
    x = (c1["pcm"].astype(np.float32) / np.float32(32768.0))
    x[16000 * 20] = np.float32("nan")
    nanwav = tmp_path / "with_nan.wav"
    nanwav.write_bytes(synth.wav_bytes(x, 16000, "f32"))
    files = [str(nanwav), project["wav"]]
    with caplog.at_level(logging.WARNING):
        got, det, msgs = _run_worker(files, str(tmp_path / "a.csv"), project["ck"])
    assert det.model.precision == "f16x2" and det.model.effective_precision() == "f16x2" and det.model.hip_context().precision == "f16x2"
    assert det.model._fp32_tmp is not None and det.model._fp32_tmp.precision == "fp32"
    assert not msgs                                                  # no file was skipped
    assert sum("fp32 mode" in r.getMessage() for r in caplog.records) == 1
    assert sum("cannot represent an input" in r.getMessage() for r in caplog.records) == 1
    want, det32, _ = _run_worker(files, str(tmp_path / "b.csv"), project["ck"], precision="fp32")
    assert got == want and got.count("with_nan.wav") >= 3 and got.count("c1_seed1001.wav") == 6
    # a second file with the status: permanent
    x[16000 * 40] = np.float32("inf")
    nan2 = tmp_path / "with_nan_and_inf.wav"
    nan2.write_bytes(synth.wav_bytes(x, 16000, "f32"))
    files3 = [str(nanwav), project["wav"], str(nan2)]
    caplog.clear()
    with caplog.at_level(logging.WARNING):
        got3, det3, msgs3 = _run_worker(files3, str(tmp_path / "c.csv"), project["ck"])
    assert not msgs3 and det3.model.effective_precision() == "fp32"
    assert sum("cannot represent this checkpoint" in r.getMessage() for r in caplog.records) == 1
    want3, _, _ = _run_worker(files3, str(tmp_path / "d.csv"), project["ck"], precision="fp32")
    assert got3 == want3 and got3.count("with_nan_and_inf.wav") >= 3


def test_worker_two_contexts_give_the_one_context_job(project, c1, tmp_path, caplog, monkeypatch):
    """ProcessWorker.run alternates a job of three or more files between two device contexts (settings.hip_file_contexts = 2): the CSV,
    the per-file signal order and the skipped-file message equal the one-context run's -- with a broken file in the list, and with a
    file in the middle that makes the f16x2 mode report SS_ERR_RANGE while the next file is already in flight on the other context
    (that file alone is run again on the fp32 side context; the file in flight is not disturbed).  The event lists are compared
    WITH the progress values: a file that is run again reports each of the reference's values once (worker.py:82-84)."""
    import logging
    import shutil
    from root.code.backend import settings
    from root.code.frontend.NNDetector import NNDetector
    from root.code.backend.worker import ProcessWorker
    from softspoken_amd.detections import DetectionProject
    from softspoken_amd import synth
    files = []
    for k in range(5):
        f = tmp_path / ("rec%d.wav" % k)
        shutil.copyfile(project["wav"], f)
        files.append(str(f))
    bad = tmp_path / "broken.wav"
    bad.write_bytes(b"RIFF....WAVEjunk")
    files.insert(2, str(bad))

    def job(n_ctx, names, csv):
        monkeypatch.setattr(settings, "hip_file_contexts", n_ctx)
        pm = _PM([f for f in names if not f.endswith("broken.wav")], csv)
        det = NNDetector(pm, checkpoint_path=project["ck"])
        plan = det.plan_detection_job()                              # (planning itself reads the headers: the broken file joins the work list below)
        w = ProcessWorker(det, DetectionProject(pm), {f: plan.get(f, np.arange(3)) for f in names})
        ev = []
        w.signals.fileStarted.connect(lambda f: ev.append(("start", os.path.basename(f))))
        w.signals.fileDone.connect(lambda f: ev.append(("done", os.path.basename(f))))
        w.signals.message.connect(lambda m: ev.append(("msg", "broken.wav" in m)))
        w.signals.overallProgressChanged.connect(lambda p: ev.append(("overall", round(p, 3))))
        w.signals.fileProgressChanged.connect(lambda p: ev.append(("prog", round(p, 3))))
        w.run()
        return open(csv).read(), ev, det

    two, ev2, det2 = job(2, files, str(tmp_path / "two.csv"))
    one, ev1, _ = job(1, files, str(tmp_path / "one.csv"))
    assert two == one and ev2 == ev1
    assert det2.model._ctx2 is not None and det2.model._ctx2.alive    # the second context was in use
    assert two.count("rec") == 5 * 6 and ("msg", True) in ev2 and ev2[-1] == ("overall", 100.0)
    # a NaN sample in the third of four files: the fourth is in flight on the other context when the fall-back closes both
    x = (c1["pcm"].astype(np.float32) / np.float32(32768.0))
    x[16000 * 20] = np.float32("nan")
    nanwav = tmp_path / "with_nan.wav"
    nanwav.write_bytes(synth.wav_bytes(x, 16000, "f32"))
    names = [files[0], files[1], str(nanwav), files[3], files[4]]
    with caplog.at_level(logging.WARNING):
        got, evn, detn = job(2, names, str(tmp_path / "nan2.csv"))
    assert detn.model.effective_precision() == "f16x2" and sum("fp32 mode" in r.getMessage() for r in caplog.records) == 1
    assert detn.model._ctx2 is not None and detn.model._ctx2.alive and detn.model._ctx2.precision == "f16x2"
    want, evw, _ = job(1, names, str(tmp_path / "nan1.csv"))
    assert got == want and evn == evw
    progs = [e[1] for e in evn if e[0] == "prog"]
    assert len(progs) == 5 * 4 and progs.count(100.0) == 5          # four values per 105-window file, none twice
    assert got.count("with_nan.wav") >= 3 and not any(e[0] == "msg" for e in evn)


_FALLBACK_SCRIPT = r"""
import os, sys, logging, numpy as np
sys.path.insert(0, {root!r})
import torch
from softspoken_amd import synth
from softspoken_amd.detections import DetectionProject
from root.code.frontend.NNDetector import NNDetector
from root.code.backend.worker import ProcessWorker
class PM:
    def __init__(self, files, det): self.files = files; self.current_project = {{'detections_file': det}}
    def get_unprocessed_list(self): return list(self.files)
sd = synth.make_state_dict(0)
for name in ("conv1_1.conv2.1.weight", "conv2_1.conv2.1.weight"):
    sd[name] = sd[name] * np.float32(400.0)
ck = os.path.join({tmp!r}, "gain400.pth")
torch.save({{"model_state_dict": synth.to_torch_state_dict(sd), "epoch": 0}}, ck)
wav = os.path.join({tmp!r}, "c1.wav")
synth.write_wav(wav, synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1)), 16000)
out = {{}}
for prec in ("f16x2", "fp32"):
    csv = os.path.join({tmp!r}, prec + ".csv")
    pm = PM([wav], csv)
    det = NNDetector(pm, checkpoint_path=ck)
    det.model.precision = prec
    w = ProcessWorker(det, DetectionProject(pm), det.plan_detection_job())
    msgs = []
    w.signals.message.connect(msgs.append)
    w.run()
    assert not msgs, msgs
    out[prec] = (open(csv).read(), det.model.effective_precision())
assert out["f16x2"][1] == "fp32" and out["fp32"][1] == "fp32"
assert out["f16x2"][0] == out["fp32"][0] and out["f16x2"][0].count("c1.wav") >= 1, out
print("FALLBACK_OK", out["f16x2"][0].count("c1.wav"))
"""


def test_worker_fallback_on_a_checkpoint_beyond_the_f16_range(tmp_path, build_all):
    """The x 400 BatchNorm-gain checkpoint of test_f16x2_reports_values_outside_the_f16_range through ProcessWorker, in the development
    build with the channel normalisation switched off (SOFTSPOKEN_NORM=0; with it the checkpoint simply runs in f16x2): the f16x2
    context reports SS_ERR_RANGE in the first file's run, the detector re-runs it in fp32, the CSV equals the fp32 detector's."""
    import subprocess, sys
    from softspoken_amd import build as hip_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e["SOFTSPOKEN_LIB"] = hip_build.DEV_LIB; e["SOFTSPOKEN_NORM"] = "0"
    r = subprocess.run([sys.executable, "-c", _FALLBACK_SCRIPT.format(root=root, tmp=str(tmp_path))], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_worker_runs_a_huge_gain_checkpoint_in_f16x2(project, tmp_path):
    """BatchNorm gains of 400 and 4e6 (round 2: SS_ERR_RANGE) through ProcessWorker with the product library: the channel
    normalisation keeps every stored value near 1, the job stays in f16x2 and its CSV equals the fp32 detector's."""
    from softspoken_amd import synth
    sd = synth.make_state_dict(0)
    for name in ("conv1_1.conv2.1.weight", "conv2_1.conv2.1.weight"):
        sd[name] = sd[name] * np.float32(400.0)
    sd["conv2_1.conv2.1.weight"] = sd["conv2_1.conv2.1.weight"] * np.float32(1e4)
    ck = str(tmp_path / "gain4e6.pth")
    torch.save({"model_state_dict": synth.to_torch_state_dict(sd), "epoch": 0}, ck)
    got, det, msgs = _run_worker([project["wav"]], str(tmp_path / "a.csv"), ck)
    assert det.model.effective_precision() == "f16x2" and not msgs
    want, _, _ = _run_worker([project["wav"]], str(tmp_path / "b.csv"), ck, precision="fp32")
    assert got == want


_DEVJOB_PRELUDE = r"""
import os, sys, logging, numpy as np
sys.path.insert(0, {root!r})
import torch
from softspoken_amd import synth, native
from softspoken_amd.detections import DetectionProject
from root.code.backend import settings
from root.code.frontend.NNDetector import NNDetector
from root.code.backend.worker import ProcessWorker
class PM:
    def __init__(self, files, det): self.files = files; self.current_project = {{'detections_file': det}}
    def get_unprocessed_list(self): return list(self.files)
tmp = {tmp!r}
def job(files, ck, csv, prec=None, hook=None, n_ctx=2):
    settings.hip_file_contexts = n_ctx
    pm = PM(files, csv)
    det = NNDetector(pm, checkpoint_path=ck)
    if prec: det.model.precision = prec
    if hook: hook(det)
    w = ProcessWorker(det, DetectionProject(pm), det.plan_detection_job())
    ev = []
    w.signals.message.connect(lambda m: ev.append(("msg", m)))
    w.signals.fileDone.connect(lambda f: ev.append(("done", os.path.basename(f))))
    w.signals.fileProgressChanged.connect(lambda p: ev.append(("prog", round(p, 3))))
    w.run()
    return open(csv).read(), det, ev
"""

_SELFCHECK_SCRIPT = _DEVJOB_PRELUDE + r"""
# conv3_1's h tensor 2^-14 of its size, undone in the conv behind it: the same function in fp32, no overflow anywhere, but with the
# channel normalisation off (SOFTSPOKEN_NORM=0) the f16 pairs of that tensor keep ~11 bits
sd = synth.make_state_dict(0)
g = np.float32(2.0 ** -14)
for k in ("weight", "bias"):
    sd["conv3_1.conv1.1." + k] = sd["conv3_1.conv1.1." + k] * g
sd["conv3_1.conv2.0.weight"] = sd["conv3_1.conv2.0.weight"] * np.float32(2.0 ** 14)
ck = os.path.join(tmp, "quiet_h3.pth")
torch.save({{"model_state_dict": synth.to_torch_state_dict(sd), "epoch": 0}}, ck)
wav = os.path.join(tmp, "c1.wav")
synth.write_wav(wav, synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1)), 16000)
logging.basicConfig(level=logging.WARNING)
got, det, ev = job([wav], ck, os.path.join(tmp, "a.csv"))
assert not [e for e in ev if e[0] == "msg"], ev
print("DELTA", det.model.selfcheck_delta, det.model.effective_precision(), det.model.hip_context().precision)
want, det32, _ = job([wav], ck, os.path.join(tmp, "b.csv"), prec="fp32")
assert got == want and got.count("c1.wav") >= 1
if {expect_fp32!r}:
    assert det.model.selfcheck_delta > 5e-5 and det.model.effective_precision() == "fp32" and det.model.hip_context().precision == "fp32"
else:
    assert det.model.selfcheck_delta <= 5e-5 and det.model.effective_precision() == "f16x2"
print("SELFCHECK_OK")
"""


def _run_script(script, tmp_path, dev, extra_env=None, **fmt):
    import subprocess, sys
    from softspoken_amd import build as hip_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    if dev:
        e["SOFTSPOKEN_LIB"] = hip_build.DEV_LIB
    e.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", script.format(root=root, tmp=str(tmp_path), **fmt)], env=e, capture_output=True, text=True, timeout=600)


def test_load_time_check_keeps_fp32_when_f16x2_loses_precision_without_overflow(tmp_path, build_all):
    """nothing at load time compared the two modes on the checkpoint actually loaded; precision lost WITHOUT an
    overflow raised no flag.  A checkpoint whose conv3_1 intermediate sits at 2^-14 of its natural size (undone by the next conv: the
    same function for the reference's fp32, NNDetector.py:21-53 + pytorch_neural_nets.py:7-41) in the development build with the channel
    normalisation off: SpecUNet_2D._selfcheck sees f16x2 and fp32 differ by more than 5e-5 on its fixed windows, the detector
    logs once and ends up in fp32, its CSV equals the fp32 detector's.  With the product library (normalisation on) the same
    checkpoint passes the check and stays in f16x2."""
    r = _run_script(_SELFCHECK_SCRIPT, tmp_path, dev=True, extra_env={"SOFTSPOKEN_NORM": "0"}, expect_fp32=True)
    assert r.returncode == 0 and "SELFCHECK_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    assert r.stderr.count("load-time check") == 1
    r = _run_script(_SELFCHECK_SCRIPT, tmp_path, dev=False, expect_fp32=False)
    assert r.returncode == 0 and "SELFCHECK_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


_NOMEM_SCRIPT = _DEVJOB_PRELUDE + r"""
ck = os.path.join(tmp, "ck.pth")
synth.save_checkpoint(ck, 0, epoch=0)
files = []
for k in range(5):
    f = os.path.join(tmp, "rec%d.wav" % k)
    synth.write_wav(f, synth.to_pcm16(synth.synth_audio(1001 + (k % 2), 30.0 + 6 * k, 16000, 1)), 16000)
    files.append(f)
def starve(det):            # the second context exists, but its first workspace allocation fails (ss_debug_fail_workspace_alloc)
    det.model.hip_context(1).debug_fail_workspace_alloc(0)
logging.basicConfig(level=logging.WARNING)
two, det2, ev2 = job(files, ck, os.path.join(tmp, "two.csv"), hook=starve)
one, det1, ev1 = job(files, ck, os.path.join(tmp, "one.csv"), n_ctx=1)
assert two == one and ev2 == ev1, (ev2, ev1)
assert not [e for e in ev2 if e[0] == "msg"] and len([e for e in ev2 if e[0] == "done"]) == 5
assert det2.model._ctx2 is None
# a file whose poll raises leaves its context usable for the next file
class Boom(Exception): pass
def poll_boom(det):
    real = det.file_poll
    state = dict(n=0)
    def fp(token, progress=None, block=True):
        state["n"] += 1
        if state["n"] == 2: raise Boom("poll failed")
        return real(token, progress, block)
    det.file_poll = fp
three, det3, ev3 = job(files, ck, os.path.join(tmp, "three.csv"), hook=poll_boom)
msgs = [e for e in ev3 if e[0] == "msg"]
assert len(msgs) == 1 and "rec1.wav" in msgs[0][1] and "poll failed" in msgs[0][1], ev3
assert [e[1] for e in ev3 if e[0] == "done"] == ["rec0.wav", "rec2.wav", "rec3.wav", "rec4.wav"], ev3
print("NOMEM_OK")
"""


def test_worker_degrades_to_one_context_when_the_second_workspace_does_not_fit(tmp_path, build_all):
    """for jobs of three or more files run() alternates the files between two device contexts; when the second
    context's workspace cannot be allocated (SS_ERR_NOMEM from run_begin: a shared card, a large SOFTSPOKEN_CHUNK) every odd file
    used to be reported as failed and skipped.  Now the second context is closed and the job continues on context 0: same CSV and
    same signals as the one-context job, no message.  And a file whose poll raises is aborted on its context (run_end), so the next
    file of that context runs instead of failing at reset."""
    r = _run_script(_NOMEM_SCRIPT, tmp_path, dev=True)
    assert r.returncode == 0 and "NOMEM_OK" in r.stdout, (r.stdout[-800:], r.stderr[-3000:])
    assert r.stderr.count("continuing with one context") == 1


def test_worker_takes_an_aiff_file_like_the_wav_with_the_same_samples(project, c1, tmp_path):
    """sf.read, which the reference loads files with (voice_activity.py:37), reads AIFF / AIFF-C as well as WAV; the drop-in walks both
    containers (round 4).  The C1 recording as big-endian AIFF, little-endian AIFF-C ('sowt') and WAV through get_audio_data, load_audio
    and one ProcessWorker job: the same duration, the same signal, the same detection rows."""
    from root.code.backend.voice_activity import get_audio_data, load_audio
    from softspoken_amd import synth
    names = {"c1.aif": synth.aiff_bytes(c1["pcm"], 16000, 16), "c1_sowt.aifc": synth.aiff_bytes(c1["pcm"], 16000, 16, b"sowt")}
    files = []
    for n, img in names.items():
        f = tmp_path / n
        f.write_bytes(img)
        files.append(str(f))
        assert get_audio_data(str(f)) == (60.0, 16000)
        data, sr = load_audio(str(f))
        assert sr == 22050 and np.array_equal(data, c1["sig"])
    got, det, msgs = _run_worker(files + [project["wav"]], str(tmp_path / "a.csv"), project["ck"])
    assert not msgs
    import pandas as pd
    df = pd.read_csv(str(tmp_path / "a.csv"))
    per = [df[df.file_name == os.path.basename(f)][["start_time", "end_time"]].reset_index(drop=True) for f in files + [project["wav"]]]
    assert len(per[0]) == 6 and per[0].equals(per[1]) and per[0].equals(per[2])
