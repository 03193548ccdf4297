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
