"""Generate tests/golden/review_case.json by running the REFERENCE's own review / export code on a small
synthetic project.  Runs only in the build container (needs /root/reference); the JSON is committed.

What is the reference's own code here, and what is a stand-in:
  * root/code/frontend/review_exporter.py -- imported and run as is (ReviewExportManager and the three
    transforms write every export file of the fixture).
  * root/code/frontend/review_detections.py -- imported as is; `filter_by_minimum_detection_len`,
    `_ensure_id_column_first`, `_assign_missing_ids`, `populate_table`, `apply_label_to_current_detection`
    and `save_review` of ReviewDetectionsScreen run unmodified on an instance made with __new__ (its
    __init__ builds the window).  The data steps of __init__ (:220-237) are replayed by calling those
    methods in the same order.
  * PySide6 is not in this image: its modules are in-memory dummies; the screen's QTableWidget is a
    list-of-strings table with the handful of methods the code above calls.  Nothing numeric or textual
    is computed by the dummies -- they only hold the strings the reference code puts in and reads back.
  * soundfile is not in this image: `sf.info` is served by the standard library's `wave` module (which
    the reference file itself uses in its first `_wav_duration`, :22-24).  librosa / sounddevice /
    torchaudio are empty modules; none of their functions is called.
  * `datetime.datetime.now()` inside review_detections is pinned to a fixed instant.

Paths inside the fixture are written with the placeholder @ROOT@ so the test can run in any directory.

Usage:  python tests/golden/make_review_golden.py
"""
import datetime as _dt
import json
import os
import shutil
import sys
import tempfile
import types
import wave

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

FIXED_NOW = _dt.datetime(2025, 3, 14, 15, 9, 26)
PROJECT = "demo"

# (relative dir, file name, seconds, rate) -- the third recording is listed but absent on disk, which
# sends the Raven exporter down its "length = last detection end" fallback
RECORDINGS = [("siteA", "rec_01.wav", 2.5, 8000), ("siteA/sub", "rec_02.wav", 4.0, 16000),
              ("siteB", "rec 03.wav", None, None)]

# detector output rows: (recording index, start, end); unsorted, with short and borderline detections
DETECTIONS = [
    (1, 0.30000000000000004, 1.7999999999999998),
    (0, 1.2000000000000002, 1.2600000000000002),      # 0.06 s  -> filtered
    (0, 0.0, 0.8999999999999999),
    (2, 10.2, 11.4006),
    (1, 2.4000000000000004, 2.5000000000000004),      # exactly the limit in floating point -> see filter
    (0, 1.5, 2.4999999999999996),
    (2, 3.5994, 3.7),                                 # 0.1006
    (1, 3.0, 3.05),                                   # filtered
]


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _DummyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Dummy,), {})
        setattr(self, name, cls)
        return cls


class QTableWidgetItem(_Dummy):
    def __init__(self, text=""):
        self._text = str(text)

    def text(self):
        return self._text


class StringTable(_Dummy):
    """Holds what the screen's QTableWidget would: header labels and one text item per cell."""

    def __init__(self):
        self.headers, self.rows, self.ncol = [], [], 0

    def setRowCount(self, n):
        self.rows = self.rows[:n] + [dict() for _ in range(n - len(self.rows))]

    def setColumnCount(self, n):
        self.ncol = n

    def setHorizontalHeaderLabels(self, labels):
        self.headers = [QTableWidgetItem(x) for x in labels]

    def insertRow(self, i):
        self.rows.insert(i, dict())

    def setItem(self, r, c, item):
        self.rows[r][c] = item

    def item(self, r, c):
        return self.rows[r].get(c)

    def rowCount(self):
        return len(self.rows)

    def columnCount(self):
        return self.ncol

    def horizontalHeaderItem(self, c):
        return self.headers[c]


def install_standins():
    for name in ("PySide6", "PySide6.QtMultimedia", "PySide6.QtCore", "PySide6.QtGui", "PySide6.QtWidgets"):
        sys.modules[name] = _DummyModule(name)
    sys.modules["PySide6.QtWidgets"].QTableWidgetItem = QTableWidgetItem
    for name in ("sounddevice", "librosa", "librosa.display", "torchaudio", "torchaudio.transforms"):
        sys.modules[name] = _DummyModule(name)
    sys.modules["librosa"].display = sys.modules["librosa.display"]
    sys.modules["torchaudio"].transforms = sys.modules["torchaudio.transforms"]
    sf = types.ModuleType("soundfile")

    class _Info:
        def __init__(self, path):
            with wave.open(str(path), "rb") as fh:
                self.frames, self.samplerate = fh.getnframes(), fh.getframerate()

    sf.info = _Info
    sys.modules["soundfile"] = sf


def write_wav(path, seconds, rate):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    n = int(round(seconds * rate))
    t = np.arange(n) / rate
    pcm = (np.sin(2 * np.pi * 440 * t) * 8000).astype("<i2")
    with wave.open(path, "wb") as fh:
        fh.setnchannels(1)
        fh.setsampwidth(2)
        fh.setframerate(rate)
        fh.writeframes(pcm.tobytes())


def build_project(root):
    """Recordings + the detections CSV exactly as the detector's DetectionProject writes it."""
    for d, f, seconds, rate in RECORDINGS:
        if seconds is not None:
            write_wav(os.path.join(root, "audio", d, f), seconds, rate)
    rows = []
    for i, (rec, s, e) in enumerate(DETECTIONS):
        d, f, _, _ = RECORDINGS[rec]
        rows.append({"ID": i + 1, "file_path": os.path.join(root, "audio", d), "file_name": f,
                     "start_time": str(s), "end_time": str(e), "erase": 0, "user_comment": "",
                     "review_datetime": pd.NaT})
    df = pd.DataFrame(rows)
    proj = os.path.join(root, "projects")
    os.makedirs(proj, exist_ok=True)
    det = os.path.join(proj, f"{PROJECT}_detections.csv")
    df.to_csv(det, index=False)
    return det, os.path.join(proj, f"{PROJECT}_review.csv"), proj


def collect(root, skip=()):
    out = {}
    for base, _, files in os.walk(os.path.join(root, "projects")):
        for f in sorted(files):
            p = os.path.join(base, f)
            rel = os.path.relpath(p, root)
            if rel in skip:
                continue
            with open(p, newline="") as fh:
                out[rel] = fh.read().replace(root, "@ROOT@")
    return out


def main():
    install_standins()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    root = tempfile.mkdtemp(prefix="ss_review_")
    try:
        os.chdir(root)                      # the exporters are handed dst="." by save_review
        import root.code.frontend.review_detections as RD

        class _Clock:
            class datetime(_dt.datetime):
                @classmethod
                def now(cls, tz=None):
                    return FIXED_NOW
        RD.datetime = _Clock

        det, rev, proj = build_project(root)
        pm = types.SimpleNamespace(current_project={"name": PROJECT, "detections_file": det, "review_file": rev},
                                   projects_folder=proj)
        scr = RD.ReviewDetectionsScreen.__new__(RD.ReviewDetectionsScreen)
        scr.project_manager, scr.parent_app_screen = pm, None
        scr.table = StringTable()
        scr.scroll = lambda *_a, **_k: None
        # data steps of __init__ (:220-237), first opening: detections CSV -> filter -> ID first -> table
        scr.csv_data = pd.read_csv(det)
        scr.filter_by_minimum_detection_len()
        scr.csv_data = scr._ensure_id_column_first(scr.csv_data)
        scr.populate_table()
        stages = {"detections_csv": open(det, newline="").read().replace(root, "@ROOT@")}
        scr.save_review(persist=True)
        stages["first_save"] = collect(root, skip=(os.path.relpath(det, root),))

        # review two rows (erase the 2nd, keep the 4th), comment on one, save through the reference
        scr.current_index = 1
        scr.apply_label_to_current_detection(1)
        scr.current_index = 3
        scr.apply_label_to_current_detection(0)
        col = list(scr.csv_data.columns).index("user_comment")
        scr.table.setItem(0, col, QTableWidgetItem('two people, "quoted", far'))
        scr.save_review(persist=True)
        stages["after_labels"] = collect(root, skip=(os.path.relpath(det, root),))

        # second opening: the review CSV is read back (:224-225), re-populated and saved again
        scr2 = RD.ReviewDetectionsScreen.__new__(RD.ReviewDetectionsScreen)
        scr2.project_manager, scr2.parent_app_screen = pm, None
        scr2.table = StringTable()
        scr2.csv_data = scr2._ensure_id_column_first(pd.read_csv(rev))
        scr2.populate_table()
        # a hand-drawn detection is inserted as text with a blank ID (:606-626)
        hdr = [h.text() for h in scr2.table.headers]
        scr2.table.insertRow(2)
        d, f, _, _ = RECORDINGS[0]
        manual = {"file_path": os.path.join(root, "audio", d), "file_name": f, "start_time": f"{2.0:.3f}",
                  "end_time": f"{2.25:.3f}"}
        for c, h in enumerate(hdr):
            scr2.table.setItem(2, c, QTableWidgetItem(manual.get(h, "")))
        scr2.save_review(persist=True)
        stages["reopened"] = collect(root, skip=(os.path.relpath(det, root),))
    finally:
        os.chdir(cwd)
        shutil.rmtree(root, ignore_errors=True)

    fixture = {"project": PROJECT, "fixed_now": FIXED_NOW.isoformat(), "minimum_detection_len": 0.1,
               "recordings": RECORDINGS, "stages": stages}
    with open(os.path.join(HERE, "review_case.json"), "w") as fh:
        json.dump(fixture, fh, indent=1)
    print("wrote review_case.json:", {k: (list(v) if isinstance(v, dict) else len(v)) for k, v in stages.items()})


if __name__ == "__main__":
    main()
