"""Background job driver of the voice detector (reference root/code/backend/worker.py:4-139).

Same class names, constructor, `signals`, `stop()` and `run()` contract as the reference, so
silencer_ui.py:234-243 can hand it to a QThreadPool unchanged.  PySide6 is optional: with it the
worker is a QRunnable with real Qt signals; without it (headless jobs, tests, the bench) `signals`
are small objects with the same connect()/emit() surface and run() is called directly.

Per file the behaviour seen by the caller is the reference's: fileStarted -> fileProgressChanged
after every batch of windows -> rows appended to detection_project.df (ID continues from the
current maximum, erase 0, empty comment/datetime) -> save_detections() -> fileDone ->
overallProgressChanged; finished at the end; stop() is honoured between files and between
batches and a file interrupted half-way leaves no rows (worker.py:72,86-87).
Differences: the file is decoded, mixed down, resampled, padded and kept in HBM by the library
(one upload per file, not per batch), averaging runs on the GPU, and a file that cannot be decoded
is reported through signals.message and skipped (the reference crashes on len(None), worker.py:60).
"""
from __future__ import annotations

import ctypes
from os.path import basename, dirname

import numpy as np
import pandas as pd

from root.code.backend import settings

try:                                            # GUI build
    from PySide6.QtCore import QObject, QRunnable, Signal
    _HAVE_QT = True
except Exception:                               # headless
    _HAVE_QT = False


if _HAVE_QT:
    class WorkerSignals(QObject):
        fileProgressChanged = Signal(float)
        overallProgressChanged = Signal(float)
        fileStarted = Signal(str)
        fileDone = Signal(str)
        finished = Signal()
        message = Signal(str)

    _Base = QRunnable
else:
    class _PlainSignal:
        def __init__(self):
            self._slots = []

        def connect(self, fn):
            self._slots.append(fn)

        def emit(self, *a):
            for fn in list(self._slots):
                fn(*a)

    class WorkerSignals:
        def __init__(self):
            for name in ("fileProgressChanged", "overallProgressChanged", "fileStarted", "fileDone", "finished", "message"):
                setattr(self, name, _PlainSignal())

    class _Base:
        def __init__(self):
            pass


class ProcessWorker(_Base):
    def __init__(self, detector, detection_project, planned_work, parent=None):
        super().__init__()
        self.signals = WorkerSignals()
        self.detector = detector
        self.detection_project = detection_project
        self.planned_work = planned_work
        self.stop_requested = False
        self._stop_word = ctypes.c_int(0)       # polled by the library between chunks

    def stop(self):
        self.stop_requested = True
        self._stop_word.value = 1

    def _next_id(self):
        df = self.detection_project.df
        if not df.empty and 'ID' in df.columns:
            top = pd.to_numeric(df['ID'], errors='coerce').max()
            if not np.isnan(top):
                return int(top) + 1
        return 1

    def run(self):
        total_files = len(self.planned_work)
        files_done = 0
        for file in list(self.planned_work.keys()):
            if self.stop_requested:
                break
            self.signals.fileStarted.emit(file)
            try:
                regions = self.detector.detect_files(
                    [file],
                    progress=lambda done, total: self.signals.fileProgressChanged.emit((done / max(total, 1)) * 100.0),
                    stop_flag=self._stop_word)
            except Exception as e:                          # undecodable / failed file: reported and skipped; it still counts towards
                self.signals.message.emit(f"{file}: {e}")   # the overall progress, which would otherwise never reach 100 %
                files_done += 1
                self.signals.overallProgressChanged.emit((files_done / total_files) * 100.0)
                continue
            if regions is None or self.stop_requested:      # interrupted: discard the partial file
                break
            file_path, file_name = dirname(file), basename(file)
            next_id = self._next_id()
            for (start_time, end_time) in regions[file]:
                self.detection_project.df.loc[len(self.detection_project.df)] = {
                    'ID': next_id, 'file_path': file_path, 'file_name': file_name,
                    'start_time': start_time, 'end_time': end_time,
                    'erase': 0, 'user_comment': '', 'review_datetime': ''}
                next_id += 1
            self.detection_project.save_detections()
            self.signals.fileDone.emit(file)
            files_done += 1
            self.signals.overallProgressChanged.emit((files_done / total_files) * 100.0)
        self.signals.finished.emit()
