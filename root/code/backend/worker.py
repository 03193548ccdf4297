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
import logging
from os.path import basename, dirname

import numpy as np
import pandas as pd

from root.code.backend import settings

_SS_ERR_NOMEM = 6                                # include/softspoken.h (softspoken_amd.native.SS_ERR_NOMEM)

# NOTE: PySide6 is not in this image (nor on the GPU box): the Qt branch below (QRunnable + Signal classes, reference
# worker.py:4-32) has never executed here; every test drives the headless classes, which have the same connect()/emit() surface.
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

    def _append_rows(self, file, regions):
        """worker.py:103-125: one row per region, IDs continuing from the current maximum.  The first row is appended with the
        reference's statement -- on an empty frame it decides what the columns' dtypes become --, the others as one block of those
        dtypes (the statement costs ~0.8 ms per row: 50 ms for a 10-minute recording's 64 rows, twice its time on the GPU)."""
        if not regions:
            return
        df = self.detection_project.df
        file_path, file_name = dirname(file), basename(file)
        next_id = self._next_id()
        s0, e0 = regions[0]
        df.loc[len(df)] = {'ID': next_id, 'file_path': file_path, 'file_name': file_name, 'start_time': s0, 'end_time': e0,
                           'erase': 0, 'user_comment': '', 'review_datetime': ''}
        if len(regions) > 1:
            n = len(regions) - 1
            block = pd.DataFrame({'ID': np.arange(next_id + 1, next_id + 1 + n, dtype=np.int64), 'file_path': file_path, 'file_name': file_name,
                                  'start_time': pd.Series([r[0] for r in regions[1:]], dtype=object),
                                  'end_time': pd.Series([r[1] for r in regions[1:]], dtype=object),
                                  'erase': 0, 'user_comment': '', 'review_datetime': ''}, columns=list(df.columns))
            block.index = range(len(df), len(df) + n)
            self.detection_project.df = pd.concat([df, block.astype(df.dtypes.to_dict())])

    def run(self):
        """The reference's signal sequence per file (worker.py:49-139), with the device ahead of the host: files alternate between
        two device contexts (settings.hip_file_contexts), so while file k's results are read, its rows appended and the CSV written,
        file k + 1 is already running on the other context's stream -- queued before file k's last launch ended, so the device never
        waits for the host between files --, file k + 2 starts as soon as k's context is free, and the samples of the files behind
        them are crossing PCIe."""
        det = self.detector
        files = list(self.planned_work.keys())
        total_files = len(files)
        files_done = 0
        # (a second context costs its creation and a second workspace: not for a job of one or two files)
        n_ctx = 2 if int(getattr(settings, "hip_file_contexts", 2)) >= 2 and total_files >= 3 else 1
        tokens, handles = {}, {}
        LATER = object()                 # a file whose turn on the one remaining context has not come yet

        def begin(i):                    # -> token, or the exception that file raised (reported when its turn comes)
            nonlocal n_ctx
            which = i % n_ctx
            try:
                return det.file_begin(files[i], handles.pop(i, None), which=which)
            except Exception as e:
                if which == 1 and getattr(e, "code", None) == _SS_ERR_NOMEM:
                    # the second context's workspace does not fit (a shared card, a large SOFTSPOKEN_CHUNK): its memory goes back and
                    # this file and every later one run on context 0, one after the other, as in round 2 -- no file is lost
                    logging.warning("second file context: %s -- continuing with one context", e)
                    det.model.drop_second_context()
                    n_ctx = 1
                    handles.clear()
                    return LATER
                return e

        def prefetch(i):
            if i < total_files and n_ctx > 1:
                try:
                    handles[i] = det.file_prefetch(files[i], which=i % n_ctx)
                except Exception:
                    pass                 # (file_begin walks the header again and raises in turn)

        for i in range(min(n_ctx, total_files)):
            if not self.stop_requested:
                tokens[i] = begin(i)
        for i in range(n_ctx, 2 * n_ctx):
            if not self.stop_requested:
                prefetch(i)
        for i, file in enumerate(files):
            if self.stop_requested:
                break
            self.signals.fileStarted.emit(file)
            regions, err = None, None
            token = tokens.pop(i, None)
            if token is LATER:                                  # (context 0 is free now: the file before this one has ended)
                token = begin(i)
            if token is None:                                   # (stop was requested before this file could start)
                break
            if isinstance(token, Exception):
                err = token
            else:
                progress = lambda done, total: self.signals.fileProgressChanged.emit((done / max(total, 1)) * 100.0)
                try:
                    det.file_poll(token, progress)
                    regions = det.file_end(token, progress)
                except Exception as e:
                    err = e
                    det.file_abort(token)                       # the context must not stay "run in flight" for the next file
            if self.stop_requested and err is None:             # interrupted: discard the partial file (worker.py:86-87)
                break
            nxt = i + n_ctx                                     # this file's context takes its next file before the rows are filed
            if nxt < total_files and not self.stop_requested and (nxt not in tokens or tokens[nxt] is LATER):
                tokens[nxt] = begin(nxt)
                if tokens[nxt] is not LATER and not isinstance(tokens[nxt], Exception):
                    prefetch(i + 2 * n_ctx)
            if err is not None:                                  # undecodable / failed file: reported and skipped; it still counts towards
                self.signals.message.emit(f"{file}: {err}")      # the overall progress, which would otherwise never reach 100 %
                files_done += 1
                self.signals.overallProgressChanged.emit((files_done / total_files) * 100.0)
                continue
            self._append_rows(file, regions)
            self.detection_project.save_detections()
            self.signals.fileDone.emit(file)
            files_done += 1
            self.signals.overallProgressChanged.emit((files_done / total_files) * 100.0)
        for token in tokens.values():                            # files still in flight when the loop was left
            if token is not None and token is not LATER and not isinstance(token, Exception):
                det.file_abort(token)
        self.signals.finished.emit()
